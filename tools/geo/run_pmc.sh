#!/bin/bash
# tools/geo/run_pmc.sh <tag> <variant>: PMC passes over the probe for one variant
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=$1; V=$2; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export ENVIDR_AMD_LIB=$GRAFT_REPO_ROOT/tools/geo/variants/$V.so
rocprofv3 -L 2>/dev/null | grep -oE "(TA|TCP|TCC|TD|SQ)_[A-Za-z0-9_]+" | sort -u > $OUT/counters.txt
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_BUSY_avr TA_TA_BUSY_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM"; do
  N=$(echo $SET | tr ' ' '_' | cut -c1-30)
  ( cd /tmp && timeout 300 rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_$N -o pmc -- python $GRAFT_REPO_ROOT/tools/geo/eval_probe.py 800 > $OUT/pmc_$N.log 2>&1 )
done
python - <<PY
import csv, glob, collections
per = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_geo_eval" in row.get("Kernel_Name", ""):
            per[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(per.items()):
    print(f"$V {k} = {sum(v)/len(v):.4g} (n={len(v)})")
PY
