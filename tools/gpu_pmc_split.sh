#!/bin/bash
# PMC of k_env_split on one 800x800 frame's records (through gpurun): tools/gpu_pmc_split.sh <tag> [lib.so]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
[ -n "$2" ] && export ENVIDR_AMD_LIB=$GRAFT_REPO_ROOT/$2
python tools/geo/split_probe.py > $OUT/split_time.txt 2>&1; cat $OUT/split_time.txt
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/tools/geo/split_probe.py > $OUT/trace.log 2>&1 )
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  N=$(echo $SET | tr ' ' '_' | cut -c1-24)
  ( cd /tmp && timeout 300 rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_$N -o pmc -- python $GRAFT_REPO_ROOT/tools/geo/split_probe.py > $OUT/pmc_$N.log 2>&1 )
done
python - <<PY
import csv, glob, collections
per = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_env_split" in row.get("Kernel_Name", ""):
            per[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(per.items()):
    print(f"{k} = {sum(v)/len(v):.5g} (n={len(v)})")
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_env_split" in row["Name"] or "k_shade" in row["Name"]:
            print(row["Name"][:60], row["Calls"], row["AverageNs"])
PY
