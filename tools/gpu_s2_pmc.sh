#!/bin/bash
# wave-state counters of the fused-pair split kernel under variants: tools/gpu_s2_pmc.sh <tag> <variant...>   ("base" = the product library)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
for v in "$@"; do
  [ "$v" = base ] && unset ENVIDR_AMD_LIB || export ENVIDR_AMD_LIB=$GRAFT_REPO_ROOT/tools/geo/variants/$v.so
  i=0
  for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU" "SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    ( cd /tmp && timeout 300 rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_${v}_$i -o pmc -- python $GRAFT_REPO_ROOT/tools/geo/split_probe.py > $OUT/pmc_${v}_$i.log 2>&1 )
  done
done
python - <<PY
import csv, glob, collections, os
out = "$OUT"
table = collections.defaultdict(dict)
for d in sorted(glob.glob(out + "/pmc_*_[0-9]")):
    v = os.path.basename(d)[4:-2]
    per = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_env_split" in row.get("Kernel_Name", ""):
                per[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, x in per.items():
        table[k][v] = sum(x) / len(x)
vs = sorted({v for t in table.values() for v in t})
lines = ["counter".ljust(28) + "".join(v.rjust(16) for v in vs)]
for k in sorted(table):
    lines.append(k.ljust(28) + "".join((f"{table[k][v]:.4g}" if v in table[k] else "-").rjust(16) for v in vs))
open(out + "/pmc_table.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
