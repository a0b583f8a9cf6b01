#!/bin/bash
# L2 / fabric transaction counters of the standalone hash gather (ours and the reference's kernel compiled for this GPU): tools/gpu_gather_pmc.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
for who in ours ref; do
  for pts in random frame; do
    A="$pts"; [ $who = ref ] && A="$pts ref"
    python tools/probe/gather_pmc.py $A > $OUT/time_${who}_$pts.txt 2>&1
    i=0
    for SET in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES"; do
      i=$((i+1))
      ( cd /tmp && timeout 300 rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_${who}_${pts}_$i -o pmc -- python $GRAFT_REPO_ROOT/tools/probe/gather_pmc.py $A > $OUT/pmc_${who}_${pts}_$i.log 2>&1 )
    done
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
            if "forward" in row.get("Kernel_Name", "").lower() or "kernel_grid" in row.get("Kernel_Name", ""):
                per[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, x in per.items():
        table[k][v] = sum(x) / len(x)
vs = sorted({v for t in table.values() for v in t})
lines = [open(f).read().strip().splitlines()[-1] for f in sorted(glob.glob(out + "/time_*.txt"))]
lines += ["", "per launch (7.7 M points x 16 levels x 8 corner rows = 985.6 M gathered rows of 8 B):", "counter".ljust(36) + "".join(v.rjust(16) for v in vs)]
for k in sorted(table):
    lines.append(k.ljust(36) + "".join((f"{table[k][v]:.4g}" if v in table[k] else "-").rjust(16) for v in vs))
open(out + "/gather_pmc.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
