#!/bin/bash
# two-group split kernel: parity tests, time, kernel stats, MFMA-busy counters.  tools/gpu_split2.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
timeout 900 python -m pytest tests/test_split_gpu.py -x -q > $OUT/pytest_split.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_split.log; tail -8 $OUT/pytest_split.log
timeout 300 python tools/geo/split_probe.py > $OUT/split_time.txt 2>&1
timeout 300 python tools/geo/split_probe.py --v1 >> $OUT/split_time.txt 2>&1; cat $OUT/split_time.txt
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/tools/geo/split_probe.py > $OUT/trace.log 2>&1 )
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE"; do
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
lines = []
for k, v in sorted(per.items()):
    lines.append(f"{k} = {sum(v)/len(v):.5g} (n={len(v)})")
m = {k: sum(v) / len(v) for k, v in per.items()}
if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
    # GRBM_GUI_ACTIVE: cycles of the launch (one count per XCD -> / 8); MFMA busy summed over 1024 SIMDs
    lines.append(f"matrix pipe busy = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8 * 1024):.4f} of the kernel's cycles (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs))")
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_env_split" in row["Name"] or "k_shade" in row["Name"] or "k_geo" in row["Name"]:
            lines.append(f"{row['Name'][:70]} calls {row['Calls']} avg {float(row['AverageNs'])/1e6:.3f} ms")
open("$OUT/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
