#!/bin/bash
# usage: tools/gpu_prof.sh <tag>   (runs on the GPU box via gpurun)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python tools/prof_frame.py 3 > $OUT/trace.log 2>&1
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" ; do
  N=$(echo $SET | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmc_$N -o pmc -- python tools/prof_frame.py 2 > $OUT/pmc_$N.log 2>&1
done
find $OUT -name "*.csv" | head -50
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        if 'render_persistent' in row.get('Kernel_Name',''):
            agg[row['Counter_Name']] += float(row['Counter_Value']); cnt[row['Counter_Name']] += 1
    for k,v in agg.items(): print(f"{k:32s} total {v:.4e} per-dispatch {v/max(cnt[k],1):.4e} (n={cnt[k]})")
for f in sorted(glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)):
    print(open(f).read()[:3000])
PY
