#!/bin/bash
# PMC of the per-ray kernels of a COLD frame (through gpurun): tools/gpu_pmc_rays.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  N=$(echo $SET | tr ' ' '_' | cut -c1-24)
  ( cd /tmp && rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_$N -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --headline-only --cold > $OUT/pmc_$N.log 2>&1 )
done
python - <<PY
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "k_geo_rays" in n or "k_geo_eval" in n:
            key = ("rays<first>" if "<true" in n else "rays<later>") if "k_geo_rays" in n else "eval"
            per[(key, row["Dispatch_Id"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
# group by kernel kind in dispatch order: print the first few dispatches of each
seen = collections.defaultdict(int)
for (key, did), d in sorted(per.items(), key=lambda kv: int(kv[0][1])):
    seen[key] += 1
    if seen[key] <= 4:
        print(key, did, {c: sum(v) for c, v in d.items()})
PY
