#!/bin/bash
# quick GPU pass (through gpurun): the tests named on the command line, then short bench runs
#   tools/gpu_quick.sh <tag> [pytest args...]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
timeout 1500 python -m pytest "$@" -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/bench.err
timeout 300 python bench.py --headline-only --hinted --steps 10 --warmup 2 > $OUT/bench_hinted.json 2> $OUT/bench_hinted.err
if [ -f tools/geo/variants/no_chunk_prediction.so ]; then
  ENVIDR_AMD_LIB=$GRAFT_REPO_ROOT/tools/geo/variants/no_chunk_prediction.so timeout 300 python bench.py --headline-only --cold --steps 10 --warmup 2 > $OUT/bench_cold_nopred.json 2> $OUT/bench_cold_nopred.err
fi
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cold -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --headline-only --cold > $OUT/trace_cold.log 2>&1 )
tail -15 $OUT/pytest.log
python - <<PY
import json
for f in ("bench", "bench_hinted", "bench_cold_nopred"):
    try:
        j = json.loads(open("$OUT/" + f + ".json").read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, "value", j["value"], "ms", j["ms_per_step"], "frame", j.get("frame", {}).get("geometry_ms"), j.get("frame", {}).get("shading_ms"),
          "samples", j["config"]["samples_per_frame"], j["config"]["samples_evaluated_per_frame"])
    for k in ("video_fixed_camera", "cold_frame", "moving_camera", "survey_density_scene"):
        if k in j: print("  ", k, {a: b for a, b in j[k].items() if a != "note"})
PY
tail -3 $OUT/bench.err
