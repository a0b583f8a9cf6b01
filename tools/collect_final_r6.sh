#!/bin/bash
# In the build container, after `gpurun -- 'bash tools/gpu_final_r6.sh <tag>'` has merged its outputs into gpurun_out/: copy what is to be
# judged into profiles/<tag>/ (tracked), refresh profiles/pmc_latest.json (bench.py reads roofline.traffic from it when its csrc_sha matches
# the kernel sources), and commit.     tools/collect_final_r6.sh <tag>
cd "$(dirname "$0")/.."
TAG=${1:-r06b}
G=gpurun_out
[ -d $G/$TAG/profile ] || { echo "no $G/$TAG/profile: the GPU run has not produced it"; exit 1; }
mkdir -p profiles/$TAG
cp $G/$TAG/profile/* profiles/$TAG/
[ -f $G/$TAG/summary.json ] && cp $G/$TAG/summary.json profiles/pmc_latest.json
[ -f $G/$TAG/pytest_gpu_all.log ] && tail -15 $G/$TAG/pytest_gpu_all.log > profiles/$TAG/pytest_gpu_all_tail.txt
for f in asan_driver.txt asan_summary.txt stress_sync.txt asan_build.log; do [ -f $G/${TAG}_asan/$f ] && cp $G/${TAG}_asan/$f profiles/$TAG/; done
for f in $G/${TAG}_asan/asan_report*; do [ -f "$f" ] && head -200 "$f" > profiles/$TAG/$(basename $f).txt; done
for f in split_time.txt summary.txt pytest_split.log; do [ -f $G/${TAG}_split/$f ] && cp $G/${TAG}_split/$f profiles/$TAG/split2_$f; done
ls profiles/$TAG | wc -l
git add profiles/$TAG profiles/pmc_latest.json
git commit -qm "profiles/$TAG: the round's final tree on the GPU (tests, smoke, bench, kernel stats, PMC passes -> pmc_latest.json, sanitizer driver, split-precision counters)" && echo committed
