export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05l; mkdir -p $OUT
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train -o t -- python $GRAFT_REPO_ROOT/tools/train_step_bench.py 20 > $OUT/trace_train.log 2>&1 )
find $OUT/trace_train -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_train_step.csv \;
rm -rf $OUT/trace_train
