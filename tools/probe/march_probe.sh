mkdir -p gpurun_out/march_probe
for v in 0 1 2 3 4 5; do echo "variant $v: $(ENVIDR_MARCH_VARIANT=$v python tools/probe/march_train_probe.py $([ $v = 0 ] && echo ref) 2>&1 | tail -1)"; done | tee gpurun_out/march_probe/out.txt
