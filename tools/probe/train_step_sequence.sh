export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/seq; mkdir -p $OUT
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/tools/train_step_bench.py 2 > $OUT/trace.log 2>&1 )
find $OUT/trace -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
rm -rf $OUT/trace
python - <<'PY'
import csv, re, os
rows = list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/seq/kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last full step: between the last two march_rays_train launches
idx = [i for i, n in enumerate(names) if "k_march_rays_train" in n]
a, b = idx[-2], idx[-1]
def short(n):
    n = re.sub(r"at::native::|\(anonymous namespace\)::|envidr::|void ", "", n)
    n = re.sub(r"vectorized_elementwise_kernel<4, |elementwise_kernel_manual_unroll<128, 4, gpu_kernel_impl_nocast<|elementwise_kernel_manual_unroll<128, 4, gpu_kernel_impl<", "EW:", n)
    return n[:70]
with open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/seq/sequence.txt", "w") as f:
    for r in rows[a:b]:
        f.write(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  {short(r['Kernel_Name'])}\n")
print(b - a, "kernels in the step")
PY
