#!/bin/bash
# Which device kernels of libenvidr_amd.so does the GPU test suite never launch?  (run through gpurun)
#   rocprofv3 --kernel-trace over `pytest -m gpu`, then the set difference against the .kd symbols of the library.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/coverage
mkdir -p $OUT
( cd /tmp && rm -rf /tmp/cov && rocprofv3 --kernel-trace --output-format csv -d /tmp/cov -o cov -- python -m pytest $GRAFT_REPO_ROOT/tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1 )
tail -2 $OUT/pytest.log
python - <<'PY'
import csv, glob, os, re, subprocess
root = os.environ["GRAFT_REPO_ROOT"]
out = root + "/gpurun_out/coverage"
def norm(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n.strip()).replace("_Float16", "half").replace("__half", "half")
    depth, cut = 0, len(n)
    for i, ch in enumerate(n):          # cut the argument list: the first '(' outside template brackets
        if ch == "<": depth += 1
        elif ch == ">": depth -= 1
        elif ch == "(" and depth == 0: cut = i; break
    return n[:cut].replace(" ", "")
launched = set()
demangled = {}
for f in glob.glob("/tmp/cov/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if n.startswith("_Z"):          # (the tracer's demangler does not know _Float16's DF16_ either)
            if n not in demangled:
                demangled[n] = subprocess.run(["c++filt", n.split()[0].replace("DF16_", "Dh")], capture_output=True, text=True).stdout
            n = demangled[n]
        launched.add(norm(n))
syms = subprocess.run("strings -n 8 %s/envidr_amd/libenvidr_amd.so | grep '\\.kd$' | sort -u" % root, shell=True, capture_output=True, text=True).stdout.split("\n")
names = set()
for s in syms:
    s = s.strip()
    i = s.find("_Z")
    if i < 0:
        if s.endswith(".kd") and s[:-3].isidentifier(): names.add(s[:-3])
        continue
    d = subprocess.run(["c++filt", s[i:-3].replace("DF16_", "Dh")], capture_output=True, text=True).stdout          # (binutils' c++filt predates _Float16's DF16_)
    names.add(norm(d))
mine = sorted(n for n in names if n)
never = [n for n in mine if n not in launched]
open(out + "/never_launched.txt", "w").write("\n".join(never) + "\n")
print(len(mine), "kernels in the library,", len(mine) - len(never), "launched by the GPU tests,", len(never), "never launched")
fam = {}
for n in never: fam.setdefault(n.split("<")[0], []).append(n)
for k, v in sorted(fam.items()): print(" ", k, len(v), "e.g.", v[0][:110])
PY
