cd /tmp; export TMPDIR=/tmp
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pm; rocprofv3 --pmc $SET --output-format csv -d /tmp/pm -o pm -- python $GRAFT_REPO_ROOT/tools/probe/scatter_only.py > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "scatter_lds" in r["Kernel_Name"]: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in d.items(): print(k, "%.4g" % (sum(v) / len(v)), len(v))
PY
done
