cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pm; rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm -o pm -- $GRAFT_REPO_ROOT/tools/probe/lds_atomic_probe > /dev/null 2>&1
python - <<PY
import csv, glob, collections
rows = []
for f in glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append(r)
by = collections.OrderedDict()
for r in rows: by.setdefault((r["Dispatch_Id"], r["Kernel_Name"][:40]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in list(by.items())[24:38]: print(k[0], "IDX_ACTIVE %.4g INSTS %.4g GRBM %.4g BANKC %.3g" % (v.get("SQ_LDS_IDX_ACTIVE", -1), v.get("SQ_INSTS_LDS", -1), v.get("GRBM_GUI_ACTIVE", -1), v.get("SQ_LDS_BANK_CONFLICT", -1)))
PY
rm -rf /tmp/pm; rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm -o pm -- python $GRAFT_REPO_ROOT/tools/probe/scatter_only.py > /dev/null 2>&1
python - <<PY
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "scatter_lds" in r["Kernel_Name"]: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in d.items(): print(k, "%.4g" % (sum(v) / len(v)), len(v))
PY
