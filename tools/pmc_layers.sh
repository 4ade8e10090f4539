cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for L in L3.0.conv2 L4.1.conv3 L1.conv1+conv2 L2.conv2; do
  for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE"; do
    rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pm -o p -- python3 $R/tools/bench_conv.py --batch 700 --iters 4 --only $L > /dev/null 2>&1
    f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$L" <<'PY'
import csv, sys, collections
f, layer = sys.argv[1], sys.argv[2]
rows = [r for r in csv.DictReader(open(f)) if "conv" in r["Kernel_Name"]]
last = max(int(r["Dispatch_Id"]) for r in rows)
acc = collections.OrderedDict()
for r in rows:
    if int(r["Dispatch_Id"]) == last:
        acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print(layer, " ".join("%s=%.0f" % kv for kv in acc.items()))
PY
  done
done
