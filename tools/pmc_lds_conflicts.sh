#!/bin/bash
# Where do the LDS bank-conflict cycles of the conv loop come from?  SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS of L2.conv2 (conv_wide_kernel,
# 256-row variant) as shipped and in the timing-only build WITHOUT in-loop staging (libsubreg_wd1.so: make variant NAME=wd1
# EXTRA=-DSUBREG_WIDE_DIAG=1; same fragment reads, same epilogue, no LDS-DMA writes inside the loop).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for LIB in libsubreg_hip.so libsubreg_wd1.so; do
  rm -rf /tmp/pmc_l; SUBREG_LIB=$R/subspace-reg_amd/subreg_hip/$LIB rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc_l -o p -- python3 $R/tools/bench_conv.py --batch 700 --iters 4 --only L2.conv2 > /dev/null 2>&1
  f=$(find /tmp/pmc_l -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$LIB" <<'PY'
import csv, sys, collections
f, lib = sys.argv[1], sys.argv[2]
rows = [r for r in csv.DictReader(open(f)) if "conv" in r["Kernel_Name"]]
last = max(int(r["Dispatch_Id"]) for r in rows)
acc = collections.OrderedDict()
for r in rows:
    if int(r["Dispatch_Id"]) == last:
        acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print("%-20s %s" % (lib, " ".join("%s=%.0f" % kv for kv in acc.items())), rows[-1]["Kernel_Name"][:40])
PY
done
