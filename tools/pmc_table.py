#!/usr/bin/env python3
"""Format tools/pmc_layers.sh's raw counter lines (gpurun_out/evidence/pmc_raw.txt) as the per-layer table of profiles/rNN_pmc_layers.txt.
Usage: pmc_table.py <pmc_raw.txt>"""
import collections
import re
import sys


def main():
    rows = collections.OrderedDict()
    for ln in open(sys.argv[1]):
        m = re.match(r"(\S+)\s+(.*)", ln.strip())
        if not m or "=" not in m.group(2):
            continue
        d = rows.setdefault(m.group(1), {})
        for kv in m.group(2).split():
            k, v = kv.split("=")
            d[k] = float(v)
    print("rocprofv3 --kernel-trace --pmc <set> -- python3 tools/bench_conv.py --batch 700 --iters 4 --only <layer>   (tools/pmc_layers.sh: three passes per layer,")
    print("counters only with --kernel-trace; last dispatch of each pass; sums over the chip; bf16, batch 700; formatted by tools/pmc_table.py)\n")
    for layer, d in rows.items():
        try:
            busy = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
            wc = d["SQ_WAVE_CYCLES"]
            print(layer)
            print("  MFMA pipe busy             %.1f %%   = SQ_VALU_MFMA_BUSY_CYCLES %.0f / (GRBM_GUI_ACTIVE %.0f / 8 XCDs x 1024 SIMDs)"
                  % (100 * busy, d["SQ_VALU_MFMA_BUSY_CYCLES"], d["GRBM_GUI_ACTIVE"]))
            print("  wave cycles: issuing %.0f %%, parked at s_waitcnt / s_barrier (SQ_WAIT_ANY) %.0f %%, issue-stalled (SQ_WAIT_INST_ANY) %.0f %%"
                  % (100 * d["SQ_ACTIVE_INST_ANY"] / wc, 100 * d["SQ_WAIT_ANY"] / wc, 100 * d["SQ_WAIT_INST_ANY"] / wc))
            print("  instructions: VALU %.1f M, SALU %.1f M, LDS %.1f M" % (d["SQ_INSTS_VALU"] / 1e6, d["SQ_INSTS_SALU"] / 1e6, d["SQ_INSTS_LDS"] / 1e6))
            print("  LDS: bank-conflict cycles %.1f M of %.1f M LDS-array cycles (SQ_LDS_IDX_ACTIVE) = %.1f %%   [SQ_ACTIVE_INST_LDS %.1f M counts QUAD-cycles]\n"
                  % (d["SQ_LDS_BANK_CONFLICT"] / 1e6, d["SQ_LDS_IDX_ACTIVE"] / 1e6, 100 * d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"],
                     d["SQ_ACTIVE_INST_LDS"] / 1e6))
        except KeyError as e:
            print(layer, "incomplete counters:", e)


if __name__ == "__main__":
    main()
