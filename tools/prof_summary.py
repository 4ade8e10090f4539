#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel (short name) and, for the conv kernels, per launch geometry
(= per layer shape).  Usage: prof_summary.py <kernel_trace.csv> [--top N] [--conv] [--images N] [--timeline N]
(--timeline N: the last N launches in start order with duration, gap to the previous kernel's end and grid;
--busy: the union of all kernel intervals beside the sum of their durations)"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void\s+", "", name)
    m = re.match(r"_ZN6subreg(\d+)([a-z_0-9]+)", name)
    if m:
        n = int(m.group(1))
        tail = name[len("_ZN6subreg") + len(m.group(1)) + n:]
        t = re.findall(r"Li(\d+)E|Lb([01])E|(DF16b|f(?=Li))", tail[:60])
        return m.group(2)[:n] + "<" + ",".join(a or b or c for a, b, c in t) + ">"
    name = re.sub(r"subreg::", "", name)
    name = re.sub(r"\(.*", "", name)
    return name[:70]


def main():
    path = sys.argv[1]
    if "--timeline" in sys.argv:
        n = int(sys.argv[sys.argv.index("--timeline") + 1])
        rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
        prev, t0, busy = None, int(rows[0]["Start_Timestamp"]), 0.0
        for r in rows:
            s0, e0 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            gap = (s0 - prev) * 1e-3 if prev is not None else 0.0
            busy += (e0 - s0) * 1e-3
            q = r.get("Stream_Id") or r.get("Queue_Id") or "?"
            print("%9.1f us  +%7.1f us  gap %6.1f  q%-3s grid %6d x%-3d  %s" % ((s0 - t0) * 1e-3, (e0 - s0) * 1e-3, gap, q,
                  int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), short(r["Kernel_Name"])[:70]))
            prev = max(prev or 0, e0)
        print("span %.1f us, kernel time %.1f us" % ((prev - t0) * 1e-3, busy))
        if "--step" in sys.argv:
            # one training step = from one `--step <kernel substring>` launch to the next: span, union of busy intervals, idle gaps
            key = sys.argv[sys.argv.index("--step") + 1]
            marks = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
            if len(marks) >= 2:
                seg = rows[marks[-2]:marks[-1]]
                iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
                cur_s, cur_e, union, gaps = iv[0][0], iv[0][1], 0, []
                for a, b in iv[1:]:
                    if a > cur_e:
                        union += cur_e - cur_s
                        gaps.append((a - cur_e, cur_e - iv[0][0]))
                        cur_s, cur_e = a, b
                    else:
                        cur_e = max(cur_e, b)
                union += cur_e - cur_s
                span = int(rows[marks[-1]]["Start_Timestamp"]) - iv[0][0]
                print("step: %d launches, span %.1f us, busy (union) %.1f us, idle %.1f us; sum of kernel times %.1f us" %
                      (len(seg), span * 1e-3, union * 1e-3, (span - union) * 1e-3, sum(b - a for a, b in iv) * 1e-3))
                print("largest idle gaps (us @ offset): " + ", ".join("%.1f@%.0f" % (g * 1e-3, o * 1e-3) for g, o in sorted(gaps, reverse=True)[:12]))
        return
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
    per, geo = defaultdict(lambda: [0, 0.0]), defaultdict(lambda: [0, 0.0])
    total = 0.0
    for r in csv.DictReader(open(path)):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        n = short(r["Kernel_Name"])
        per[n][0] += 1
        per[n][1] += d
        total += d
        if "conv_fwd" in n:
            wg = int(r["Workgroup_Size_X"])
            k = (n, int(r["Grid_Size_X"]) // wg, int(r["Grid_Size_Y"]), r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"])
            geo[k][0] += 1
            geo[k][1] += d
    print("total kernel time %.1f ms" % (total * 1e-3))
    for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%-64s calls %6d  total %9.1f us  avg %9.1f us  %5.1f%%" % (n[:64], c, t, t / c, 100 * t / total))
    if "--busy" in sys.argv:
        # union of the busy intervals of the whole trace (kernels of concurrent lanes / streams overlap: the sum of their durations
        # exceeds the wall time, the union cannot) - what a two-lane trace is compared with the bench line's own wall time by
        iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(path)))
        cur_s, cur_e, union = iv[0][0], iv[0][1], 0
        for a, b in iv[1:]:
            if a > cur_e:
                union += cur_e - cur_s
                cur_s, cur_e = a, b
            else:
                cur_e = max(cur_e, b)
        union += cur_e - cur_s
        print("busy (union of all kernel intervals) %.1f ms of a %.1f ms trace span; sum of kernel durations %.1f ms (overlap factor %.3f)"
              % (union * 1e-6, (iv[-1][1] - iv[0][0]) * 1e-6, total * 1e-3, total * 1e3 / max(union, 1)))
    if "--images" in sys.argv:
        # roofline.frac of bench.py from the trace alone: the forward's kernels (conv family + avgpool) over the images forwarded
        # in the traced process (warm-up included).  Only meaningful for a ONE-lane run (SUBREG_EVAL_LANES=1): two lanes overlap.
        n_img = float(sys.argv[sys.argv.index("--images") + 1])
        fwd = sum(t for n, (c, t) in per.items() if n.startswith(("conv", "avgpool")))
        tf = n_img * 8.1219e9 / (fwd * 1e-6) / 1e12
        print("forward kernels (conv*, avgpool): %.1f ms over %.0f images -> %.1f TFLOP/s algorithmic = %.4f of the 2500 TFLOP/s bf16 peak"
              % (fwd * 1e-3, n_img, tf, tf / 2500.0))
    if "--conv" in sys.argv:
        print("\nconv launches by geometry (grid_m x grid_n, vgpr, agpr, lds):")
        for k, (c, t) in sorted(geo.items(), key=lambda kv: -kv[1][1])[:40]:
            print("%-44s grid %5dx%d v%s a%s lds%s calls %5d avg %9.1f us total %9.1f us" % (k[0][:44], k[1], k[2], k[3], k[4], k[5], c, t / c, t))


if __name__ == "__main__":
    main()
