#!/usr/bin/env python3
"""Where is the GPU idle?  Reads a rocprofv3 --kernel-trace CSV, takes the union of all kernel intervals and lists the idle gaps between
them: total by size class, and the largest ones with the kernels on either side.  Usage: prof_gaps.py <kernel_trace.csv> [--top N]
[--skip-ms T] (drop the first T ms of the trace: warm-up) [--around N W] (the launches within W ms of the N-th largest gap)."""
import csv
import sys
from collections import defaultdict

from prof_summary import short


def main():
    path = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
    skip = float(sys.argv[sys.argv.index("--skip-ms") + 1]) if "--skip-ms" in sys.argv else 0.0
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(path))))
    t0 = rows[0][0] + int(skip * 1e6)
    rows = [r for r in rows if r[0] >= t0]
    cur_e, last_name = rows[0][1], rows[0][2]
    gaps = []                       # (length ns, offset ns, kernel before, kernel after)
    busy = rows[0][1] - rows[0][0]
    cur_s = rows[0][0]
    for s, e, name in rows[1:]:
        if s > cur_e:
            gaps.append((s - cur_e, cur_e - rows[0][0], last_name, name))
            cur_s = s
        if e > cur_e:
            busy += e - max(cur_e, s)
            cur_e, last_name = e, name
    span = cur_e - rows[0][0]
    print("span %.1f ms, busy (union) %.1f ms, idle %.1f ms (%.2f %%), %d launches, %d gaps" %
          (span * 1e-6, busy * 1e-6, (span - busy) * 1e-6, 100.0 * (span - busy) / span, len(rows), len(gaps)))
    classes = [(2e3, "< 2 us"), (1e4, "2-10 us"), (5e4, "10-50 us"), (2e5, "50-200 us"), (1e6, "0.2-1 ms"), (1e7, "1-10 ms"), (1e18, "> 10 ms")]
    tot = defaultdict(lambda: [0, 0])
    for g in gaps:
        for lim, label in classes:
            if g[0] < lim:
                tot[label][0] += 1
                tot[label][1] += g[0]
                break
    for _, label in classes:
        n, t = tot[label]
        print("  gaps %-10s %7d   %9.2f ms   %5.2f %% of the span" % (label, n, t * 1e-6, 100.0 * t / span))
    # gaps by the pair of kernels around them
    pairs = defaultdict(lambda: [0, 0])
    for g in gaps:
        k = (g[2][:48], g[3][:48])
        pairs[k][0] += 1
        pairs[k][1] += g[0]
    print("idle time by (kernel before -> kernel after):")
    for k, (n, t) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:top]:
        print("  %9.2f ms  %6d x %8.1f us   %s  ->  %s" % (t * 1e-6, n, t * 1e-3 / n, k[0], k[1]))
    print("largest gaps:")
    for g in sorted(gaps, reverse=True)[:top]:
        print("  %9.1f us at %9.1f ms   %s  ->  %s" % (g[0] * 1e-3, g[1] * 1e-6, g[2][:48], g[3][:48]))
    if "--around" in sys.argv:
        # the launches within `w` ms before / after the end of the n-th largest gap
        n, w = int(sys.argv[sys.argv.index("--around") + 1]), float(sys.argv[sys.argv.index("--around") + 2])
        g = sorted(gaps, reverse=True)[n]
        t_gap = rows[0][0] + g[1]
        print("launches within %.1f ms of the start of gap %d (%.1f us at %.1f ms):" % (w, n, g[0] * 1e-3, g[1] * 1e-6))
        prev_e = None
        for s_, e_, name in rows:
            if s_ < t_gap - w * 1e6 or s_ > t_gap + g[0] + w * 1e6:
                continue
            print("  %10.1f us  +%8.1f us  idle before %8.1f   %s" % ((s_ - t_gap) * 1e-3, (e_ - s_) * 1e-3,
                  (s_ - prev_e) * 1e-3 if prev_e is not None and s_ > prev_e else 0.0, name[:80]))
            prev_e = max(prev_e or 0, e_)


if __name__ == "__main__":
    main()
