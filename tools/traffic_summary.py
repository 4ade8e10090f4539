#!/usr/bin/env python3
"""HBM traffic of the conv stack from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/bench_conv.py.

  traffic_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <batch> [dtype]
Per MI355X_MICROARCH.md (HBM / rocprofv3): counters are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide
coalesced stream => doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  Dispatches are grouped by (kernel, grid) =
one group per layer of bench_conv.LAYERS (in launch order); groups with count 2 in the stack are weighted accordingly.
Prints per-layer bytes and writes profiles/traffic.json {dtype: bytes per image}."""
import csv
import json
import os
import sys
from collections import OrderedDict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
from bench_conv import LAYERS  # noqa: E402


def is_conv_kernel(name):
    """every kernel a bench_conv.py layer call launches: conv_fwd_kernel, conv_wide_kernel, the layer-1 kernels (conv_first_kernel,
    conv64_resident_kernel, conv64_pool_img_kernel, conv64_fused_first_kernel, conv64_wide_kernel, ...) - anything named conv*_kernel"""
    return "conv" in name and "kernel" in name and "splitk_reduce" not in name


def per_dispatch(path, counter):
    d, names = OrderedDict(), {}
    for r in csv.DictReader(open(path)):
        if not is_conv_kernel(r["Kernel_Name"]) or r["Counter_Name"] != counter:
            continue
        i = int(r["Dispatch_Id"])
        d[i] = d.get(i, 0.0) + float(r["Counter_Value"])
        names[i] = r["Kernel_Name"].split("<")[0].split("(")[0].replace("void ", "").replace("subreg::", "")
    ids = sorted(d)
    return [d[k] for k in ids], [names[k] for k in ids]


def git_head():
    try:
        import subprocess
        return subprocess.run(["git", "-C", REPO, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception:
        return None


def main():
    fpath, wpath, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
    dtype = sys.argv[4] if len(sys.argv) > 4 else "bf16"
    (f, fn), (w, wn) = per_dispatch(fpath, "FETCH_SIZE"), per_dispatch(wpath, "WRITE_SIZE")
    if len(f) != len(w) or len(f) % len(LAYERS) != 0:
        # a kernel this script does not know about (or a layer that takes two launches): say which, do not write anything
        from collections import Counter
        print("traffic_summary: %d FETCH / %d WRITE dispatches for %d layers - kernels seen: %s" % (len(f), len(w), len(LAYERS), dict(Counter(fn))))
        sys.exit(2)
    per = len(f) // len(LAYERS)               # bench_conv launches every layer (3 warm-up + iters) times, in order
    total = 0.0
    print("%-26s %12s %12s   %s" % ("layer", "read MB", "write MB", "kernel"))
    for li, (name, *_rest, count) in enumerate(LAYERS):
        rd = 2.0 * 1024 * sum(f[li * per:(li + 1) * per]) / per
        wr = 1024 * sum(w[li * per:(li + 1) * per]) / per
        total += (rd + wr) * count
        print("%-26s %12.1f %12.1f   x%d  %s" % (name, rd / 1e6, wr / 1e6, count, fn[li * per]))
    per_img = total / batch
    print("conv stack HBM traffic: %.1f MB per forward of %d images = %.2f MB/image (algorithmic minimum 12.79 MB bf16)" % (total / 1e6, batch, per_img / 1e6))
    out = os.path.join(REPO, "profiles", "traffic.json")
    cur = json.load(open(out)) if os.path.exists(out) else {}
    cur[dtype] = {"bytes_per_image": per_img, "batch": batch, "head": os.environ.get("SUBREG_EVIDENCE_HEAD") or git_head(),
                  "note": "sum over the 22 conv launches of one B=%d forward; FETCH_SIZE doubled per MI355X_MICROARCH.md" % batch}
    json.dump(cur, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
