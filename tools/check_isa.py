#!/usr/bin/env python3
"""Build-time ISA check of libsubreg_hip (run by subspace-reg_amd/Makefile on the .s files `hipcc -save-temps=obj` leaves in build/).

The LDS-DMA helpers (csrc/subreg_common.h::dma16 and friends) write M0 inside an asm statement and declare it clobbered instead of
saving / restoring it.  That is only sound while the COMPILER never keeps a value of its own in M0, so every kernel's ISA is checked:
  * `m0` may appear only inside ;;#ASMSTART ... ;;#ASMEND blocks,
  * no implicit M0 users anywhere: s_set_gpr_idx_*, v_movrel*, ds_*addtid*, s_sendmsg with an M0 payload is not used by this library.
Exit status 1 (and the offending lines) on a violation, so a compiler bump or a code change that breaks the assumption stops the build.
"""
import glob
import os
import re
import sys

IMPLICIT = re.compile(r"\b(s_set_gpr_idx|v_movrel|ds_read_addtid|ds_write_addtid|s_movrel)")


def check(path):
    bad = []
    in_asm = False
    for n, line in enumerate(open(path, errors="replace"), 1):
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        code = t.split(";", 1)[0]
        if not code or code.startswith("."):
            continue
        if IMPLICIT.search(code):
            bad.append((n, t))
        elif not in_asm and re.search(r"\bm0\b", code):
            bad.append((n, t))
    return bad


def main():
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "subspace-reg_amd", "build")
    files = sorted(glob.glob(os.path.join(d, "*-hip-amdgcn-amd-amdhsa-gfx950.s")))
    if not files:
        print("check_isa: no device .s files in %s (build with -save-temps=obj)" % d)
        return 1
    rc = 0
    # every kernel source named on the command line (the Makefile passes csrc/*.hip) must have its ISA here, and that ISA must be
    # at least as new as the source: a build directory from before -save-temps, or an incremental rebuild of one file, must not pass
    # on the strength of the files that happen to exist
    for src in sys.argv[2:]:
        stem = os.path.splitext(os.path.basename(src))[0]
        isa = os.path.join(d, stem + "-hip-amdgcn-amd-amdhsa-gfx950.s")
        if not os.path.exists(isa):
            rc = 1
            print("check_isa: no ISA for %s in %s (object built without -save-temps=obj?)" % (src, d))
        elif os.path.getmtime(isa) + 1.0 < os.path.getmtime(src):
            rc = 1
            print("check_isa: %s is older than %s" % (isa, src))
    if rc:
        return rc
    for f in files:
        bad = check(f)
        if bad:
            rc = 1
            print("check_isa: %s: %d line(s) touch M0 outside the LDS-DMA statements" % (os.path.basename(f), len(bad)))
            for n, t in bad[:8]:
                print("   %d: %s" % (n, t))
    if rc == 0:
        print("check_isa: %d files, M0 only inside the LDS-DMA statements, no implicit M0 users" % len(files))
    return rc


if __name__ == "__main__":
    sys.exit(main())
