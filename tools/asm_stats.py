#!/usr/bin/env python3
"""Instruction-mix summary per kernel of a hipcc -S --cuda-device-only .s file."""
import re, sys
s = open(sys.argv[1]).read()
labels = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):\s*;", s, flags=re.M)]
for i, (pos, name) in enumerate(labels):
    end = s.find("s_endpgm", pos)
    body = s[pos:end]
    c = lambda pat: len(re.findall(pat, body))
    short = re.sub(r"_ZN6subreg\d+", "", name)[:60]
    print("%-60s mfma %4d dsr128 %4d dsr_other %3d dsw %3d gload %3d gstore %3d flat %d scratch %d barrier %3d waitcnt %4d valu~ %5d lines %6d" % (
        short, c(r"v_mfma"), c(r"ds_read_b128"), c(r"ds_read_(?!b128)"), c(r"ds_write"), c(r"global_load"), c(r"global_store"),
        c(r"flat_"), c(r"scratch_"), c(r"s_barrier"), c(r"s_waitcnt"), c(r"\n\s+v_(?!mfma)"), body.count("\n")))
