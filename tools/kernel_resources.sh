#!/bin/bash
# Print VGPR/AGPR/SGPR/occupancy/spill/LDS per kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
f=${1:?usage: kernel_resources.sh file.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import sys,re,subprocess
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r"Function Name: (\S+)",line)
    if m:
        cur={"name":subprocess.run(["/usr/bin/c++filt",m.group(1)],capture_output=True,text=True).stdout.strip()}; rows.append(cur); continue
    m=re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/block\]| \[waves/SIMD\])?: (\d+)",line)
    if m and cur is not None: cur[m.group(1).strip()]=int(m.group(2))
for r in rows:
    n=re.sub(r"subreg::","",r["name"]); n=re.sub(r"\(.*","",n)
    print("%-70s V=%3d A=%3d S=%3d occ=%d spill=%d/%d" % (n[:70], r.get("VGPRs",-1), r.get("AGPRs",-1), r.get("TotalSGPRs",-1), r.get("Occupancy",-1), r.get("VGPRs Spill",-1), r.get("SGPRs Spill",-1)))
'
# M0 check (csrc/subreg_common.h::dma16 declares m0 clobbered instead of saving / restoring it): every mention of m0 in the ISA
# must be one of the statements' own `s_mov_b32 m0, ...`.  Prints the offending lines, if any.
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only "$f" -o /tmp/_m0check.s 2>/dev/null &&
  { n=$(grep "m0" /tmp/_m0check.s | grep -v "s_mov_b32 m0, \|s_mov_b32 s[0-9]*, m0$\|s_mov_b32 vcc_[lohi]*, m0$" | wc -l); echo "m0 outside the LDS-DMA statements: $n line(s)"; grep -n "m0" /tmp/_m0check.s | grep -v "s_mov_b32 m0, \|s_mov_b32 s[0-9]*, m0$\|s_mov_b32 vcc_[lohi]*, m0$" | head -5; }
