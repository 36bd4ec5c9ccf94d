#!/usr/bin/env python3
"""Compile the C-ABI translation unit with -Rpass-analysis=kernel-resource-usage and print one line per kernel
(registers, spills, scratch, occupancy).  Usage: python tools/kernel_resources.py [name-filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
       "-Wno-unused-value", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "mini_amd/csrc/mgx_capi.hip"),
       "-o", "/tmp/kres.so", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line) or re.search(r" Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem)
    if flt and flt not in dem: continue
    print("%-70s vgpr %3d agpr %3d spill(v) %3d spill(s) %3d scratch %4d sgpr %3d occ %d" % (
        dem[:70], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("VGPRs Spill", -1), r.get("SGPRs Spill", -1),
        r.get("ScratchSize", -1), r.get("TotalSGPRs", -1), r.get("Occupancy", -1)))
