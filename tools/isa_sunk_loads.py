#!/usr/bin/env python3
"""Loads the code generator has sunk under a branch, per kernel of the product library's device code.

`x = cond ? load(p) : y` with no other use of the load becomes `if (cond) { load; s_waitcnt vmcnt(0) }` in the binary: hipcc's
s_waitcnt insertion is static, so the wait inside the branch drains EVERY load in flight (the prefetched stream included) and every
such gather is a round trip of its own.  Round 5 found 70 of these in k_nr_edges and 97 in k_bfs_push<false, 0> (the epilogue's
few-marks path) although the sources said "unconditional load"; a relaxed atomic load of wavefront scope (__hip_atomic_load) is the
same instruction and stays where it is written (mgx/nreduce.hpp: nr_load_pinned).

The signature counted here: s_and_saveexec ... {global_load | ds_read} ... s_waitcnt vmcnt(0) ... s_or exec, within 16 instructions.
Legitimate sites exist (a rare path that really is conditional: a cold probe, block 0 opening a level); the count is a lead, not a verdict.

usage: python tools/isa_sunk_loads.py [substring of the demangled kernel name ...]      (build container: compiles the device code to assembly, ~20 s)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def device_asm(path=None):
    out = path or os.path.join(tempfile.mkdtemp(prefix="mgx_isa_"), "capi.s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-value", "-I" + os.path.join(ROOT, "include"),
           "-S", "--cuda-device-only", "-o", out, os.path.join(ROOT, "mini_amd", "csrc", "mgx_capi.hip")]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def sunk_loads(asm_path):
    """{mangled kernel name: sites}"""
    body, cur = {}, None
    for line in open(asm_path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1); body[cur] = []; continue
        if cur is None:
            continue
        t = line.strip()
        if not t or t.startswith(";") or (t.startswith(".") and not t.startswith(".LBB")):
            continue
        body[cur].append(t)
    res = {}
    for k, ins in body.items():
        c, i = 0, 0
        while i < len(ins):
            if ins[i].startswith("s_and_saveexec"):
                j, has_load, hit = i + 1, False, False
                while j < len(ins) and j < i + 16:
                    if ins[j].startswith("global_load") or ins[j].startswith("ds_read"):
                        has_load = True
                    if has_load and "s_waitcnt vmcnt(0)" in ins[j]:
                        hit = True
                    if ins[j].startswith("s_or_b64 exec"):
                        break
                    j += 1
                c += hit
                i = j
            else:
                i += 1
        res[k] = c
    return res


def demangle(names):
    out = subprocess.run(["c++filt"] + list(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


if __name__ == "__main__":
    res = sunk_loads(device_asm())
    names = demangle(list(res))
    pats = sys.argv[1:]
    for k, c in sorted(res.items(), key=lambda kc: -kc[1]):
        d = names.get(k, k)
        if (pats and any(p in d for p in pats)) or (not pats and c > 0):
            print("%4d  %s" % (c, d[:150]))
