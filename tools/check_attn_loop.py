#!/usr/bin/env python3
"""Build-time check of attention_p.hip's ISA: inside the tile loops of every attn64q / attn64p instantiation there must be NO scratch
operation and NO compiler-inserted `s_waitcnt vmcnt` (only the counted waits of the asm statements).  Either one drains the
LDS-DMA pipeline once per tile (a spilled register's reload counts on vmcnt with the DMA loads; a pending compiler-visible
load in front of the loop leaves its wait inside the loop): measured 145 / 165 us against 102 / 110.
    python tools/check_attn_loop.py          (compiles to /tmp; exit code 1 on a finding)"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/check_attn_loop.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", root + "/include", "-I",
                       root + "/ditto_tts_amd/csrc", "-w", "-fno-honor-nans", "-fno-slp-vectorize", "-S", "--cuda-device-only",
                       root + "/ditto_tts_amd/csrc/attention_p.hip", "-o", out])
s = open(out).read()
bad = 0
for name in re.findall(r"^(_ZN\S*attn64[pq]_kernel\S+):", s, re.M):
    i = s.index(name + ":"); j = s.index(".end_amdhsa_kernel", i)
    blocks = re.split(r"\n(\.LBB\d+_\d+):", s[i:j])
    for k in range(1, len(blocks), 2):
        b = blocks[k + 1]
        n = b.count("v_mfma")
        q = "attn64q" in name
        if q and (n not in (4, 28) or not b.count("v_exp") or b.count("v_exp") > 100):      # attn64q: the steady loop's two blocks (28 + 4 MFMAs)
            continue
        if not q and (n != 16 or "Loop" not in b.split("\n")[0] or "Li0EEEv" not in name):   # attn64p: the 16-MFMA blocks of its tile loops (product instantiations)
            continue
        in_asm, waits = False, 0
        for l in b.split("\n"):
            if "#ASMSTART" in l: in_asm = True
            elif "#ASMEND" in l: in_asm = False
            elif "vmcnt" in l and not in_asm: waits += 1
        scratch = b.count("scratch_")
        flag = "" if not (waits or scratch) else "   <-- PROBLEM"
        bad += bool(waits or scratch)
        print(f"{name[-44:]:46s} {blocks[k]:10s} mfma {n:2d}  scratch {scratch}  compiler vmcnt waits {waits}{flag}")
sys.exit(1 if bad else 0)
