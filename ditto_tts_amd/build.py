"""Build libditto_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m ditto_tts_amd.build [--force]

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libditto_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# per-file extras.  attention.hip: without -fno-honor-nans hipcc canonicalises every MFMA output (v_max x,x)
# before the row-max fmaxf chain (+32 VALU per KV tile); the softmax has no NaN semantics to preserve.
# attention_bwd.hip: the SLP vectoriser pairs the P / dS arithmetic into v_pk_add_f32 / v_pk_mul_f32, and packed fp32
# instructions occupy the matrix pipe (a lone wave showed zero MFMA / vector overlap: removing the 24 MFMAs of a tile saved
# exactly 24 x 32 cycles); scalar v_sub / v_mul issue beside the MFMAs.
EXTRA = {"attention.hip": ["-fno-honor-nans"], "attention_p.hip": ["-fno-honor-nans", "-fno-slp-vectorize"],
         "attention_bwd.hip": ["-fno-slp-vectorize"], "attention_train.hip": ["-fno-honor-nans", "-fno-slp-vectorize"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC,
         "-Wall", "-Wno-unused-function"]


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(INCLUDE, "ditto_hip.h"))
    return hdrs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    srcs, hdrs = _sources(), _deps()
    flags = FLAGS
    objs, jobs = [], []
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([HIPCC, *flags, *EXTRA.get(os.path.basename(s), []), "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
