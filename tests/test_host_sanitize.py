"""Host-side sanitizer pass.  The HOST code of the three files that do the plan / arena /
workspace offset arithmetic and the argument validation — csrc/ditto_api.hip, csrc/ditto_train.hip, csrc/slp.hip — is
rebuilt with -fsanitize=address,undefined (device code unchanged: GPU sanitizers are not available on this pool) and
linked with the ordinary kernel objects into tests/host_sanitize/driver.cpp, which drives every *_bytes query, the
ditto_model_create / ditto_slp_create argument validation and the error paths over a grid of (d, L, h, B, N, T)
including the ragged and the fp8 cases.  No GPU is needed (and none is used)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT
from ditto_tts_amd import build as B

HERE = os.path.join(ROOT, "tests", "host_sanitize")
OUT = os.path.join(ROOT, "build", "host_sanitize")
SAN_FILES = ("ditto_api.hip", "ditto_train.hip", "slp.hip")
SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-sanitize-recover=undefined",
             "-fno-omit-frame-pointer", "-g", "-O1"]


def _stale(target, deps):
    return not os.path.exists(target) or any(os.path.getmtime(d) > os.path.getmtime(target) for d in deps)


def _build_driver():
    if not os.path.exists(B.HIPCC):
        pytest.skip("hipcc not found")
    B.build(verbose=False)                                  # the ordinary objects of the kernel files
    os.makedirs(OUT, exist_ok=True)
    hdrs = B._deps()
    objs = []
    for src in B._sources():
        name = os.path.basename(src)
        if name in SAN_FILES:
            obj = os.path.join(OUT, name[:-4] + ".asan.o")
            if _stale(obj, [src] + hdrs):
                cmd = [B.HIPCC, "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-I", B.INCLUDE, "-I", B.CSRC, *SAN_FLAGS,
                       "-c", src, "-o", obj]
                r = subprocess.run(cmd, capture_output=True, text=True)
                assert r.returncode == 0, f"sanitizer build of {name} failed:\n{r.stderr[-4000:]}"
            objs.append(obj)
        else:
            objs.append(src[:-4] + ".o")
    exe = os.path.join(OUT, "driver")
    drv = os.path.join(HERE, "driver.cpp")
    if _stale(exe, objs + [drv, os.path.join(B.INCLUDE, "ditto_hip.h")]):
        cmd = [B.HIPCC, "--offload-arch=gfx950", "-std=c++17", "-I", B.INCLUDE, *SAN_FLAGS, "-x", "c++", drv, "-x", "none",
               *objs, "-o", exe]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, f"link of the sanitizer driver failed:\n{r.stderr[-4000:]}"
    return exe


@pytest.mark.timeout(900)
def test_host_code_is_clean_under_asan_and_ubsan():
    exe = _build_driver()
    import torch
    env = dict(os.environ)
    # the HIP runtime's own start-up allocations are not ours to judge
    supp = os.path.join(HERE, "lsan.supp")
    env["ASAN_OPTIONS"] = "detect_leaks=1:abort_on_error=0:halt_on_error=1:strict_string_checks=1"
    env["LSAN_OPTIONS"] = f"suppressions={supp}:print_suppressions=0"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    args = [exe] + (["--have-gpu"] if torch.cuda.device_count() > 0 else [])
    r = subprocess.run(args, capture_output=True, text=True, env=env, timeout=600)
    tail = (r.stdout + "\n" + r.stderr)[-6000:]
    assert r.returncode == 0, f"sanitizer driver failed (rc {r.returncode}):\n{tail}"
    assert "host sanitizer driver: ok" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, tail
