"""ISA-level invariants of the hand-scheduled kernels, checked on the cross-compiled code (no GPU needed)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_attention_tile_loops_hold_no_scratch_and_no_compiler_vmcnt_wait():
    """attn64q's tile loops run behind COUNTED `s_waitcnt vmcnt` of their LDS-DMA loads.  A spilled register reloaded in the loop, or a
    compiler-visible load left pending in front of it, puts another vmcnt wait inside the loop and turns every counted wait into a
    drain (measured 145 / 165 us against 102 / 110).  tools/check_attn_loop.py compiles attention_p.hip and scans the loop blocks of
    every instantiation for both."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_attn_loop.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("mfma 28") >= 6 and "PROBLEM" not in r.stdout, r.stdout
