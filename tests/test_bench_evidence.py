"""bench.py's evidence bookkeeping (host arithmetic, no GPU): the committed PMC traffic figure is only quoted for the kernel this tree
launches, and the committed C3 digest expectation names the sources it was taken on."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(m)      # (module level only defines things: main() runs under __name__ == "__main__")
    finally:
        sys.argv = argv
    return m


def test_committed_pmc_traffic_is_of_this_trees_kernels():
    """a traffic figure measured on another build's kernel is refused, not reused.  The newest committed pass must
    name, for every class bench.py prices, the kernel this tree launches (so the round's last PMC pass was taken on this tree's
    dispatch), and a pass that names another kernel is refused."""
    b = _bench()
    from ditto_tts_amd.config import PRESETS
    cfg = PRESETS["C2"]["cfg"]
    got, src = b.pmc_traffic("gemm_gated_mlp", 32, 1024, 1024, cfg)
    assert got and got > 261e6 and "offline PMC pass" in src, (got, src)
    for cls in ("attn_self", "attn_cross", "gemm_qkv_rope", "gemm_q_proj"):
        got, src = b.pmc_traffic(cls, 32, 1024, 1024, cfg)
        assert got, (cls, src)
    assert b.pmc_traffic("gemm_gated_mlp", 8, 1024, 1024, cfg) == (None, None)      # another workload: no figure
    keep = dict(b.TRAFFIC_KERNEL)
    try:
        b.TRAFFIC_KERNEL["attn_self"] = "some_other_kernel<"
        got, src = b.pmc_traffic("attn_self", 32, 1024, 1024, cfg)
        assert got is None and src.startswith("refused"), (got, src)
    finally:
        b.TRAFFIC_KERNEL.clear(); b.TRAFFIC_KERNEL.update(keep)


def test_c3_digest_expectation_names_its_sources():
    """profiles/c3_digest_expect.json (bench.py --write-c3-expect at N = 1) carries the hash of the kernel sources it was taken on; a run
    on other sources does not compare against it.  The hash function is stable and covers csrc/ + the C-ABI header."""
    b = _bench()
    h = b.tree_sha16()
    assert len(h) == 16 and h == b.tree_sha16()
    d = json.load(open(os.path.join(ROOT, "profiles", "c3_digest_expect.json")))
    assert set(d) >= {"tree_sha16", "latents_digest", "global_batch", "steps"} and len(d["tree_sha16"]) == 16
    if d["tree_sha16"] != h:      # legitimate between a kernel change and the round's next N = 1 run: say so, do not fail
        print(f"note: profiles/c3_digest_expect.json is of tree {d['tree_sha16']}, the sources are {h}: rerun bench.py --write-c3-expect")
