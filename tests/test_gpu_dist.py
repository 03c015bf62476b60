"""-m gpu: the multi-GPU host path on the one GPU a test box has — RCCL ("nccl") process group of world size 1 on
127.0.0.1.  The 2-rank behaviour is covered on CPU/gloo (tests/test_dist_gloo.py); this checks that the same code runs on
the real backend with CUDA tensors (RCCL initialisation, point-to-point group ops, the gradient bucket path)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from ditto_tts_amd.config import DiTTOConfig
from ditto_tts_amd.dist import GradSync, allreduce_gradients, gather_batch, sample_sharded, scatter_batch
from ditto_tts_amd.modules import DiTTO
from ditto_tts_amd.sampler import SpeechGenerator
from ditto_tts_amd.synth import hash_normal, synthetic_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def rccl_world1():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


@torch.no_grad()
def test_sharded_sampling_on_rccl_equals_direct(rccl_world1):
    cfg = DiTTOConfig(128, 1, 2, 64, 128, 6)
    m = DiTTO(128, 1, 2, 64, 128, 6)
    m.load_state_dict(synthetic_state_dict(cfg, 3))
    sg = SpeechGenerator(ditto_model=m.to(DEV).eval(), device=DEV)
    B, N, T = 3, 32, 16
    text = hash_normal((B, T, 128), "text", 1).to(DEV)
    xT = hash_normal((B, N, 128), "xT", 1).to(DEV)
    noises = [hash_normal((B, N, 128), f"z{i}", 2).to(DEV) for i in range(6)]

    def fn(text_shard, x_shard, first):
        return sg._SpeechGenerator__sample_latents(text_shard, x_shard, cond_by_audio=True,
                                                   noises=lambda i: noises[i][first:first + x_shard.shape[0]])

    want = fn(text, xT, 0)                                # the UNMODIFIED direct call on the caller's fp32 text
    got = sample_sharded(fn, text, xT, (T, 128), (N, 128), DEV)
    assert torch.equal(got, want)
    # opt-in bf16 transport (SURVEY.md 8e's 50 MB per peer): every rank computes on the rounded text
    got16 = sample_sharded(fn, text, xT, (T, 128), (N, 128), DEV, text_dtype=torch.bfloat16)
    assert torch.equal(got16, fn(text.to(torch.bfloat16).float(), xT, 0)) and not torch.equal(got16, got)
    sh = scatter_batch(text, (T, 128), torch.float32, DEV)
    assert torch.equal(gather_batch(sh, B), text)
    # an EMPTY batch through the same collectives (what a rank with an empty shard does at world > B)
    e = scatter_batch(text[:0], (T, 128), torch.float32, DEV)
    assert e.shape == (0, T, 128) and gather_batch(e, 0).shape == (0, T, 128)


@torch.no_grad()
def test_pinned_class_restores_and_refuses_what_it_cannot_honour(rccl_world1):
    """hip.batch_class nests (exit restores what was in force on entry) WITHOUT touching the process-wide switch (ABI 9: the
    pin is a property of this thread's calls), and a launch that cannot take a pinned full-row class — fewer rows than one
    64-row tile — raises instead of silently running the tiled kernels (whose bits differ)."""
    from ditto_tts_amd import hip
    assert hip.get_option("fr_class_rows") == 0 and hip.current_opts().class_rows == 0
    with hip.batch_class(32768):
        with hip.batch_class(512):
            assert hip.current_opts().class_rows == 512
        assert hip.current_opts().class_rows == 32768
        assert hip.get_option("fr_class_rows") == 0       # nothing process-wide moved
    assert hip.current_opts().class_rows == 0
    cfg = DiTTOConfig(768, 1, 12, 64, 768, 6)
    m = DiTTO(768, 1, 12, 64, 768, 6)
    m.load_state_dict(synthetic_state_dict(cfg, 3))
    m = m.to(DEV).eval()
    x = hash_normal((1, 48, 768), "x", 1).to(DEV)
    text = hash_normal((1, 16, 768), "t", 1).to(DEV)
    t = torch.zeros(1, dtype=torch.long, device=DEV)
    plain = m(x, text, t)
    with hip.batch_class(32768):                          # a C2-sized class on a 48-row launch
        with pytest.raises(hip.DittoHipError, match="pinned"):
            m(x, text, t)
    assert torch.equal(m(x, text, t), plain)              # ... and the switch is back


def test_gradient_bucket_path_on_rccl(rccl_world1):
    ps = [torch.nn.Parameter(torch.zeros(5, 7, device=DEV)), torch.nn.Parameter(torch.zeros(3, device=DEV))]
    ps[0].grad = hash_normal((5, 7), "g", 1).to(DEV)
    before = ps[0].grad.clone()
    assert allreduce_gradients(ps) == 0          # world size 1: nothing to do, gradients untouched
    assert torch.equal(ps[0].grad, before) and ps[1].grad is None
    # the bucket arithmetic itself, through RCCL collectives of world size 1
    flat = torch.arange(8, dtype=torch.float32, device=DEV)
    shard = torch.empty(8, device=DEV)
    dist.reduce_scatter_tensor(shard, flat)
    out = torch.empty(8, device=DEV)
    dist.all_gather_into_tensor(out, shard)
    assert torch.equal(out, flat)


def test_overlapped_gradient_exchange_on_rccl(rccl_world1):
    """dist.GradSync through DiTTO.set_grad_sync on the real backend (the reduce-scatter + all-gather branch had
    never executed on RCCL): the backward runs in layer pieces, each piece's bucket is exchanged on a side stream behind an event
    while the next piece computes, finish() joins the streams.  World size 1, `exchange_at_world1` — the mean over one rank is the
    identity, so every gradient must equal the plain backward's BIT FOR BIT, through padding (buckets of 1 MiB over tensors of
    odd sizes), the 1/W scale and the copy back."""
    import torch.nn.functional as F
    from ditto_tts_amd.synth import synthetic_inputs
    cfg = DiTTOConfig(256, 4, 4, 64, 256, 20)
    x, text, t = (z.to(DEV) for z in synthetic_inputs(cfg, 2, 128, 48, seed=41))
    target = hash_normal((2, 128, 256), "noise", 42).to(DEV)

    def grads(sync, per_piece=1):
        m = DiTTO(256, 4, 4, 64, 256, 20)
        m.load_state_dict(synthetic_state_dict(cfg, 5))
        m = m.to(DEV).eval()
        m.set_grad_sync(sync, layers_per_piece=per_piece)
        F.mse_loss(m(x, text, t), target).backward()
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

    plain = grads(None)
    for per_piece, bucket in ((1, 1 << 20), (2, 3 << 20), (4, 1 << 30)):
        sync = GradSync(bucket_bytes=bucket, exchange_at_world1=True)
        assert sync.two_phase
        got = grads(sync, per_piece)
        assert sync.last_buckets >= 1 and (bucket > (1 << 20) or sync.last_buckets >= 3)
        assert got.keys() == plain.keys()
        for n, g in plain.items():
            assert torch.equal(got[n], g), (per_piece, bucket, n)
    # the post-hoc form on the same backend: a forced exchange at world 1 leaves the gradients untouched too
    ps = [torch.nn.Parameter(torch.zeros(5, 7, device=DEV)), torch.nn.Parameter(torch.zeros(3, device=DEV))]
    for i, p in enumerate(ps):
        p.grad = hash_normal(tuple(p.shape), f"g{i}", 1).to(DEV)
    before = [p.grad.clone() for p in ps]
    s2 = GradSync(bucket_bytes=64, overlap=False, exchange_at_world1=True)
    s2.reduce([ps[0].grad]); s2.reduce([ps[1].grad])
    assert s2.finish() == 2 and all(torch.equal(p.grad, b) for p, b in zip(ps, before))


# ---------------------------------------------------------------------------------------------------------------
# world = 2 on RCCL (BASELINE configs[2] in miniature): needs two GPUs, skipped on the one-GPU test box.  Each rank is
# its own process on its own device; rank 0 holds the global batch, dist.sample_sharded scatters text / x_T over
# RCCL, both ranks run the HIP denoise loop, latents are gathered — and must equal the world-1 result BIT FOR BIT.
def _rccl_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        cfg = DiTTOConfig(256, 2, 4, 64, 256, 6)
        m = DiTTO(256, 2, 4, 64, 256, 6)
        m.load_state_dict(synthetic_state_dict(cfg, 3))
        sg = SpeechGenerator(ditto_model=m.to(dev).eval(), device=dev)
        B, N, T = 5, 160, 48                                  # 5 utterances over 2 ranks: shards of 3 and 2
        text = hash_normal((B, T, 256), "text", 1).to(dev) if rank == 0 else None
        xT = hash_normal((B, N, 256), "xT", 1).to(dev) if rank == 0 else None

        def fn(text_shard, x_shard, first):
            noises = lambda i: hash_normal((B, N, 256), f"z{i}", 2)[first:first + x_shard.shape[0]]
            with torch.no_grad():
                return sg._SpeechGenerator__sample_latents(text_shard, x_shard, cond_by_audio=True, noises=noises)

        def fn_seeded(text_shard, x_shard, first):       # per-utterance seeds = global utterance index: no noise tensors at all
            seeds = torch.arange(first, first + x_shard.shape[0], device=dev) + 1000
            with torch.no_grad():
                return sg._SpeechGenerator__sample_latents(text_shard, x_shard, seeds=seeds)

        phases = {}
        got = sample_sharded(fn, text, xT, (T, 256), (N, 256), dev, phases=phases,
                             sync=lambda: torch.cuda.synchronize(dev))
        assert set(phases) == {"scatter_s", "loop_s", "gather_s"}
        got2 = sample_sharded(fn_seeded, text, xT, (T, 256), (N, 256), dev)
        if rank == 0:
            from ditto_tts_amd.hip import batch_class
            with batch_class(B * N):                           # sample_sharded pins the unsplit batch's kernel class
                want, want2 = fn(text, xT, 0), fn_seeded(text, xT, 0)   # the direct call on the caller's fp32 text
            q.put((bool(torch.equal(got, want)) and bool(torch.equal(got2, want2)), bool(torch.isfinite(got).all()),
                   tuple(got.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the driver's multi-GPU node)")
def test_world2_rccl_sharded_sampling_is_bitwise_world1():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    same, finite, shape = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert same and finite and shape == (5, 160, 256)
