"""N > 1 path on CPU: gloo at world sizes 2 and 4.  Batch sharding must not change any utterance's result
(per-utterance outputs bit-identical for every world size).  The per-rank "sampler" here is
the oracle's CPU loop on a tiny model (tests may use the oracle); the product's scatter/gather code
(ditto_tts_amd/dist.py) is what is under test: uneven shards (5 utterances over 2 and over 4 ranks), EMPTY shards
(2 utterances over 4 ranks), the exact fp32 text transport (== the direct call on the caller's text) and the opt-in
bf16 transport (== the direct call on the rounded text, at every world size)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ditto_tts_amd.config import DiTTOConfig
from ditto_tts_amd.dist import (GradSync, _two_phase_ok, allreduce_gradients, gather_batch, sample_sharded, scatter_batch,
                                 shard_bounds)
from ditto_tts_amd.synth import hash_normal, synthetic_state_dict

CFG = DiTTOConfig(64, 1, 1, 32, 64, 4)
N, T, STEPS = 8, 6, 4


def _sample_fn(sd):
    from oracle import ditto_oracle as O

    def fn(text, xT, first):
        outs = []
        for j in range(text.shape[0]):   # per-utterance noise keyed by GLOBAL utterance index -> W-independent
            g = first + j
            noises = [hash_normal((1, N, 64), f"z{g}_{i}", 9) for i in range(STEPS)]
            x, _ = O.sample_latents(sd, 1, 1, xT[j:j + 1], text[j:j + 1], STEPS, noises)
            outs.append(x)
        return torch.cat(outs) if outs else xT
    return fn


def _worker(rank, world, port, q, B):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        sd = synthetic_state_dict(CFG, seed=4)
        text = hash_normal((B, T, 64), "text", 1) if rank == 0 else None
        xT = hash_normal((B, N, 64), "xT", 1) if rank == 0 else None
        # plain scatter / gather round trip
        sh = scatter_batch(text, (T, 64), torch.float32, "cpu")
        lo, hi = shard_bounds(B, world, rank)
        assert sh.shape[0] == hi - lo
        back = gather_batch(sh, B)
        if rank == 0:
            assert torch.equal(back, text)
        out = sample_sharded(_sample_fn(sd), text, xT, (T, 64), (N, 64), "cpu")
        out16 = sample_sharded(_sample_fn(sd), text, xT, (T, 64), (N, 64), "cpu", text_dtype=torch.bfloat16)
        if rank == 0:
            q.put((out, out16))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(target, world, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q, *args)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return got


def test_shard_bounds_cover_the_batch_in_order():
    for total in (0, 1, 2, 5, 8, 31, 256):
        for world in (1, 2, 3, 4, 8):
            b = [shard_bounds(total, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == total
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,B", [(2, 5), (4, 5), (4, 2)])      # shards 3+2; 2+1+1+1; 1+1+0+0 (two EMPTY shards)
def test_sharded_sampling_equals_the_direct_call_bitwise(world, B):
    sd = synthetic_state_dict(CFG, seed=4)
    text, xT = hash_normal((B, T, 64), "text", 1), hash_normal((B, N, 64), "xT", 1)
    torch.set_num_threads(1)
    want = _sample_fn(sd)(text, xT, 0)                               # the UNMODIFIED direct call: fp32 text, as the reference
    want16 = _sample_fn(sd)(text.to(torch.bfloat16).float(), xT, 0)  # opt-in bf16 transport: every rank on the rounded text
    got, got16 = _spawn(_worker, world, B)
    assert torch.equal(got, want)
    assert torch.equal(got16, want16)
    assert not torch.equal(want, want16)                             # (the two transports really differ)


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ps = [torch.nn.Parameter(torch.zeros(s)) for s in ((7, 5), (3,), (11, 2), (4,))]
        for i, p in enumerate(ps):
            if i != 3:                                   # one parameter without a gradient (like attn.out_proj)
                p.grad = hash_normal(tuple(p.shape), f"g{i}", rank)
        two_phase = _two_phase_ok()                      # the branch is chosen by CAPABILITY: this torch's gloo has both collectives,
        nb = allreduce_gradients(ps, bucket_bytes=160)   # tiny buckets: forces several, with world-size padding
        # the same gradients through the OVERLAPPED form: pieces arrive in backward order (top layer first), a bucket is
        # exchanged as soon as it is full, finish() takes the rest
        qs = [torch.nn.Parameter(torch.zeros(s)) for s in ((7, 5), (3,), (11, 2), (4,))]
        gs = [hash_normal(tuple(p.shape), f"g{i}", rank) if i != 3 else None for i, p in enumerate(qs)]
        sync = GradSync(bucket_bytes=100)
        sync.reduce([gs[2], gs[3]])                      # (a None in a piece is skipped, like a parameter without a gradient)
        sync.reduce([gs[1]])
        sync.reduce([gs[0]])
        nb2 = sync.finish()
        # the one-all_reduce fallback of a backend without the two collectives must give the same means
        from ditto_tts_amd.dist import _mean_bucket_
        fb = [hash_normal(tuple(p.shape), f"g{i}", rank) for i, p in enumerate(ps) if i != 3]
        _mean_bucket_(fb, world, None, two_phase=False)
        if rank == 0:
            q.put((nb, [None if p.grad is None else p.grad.clone() for p in ps], two_phase, nb2, gs, fb))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4])
def test_gradient_allreduce_is_the_mean_over_ranks(world):
    """allreduce_gradients (after the backward) and GradSync (overlapped with it: VERDICT r4 item 7) on the reduce-scatter +
    all-gather branch — the one RCCL takes — at world 2 and 4: padding to a multiple of W (35 + 3 + 22 elements in buckets of
    40 / 25 floats), bucket boundaries, the 1/W scale; and the all_reduce fallback gives the same means."""
    nb, got, two_phase, nb2, sync_got, fb = _spawn(_grad_worker, world)
    assert two_phase, "this torch's gloo was expected to have reduce_scatter_tensor / all_gather_into_tensor"
    assert nb >= 2 and got[3] is None and nb2 >= 2 and sync_got[3] is None
    k = 0
    for i, shape in enumerate(((7, 5), (3,), (11, 2))):
        want = sum(hash_normal(shape, f"g{i}", r) for r in range(world)) / world
        assert torch.allclose(got[i], want, rtol=0, atol=1e-6)
        assert torch.allclose(sync_got[i], want, rtol=0, atol=1e-6)
        assert torch.allclose(fb[k], want, rtol=0, atol=1e-6)
        k += 1
