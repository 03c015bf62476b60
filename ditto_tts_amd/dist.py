"""Batch-parallel sampling over the GPUs of one node: one process per GPU, RCCL over xGMI for the
scatter of conditioning / gather of latents ONLY.  There is no collective inside the
step loop: every utterance's trajectory depends only on its own (x_T, text_emb, noise) and the
replicated weights.

xGMI is point-to-point (7 links x ~153 GB/s per GPU), so root<->peer transfers are issued as one batch of
send/recv pairs (`torch.distributed.batch_isend_irecv` = grouped ncclSend/ncclRecv on RCCL): each peer's
shard rides its own direct link instead of a ring bound by one link.

Training adds ONE collective per step: the data-parallel gradient mean
(`allreduce_gradients`), issued as reduce-scatter + all-gather over flat buckets so that on RCCL every GPU
exchanges 1/W of a bucket with each peer over its own xGMI link (a ring would be bound by one link).

Works on any backend: "nccl" (= RCCL on ROCm) on the GPU node, "gloo" in the CPU tests.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of `total` utterances for `rank`; the first (total % world) ranks get one more."""
    q, r = divmod(total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def scatter_batch(full: Optional[torch.Tensor], shape_tail, dtype, device, root: int = 0, group=None) -> torch.Tensor:
    """Root holds `full` [B_total, *shape_tail]; every rank returns its contiguous shard (root keeps a view)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    meta = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == root:
        meta[0] = full.shape[0]
    dist.broadcast(meta, src=root, group=group)
    total = int(meta.item())
    lo, hi = shard_bounds(total, world, rank)
    if rank == root:
        ops = []
        for r in range(world):
            if r == root:
                continue
            a, b = shard_bounds(total, world, r)
            if b > a:
                ops.append(dist.P2POp(dist.isend, full[a:b].contiguous(), r, group))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return full[lo:hi]
    out = torch.empty((hi - lo, *shape_tail), dtype=dtype, device=device)
    if hi > lo:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, out, root, group)]):
            w.wait()
    return out


def gather_batch(shard: torch.Tensor, total: int, root: int = 0, group=None) -> Optional[torch.Tensor]:
    """Inverse of scatter_batch: root returns [total, ...], the others None."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if rank == root:
        full = torch.empty((total, *shard.shape[1:]), dtype=shard.dtype, device=shard.device)
        lo, hi = shard_bounds(total, world, rank)
        full[lo:hi] = shard
        ops, bufs = [], []
        for r in range(world):
            if r == root:
                continue
            a, b = shard_bounds(total, world, r)
            if b > a:
                buf = torch.empty((b - a, *shard.shape[1:]), dtype=shard.dtype, device=shard.device)
                bufs.append((a, b, buf))
                ops.append(dist.P2POp(dist.irecv, buf, r, group))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        for a, b, buf in bufs:
            full[a:b] = buf
        return full
    if shard.shape[0] > 0:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, shard.contiguous(), root, group)]):
            w.wait()
    return None


def sample_sharded(sample_fn: Callable[[torch.Tensor, torch.Tensor, int], torch.Tensor],
                   text_full: Optional[torch.Tensor], xT_full: Optional[torch.Tensor], text_tail, x_tail,
                   device, root: int = 0, group=None, phases: Optional[dict] = None,
                   sync: Optional[Callable[[], None]] = None, pin_class: bool = True,
                   text_dtype: torch.dtype = torch.float32) -> Optional[torch.Tensor]:
    """Scatter (text_emb, x_T) from root, run `sample_fn(text_shard, xT_shard, first_global_index)` on every
    rank (the whole denoise loop — no communication inside), gather the final latents on root.

    `text_dtype`: what the text conditioning travels as.  torch.float32 (default) is exact: every rank computes on the
    caller's fp32 text, as the reference does and as the plain single-process sampler does, so the gathered latents equal
    `sample_fn(text_full, xT_full, 0)` bit for bit at every world size.  torch.bfloat16 halves the scatter (SURVEY.md §8e:
    50.3 MB instead of 100.7 MB per peer at C3 — 0.3 ms of a 0.6 s loop on a 153 GB/s xGMI link); the engine's K/V GEMM
    rounds the text to bf16 anyway, but GlobalAdaLN pools it in fp32 (src/components/DiT.py:27), so the result then equals
    `sample_fn(text_full.bfloat16().float(), ...)` — still independent of the world size (every rank, root included,
    computes on the rounded values), but NOT the plain sampler's bits.  x_T and the latents always travel as fp32.

    `pin_class` (GPU shards only): run `sample_fn` under `hip.batch_class(rows of the GLOBAL batch)` — since ABI 9 a scope of
    THIS thread's library calls (ditto_call_opts_push / _pop), not a process-wide switch — so that every
    shard's launches pick the kernel class the unsplit batch would pick (the full-row GEMM sums in another order than
    the tiled one) and the gathered latents are bit-identical at every world size.  A shard too small to take a pinned
    full-row class (fewer than 64 rows per launch) raises instead of silently running another class.

    `phases` (optional dict) receives this rank's wall seconds of the three phases, {"scatter_s", "loop_s",
    "gather_s"}; `sync` (e.g. `torch.cuda.synchronize`) is called at each phase boundary so the figures are device
    time, not enqueue time."""
    import time
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    tick = (lambda: (sync() if sync else None, time.perf_counter())[1]) if phases is not None else (lambda: 0.0)
    t0 = tick()
    if text_dtype == torch.float32:
        text = scatter_batch(text_full if rank == root else None, text_tail, torch.float32, device, root, group)
    else:
        text_t = text_full.to(text_dtype) if rank == root else None
        text = scatter_batch(text_t, text_tail, text_dtype, device, root, group).to(torch.float32)
        del text_t
    xT = scatter_batch(xT_full, x_tail, torch.float32, device, root, group)
    meta = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == root:
        meta[0] = text_full.shape[0]
    dist.broadcast(meta, src=root, group=group)
    total = int(meta.item())
    lo, _ = shard_bounds(total, world, rank)
    t1 = tick()
    if text.shape[0] == 0:
        out = xT
    elif pin_class and xT.is_cuda:
        from .hip import batch_class
        with batch_class(total * int(x_tail[0])):
            out = sample_fn(text, xT, lo)
    else:
        out = sample_fn(text, xT, lo)
    t2 = tick()
    full = gather_batch(out, total, root, group)
    t3 = tick()
    if phases is not None:
        phases.update(scatter_s=t1 - t0, loop_s=t2 - t1, gather_s=t3 - t2)
    return full


_TWO_PHASE = {}   # backend name -> does it have reduce_scatter_tensor + all_gather_into_tensor (probed on first use)


def _two_phase_ok(group=None) -> bool:
    """Capability, not backend name: RCCL has reduce_scatter_tensor / all_gather_into_tensor, and so does this torch's gloo;
    a backend without them answers with an exception on the first (tiny) call and falls back to one all_reduce per bucket."""
    be = dist.get_backend(group)
    if be not in _TWO_PHASE:
        world = dist.get_world_size(group)
        dev = torch.device("cuda", torch.cuda.current_device()) if be == "nccl" else torch.device("cpu")
        try:
            flat = torch.zeros(world, dtype=torch.float32, device=dev)
            shard = torch.empty(1, dtype=torch.float32, device=dev)
            dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, group=group)
            dist.all_gather_into_tensor(flat, shard, group=group)
            _TWO_PHASE[be] = True
        except (RuntimeError, NotImplementedError):
            _TWO_PHASE[be] = False
    return _TWO_PHASE[be]


def _mean_flat_(flat: torch.Tensor, world: int, group=None, two_phase: bool = True):
    """flat (fp32, numel a multiple of `world`) <- its mean over the group, in place.  two_phase: reduce-scatter (every rank sums
    1/W of the bucket: on RCCL each peer's slice rides its own xGMI link) + scale + all-gather; else one all_reduce."""
    if two_phase:
        shard = torch.empty(flat.numel() // world, dtype=torch.float32, device=flat.device)
        dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, group=group)
        shard.mul_(1.0 / world)
        dist.all_gather_into_tensor(flat, shard, group=group)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.mul_(1.0 / world)


def _mean_bucket_(tensors: List[torch.Tensor], world: int, group=None, two_phase: bool = True):
    """The tensors of one bucket <- their means over the group (packed into one flat fp32 buffer padded to a multiple of W)."""
    n = sum(t.numel() for t in tensors)
    padded = (n + world - 1) // world * world
    flat = torch.zeros(padded, dtype=torch.float32, device=tensors[0].device)
    off = 0
    for t in tensors:
        flat[off:off + t.numel()].copy_(t.reshape(-1))
        off += t.numel()
    _mean_flat_(flat, world, group, two_phase)
    off = 0
    for t in tensors:
        t.copy_(flat[off:off + t.numel()].view_as(t))
        off += t.numel()


class GradSync:
    """The data-parallel gradient mean of one training step, OVERLAPPED with the backward that produces the gradients.

    The library's backward writes gradients layer by layer, top layer first (ditto_train_backward_layers); `reduce(tensors)` is
    called with a piece's gradients as soon as that piece is ENQUEUED on the compute stream: the tensors join the open bucket,
    and a bucket that has reached `bucket_bytes` is exchanged at once — on CUDA on a side stream that waits for an event
    recorded on the compute stream, so that the reduce-scatter / all-gather of the upper layers runs while the lower layers
    are still being computed (at W = 8 the 552 MB of DiTTO-S gradients are ~1 GB moved per GPU per step against a ~36 ms
    backward).  `finish()` exchanges what is left and makes the compute stream wait for the exchange.  Bucket contents depend on
    the ORDER of the reduce() calls only — a property of the model — so every rank forms the same buckets.  Same arithmetic as
    allreduce_gradients (reduce-scatter + scale + all-gather per bucket): the means are identical to the un-overlapped call's
    when the buckets are."""

    def __init__(self, group=None, bucket_bytes: int = 96 << 20, overlap: bool = True, exchange_at_world1: bool = False):
        """`exchange_at_world1`: run the collectives even in a group of one rank (the mean over one rank is the identity): the
        whole path — events, side stream, padding, reduce-scatter + all-gather on RCCL — on a single-GPU box (tests)."""
        self.group, self.bucket_bytes, self.overlap = group, int(bucket_bytes), overlap
        self.world = dist.get_world_size(group)
        self.exchange_at_world1 = exchange_at_world1
        self.two_phase = _two_phase_ok(group) if (self.world > 1 or exchange_at_world1) else True
        self._open: List[torch.Tensor] = []
        self._open_bytes = 0
        self._comm = None
        self._keep = []            # tensors the side stream still works on
        self.buckets = 0           # exchanged in the step in progress (reset by finish)
        self.last_buckets = 0

    def _stream(self, dev):
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=dev)
        return self._comm

    def _flush(self):
        if not self._open:
            return
        ts, self._open, self._open_bytes = self._open, [], 0
        self.buckets += 1
        if self.world == 1 and not self.exchange_at_world1:
            return
        if ts[0].is_cuda and self.overlap:
            cur = torch.cuda.current_stream(ts[0].device)
            comm = self._stream(ts[0].device)
            ev = torch.cuda.Event()
            ev.record(cur)                       # everything enqueued so far: these gradients' producers included
            comm.wait_event(ev)
            with torch.cuda.stream(comm):
                _mean_bucket_(ts, self.world, self.group, self.two_phase)
            for t in ts:
                t.record_stream(comm)
            self._keep.append(ts)
        else:
            _mean_bucket_(ts, self.world, self.group, self.two_phase)

    def reduce(self, tensors: List[torch.Tensor]):
        for t in tensors:
            if t is None:
                continue
            self._open.append(t)
            self._open_bytes += t.numel() * 4
        if self._open_bytes >= self.bucket_bytes:
            self._flush()

    def abort(self):
        """Drop the step's open state after an exception in the backward or in a collective: the tensors of the dead step must not
        join the next step's buckets (another size than the other ranks' -> a collective mismatch or silently averaged stale
        gradients).  The compute stream still waits for whatever exchange was enqueued, so buffers are not reused under it."""
        self._open, self._open_bytes = [], 0
        if self._comm is not None and self._keep:
            torch.cuda.current_stream(self._keep[0][0].device).wait_stream(self._comm)
        self._keep = []
        self.buckets = 0

    def finish(self) -> int:
        self._flush()
        if self._comm is not None and self._keep:
            torch.cuda.current_stream(self._keep[0][0].device).wait_stream(self._comm)
        self._keep = []
        self.last_buckets, self.buckets = self.buckets, 0
        return self.last_buckets


def allreduce_gradients(params, bucket_bytes: int = 256 << 20, group=None) -> int:
    """Average `.grad` of `params` over the data-parallel group, in place, AFTER the backward; returns the number of buckets.
    (The overlapped form is GradSync, driven from inside the backward: DiTTO.set_grad_sync.)

    Gradients are packed into flat fp32 buckets of about `bucket_bytes` (large on purpose: 288 GB of HBM per GPU,
    and per-link-bound xGMI favours few large transfers; the whole DiTTO-S gradient is 552 MB = 3 buckets), each
    padded to a multiple of the world size.  Where the backend has reduce_scatter_tensor / all_gather_into_tensor (RCCL, and
    this torch's gloo: probed once, by capability) a bucket is reduced as reduce-scatter + all-gather; elsewhere as one
    all_reduce.  Parameters without a gradient (the dead `attn.out_proj`, frozen codec weights) are skipped — identically on
    every rank, because which parameters get gradients is a property of the model, not of the data."""
    world = dist.get_world_size(group)
    plist = [p for p in params if p.grad is not None]
    if world == 1 or not plist:
        return 0
    two_phase = _two_phase_ok(group)
    nb, i = 0, 0
    while i < len(plist):
        chunk, size = [], 0
        while i < len(plist) and (not chunk or size + plist[i].grad.numel() * 4 <= bucket_bytes):
            chunk.append(plist[i])
            size += plist[i].grad.numel() * 4
            i += 1
        _mean_bucket_([p.grad for p in chunk], world, group, two_phase)
        nb += 1
    return nb
