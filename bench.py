#!/usr/bin/env python3
"""bench.py — DiT denoise throughput on MI355X (BASELINE.json metric), one JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2] [--batch 32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Launching.  `--gpus N` with no WORLD_SIZE in the environment makes THIS process a launcher: before any GPU call it
checks that the node shows N devices and starts N ranks as a child `python -m torch.distributed.run` (never a re-exec
of a process that touched the GPU), then exits with the child's code.  Under torchrun, WORLD_SIZE must equal
--gpus.  `n_gpus` in the JSON line is the RCCL world size that actually joined; fewer than N ranks is an error, not a
smaller measurement.

A "step" = one reverse-diffusion step (DiTTO.forward + DDPM update + the step's N(0,1) draw) over one batch
of B utterances per GPU, synthetic latents of the named shape, closed-form random weights.  Every 50th step
starts a new utterance batch (fresh x_T and the step-invariant text work: cross-attention K/V of all layers
+ text AdaLN modulation), so the one-off per-utterance work is inside the timed region, amortised as in a
real 50-step sampling loop.  value = (B x world) x K / max-over-ranks wall time  ("scaling": "weak", B per GPU fixed;
no collective in the data path, SURVEY.md §8e — the only collectives are the timing barrier and the max-reduce).

Besides the contract's line the JSON carries:
  loops        5 more timed loops of K steps, HIP-event-timed on the compute stream: their ms/step and the median
  sweep        B in {1, 8, 32} per GPU (N = 1 runs only)
  c3_strong    BASELINE config 3 as a STRONG-scaling point: rank 0 holds a global batch of 256 utterances, scatters
               text / x_T (grouped RCCL send/recv, ditto_tts_amd/dist.py), every rank runs the 50-step loop over its
               shard in micro-batches of 32, latents are gathered on rank 0; scatter and gather are INSIDE the timed
               region and reported per phase
  roofline     dominant kernel class, timed live with HIP events on the launch stream (libditto_hip's per-class
               event profiler, an eager pass of the same steps); achieved = algorithmic FLOPs per launch / average
               launch duration; peak = 2.5 PFLOP/s dense bf16 (5 PFLOP/s for the fp8 GEMM classes of C5)
  cpu_baseline the fp32 oracle (proved equal to the reference, tests/test_oracle_*) on the host cores, rank 0, N = 1
               only, bounded samples at B = 1 and B = 4
  parity       the line validates itself: after the timed region the GPU model runs ONE forward at the timed batch size
               (so on the timed kernels: full-row path at B = 32) whose utterance 0 is the oracle's input
               (synthetic_inputs(cfg, 1, N, T, seed = 7), the seed-1234 weights); rel-L2 / max-abs of that utterance
               against the fp32 oracle, tolerance 2e-2 (6e-2 for the fp8 linear path).  Above tolerance the process
               exits non-zero AFTER printing the line.
  other_configs (N = 1 runs) the claims that used to live only in profiles/: C4 (N = 4096, B = 8), C5 fp8 and C5 bf16
               (24L, d = 1024, B = 16) as short timed loops with their own dominant-kernel roofline, and train_step =
               forward + backward + AdamW of C2 at B = 32 (reference src/TrainDiTTO.py:85-91), all witnessed by the
               same run
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
PEAK_FP8_TFLOPS = 5000.0    # ~5 PF dense fp8 (block-scaled MFMA)
PEAK_HBM_GBS = 8000.0
FP8_CLASSES = ("gemm_qkv_rope", "gemm_gated_mlp", "gemm_fc2")   # the GEMMs DITTO_CFG_FP8_LINEAR moves to fp8


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C2", choices=["C1", "C2", "C4", "C5", "C5_bf16", "shipped"])
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU (default: the preset's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--loops", type=int, default=5, help="extra HIP-event-timed loops of K steps (median reported)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the B in {1, 8, 32} sweep (N = 1 only)")
    ap.add_argument("--global-batch", type=int, default=256, help="C3 strong-scaling point: utterances held by rank 0")
    ap.add_argument("--no-c3", action="store_true", help="skip the C3 strong-scaling point")
    ap.add_argument("--write-c3-expect", action="store_true",
                    help="N = 1 only: record this tree's C3 latents digest in profiles/c3_digest_expect.json (what N = 2 / 4 / 8 must reproduce)")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run oracle parity check of the timed shape")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the C4 / C5 / training-step side measurements (N = 1 runs of the default config only)")
    ap.add_argument("--side-steps", type=int, default=10, help="timed steps of each side configuration")
    ap.add_argument("--noise", choices=["seeded", "torch"], default="seeded",
                    help="the step's N(0,1) draw: 'seeded' = the library's per-utterance counter-based generator inside the "
                         "update kernel (ditto_p_sample_seeded: what batch-sharded sampling uses, independent of the world "
                         "size); 'torch' = torch's generator filling a noise tensor (the reference's randn_like)")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="off",
                    help="replay the step from a HIP graph.  Measured: no gain at B = 1 / 8 (2.35 / 4.59 ms per step "
                         "either way: small batches are bound by per-kernel latency on sparse grids, not by launches)")
    return ap.parse_args()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args):
    """No WORLD_SIZE and --gpus N > 1: start N ranks as a child torchrun.  This process makes NO GPU call
    (torch.cuda.device_count() does not initialise the device on this image)."""
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but this node shows {n_dev} GPU(s); refusing to measure fewer "
                         f"ranks than asked for")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def kernel_flops(cfg, B, N, T):
    """Algorithmic FLOPs of ONE launch of each kernel class (2 FLOP / MAC)."""
    d, M = cfg.hidden_dim, B * N
    return {
        "gemm_qkv_rope": 2.0 * M * 3 * d * d,
        "gemm_q_proj": 2.0 * M * d * d,
        "gemm_out_proj": 2.0 * M * d * d,
        "gemm_gated_mlp": 2.0 * M * 8 * d * d,
        "gemm_fc2": 2.0 * M * d * 4 * d,
        "gemm_final": 2.0 * M * d * 2 * d,
        "attn_self": 4.0 * M * N * d,
        "attn_cross": 4.0 * M * T * d,
    }


def stream_is_bf16(cfg, B, N):
    """Does a forward of B x N rows carry its residual stream as bf16?  Answered by the library itself (ditto_full_row_plan_opts:
    csrc/ditto_api.hip ditto_forward's rule — "residual_bf16" on, d = 768, head_dim 64, bf16 linears, both fused launches on the
    128-row full-row kernel = the full-row class)."""
    from ditto_tts_amd import hip
    try:
        return hip.stream_is_bf16(cfg, B, N)
    except Exception:  # noqa: BLE001 - an older library selected with DITTO_HIP_LIB
        return False


def kernel_bytes(cfg, B, N, T, seeded=True):
    """Algorithmic HBM bytes of ONE launch of the memory-bound classes."""
    d, M = cfg.hidden_dim, B * N
    hs = 2.0 if stream_is_bf16(cfg, B, N) else 4.0       # bytes per element of the residual stream h
    return {
        "layernorm": M * d * (hs + 2.0),                # h in, bf16 out
        # fp32 x in; h out + bf16 raw copy (proj_in's operand) + block 0's norm1 output (bf16; fused since round 4)
        "adaln": M * d * (4 + hs + 2.0 + (0.0 if cfg.fp8_linear else 2.0)),
        # seeded kernel: x and eps in, x out (the noise is generated in registers); noise-tensor kernel: + z in
        "p_sample_update": M * d * 4 * (3.0 if seeded else 4.0),
    }


def floor_table(cfg, B, N, T, classes):
    """profiles/r05_floor_table.json: per kernel class the time (us) its launch takes with the removable overheads knocked out
    in diagnostic builds (main loop without epilogue, operands L2-hot, no stores: tools/round4/r04_run5.sh) —
    measured OFFLINE on the timed shape, so only reported for it.  step_floor_ms = the step if every class ran at its floor
    (classes without a floor entry at their time in THIS run); step_frac_at_floor = the executed FLOPs at that time / 2.5 PF."""
    if not (B == 32 and N == 1024 and T == 1024 and cfg.hidden_dim == 768 and cfg.num_layers == 12 and not cfg.fp8_linear):
        return None
    try:
        tab = json.load(open(os.path.join(ROOT, "profiles", "r05_floor_table.json")))
    except (OSError, ValueError):
        return None
    us = 0.0
    for name, e in classes.items():
        per = tab.get(name, {}).get("floor_us")
        us += e["launches_per_step"] * (per if per is not None else e["avg_ms"] * 1e3)
    one_off = cfg.flops_text_kv(T) / cfg.diffusion_steps
    fl = B * (cfg.flops_per_utt_step(N, T, cached_kv=True) + one_off)
    return {"step_floor_ms": us * 1e-3, "step_frac_at_floor": fl / (us * 1e-6) / 1e12 / PEAK_BF16_TFLOPS,
            "per_class_floor_us": {k: v.get("floor_us") for k, v in tab.items() if isinstance(v, dict)},
            "source": "profiles/r05_floor_table.json (offline diagnostic builds on this shape; DESIGN.md section 9)"}


def class_peak(cfg, name):
    return PEAK_FP8_TFLOPS if (cfg.fp8_linear and name in FP8_CLASSES) else PEAK_BF16_TFLOPS


def step_min_seconds(cfg, B, N, T, one_off_per_step):
    """Time the executed FLOPs of one step need at the MFMA peak of the type each part runs in (fp8 GEMMs of C5 against
    5 PF, everything else against 2.5 PF): the denominator-free form of SURVEY §8d's time-weighted fraction."""
    d, L = cfg.hidden_dim, cfg.num_layers
    kf = kernel_flops(cfg, B, N, T)
    per_layer = {k: kf[k] for k in ("gemm_qkv_rope", "gemm_q_proj", "gemm_out_proj", "gemm_gated_mlp", "gemm_fc2",
                                    "attn_self", "attn_cross")}
    sec = sum(L * f / (class_peak(cfg, k) * 1e12) for k, f in per_layer.items())
    sec += kf["gemm_final"] / (PEAK_BF16_TFLOPS * 1e12)
    sec += B * one_off_per_step / (PEAK_BF16_TFLOPS * 1e12)      # the text K/V GEMM is bf16 in every config
    return sec


# the kernel this tree launches per class at C2, B = 32 (what a committed PMC pass must have measured to be quoted)
TRAFFIC_KERNEL = {"gemm_gated_mlp": "gemm256_kernel<3", "gemm_qkv_rope": "gemm256_kernel<2", "gemm_q_proj": "gemm_lnq_kernel<32, 4, true, 768, 8",
                  "gemm_out_proj": "gemm_frd_kernel", "gemm_fc2": "gemm_frd_kernel", "gemm_final": "gemm192_kernel<4",
                  "attn_self": "attn64q_kernel<true", "attn_cross": "attn64q_kernel<false"}


def pmc_traffic(kernel_class, B, N, T, cfg):
    """(HBM bytes per launch of `kernel_class`, where the figure comes from): the committed rocprofv3 PMC passes
    (profiles/*_pmc_traffic.json, tools/pmc_traffic.py: separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command,
    gfx950 x2 correction on the read side).  bench.py cannot run under the counter collector itself, so the figure is an
    OFFLINE one — not a measurement of this run — and is labelled so in the line ("traffic_source"); it is only reported
    when the workload is the one those passes measured (C2, B=32), else null."""
    if not (B == 32 and N == 1024 and T == 1024 and cfg.hidden_dim == 768 and cfg.num_layers == 12):
        return None, None
    pdir = os.path.join(ROOT, "profiles")
    try:
        names = sorted((f for f in os.listdir(pdir) if f.endswith("pmc_traffic.json")), reverse=True)  # newest first
    except OSError:
        return None, None
    want = TRAFFIC_KERNEL.get(kernel_class)
    for name in names:
        try:
            d = json.load(open(os.path.join(pdir, name)))
            rec = d[kernel_class]
            # a pass taken on another build's kernel says nothing about this one: the recorded kernel name must be the one this
            # tree launches for the class, else the figure is refused, not quietly reused
            if want and want not in rec.get("kernel", ""):
                return None, f"refused: profiles/{name} measured {rec.get('kernel', '?')[:60]!r}, this tree launches {want!r} for {kernel_class}"
            return rec["traffic_bytes"], f"offline PMC pass profiles/{name} (not this run)"
        except (OSError, KeyError, ValueError, TypeError):
            continue
    return None, None


def tree_sha16():
    """sha256 (first 16 hex digits) over the kernel sources and the C-ABI header: identifies the code that decides the bits
    (a rebuilt .so is not byte-identical, the sources are)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "ditto_tts_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(csrc, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "ditto_hip.h"), "rb").read())
    return h.hexdigest()[:16]


def usable_cores():
    """CPUs this process may actually use: the scheduler affinity capped by the cgroup CPU quota (the GPU box
    shows 256 hardware threads behind a 16-CPU quota; running 256 torch threads there is 70x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(cfg, N, T):
    """fp32 oracle on the host cores (bounded samples): B = 1, 12 timed steps after 1 warm-up; B = 4, 3 timed steps
    after 1 warm-up (SURVEY §8d / BASELINE.md §2 step 3).  The top-level value is the B = 1 figure."""
    from ditto_tts_amd.synth import hash_normal, synthetic_inputs, synthetic_state_dict
    from oracle import ditto_oracle as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    sd = synthetic_state_dict(cfg, seed=1234)
    betas, alphas, acp = O.sampler_tables(cfg.diffusion_steps)

    def run(B, steps):
        x, text, _ = synthetic_inputs(cfg, B, N, T, seed=7)
        z = hash_normal(tuple(x.shape), "z", 7)
        ts = []
        with torch.no_grad():
            for i in range(steps + 1):
                t = torch.full((B,), cfg.diffusion_steps - 1 - i, dtype=torch.long)
                t0 = time.perf_counter()
                eps = O.ditto_forward(sd, cfg.num_layers, cfg.num_heads, x, text, t)
                x = O.p_sample_update(x, eps, t, betas, alphas, acp, z)
                ts.append(time.perf_counter() - t0)
        ts = sorted(ts[1:])
        return ts[len(ts) // 2], ts[0], ts[-1]

    med1, min1, max1 = run(1, 12)
    try:
        quota = open("/sys/fs/cgroup/cpu.max").read().strip()
    except OSError:
        quota = None
    out = {"value": 1.0 / med1, "unit": "utterance-steps/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"B=1, 12 steps (median) after 1 warm-up of the same {cfg.num_layers}L/d={cfg.hidden_dim}/"
                     f"N={N}/T={T} step, fp32 torch-CPU oracle (uncached text K/V, as the reference)",
           "s_per_step": med1, "gflops": cfg.flops_per_utt_step(N, T, cached_kv=False) / med1 / 1e9,
           # the host is a shared, quota-limited cgroup: the figure moved 2.2x between rounds with identical code, so it carries its
           # spread: the fastest and slowest of the 12 steps and the quota it ran under
           "s_per_step_min": min1, "s_per_step_max": max1, "value_best": 1.0 / min1, "spread": max1 / min1, "cgroup_cpu_max": quota}
    try:
        cpu_name = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except (OSError, IndexError):
        cpu_name = None
    out["cpu_model"] = cpu_name
    med4 = run(4, 3)[0]
    out["b4"] = {"value": 4.0 / med4, "s_per_step": med4, "sample": "B=4, 3 steps (median) after 1 warm-up",
                 "gflops": 4 * cfg.flops_per_utt_step(N, T, cached_kv=False) / med4 / 1e9}
    return out


class StepRunner:
    """One (model, B) working set: state, noise, t, conditioning, and the step function of the timed region."""

    def __init__(self, eng, sg, cfg, B, N, T, dev, seed, use_graph=False, seeded=True):
        self.eng, self.sg, self.cfg, self.B, self.N, self.T, self.dev = eng, sg, cfg, B, N, T, dev
        self.seeded = seeded and not use_graph          # the step index is a kernel argument: not replayable from a graph
        self.seeds = torch.arange(B, device=dev, dtype=torch.long) + 7919 * seed
        self.batches = 0
        self.S = cfg.diffusion_steps
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(seed)
        self.text = torch.randn(B, T, cfg.text_dim, device=dev, generator=self.gen)
        self.x = torch.empty(B, N, cfg.hidden_dim, device=dev)
        self.z = torch.empty_like(self.x)
        self.t = torch.empty(B, device=dev, dtype=torch.long)
        self.cond, self.graph, self.use_graph = None, None, use_graph

    def step(self, i, eager=False):
        k = i % self.S
        sg, eng = self.sg, self.eng
        if k == 0 or self.cond is None:       # new utterance batch: x_T and the step-invariant text work
            self.batches += 1
            if self.seeded:
                self.seeds += self.B               # new utterances, new seeds
                eng.noise_normal_(self.x, self.seeds, 0xFFFFFFFF)
            else:
                self.x.normal_(generator=self.gen)
            if self.cond is None:
                self.cond = eng.prepare_text(self.text, self.N)
            else:                              # same buffers (a graph is bound to them), new contents
                eng.prepare_text_into(self.text, self.N, self.cond)
        self.t.fill_(self.S - 1 - k)
        if self.seeded:                       # the step's N(0,1) draw happens inside the update kernel
            eng.p_sample_seeded_(self.x, self.cond, self.t, self.seeds, self.S - 1 - k, sg.betas, sg.alphas,
                                 sg.alphas_cumprod)
            return
        self.z.normal_(generator=self.gen)    # the step's N(0,1) draw (reference: randn_like per step)
        if self.use_graph and not eager:
            if self.graph is None:
                self.graph = eng.capture_p_sample(self.x, self.cond, self.t, self.z, sg.betas, sg.alphas,
                                                  sg.alphas_cumprod)
            self.graph.replay()
        else:
            eng.p_sample_(self.x, self.cond, self.t, self.z, sg.betas, sg.alphas, sg.alphas_cumprod)

    def run(self, first, n):
        for i in range(first, first + n):
            self.step(i)


def profile_classes(eng, runner, cfg, B, N, T, profile_steps):
    """Per-kernel-class HIP-event timing (libditto_hip's profiler, events on the launch stream) of `profile_steps` pure
    denoise steps of `runner`, eager.  Returns (roofline of the dominant class, every class)."""
    dev = runner.dev
    eng.profile_enable(True)
    for i in range(profile_steps):
        runner.step(1 + i, eager=True)     # k != 0: pure denoise steps, eager so events bracket launches
    torch.cuda.synchronize(dev)
    prof = eng.profile_read()
    eng.profile_enable(False)
    kf, kb = kernel_flops(cfg, B, N, T), kernel_bytes(cfg, B, N, T, runner.seeded)
    classes = {}
    tot_ms = sum(ms for _, ms in prof.values()) or 1.0
    for name, (n, ms) in prof.items():
        if n == 0:
            continue
        avg = ms / n
        ent = {"launches_per_step": n / profile_steps, "avg_ms": avg, "share": ms / tot_ms}
        if name in kf:
            ent["tflops"] = kf[name] / (avg * 1e-3) / 1e12
            ent["peak_tflops"] = class_peak(cfg, name)
            ent["frac_mfma"] = ent["tflops"] / ent["peak_tflops"]
        if name in kb:
            ent["gbs"] = kb[name] / (avg * 1e-3) / 1e9
            ent["frac_hbm"] = ent["gbs"] / PEAK_HBM_GBS
        classes[name] = ent
    dom = max(classes, key=lambda k: classes[k]["share"])
    e = classes[dom]
    traffic, source = pmc_traffic(dom, B, N, T, cfg)
    if "tflops" in e:
        roof = {"bound": "mfma", "kernel": dom, "achieved": e["tflops"], "peak": e["peak_tflops"],
                "unit": "TFLOP/s", "frac": e["frac_mfma"], "traffic": traffic, "traffic_source": source,
                "avg_launch_ms": e["avg_ms"], "flops_per_launch": kf[dom]}
    else:
        roof = {"bound": "hbm", "kernel": dom, "achieved": e["gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": e["frac_hbm"], "traffic": traffic, "traffic_source": source, "avg_launch_ms": e["avg_ms"],
                "bytes_per_launch": kb[dom]}
    return roof, classes


def parity_check(model, cfg, B, N, T, dev):
    """One GPU forward at the TIMED batch size (so on the timed kernels) whose utterance 0 is the oracle's input, against
    the fp32 oracle on the host cores.  Tolerance: SURVEY.md 8c (bf16 operands / fp32 accumulate: rel-L2 <= 2e-2; the
    fp8 linear path of C5: 6e-2)."""
    from ditto_tts_amd.synth import synthetic_inputs, synthetic_state_dict
    from oracle import ditto_oracle as O
    torch.set_num_threads(usable_cores())
    x0, text0, t0 = synthetic_inputs(cfg, 1, N, T, seed=7)
    with torch.no_grad():
        want = O.ditto_forward(synthetic_state_dict(cfg, seed=1234), cfg.num_layers, cfg.num_heads, x0, text0, t0)
        if B > 1:
            x1, text1, t1 = synthetic_inputs(cfg, B - 1, N, T, seed=8)
            x, text, t = torch.cat([x0, x1]), torch.cat([text0, text1]), torch.cat([t0, t1])
        else:
            x, text, t = x0, text0, t0
        got = model(x.to(dev), text.to(dev), t.to(dev))[:1].float().cpu()
        # the OTHER residual stream on the same inputs (the timed one is whatever "residual_bf16" says for this class): both
        # figures stand next to the timed number
        from ditto_tts_amd import hip
        other = None
        if stream_is_bf16(cfg, B, N) or (hip.get_option("residual_bf16") == 0 and cfg.hidden_dim == 768 and not cfg.fp8_linear):
            prev = hip.get_option("residual_bf16")
            hip.set_option("residual_bf16", 1 - prev)
            try:
                alt = model(x.to(dev), text.to(dev), t.to(dev))[:1].float().cpu()
            finally:
                hip.set_option("residual_bf16", prev)
            other = {"stream": "fp32" if prev else "bf16",
                     "rel_l2": float(torch.linalg.norm(alt.double() - want.double()) / torch.linalg.norm(want.double()))}
        # the LOOP on the timed kernels: `loop_steps` reverse-diffusion steps (forward + update, injected noise) of the same
        # batch, utterance 0 against the oracle's restated loop (src/model/SpeechGenerator.py:135-163)
        loop_steps = 10
        from ditto_tts_amd.sampler import SpeechGenerator
        from ditto_tts_amd.synth import hash_normal
        S = cfg.diffusion_steps
        sg = SpeechGenerator(ditto_model=model, device=dev, diffusion_steps=S)
        eng = model.engine(dev)
        betas, alphas, acp = O.sampler_tables(S)
        sd = synthetic_state_dict(cfg, seed=1234)
        xg, cond = x.to(dev).clone(), eng.prepare_text(text.to(dev), N)
        xo = x0.clone()
        zg = torch.zeros_like(xg)
        tt = torch.empty(B, device=dev, dtype=torch.long)
        for i in range(loop_steps):
            tv = S - 1 - i
            z0 = hash_normal(tuple(x0.shape), f"loopz{i}", 7)
            zg.normal_(); zg[:1] = z0.to(dev)
            tt.fill_(tv)
            eng.p_sample_(xg, cond, tt, zg, sg.betas, sg.alphas, sg.alphas_cumprod)
            to = torch.full((1,), tv, dtype=torch.long)
            xo = O.p_sample_update(xo, O.ditto_forward(sd, cfg.num_layers, cfg.num_heads, xo, text0, to), to, betas, alphas, acp, z0)
        gl = xg[:1].float().cpu()
        loop_rel = float(torch.linalg.norm(gl.double() - xo.double()) / torch.linalg.norm(xo.double()))
        del sg, xg, zg, cond
    dlt = (got.double() - want.double())
    rel = float(torch.linalg.norm(dlt) / torch.linalg.norm(want.double()))
    tol = 6e-2 if cfg.fp8_linear else 2e-2
    ok = bool(rel <= tol and torch.isfinite(got).all() and loop_rel <= tol and torch.isfinite(gl).all())
    return {"rel_l2": rel, "max_abs": float(dlt.abs().max()), "tol": tol, "ok": ok,
            "ref_std": float(want.std()), "other_stream": other,
            "loop_rel_l2": loop_rel, "loop_steps": loop_steps,
            "what": f"utterance 0 of one GPU forward at B={B} (the timed kernels) vs the fp32 CPU oracle on "
                    f"synthetic_inputs(cfg, 1, N={N}, T={T}, seed=7), weights synthetic_state_dict(seed=1234); loop_rel_l2: the same "
                    f"utterance after {loop_steps} steps of the sampling loop (t = {S - 1} .. {S - loop_steps}, injected noise) on "
                    f"the timed kernels vs the oracle's restated loop; other_stream: the forward on the other residual stream"}


def side_config(name, dev, steps, profile_steps, state_cache):
    """A short timed loop of another BASELINE configuration in this same process (N = 1 runs): ms/step from HIP events on
    the compute stream over `steps` steps after 3 warm-up steps (the text work of a new utterance batch is in step 0 of
    the loop, as in the headline), the step's executed TFLOP/s, and the dominant kernel's roofline."""
    from ditto_tts_amd.config import PRESETS
    from ditto_tts_amd.modules import DiTTO
    from ditto_tts_amd.sampler import SpeechGenerator
    from ditto_tts_amd.synth import synthetic_state_dict
    p = PRESETS[name]
    cfg, N, T, B = p["cfg"], p["N"], p["T"], p["B"]
    S = cfg.diffusion_steps
    key = (cfg.hidden_dim, cfg.num_layers, cfg.num_heads)
    if key not in state_cache:
        state_cache[key] = synthetic_state_dict(cfg, seed=1234)
    model = DiTTO(cfg.hidden_dim, cfg.num_layers, cfg.num_heads, cfg.time_dim, cfg.text_dim, S, fp8_linear=cfg.fp8_linear)
    model.load_state_dict(state_cache[key])
    model = model.to(dev).eval()
    sg = SpeechGenerator(ditto_model=model, device=dev, diffusion_steps=S)
    eng = model.engine(dev)
    with torch.no_grad():
        r = StepRunner(eng, sg, cfg, B, N, T, dev, 3000, False, True)
        r.run(0, 3)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r.run(0, steps)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / steps
        roof, classes = profile_classes(eng, r, cfg, B, N, T, profile_steps)
    one_off = cfg.flops_text_kv(T) * ((steps + S - 1) // S) / steps
    fl = B * (cfg.flops_per_utt_step(N, T, cached_kv=True) + one_off)
    out = {"config": name, "workload": f"{cfg.num_layers}L d={cfg.hidden_dim} h={cfg.num_heads} N={N} T={T} B={B}",
           "dtype": "fp8(e4m3) linear + bf16 attention" if cfg.fp8_linear else "bf16", "steps": steps,
           "ms_per_step": ms, "value": B / (ms * 1e-3), "unit": "utterance-steps/s",
           "step_tflops": fl / (ms * 1e-3) / 1e12,
           "step_frac_of_mfma_peak": step_min_seconds(cfg, B, N, T, one_off) / (ms * 1e-3),
           "roofline": roof,
           "kernel_classes": {k: {kk: v[kk] for kk in ("avg_ms", "share", "frac_mfma", "frac_hbm") if kk in v}
                              for k, v in classes.items()}}
    del r, eng, sg, model
    torch.cuda.empty_cache()
    return out


def train_step_line(dev, state_cache, steps=3):
    """forward + backward + AdamW of DiTTO-S (C2 shape, B = 32 utterances, N = T = 1024) through the reference's training
    closure shape (src/TrainDiTTO.py:85-91: model.train(), MSE against the noise, loss.backward(), optimizer step);
    phase medians over `steps` steps after 1 warm-up (wall clock around synchronised phases), then the step time of `steps`
    more steps run back to back."""
    import torch.nn.functional as F
    from ditto_tts_amd.config import PRESETS
    from ditto_tts_amd.modules import DiTTO
    p = PRESETS["C2"]
    cfg, N, T, B = p["cfg"], p["N"], p["T"], p["B"]
    key = (cfg.hidden_dim, cfg.num_layers, cfg.num_heads)
    m = DiTTO(cfg.hidden_dim, cfg.num_layers, cfg.num_heads, cfg.time_dim, cfg.text_dim, cfg.diffusion_steps)
    m.load_state_dict(state_cache[key])
    m = m.to(dev).train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True)
    g = torch.Generator(device=dev).manual_seed(1)
    d = cfg.hidden_dim
    x = torch.randn(B, N, d, device=dev, generator=g)
    text = torch.randn(B, T, d, device=dev, generator=g)
    noise = torch.randn(B, N, d, device=dev, generator=g)
    t = torch.randint(0, cfg.diffusion_steps, (B,), device=dev, generator=g)
    times = {"fwd": [], "bwd": [], "opt": []}
    loss = None
    for i in range(steps + 1):
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        loss = F.mse_loss(m(x, text, t), noise)
        torch.cuda.synchronize(dev); t1 = time.perf_counter()
        opt.zero_grad(); loss.backward()
        torch.cuda.synchronize(dev); t2 = time.perf_counter()
        opt.step()
        torch.cuda.synchronize(dev); t3 = time.perf_counter()
        if i:
            times["fwd"].append(t1 - t0); times["bwd"].append(t2 - t1); times["opt"].append(t3 - t2)
    med = {k: sorted(v)[len(v) // 2] * 1e3 for k, v in times.items()}
    phase_sum = sum(med.values())
    # the step as a training loop runs it: `steps` whole steps back to back, ONE synchronisation at the end (the three
    # synchronisations above drain the GPU between the phases and cost the step ~1 ms)
    torch.cuda.synchronize(dev); t0 = time.perf_counter()
    for _ in range(steps):
        loss = F.mse_loss(m(x, text, t), noise)
        opt.zero_grad(); loss.backward()
        opt.step()
    torch.cuda.synchronize(dev)
    tot = (time.perf_counter() - t0) / steps * 1e3
    fl = 3 * cfg.flops_per_utt_step(N, T, cached_kv=False) * B
    out = {"config": "C2 training step", "workload": f"12L d=768 h=12 N={N} T={T} B={B}: forward (tape) + backward + fused AdamW",
           "dtype": "bf16 operands, fp32 accumulate / gradients / master weights", "steps": steps,
           "fwd_ms": med["fwd"], "bwd_ms": med["bwd"], "optimizer_ms": med["opt"], "phase_sum_ms": phase_sum,
           "ms_per_step": tot, "timing": "ms_per_step: whole steps back to back, one synchronisation per `steps`; fwd / bwd / "
                                         "optimizer: medians of separately synchronised phases (phase_sum_ms)",
           "value": B / tot * 1e3, "unit": "utterances/s", "tflops_3x_forward": fl / (tot * 1e-3) / 1e12,
           "frac_of_mfma_peak": fl / (tot * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, "loss_finite": bool(torch.isfinite(loss).item()),
           "peak_mem_gib": torch.cuda.max_memory_allocated(dev) / 2**30}
    del m, opt, x, text, noise, loss
    torch.cuda.empty_cache()
    # the line validates itself: every parameter gradient of a 2-layer model of the SAME widths, kernel class pinned to this
    # batch, train mode with dropout, against fp32 autograd of the CPU oracle (oracle/checks.py; after the timed region).
    # B = 4, N = 768 = 3 072 rows: 12 x 12 = 144 tiles of 256 x 256 in the fc2 dgrad, the fewest that take the FUSED gated-MLP
    # backward (EPI_GATED_BWD) and the training forward's own gated instantiation (EPI_GATED_PRE) the timed step runs — at the
    # 512 rows of the round-4 check those two kernels were not in the self-check
    try:
        from oracle.checks import train_grad_parity
        torch.set_num_threads(usable_cores())
        gp = train_grad_parity(dev, shape=dict(B=4, N=768, T=192))
        out["grad_parity"] = {k: gp[k] for k in ("worst_rel_l2", "tensor", "median_rel_l2", "loss_rel", "n_tensors", "tol",
                                                 "ok", "what")}
    except Exception as e:  # noqa: BLE001 - a checker failure must not lose the timed numbers; it is reported instead
        out["grad_parity"] = {"ok": False, "error": repr(e)}
    torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path exists in ditto_tts_amd)")
    # stdout carries ONE JSON line and nothing else: RCCL prints a version banner on the C-level stdout at exit, so
    # file descriptor 1 is pointed at stderr for the life of the process and the line goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if env_world is None:
        os.environ["MASTER_PORT"] = str(free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm
    if dist.get_world_size() != args.gpus:
        raise SystemExit(f"bench.py: {dist.get_world_size()} ranks joined, --gpus {args.gpus} asked")

    from ditto_tts_amd.config import PRESETS
    from ditto_tts_amd.dist import sample_sharded
    from ditto_tts_amd.modules import DiTTO
    from ditto_tts_amd.sampler import SpeechGenerator
    from ditto_tts_amd.synth import synthetic_state_dict

    p = PRESETS[args.config]
    cfg, N, T = p["cfg"], p["N"], p["T"]
    B = args.batch or p["B"]
    S = cfg.diffusion_steps

    model = DiTTO(cfg.hidden_dim, cfg.num_layers, cfg.num_heads, cfg.time_dim, cfg.text_dim, S,
                  fp8_linear=cfg.fp8_linear)
    model.load_state_dict(synthetic_state_dict(cfg, seed=1234))
    model = model.to(dev).eval()
    sg = SpeechGenerator(ditto_model=model, device=dev, diffusion_steps=S)
    eng = model.engine(dev)

    use_graph = args.graph == "on" or (args.graph == "auto" and B * N <= 8192)
    main_run = StepRunner(eng, sg, cfg, B, N, T, dev, 1000 + rank, use_graph, args.noise == "seeded")
    seeded_noise = main_run.seeded

    def sync():
        torch.cuda.synchronize(dev)
        dist.barrier()

    def max_over_ranks(v):
        el = torch.tensor([v], device=dev, dtype=torch.float64)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item())

    with torch.no_grad():
        # ---- the contract's timed region: W warm-up steps, then EXACTLY K steps between barrier + synchronize ----
        main_run.run(0, args.warmup)
        sync()
        t0 = time.perf_counter()
        main_run.run(0, args.steps)
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
        dist.barrier()
        # every rank's own time for the K steps (a straggler GPU or link shows here the first time a node is available);
        # the contract's figure is the MAX
        per_rank = torch.zeros(world, device=dev, dtype=torch.float64)
        per_rank[rank] = elapsed
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
        rank_ms = [float(v) / args.steps * 1e3 for v in per_rank.tolist()]
        elapsed = max_over_ranks(elapsed)

        # ---- extra loops, HIP-event-timed on the compute stream (median of >= 5) ----
        loop_ms = []
        for _ in range(max(args.loops, 0)):
            sync()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            main_run.run(0, args.steps)
            e1.record()
            e1.synchronize()
            loop_ms.append(max_over_ranks(e0.elapsed_time(e1)) / args.steps)

        # ---- roofline pass (rank 0): per-kernel-class HIP-event timing of the same steps, eager ----
        roof, classes = None, {}
        if rank == 0 and args.profile_steps > 0:
            roof, classes = profile_classes(eng, main_run, cfg, B, N, T, args.profile_steps)
        # ---- in-run parity of the timed shape against the oracle (rank 0; every world size) ----
        parity = None
        if rank == 0 and not args.no_parity:
            parity = parity_check(model, cfg, B, N, T, dev)
        dist.barrier()

        # ---- B sweep (N = 1 runs): the same step at B in {1, 8, 32} per GPU ----
        sweep = []
        if world == 1 and not args.no_sweep:
            for b in (1, 8, 32):
                if b == B:
                    ms = sorted(loop_ms)[len(loop_ms) // 2] if loop_ms else elapsed / args.steps * 1e3
                else:
                    r = StepRunner(eng, sg, cfg, b, N, T, dev, 2000 + b, use_graph, args.noise == "seeded")
                    r.run(0, 5)
                    torch.cuda.synchronize(dev)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    r.run(0, args.steps)
                    e1.record()
                    e1.synchronize()
                    ms = e0.elapsed_time(e1) / args.steps
                    del r
                fl = b * (cfg.flops_per_utt_step(N, T, cached_kv=True) +
                          cfg.flops_text_kv(T) * ((args.steps + S - 1) // S) / args.steps)
                sweep.append({"batch_per_gpu": b, "ms_per_step": ms, "value": b / (ms * 1e-3),
                              "step_tflops": fl / (ms * 1e-3) / 1e12})
                if b == 1 and b != B:
                    # the explicit split-K rule of rounds 1-3 (hip.set_low_latency: a workgroup target, so the K partition depends
                    # on the batch size).  The default entry above already runs the low-latency CLASS of round 4 (splits by K only)
                    from ditto_tts_amd import hip as _hip
                    _hip.set_low_latency(True)
                    try:
                        r = StepRunner(eng, sg, cfg, b, N, T, dev, 2000 + b, use_graph, args.noise == "seeded")
                        r.run(0, 5)
                        torch.cuda.synchronize(dev)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        r.run(0, args.steps)
                        e1.record()
                        e1.synchronize()
                        ms_ll = e0.elapsed_time(e1) / args.steps
                        del r
                    finally:
                        _hip.set_low_latency(False)
                    sweep.append({"batch_per_gpu": b, "mode": "explicit split-K rule (hip.set_low_latency)", "ms_per_step": ms_ll,
                                  "value": b / (ms_ll * 1e-3), "step_tflops": fl / (ms_ll * 1e-3) / 1e12})

        # ---- C3 as a strong-scaling point: global batch on rank 0 -> scatter -> 50-step loops -> gather ----
        c3 = None
        if not args.no_c3 and args.config == "C2":
            G, mb = args.global_batch, p["B"]
            gen = torch.Generator(device=dev)
            gen.manual_seed(77)
            text_full = torch.randn(G, T, cfg.text_dim, device=dev, generator=gen) if rank == 0 else None
            xT_full = torch.randn(G, N, cfg.hidden_dim, device=dev, generator=gen) if rank == 0 else None

            def shard_loop(text_s, x_s, first):
                """The 50-step loop over this rank's shard, micro-batches of `mb` utterances; every step's noise comes from
                the per-utterance generator keyed by the GLOBAL utterance index (ditto_p_sample_seeded), so the gathered
                latents are the same bits at every world size."""
                out = torch.empty_like(x_s)
                tt = torch.empty(mb, device=dev, dtype=torch.long)
                for a in range(0, x_s.shape[0], mb):
                    b_ = min(a + mb, x_s.shape[0])
                    n = b_ - a
                    seeds = torch.arange(first + a, first + b_, device=dev, dtype=torch.long) + 5000
                    x = x_s[a:b_].clone()
                    cond = eng.prepare_text(text_s[a:b_], N)
                    for t_val in reversed(range(S)):
                        tt[:n].fill_(t_val)
                        eng.p_sample_seeded_(x, cond, tt[:n], seeds, t_val, sg.betas, sg.alphas, sg.alphas_cumprod)
                    out[a:b_] = x
                return out

            sync()
            phases = {}
            t0 = time.perf_counter()
            lat = sample_sharded(shard_loop, text_full, xT_full, (T, cfg.text_dim), (N, cfg.hidden_dim), dev,
                                 phases=phases, sync=lambda: torch.cuda.synchronize(dev), text_dtype=torch.bfloat16)
            torch.cuda.synchronize(dev)
            total = time.perf_counter() - t0
            dist.barrier()
            total = max_over_ranks(total)
            ph = {k: max_over_ranks(v) for k, v in sorted(phases.items())}
            ok, digest = True, None
            if rank == 0:
                ok = bool(torch.isfinite(lat).all().item()) and tuple(lat.shape) == (G, N, cfg.hidden_dim)
                # a world-size-independent fingerprint of the gathered latents (same value at N = 1, 2, 4, 8 <=> sharding
                # changed no bit): the sum of the fp32 bit patterns, mod 2^63
                digest = int(lat.view(torch.int32).to(torch.int64).sum().item() & 0x7FFFFFFFFFFFFFFF)
            # the checkable prediction: sharding changes no bit, so the digest at N = 2 / 4 / 8 must be the N = 1
            # value of the SAME source tree; the N = 1 run of the final tree recorded it (--write-c3-expect)
            tree = tree_sha16()
            expect = None
            try:
                e = json.load(open(os.path.join(ROOT, "profiles", "c3_digest_expect.json")))
                if e.get("tree_sha16") == tree and e.get("global_batch") == G and e.get("steps") == S:
                    expect = int(e["latents_digest"])
            except (OSError, ValueError, KeyError, TypeError):
                pass
            if rank == 0 and world == 1 and args.write_c3_expect:
                json.dump({"tree_sha16": tree, "global_batch": G, "steps": S, "latents_digest": digest,
                           "note": "bench.py --write-c3-expect at N = 1: the C3 latents digest every N must reproduce for this source tree"},
                          open(os.path.join(ROOT, "profiles", "c3_digest_expect.json"), "w"), indent=1)
                expect = digest
            c3 = {"global_batch": G, "micro_batch": mb, "steps": S, "scaling": "strong", "n_gpus": world,
                  "tree_sha16": tree, "digest_expected_from_n1": expect,
                  "digest_matches_n1": (None if expect is None or digest is None else bool(digest == expect)),
                  "prediction": "latents_digest(N) == latents_digest(1) of the same tree; value(N) >= 0.9 * N * value(1) of the weak-scaling line",
                  "value": G * S / total, "unit": "utterance-steps/s", "total_s": total, **ph,
                  "comm_bytes_per_peer": (G // world) * (T * cfg.text_dim * 2 + 2 * N * cfg.hidden_dim * 4) if world > 1 else 0,
                  "latents_finite_and_complete": ok, "latents_digest": digest,
                  "note": "scatter (text bf16 + x_T fp32, grouped RCCL send/recv) and gather (latents fp32) are inside "
                          "total_s; phase figures are the max over ranks"}
            del text_full, xT_full, lat

    # ---- the other configurations and the training step, witnessed by this same run (N = 1, default config) ----
    other = None
    if world == 1 and args.config == "C2" and not args.no_other_configs:
        del main_run
        torch.cuda.empty_cache()
        cache = {(cfg.hidden_dim, cfg.num_layers, cfg.num_heads): model.state_dict()}
        other = []
        for name in ("C4", "C5", "C5_bf16"):
            other.append(side_config(name, dev, args.side_steps, min(args.profile_steps, 3) or 1, cache))
        other.append(train_step_line(dev, cache))

    ms_per_step = elapsed / args.steps * 1e3
    value = B * world * args.steps / elapsed
    # executed FLOPs: cached-KV step count + the one-off text K/V GEMM amortised over the steps it was run for
    n_pre = (args.steps + S - 1) // S
    one_off = cfg.flops_text_kv(T) * n_pre / args.steps
    step_flops = B * (cfg.flops_per_utt_step(N, T, cached_kv=True) + one_off)
    step_tflops = step_flops / (ms_per_step * 1e-3) / 1e12
    step_frac = step_min_seconds(cfg, B, N, T, one_off) / (ms_per_step * 1e-3)

    if rank == 0:
        out = {
            "metric": f"DiT denoise steps/sec ({cfg.num_layers}L, d={cfg.hidden_dim}, latent_len={N})",
            "value": value, "unit": "utterance-steps/s", "n_gpus": dist.get_world_size(), "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp8(e4m3) linear + bf16 attention" if cfg.fp8_linear else "bf16", "data": "synthetic",
            "config": {"workload": f"{args.config}: DiTTO {cfg.num_layers}L d={cfg.hidden_dim} h={cfg.num_heads} "
                                   f"N={N} T={T}, {S}-step DDPM sampling loop (forward + update + noise draw), "
                                   f"B={B} utterances per GPU, text K/V cached per utterance batch",
                       "batch_per_gpu": B, "global_batch": B * world, "latent_len": N, "text_len": T,
                       "parallelism": f"batch-parallel x{world}, weights replicated, no data-path collective",
                       "hip_graph": bool(use_graph),
                       "noise": "per-utterance Philox4x32-10 inside the update kernel" if seeded_noise else "torch generator -> noise tensor"},
            "step_tflops_per_gpu": step_tflops, "step_frac_of_mfma_peak": step_frac,
            "per_rank_ms_per_step": rank_ms, "rank_min_ms": min(rank_ms), "rank_max_ms": max(rank_ms),
            "loops": {"n": len(loop_ms), "timer": "HIP events on the compute stream, max over ranks",
                      "ms_per_step": loop_ms,
                      "median_ms_per_step": sorted(loop_ms)[len(loop_ms) // 2] if loop_ms else None,
                      "median_value": (B * world / (sorted(loop_ms)[len(loop_ms) // 2] * 1e-3)) if loop_ms else None},
            "sweep": sweep or None,
            "c3_strong": c3,
            "roofline": roof,
            "floor": floor_table(cfg, B, N, T, classes) if classes else None,
            "residual_stream": "bf16 (fp32 in accumulators and LayerNorm statistics)" if stream_is_bf16(cfg, B, N) else "fp32",
            "parity": parity,
            "other_configs": other,
            "kernel_classes": classes,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, N, T)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0 and parity is not None and not parity["ok"]:
        raise SystemExit(f"bench.py: parity of the timed shape against the oracle FAILED: rel-L2 {parity['rel_l2']:.3e} > "
                         f"{parity['tol']:.0e}")


if __name__ == "__main__":
    main()
