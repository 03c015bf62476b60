#!/usr/bin/env python3
"""bench.py — DiT denoise throughput on MI355X (BASELINE.json metric), one JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2] [--batch 32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one reverse-diffusion step (DiTTO.forward + DDPM update + the step's N(0,1) draw) over one batch
of B utterances per GPU, synthetic latents of the named shape, closed-form random weights.  Every 50th step
starts a new utterance batch (fresh x_T and the step-invariant text work: cross-attention K/V of all layers
+ text AdaLN modulation), so the one-off per-utterance work is inside the timed region, amortised as in a
real 50-step sampling loop.  value = (B x world) x K / max-over-ranks wall time.

Multi-GPU: batch-parallel, weak scaling (B per GPU fixed); no collective in the data path —
the only collectives are the timing barrier and the max-reduce of the elapsed time.

roofline: the dominant kernel class of the step, timed live with HIP events on the launch stream
(libditto_hip's per-class event profiler, a separate eager pass of the same steps right after the timed
region); achieved = algorithmic FLOPs per launch / average launch duration, peak = 2.5 PFLOP/s dense bf16.
cpu_baseline: the fp32 oracle (proved equal to the reference, tests/test_oracle_*) on the host cores,
rank 0, N=1 only, bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C2", choices=["C1", "C2", "C4", "C5", "C5_bf16", "shipped"])
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU (default: the preset's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="off",
                    help="replay the step from a HIP graph.  Measured: no gain at B = 1 / 8 (2.35 / 4.59 ms per step "
                         "either way: small batches are bound by per-kernel latency on sparse grids, not by launches)")
    return ap.parse_args()


def kernel_flops(cfg, B, N, T):
    """Algorithmic FLOPs of ONE launch of each kernel class (2 FLOP / MAC)."""
    d, M = cfg.hidden_dim, B * N
    return {
        "gemm_qkv_rope": 2.0 * M * 3 * d * d,
        "gemm_d_x_d": 2.0 * M * d * d,
        "gemm_gated_mlp": 2.0 * M * 8 * d * d,
        "gemm_fc2": 2.0 * M * d * 4 * d,
        "gemm_final": 2.0 * M * d * 2 * d,
        "attn_self": 4.0 * M * N * d,
        "attn_cross": 4.0 * M * T * d,
    }


def kernel_bytes(cfg, B, N, T):
    """Algorithmic HBM bytes of ONE launch of the memory-bound classes."""
    d, M = cfg.hidden_dim, B * N
    return {
        "layernorm": M * d * (4 + 2.0),                 # fp32 in, bf16 out
        "adaln": M * d * (4 + 4 + 2.0),                 # fp32 in, fp32 out + bf16 raw copy
        "p_sample_update": M * d * 4 * 4.0,             # x, eps, z in; x out
    }


def pmc_traffic(kernel_class, B, N, T, cfg):
    """HBM bytes per launch of `kernel_class` from the committed rocprofv3 PMC passes (profiles/r01_v8_pmc_traffic.json, tools/pmc_traffic.py:
    separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, gfx950 x2 correction on the read side).
    bench.py cannot run under the counter collector itself, so the figure is the offline one; it is only reported
    when the workload is the one those passes measured (C2, B=32), else null."""
    if not (B == 32 and N == 1024 and T == 1024 and cfg.hidden_dim == 768 and cfg.num_layers == 12):
        return None
    for name in ("r01_v8_pmc_traffic.json", "r01_v5_pmc_traffic.json", "r01_pmc_traffic.json"):   # newest counter passes first
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            return d[kernel_class]["traffic_bytes"]
        except (OSError, KeyError, ValueError):
            continue
    return None


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


def cpu_baseline(cfg, N, T, steps=12):
    """fp32 oracle on the host cores: B = 1, `steps` timed steps after 1 warm-up (bounded sample)."""
    from ditto_tts_amd.synth import hash_normal, synthetic_inputs, synthetic_state_dict
    from oracle import ditto_oracle as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    sd = synthetic_state_dict(cfg, seed=1234)
    x, text, _ = synthetic_inputs(cfg, 1, N, T, seed=7)
    betas, alphas, acp = O.sampler_tables(cfg.diffusion_steps)
    z = hash_normal(tuple(x.shape), "z", 7)
    ts = []
    with torch.no_grad():
        for i in range(steps + 1):
            t = torch.full((1,), cfg.diffusion_steps - 1 - i, dtype=torch.long)
            t0 = time.perf_counter()
            eps = O.ditto_forward(sd, cfg.num_layers, cfg.num_heads, x, text, t)
            x = O.p_sample_update(x, eps, t, betas, alphas, acp, z)
            ts.append(time.perf_counter() - t0)
    ts = sorted(ts[1:])
    med = ts[len(ts) // 2]
    return {"value": 1.0 / med, "unit": "utterance-steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"B=1, {steps} steps (median) after 1 warm-up of the same {cfg.num_layers}L/d={cfg.hidden_dim}/"
                      f"N={N}/T={T} step, fp32 torch-CPU oracle (uncached text K/V, as the reference)",
            "s_per_step": med,
            "gflops": cfg.flops_per_utt_step(N, T, cached_kv=False) / med / 1e9}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path exists in ditto_tts_amd)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm

    from ditto_tts_amd.config import PRESETS
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
    sg = SpeechGenerator(ditto_model=model, device=dev)
    eng = model.engine(dev)

    gen = torch.Generator(device=dev)
    gen.manual_seed(1000 + rank)
    text = torch.randn(B, T, cfg.text_dim, device=dev, generator=gen)
    x = torch.empty(B, N, cfg.hidden_dim, device=dev)
    z = torch.empty_like(x)
    t_tensor = torch.empty(B, device=dev, dtype=torch.long)
    state = {"cond": None}

    use_graph = args.graph == "on" or (args.graph == "auto" and B * N <= 8192)
    state["graph"] = None

    def step(i, eager=False):
        k = i % S
        if k == 0 or state["cond"] is None:       # new utterance batch: x_T and the step-invariant text work
            x.normal_(generator=gen)
            if state["cond"] is None:
                state["cond"] = eng.prepare_text(text, N)
            else:                                  # same buffers (the graph is bound to them), new contents
                eng.prepare_text_into(text, N, state["cond"])
        t_tensor.fill_(S - 1 - k)
        z.normal_(generator=gen)                  # the step's N(0,1) draw (reference: randn_like per step)
        if use_graph and not eager:
            if state["graph"] is None:
                state["graph"] = eng.capture_p_sample(x, state["cond"], t_tensor, z, sg.betas, sg.alphas,
                                                      sg.alphas_cumprod)
            state["graph"].replay()
        else:
            eng.p_sample_(x, state["cond"], t_tensor, z, sg.betas, sg.alphas, sg.alphas_cumprod)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    with torch.no_grad():
        for i in range(args.warmup):
            step(i)
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
        if world > 1:
            dist.barrier()
            el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            elapsed = float(el.item())

        # ---- roofline pass (rank 0): per-kernel-class HIP-event timing of the same steps, eager ----
        roof, classes = None, {}
        if rank == 0 and args.profile_steps > 0:
            eng.profile_enable(True)
            for i in range(args.profile_steps):
                step(1 + i, eager=True)              # k != 0: pure denoise steps, eager so events bracket launches
            torch.cuda.synchronize(dev)
            prof = eng.profile_read()
            eng.profile_enable(False)
            kf, kb = kernel_flops(cfg, B, N, T), kernel_bytes(cfg, B, N, T)
            tot_ms = sum(ms for _, ms in prof.values()) or 1.0
            for name, (n, ms) in prof.items():
                if n == 0:
                    continue
                avg = ms / n
                ent = {"launches_per_step": n / args.profile_steps, "avg_ms": avg, "share": ms / tot_ms}
                if name in kf:
                    ent["tflops"] = kf[name] / (avg * 1e-3) / 1e12
                    ent["frac_mfma"] = ent["tflops"] / PEAK_BF16_TFLOPS
                if name in kb:
                    ent["gbs"] = kb[name] / (avg * 1e-3) / 1e9
                    ent["frac_hbm"] = ent["gbs"] / PEAK_HBM_GBS
                classes[name] = ent
            dom = max(classes, key=lambda k: classes[k]["share"])
            e = classes[dom]
            if "tflops" in e:
                roof = {"bound": "mfma", "kernel": dom, "achieved": e["tflops"], "peak": PEAK_BF16_TFLOPS,
                        "unit": "TFLOP/s", "frac": e["frac_mfma"], "traffic": pmc_traffic(dom, B, N, T, cfg),
                        "avg_launch_ms": e["avg_ms"], "flops_per_launch": kf[dom]}
            else:
                roof = {"bound": "hbm", "kernel": dom, "achieved": e["gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": e["frac_hbm"], "traffic": pmc_traffic(dom, B, N, T, cfg), "avg_launch_ms": e["avg_ms"],
                        "bytes_per_launch": kb[dom]}

    ms_per_step = elapsed / args.steps * 1e3
    value = B * world * args.steps / elapsed
    # executed FLOPs: cached-KV step count + the one-off text K/V GEMM amortised over the steps it was run for
    n_pre = (args.steps + S - 1) // S
    step_flops = B * (cfg.flops_per_utt_step(N, T, cached_kv=True) + cfg.flops_text_kv(T) * n_pre / args.steps)
    step_tflops = step_flops / (ms_per_step * 1e-3) / 1e12

    if rank == 0:
        out = {
            "metric": f"DiT denoise steps/sec ({cfg.num_layers}L, d={cfg.hidden_dim}, latent_len={N})",
            "value": value, "unit": "utterance-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp8(e4m3) linear + bf16 attention" if cfg.fp8_linear else "bf16", "data": "synthetic",
            "config": {"workload": f"{args.config}: DiTTO {cfg.num_layers}L d={cfg.hidden_dim} h={cfg.num_heads} "
                                   f"N={N} T={T}, {S}-step DDPM sampling loop (forward + update + noise draw), "
                                   f"B={B} utterances per GPU, text K/V cached per utterance batch",
                       "batch_per_gpu": B, "global_batch": B * world, "latent_len": N, "text_len": T,
                       "parallelism": f"batch-parallel x{world}, weights replicated, no data-path collective",
                       "hip_graph": bool(use_graph)},
            "step_tflops_per_gpu": step_tflops, "step_frac_of_mfma_peak": step_tflops / PEAK_BF16_TFLOPS,
            "roofline": roof,
            "kernel_classes": classes,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, N, T)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
