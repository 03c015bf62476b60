#!/usr/bin/env python3
"""Turn the two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; one --pmc pass each, no trace flags) over
`bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0` into profiles/<round>_pmc_traffic.json: per
kernel class, average KB per launch and traffic_bytes = 2*FETCH*1024 + WRITE*1024 (gfx950: FETCH_SIZE counts a wide
streaming read at half its bytes, MI355X_MICROARCH.md §HBM).
    python tools/pmc_traffic.py <fetch_csv> <write_csv> <out_json>"""
import collections
import csv
import json
import sys

CLASSES = [("gemm_gated_mlp", "gemm256_kernel<3"), ("gemm_qkv_rope", "gemm256_kernel<2"), ("gemm_q_proj", "gemm192_kernel<0"),
           ("gemm_q_proj", "gemm_lnq_kernel"),      # round 4: norm2 fused into the q-projection (the class keeps its name)
           ("gemm_final", "gemm192_kernel<4"), ("attn_self", "attn64v2_kernel<true"), ("attn_cross", "attn64v2_kernel<false"),
           ("attn_self", "attn64p_kernel<true"), ("attn_cross", "attn64p_kernel<false"),   # round 6 (a later match overrides an earlier one)
           ("attn_self", "attn64q_kernel<true"), ("attn_cross", "attn64q_kernel<false"),
           ("layernorm", "ln_kernel<3, 0>"), ("adaln", "ln_kernel<3, 1>"), ("p_sample_update", "p_sample_update_seeded_kernel")]
# the cross out-projection (+ norm3) and fc2 (+ next norm1) are the SAME kernel (gemm_fr_kernel<true, true>) at K = d and
# K = 4d: its launches are told apart by their own read traffic (fc2 reads a 4x wider A), position by position in both passes
FR = "gemm_frd_kernel<true, true"       # the full-row kernel with W fetched straight into registers (gemm_frd.hip; round 4: <.., HB>)


def avg(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].replace("(anonymous namespace)::", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def series(path, counter, pat):
    rows = [(int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in csv.DictReader(open(path))
            if r["Counter_Name"] == counter and pat in r["Kernel_Name"]]
    return [v for _, v in sorted(rows)]


fetch, write = avg(sys.argv[1], "FETCH_SIZE"), avg(sys.argv[2], "WRITE_SIZE")
out = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, bench.py --steps 2 --warmup 1, C2 B=32). "
                "Units KB per launch. gfx950 correction: hbm_bytes = 2*FETCH*1024 + WRITE*1024; FETCH counts L2 misses "
                "served by the Infinity Cache too."}
for cls, pat in CLASSES:
    ks = [k for k in fetch if pat in k]
    if not ks:
        continue
    k = ks[0]
    f, w = fetch[k], write.get(k, 0.0)
    out[cls] = {"kernel": k[:90], "fetch_kb": round(f, 1), "write_kb": round(w, 1), "traffic_bytes": int(2 * f * 1024 + w * 1024)}
fs, ws = series(sys.argv[1], "FETCH_SIZE", FR), series(sys.argv[2], "WRITE_SIZE", FR)
if not fs:
    FR = "gemm_fr_kernel<true, true>"       # older builds / fr_tile 128
    fs, ws = series(sys.argv[1], "FETCH_SIZE", FR), series(sys.argv[2], "WRITE_SIZE", FR)
if fs and len(fs) == len(ws):
    mid = (min(fs) + max(fs)) / 2
    for cls, sel in (("gemm_out_proj", lambda v: v < mid), ("gemm_fc2", lambda v: v >= mid)):
        idx = [i for i, v in enumerate(fs) if sel(v)]
        if idx:
            f, w = sum(fs[i] for i in idx) / len(idx), sum(ws[i] for i in idx) / len(idx)
            out[cls] = {"kernel": FR + (" K=d" if cls == "gemm_out_proj" else " K=4d"), "launches": len(idx), "fetch_kb": round(f, 1),
                        "write_kb": round(w, 1), "traffic_bytes": int(2 * f * 1024 + w * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
