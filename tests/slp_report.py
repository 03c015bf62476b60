"""Parity figures and timing of the speech-length predictor's decoder stack (ditto_slp_forward) on one MI355X.
Lives under tests/ because it imports oracle/ (the checker), which only tests / smoke / bench's cpu_baseline may do.
    python tests/slp_report.py            # -> one JSON line per shape
Parity is against oracle.slp_decode at sizes the CPU finishes in seconds; timing uses HIP events on the current stream."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ditto_tts_amd.slp import SLP  # noqa: E402
from ditto_tts_amd.synth import hash_normal, synthetic_slp_state_dict  # noqa: E402


def rel_l2(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float(torch.linalg.norm(a - b) / torch.linalg.norm(b))


def flops(d, nhead, nl, B, S, T):
    ff = d * nhead
    per = 2 * S * d * 3 * d + 4 * S * S * d + 2 * S * d * d          # self: in_proj, scores + PV (full square), out
    per += 2 * S * d * d + 2 * T * d * 2 * d + 4 * S * T * d + 2 * S * d * d
    per += 4 * S * d * ff
    return B * nl * per


def main():
    from oracle import ditto_oracle as O   # checker only
    for (d, nhead, nl, B, S, T, check) in [(1472, 4, 4, 2, 128, 32, True), (1472, 1, 1, 8, 512, 128, True),
                                           (1472, 4, 4, 8, 2048, 128, False), (1472, 1, 1, 8, 2048, 128, False)]:
        ncls = 11
        m = SLP(ncls, nhead, nl, hidden_size=d)
        sd = synthetic_slp_state_dict(d, nhead, nl, ncls, 3)
        m.load_state_dict(sd)
        m = m.to("cuda").eval()
        zt, za = hash_normal((B, T, d), "t", 3), hash_normal((B, S, d), "a", 3)
        ztd, zad = zt.cuda(), za.cuda()
        logits, dec = m.decode(ztd, zad, return_decoded=True)
        rec = {"d_model": d, "nhead": nhead, "layers": nl, "B": B, "S": S, "T": T}
        if check:
            t0 = time.time()
            wl, wd = O.slp_decode(sd, nl, nhead, zt, za)
            rec.update(cpu_oracle_s=round(time.time() - t0, 2), rel_l2_decoded=rel_l2(dec, wd),
                       rel_l2_logits=rel_l2(logits, wl))
        for _ in range(2):
            m.decode(ztd, zad)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            m.decode(ztd, zad)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        rec.update(ms=round(ms, 3), tflops=round(flops(d, nhead, nl, B, S, T) / ms / 1e9, 1))
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
