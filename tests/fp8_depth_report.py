"""Report (not a test): rel-L2 of the bf16 and fp8-linear HIP paths against the fp32 oracle at 24 layers, d = 1024 (the C5
shape).  Lives under tests/ because it uses the oracle.   python tests/fp8_depth_report.py"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ditto_tts_amd.config import DiTTOConfig
from ditto_tts_amd.modules import DiTTO
from ditto_tts_amd.synth import synthetic_inputs, synthetic_state_dict
from oracle import ditto_oracle as O
L = 24
cfg = DiTTOConfig(1024, L, 16, 256, 1024, 50)
sd = synthetic_state_dict(cfg, 6)
x, text, t = synthetic_inputs(cfg, 1, 256, 128, seed=4)
want = O.ditto_forward(sd, L, 16, x, text, t)
for fp8 in (False, True):
    m = DiTTO(1024, L, 16, 256, 1024, 50, fp8_linear=fp8); m.load_state_dict(sd); m = m.cuda().eval()
    with torch.no_grad(): out = m(x.cuda(), text.cuda(), t.cuda()).cpu()
    print("24L d=1024 fp8" if fp8 else "24L d=1024 bf16", "rel-L2", float(torch.linalg.norm(out.double()-want.double())/torch.linalg.norm(want.double())))
