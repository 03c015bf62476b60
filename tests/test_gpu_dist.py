"""-m gpu: the multi-GPU host path on the one GPU a test box has — RCCL ("nccl") process group of world size 1 on
127.0.0.1.  The 2-rank behaviour is covered on CPU/gloo (tests/test_dist_gloo.py); this checks that the same code runs on
the real backend with CUDA tensors (RCCL initialisation, point-to-point group ops, the gradient bucket path)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from ditto_tts_amd.config import DiTTOConfig
from ditto_tts_amd.dist import allreduce_gradients, gather_batch, sample_sharded, scatter_batch
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

    want = fn(text, xT, 0)
    got = sample_sharded(fn, text, xT, (T, 128), (N, 128), DEV)
    assert torch.equal(got, want)
    sh = scatter_batch(text, (T, 128), torch.float32, DEV)
    assert torch.equal(gather_batch(sh, B), text)


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
