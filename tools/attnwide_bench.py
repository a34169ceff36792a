#!/usr/bin/env python3
"""Timing of the wide-head (hd 64 / 128) attention of the latent UNet (B=64: 256 tokens x 512 ch, 64 tokens x 1024 ch).
Developer tool: `python tools/attnwide_bench.py [--lib path/to/alt/libdm_hip.so]`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from diffusion_model_amd import _lib  # noqa: E402

if "--lib" in sys.argv:
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from diffusion_model_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
heads = 8


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


for B, N, C in ((64, 256, 512), (64, 64, 1024), (176, 256, 512), (176, 64, 1024)):
    g = torch.Generator().manual_seed(0)
    q, k, v, d_o = ((torch.randn(B, N, C, generator=g) * 0.7).to(dev).to(torch.bfloat16) for _ in range(4))
    for p in (0.1, 0.0):
        o, lse = K.attn_fwd(q, k, v, heads, drop_p=p, drop_seed=5)
        tf = timeit(lambda: K.attn_fwd(q, k, v, heads, drop_p=p, drop_seed=5))
        tb = timeit(lambda: K.attn_bwd(q, k, v, o, lse, d_o, heads, drop_p=p, drop_seed=5))
        fl = 4.0 * B * heads * N * N * (C // heads)
        print(f"B={B} N={N} C={C} drop {p}: fwd {tf:7.1f} us ({fl / tf / 1e6:6.1f} TF/s)  bwd {tb:7.1f} us "
              f"({2.5 * fl / tb / 1e6:6.1f} TF/s)", flush=True)
