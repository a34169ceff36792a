#!/usr/bin/env python3
"""Timing of the L2 self-attention (B=64, N=2304, 8 heads x 16) forward / backward variants.  Developer tool."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from diffusion_model_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
B, N, C, heads = int(os.environ.get("B", 64)), 2304, 128, 8
g = torch.Generator().manual_seed(0)
q, k, v, d_o = ((torch.randn(B, N, C, generator=g) * 0.7).to(dev).to(torch.bfloat16) for _ in range(4))


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


for p in (0.1, 0.0):
    o, lse = K.attn_fwd(q, k, v, heads, drop_p=p, drop_seed=5)
    print(f"drop {p}: fwd {timeit(lambda: K.attn_fwd(q, k, v, heads, drop_p=p, drop_seed=5)):8.1f} us", flush=True)
    for sp in (False, True):
        t = timeit(lambda: K.attn_bwd(q, k, v, o, lse, d_o, heads, drop_p=p, drop_seed=5, single_pass=sp))
        print(f"drop {p}: bwd single_pass={sp}: {t:8.1f} us", flush=True)
