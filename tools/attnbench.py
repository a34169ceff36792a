"""Times the flash attention kernels at the UNet's self-attention shapes (B=64 per GPU):
python tools/attnbench.py [--drop 0.1] [--iters 20]"""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--drop", type=float, default=0.1)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--lib", default=None)
a = ap.parse_args()
from diffusion_model_amd import _lib  # noqa: E402
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
from diffusion_model_amd import kernels as K  # noqa: E402
dev = torch.device("cuda:0")
SHAPES = [("L2 self 48x48 C128", 2304, 128), ("L3 self 24x24 C256", 576, 256)]


def timeit(fn, iters):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, N, C in SHAPES:
    g = torch.Generator(device="cpu").manual_seed(1)
    qkv = (torch.randn(a.batch, N, 3 * C, generator=g) * 1.0).to(dev).to(torch.bfloat16)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    o, lse = K.attn_fwd(q, k, v, 8, drop_p=a.drop, drop_seed=7)
    d_o = torch.randn_like(o)
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = dqkv[..., :C], dqkv[..., C:2 * C], dqkv[..., 2 * C:]
    tf = timeit(lambda: K.attn_fwd(q, k, v, 8, drop_p=a.drop, drop_seed=7), a.iters)
    tb = timeit(lambda: K.attn_bwd(q, k, v, o, lse, d_o, 8, dq=dq, dk=dk, dv=dv, drop_p=a.drop, drop_seed=7), a.iters)
    el = a.batch * 8 * N * N
    print(f"{name:22s} drop={a.drop}: fwd {tf:8.1f} us ({el / tf / 1e6:5.2f} T elem/s)   bwd(delta+kv+q) {tb:8.1f} us", flush=True)
