#!/usr/bin/env python3
"""Achieved GB/s of the HBM-bound kernels at the STL10 B=64 shapes, against ALGORITHMIC bytes
(GN+SiLU 4 B/elem fwd, 6 B/elem bwd at bf16; optimizer 36 B/param; sampler step 20 B/elem;
q-sample 16 B/elem).  Developer tool; its output is committed under profiles/."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_model_amd import kernels as K

dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) * 1e-3 / reps


def report(name, sec, nbytes):
    print(f"{name:44s} {sec*1e6:9.1f} us  {nbytes/sec/1e9:8.0f} GB/s  ({nbytes/1e6:.1f} MB algorithmic)", flush=True)


B = 64
for H, C in [(96, 64), (48, 128), (24, 256), (48, 256), (96, 128)]:
    x = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    y = torch.empty_like(x); dx = torch.empty_like(x)
    _, stats = K.gn_silu_fwd(x, gam, bet, 8, out=y)
    n = x.numel()
    report(f"gn_silu_fwd  {C}@{H}x{H}", timeit(lambda: K.gn_silu_fwd(x, gam, bet, 8, out=y)), 4 * n)
    report(f"gn_silu_bwd  {C}@{H}x{H}", timeit(lambda: K.gn_silu_bwd(x, dy, gam, bet, stats, 8, dg, db, out=dx)), 6 * n)

N = 4177539 // 4 * 4 + 4
p, g, m, v, ema = (torch.randn(N, device=dev) * 0.01 for _ in range(5))
v = v.abs()
norm = torch.zeros(2, device=dev)
report("grad_norm_clipcoef (4.18 M params)", timeit(lambda: K.grad_norm_clipcoef(g, 1.0, out=norm)), 4 * N)
report("adamw_ema_step (4.18 M params)",
       timeit(lambda: K.adamw_ema_step(p, g, m, v, ema, norm, lr=5e-5, beta1=0.9, beta2=0.99, eps=1e-8, wd=1e-4,
                                       step=3, ema_decay=0.995)), 32 * N)
for S in (88, 704):
    z, vu, vc, nz = (torch.randn(S, 3, 96, 96, device=dev) for _ in range(4))
    coef = torch.rand(S, 8, device=dev)
    out = torch.empty_like(z)
    report(f"cfg_ddpm_step S={S}", timeit(lambda: K.cfg_ddpm_step(z, vu, vc, nz, coef, out=out)), 20 * z.numel())
    coef2 = coef.clone(); coef2[:, 6] = 1.0   # second-order rows: reads z, v_u, v_c, x0_prev; writes z', x0
    x0o = torch.empty_like(z)
    report(f"cfg_dpmpp2m_step S={S}", timeit(lambda: K.cfg_dpmpp2m_step(z, vu, vc, nz, coef2, out=out, x0_out=x0o)),
           24 * z.numel())
x, e = torch.randn(B, 3, 96, 96, device=dev), torch.randn(B, 3, 96, 96, device=dev)
ab = torch.rand(B, device=dev)
report("qsample_vtarget B=64", timeit(lambda: K.qsample_vtarget(x, e, ab)), 16 * x.numel())
