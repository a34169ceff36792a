#!/usr/bin/env python3
"""Sweep dm_gemm over shapes / flags against torch (developer tool)."""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from diffusion_model_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
bad = 0
for M, Kd, N in itertools.product([2, 64, 288, 1152, 4608, 9216], [64, 128, 192, 256, 384, 512, 768], [64, 128, 256, 384, 768]):
    for flags in range(8):
        strided, use_res, f32out = flags & 1, flags & 2, flags & 4
        ld = Kd + 64 if strided else Kd
        xb = torch.randn(M, ld, device=dev).to(torch.bfloat16)
        x = xb[:, 32:32 + Kd] if strided else xb
        w = torch.randn(N, Kd, device=dev) * 0.05
        bias = torch.randn(N, device=dev)
        wp = K.pack_weight(w)
        res = torch.randn(M, N, device=dev).to(torch.bfloat16) if use_res else None
        ref = x.float() @ w.to(torch.bfloat16).float().t() + bias
        if res is not None:
            ref = ref + res.float()
        if f32out:
            y = torch.full((M, N), float("nan"), device=dev)
            K.gemm(x, wp, N, bias=bias, res=res, out_f32=y)
        else:
            y = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
            K.gemm(x, wp, N, bias=bias, res=res, out=y)
        err = (y.float() - ref).abs().max().item() / (ref.abs().max().item() + 1e-6)
        if not (err < 2e-2):
            bad += 1
            print("BAD", M, Kd, N, "strided" if strided else "", "res" if use_res else "", "f32" if f32out else "", err,
                  flush=True)
# second sweep: SiLU on the input, per-batch channel add, no bias, fp32 accumulate-out
for M, Kd, N in itertools.product([2, 288, 1152, 4608], [64, 128, 256], [64, 128, 256, 768]):
    for flags in range(8):
        silu, cadd, nobias = flags & 1, flags & 2, flags & 4
        x = torch.randn(M, Kd, device=dev).to(torch.bfloat16)
        w = torch.randn(N, Kd, device=dev) * 0.05
        bias = None if nobias else torch.randn(N, device=dev)
        wp = K.pack_weight(w)
        xin = torch.nn.functional.silu(x.float()).to(torch.bfloat16).float() if silu else x.float()
        ref = xin @ w.to(torch.bfloat16).float().t()
        if bias is not None:
            ref = ref + bias
        ca = None
        rpb = 0
        if cadd and M % 2 == 0:
            ca = torch.randn(2, N, device=dev)
            rpb = M // 2
            ref = ref + ca.repeat_interleave(rpb, 0)
        y = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        K.gemm(x, wp, N, bias=bias, chan_add=ca, rows_per_batch=rpb, act_silu_in=bool(silu), out=y)
        err = (y.float() - ref).abs().max().item() / (ref.abs().max().item() + 1e-6)
        if not (err < 2e-2):
            bad += 1
            print("BAD2", M, Kd, N, "silu" if silu else "", "cadd" if cadd else "", "nobias" if nobias else "", err,
                  flush=True)
print("done, bad =", bad)
