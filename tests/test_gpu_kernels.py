"""GPU: every C-ABI kernel against a plain PyTorch fp32 statement of the same op on the same
(bf16-rounded) inputs.  Tolerances: bf16-output kernels rtol 1e-2 (+1e-2*max|ref| floor, one
bf16 ulp is 0.4-0.8 %); fp32 kernels 1e-5..1e-4 as written per test."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close, rel_l2

pytestmark = pytest.mark.gpu

BF = 1e-2


def dev():
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev())


def nhwc(t):  # NCHW fp32 -> NHWC bf16
    return t.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)


def nchw(t):  # NHWC bf16 -> NCHW fp32
    return t.float().permute(0, 3, 1, 2).contiguous()


def q16(t):  # bf16-round in fp32
    return t.to(torch.bfloat16).float()


@pytest.fixture(scope="module")
def K():
    from diffusion_model_amd import kernels
    return kernels


# ------------------------------------------------------------------ conv forward
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 16, 16, 64, 64), (1, 32, 48, 128, 128), (2, 24, 24, 128, 256),
                                            (2, 8, 8, 64, 128), (1, 96, 96, 64, 64), (2, 12, 12, 64, 64), (1, 20, 36, 64, 128),
                                            # 8x8 images, B >= 4: four images per workgroup tile (mode 0), ragged last quad
                                            (8, 8, 8, 128, 256), (5, 8, 8, 64, 64), (64, 8, 8, 64, 1024)])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_conv3x3_fwd(K, B, H, W, Cin, Cout, mode):
    if mode == 1 and (H % 2 or W % 2):
        pytest.skip("stride-2 needs 16-divisible input")
    if mode == 2 and H >= 96:
        pytest.skip("big")
    x = q16(rnd(B, Cin, H, W, seed=1))
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=1.0 / math.sqrt(9 * Cin))
    bias = rnd(Cout, seed=3, scale=0.1)
    ca = rnd(B, Cout, seed=4, scale=0.1)
    xi = F.interpolate(x, scale_factor=2, mode="nearest") if mode == 2 else x
    ref = F.conv2d(xi, q16(w), bias, stride=2 if mode == 1 else 1, padding=1) + ca[:, :, None, None]
    res = q16(rnd(*ref.shape, seed=5))
    wp = K.pack_weight(w)
    y = K.conv3x3(nhwc(x), wp, Cout, bias=bias, chan_add=ca, res=nhwc(res), mode=mode)
    assert_close(nchw(y), ref + res, BF, f"conv mode {mode}")


def test_conv3x3_silu_in_f32_out_and_slices(K):
    B, H, W, Cin, Cout = 2, 16, 16, 64, 64
    x = q16(rnd(B, Cin, H, W, seed=1))
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=1.0 / math.sqrt(9 * Cin))
    ref = F.conv2d(q16(F.silu(x)), q16(w), None, padding=1)
    # input read from, and output written into, channel slices of wider buffers
    xb = torch.zeros(B, H, W, 192, dtype=torch.bfloat16, device=dev())
    xb[..., 64:128] = nhwc(x)
    yb = torch.full((B, H, W, 128), 7.0, dtype=torch.bfloat16, device=dev())
    K.conv3x3(xb[..., 64:128], K.pack_weight(w), Cout, act_silu_in=True, out=yb[..., 64:128])
    assert_close(nchw(yb[..., 64:128]), ref, BF, "slice out")
    assert float((yb[..., :64].float() - 7.0).abs().max()) == 0.0
    yf = torch.empty(B, H, W, Cout, dtype=torch.float32, device=dev())
    K.conv3x3(xb[..., 64:128], K.pack_weight(w), Cout, act_silu_in=True, out_f32=yf)
    assert_close(yf.permute(0, 3, 1, 2), ref, 2e-3, "fp32 out")


# ------------------------------------------------------------------ conv data-gradient
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 16, 16, 64, 128), (1, 48, 48, 128, 64), (1, 24, 24, 256, 128), (2, 12, 12, 64, 64), (1, 20, 36, 128, 64),
                                            (6, 8, 8, 128, 64)])
def test_conv3x3_dgrad_s1(K, B, H, W, Cin, Cout):
    x = rnd(B, Cin, H, W, seed=1).requires_grad_()
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=1.0 / math.sqrt(9 * Cin))
    dy = q16(rnd(B, Cout, H, W, seed=3))
    (ref,) = torch.autograd.grad(F.conv2d(x, q16(w), padding=1), x, dy)
    dx = K.conv3x3(nhwc(dy), K.pack_weight(w, transpose=True), Cin, mode=0)
    assert_close(nchw(dx), ref, BF, "dgrad s1")


def test_conv3x3_dgrad_s2_zero_insert(K):
    B, H, W, Cin, Cout = 2, 32, 32, 64, 64
    x = rnd(B, Cin, H, W, seed=1).requires_grad_()
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=1.0 / math.sqrt(9 * Cin))
    dy = q16(rnd(B, Cout, H // 2, W // 2, seed=3))
    (ref,) = torch.autograd.grad(F.conv2d(x, q16(w), stride=2, padding=1), x, dy)
    dx = K.conv3x3(nhwc(dy), K.pack_weight(w, transpose=True), Cin, mode=3)
    assert_close(nchw(dx), ref, BF, "dgrad s2")


def test_conv3x3_dgrad_upsample_pool2(K):
    B, H, W, Cin, Cout = 2, 16, 16, 128, 128
    x = rnd(B, Cin, H, W, seed=1).requires_grad_()
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=1.0 / math.sqrt(9 * Cin))
    dy = q16(rnd(B, Cout, 2 * H, 2 * W, seed=3))
    y = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), q16(w), padding=1)
    (ref,) = torch.autograd.grad(y, x, dy)
    dx = K.conv3x3(nhwc(dy), K.pack_weight(w, transpose=True), Cin, mode=0, pool2=True)
    assert tuple(dx.shape) == (B, H, W, Cin)
    assert_close(nchw(dx), ref, BF, "dgrad up")


# ------------------------------------------------------------------ gemm
@pytest.mark.parametrize("M,Kd,N", [(64, 256, 256), (1000, 64, 128), (4608, 128, 384), (300, 32, 64),
                                    (2, 512, 64), (513, 64, 32), (700, 256, 16)])
def test_gemm(K, M, Kd, N):
    x = q16(rnd(M, Kd, seed=1))
    w = rnd(N, Kd, seed=2, scale=1.0 / math.sqrt(Kd))
    bias = rnd(N, seed=3, scale=0.1)
    ref = x @ q16(w).t() + bias
    y = K.gemm(x.to(torch.bfloat16), K.pack_weight(w), N, bias=bias)
    assert_close(y.float(), ref, BF, "gemm")
    yf = torch.empty(M, N, dtype=torch.float32, device=dev())
    K.gemm(x.to(torch.bfloat16), K.pack_weight(w), N, bias=bias, out_f32=yf)
    assert_close(yf, ref, 2e-3, "gemm f32 out")
    # data-gradient operand
    dy = q16(rnd(M, N, seed=4))
    dx = K.gemm(dy.to(torch.bfloat16), K.pack_weight(w, transpose=True), Kd)
    assert_close(dx.float(), dy @ q16(w), BF, "gemm dgrad")


def test_gemm_chan_add_rows_per_batch(K):
    B, R, Kd, N = 3, 50, 64, 64
    x = q16(rnd(B * R, Kd, seed=1))
    w = rnd(N, Kd, seed=2, scale=0.2)
    ca = rnd(B, N, seed=3)
    ref = (x @ q16(w).t()).view(B, R, N) + ca[:, None, :]
    y = K.gemm(x.to(torch.bfloat16), K.pack_weight(w), N, chan_add=ca, rows_per_batch=R)
    assert_close(y.float().view(B, R, N), ref, BF, "gemm chan_add")


# ------------------------------------------------------------------ weight gradient
@pytest.mark.parametrize("B,H,W,Cin,Cout,mode", [(2, 16, 16, 64, 64, 0), (3, 32, 32, 128, 64, 0),
                                                 (2, 24, 24, 128, 256, 0), (2, 32, 32, 64, 64, 1),
                                                 (2, 16, 16, 128, 128, 2), (2, 8, 8, 64, 64, 0),
                                                 (1, 96, 96, 64, 64, 0), (2, 12, 12, 64, 64, 0), (1, 20, 36, 64, 128, 0), (2, 12, 12, 64, 64, 2), (2, 24, 24, 64, 64, 1)])
def test_wgrad_conv(K, B, H, W, Cin, Cout, mode):
    x = q16(rnd(B, Cin, H, W, seed=1))
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=0.05).requires_grad_()
    bias = torch.zeros(Cout, device=dev(), requires_grad=True)
    xi = F.interpolate(x, scale_factor=2, mode="nearest") if mode == 2 else x
    y = F.conv2d(xi, w, bias, stride=2 if mode == 1 else 1, padding=1)
    dy = q16(rnd(*y.shape, seed=3))
    gw, gb = torch.autograd.grad(y, [w, bias], dy)
    dw = torch.ones_like(gw)  # accumulate semantics: starts at 1
    db = torch.ones(Cout, device=dev())
    K.wgrad(nhwc(x), nhwc(dy), dw, db, taps=9, mode=mode)
    assert_close(dw - 1.0, gw, 5e-3, f"wgrad mode {mode}")
    assert_close(db - 1.0, gb, 5e-3, "bias grad")


@pytest.mark.parametrize("M,Kd,N", [(64, 256, 256), (1000, 64, 128), (4608, 128, 384), (300, 32, 64), (2, 512, 64),
                                    (1000, 384, 128), (777, 192, 64)])   # 128-wide ci tiles (three of them); 192: 64-wide path
def test_wgrad_linear(K, M, Kd, N):
    x = q16(rnd(M, Kd, seed=1))
    dy = q16(rnd(M, N, seed=2))
    dw = torch.zeros(N, Kd, device=dev())
    db = torch.zeros(N, device=dev())
    K.wgrad(x.to(torch.bfloat16), dy.to(torch.bfloat16), dw, db, taps=1)
    assert_close(dw, dy.t() @ x, 5e-3, "linear wgrad")
    assert_close(db, dy.sum(0), 5e-3, "linear bias grad")


def test_wgrad_silu_in_and_kvalid(K):
    B, H, W, Cout = 2, 16, 16, 64
    x = torch.zeros(B, 64, H, W, device=dev())
    x[:, :3] = q16(rnd(B, 3, H, W, seed=1))
    w = rnd(Cout, 3, 3, 3, seed=2, scale=0.1).requires_grad_()
    y = F.conv2d(q16(F.silu(x[:, :3])), w, padding=1)
    dy = q16(rnd(*y.shape, seed=3))
    (gw,) = torch.autograd.grad(y, w, dy)
    dw = torch.zeros_like(gw)
    K.wgrad(nhwc(x), nhwc(dy), dw, None, taps=9, mode=0, act_silu_in=True, k_valid=3)
    assert_close(dw, gw, 5e-3, "wgrad k_valid")


# ------------------------------------------------------------------ GroupNorm + SiLU
@pytest.mark.parametrize("B,H,W,C", [(2, 12, 12, 64), (3, 24, 24, 128), (2, 16, 16, 256), (1, 96, 96, 64),
                                     (2, 16, 16, 512), (2, 8, 8, 1024)])   # last two: latent-model widths (utils.py:145-156)
@pytest.mark.parametrize("with_add", [False, True])
def test_gn_silu(K, B, H, W, C, with_add):
    x = q16(rnd(B, C, H, W, seed=1) * 1.5 + 0.3).requires_grad_()
    gamma = (1 + 0.1 * rnd(C, seed=2)).requires_grad_()
    beta = (0.1 * rnd(C, seed=3)).requires_grad_()
    ca = rnd(B, C, seed=4) if with_add else None
    xin = x + ca[:, :, None, None] if with_add else x
    ref = F.silu(F.group_norm(xin, 8, gamma, beta, 1e-5))
    dy = q16(rnd(B, C, H, W, seed=5))
    gx, gg, gb = torch.autograd.grad(ref, [x, gamma, beta], dy)
    xq = nhwc(x.detach())
    y, stats = K.gn_silu_fwd(xq, gamma.detach(), beta.detach(), 8, chan_add=ca)
    assert_close(nchw(y), ref, BF, "gn fwd")
    xg = xin.detach().view(B, 8, -1)
    assert_close(stats[..., 0], xg.mean(-1), 1e-4, "mean", atol_frac=1e-4)
    assert_close(stats[..., 1], torch.rsqrt(xg.var(-1, unbiased=False) + 1e-5), 1e-4, "rstd")
    dgamma = torch.zeros(C, device=dev())
    dbeta = torch.zeros(C, device=dev())
    dx = K.gn_silu_bwd(xq, nhwc(dy), gamma.detach(), beta.detach(), stats, 8, dgamma, dbeta, chan_add=ca)
    assert_close(nchw(dx), gx, BF, "gn dx")
    assert_close(dgamma, gg, 5e-3, "dgamma")
    assert_close(dbeta, gb, 5e-3, "dbeta")


# ------------------------------------------------------------------ attention
def ref_attn(q, k, v, heads):
    B, Nq, C = q.shape
    hd = C // heads
    qh = q.view(B, Nq, heads, hd).transpose(1, 2)
    kh = k.view(B, -1, heads, hd).transpose(1, 2)
    vh = v.view(B, -1, heads, hd).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) / math.sqrt(hd)
    o = torch.softmax(s, -1) @ vh
    return o.transpose(1, 2).reshape(B, Nq, C), torch.logsumexp(s, -1)


@pytest.mark.parametrize("B,Nq,Nk,C", [(2, 576, 576, 256), (2, 64, 64, 128), (1, 2304, 2304, 128), (2, 200, 1, 64),
                                       (2, 130, 4, 128), (1, 576, 77, 256), (2, 96, 96, 64),
                                       (2, 300, 16, 256), (1, 100, 17, 64), (3, 2304, 1, 128), (2, 130, 1, 256),
                                       # head sizes 64 / 128: the latent model's widths (utils.py:145-156)
                                       (2, 256, 256, 512), (2, 64, 64, 1024), (1, 100, 37, 512), (2, 45, 3, 1024),
                                       (1, 200, 150, 1024), (1, 330, 300, 512),   # several staged key images, ragged tails
                                       (2, 130, 1, 512), (2, 64, 1, 1024)])
def test_attention_fwd_bwd(K, B, Nq, Nk, C):
    heads = 8
    q = q16(rnd(B, Nq, C, seed=1)).requires_grad_()
    k = q16(rnd(B, Nk, C, seed=2)).requires_grad_()
    v = q16(rnd(B, Nk, C, seed=3)).requires_grad_()
    ref, lse_ref = ref_attn(q, k, v, heads)
    d_o = q16(rnd(B, Nq, C, seed=4))
    gq, gk, gv = torch.autograd.grad(ref, [q, k, v], d_o)
    qb, kb, vb = (t.detach().to(torch.bfloat16) for t in (q, k, v))
    o, lse = K.attn_fwd(qb, kb, vb, heads)
    assert_close(o.float(), ref, BF, "attn o")
    assert_close(lse, lse_ref, 2e-3, "lse", atol_frac=2e-3)
    dq, dk, dv = K.attn_bwd(qb, kb, vb, o, lse, d_o.to(torch.bfloat16), heads)
    assert_close(dv.float(), gv, 1.5e-2, "dv")
    if Nk > 1:  # with one key softmax==1 and dq, dk are identically ~0
        assert_close(dq.float(), gq, 1.5e-2, "dq")
        assert_close(dk.float(), gk, 1.5e-2, "dk")
    else:
        assert float(dq.float().abs().max()) < 1e-2 * float(gv.abs().max())


@pytest.mark.parametrize("B,N,C", [(1, 640, 128), (1, 320, 256)])
def test_attention_wide_score_range(K, B, N, C):
    """Scores spread over tens of units (exp2 domain): the forward's lazily moved reference maximum must take its
    rescale path in later key blocks, and the hi/lo-split scaled operand must keep exp(s - lse) accurate."""
    heads = 8
    q = q16(3.0 * rnd(B, N, C, seed=11)).requires_grad_()
    k = q16(3.0 * rnd(B, N, C, seed=12)).requires_grad_()
    with torch.no_grad():  # a late key that tops every earlier score of every query of head 0
        k[:, N - 5, :C // heads] = 6.0 * torch.sign(q[:, :, :C // heads].mean(1))
    v = q16(rnd(B, N, C, seed=13)).requires_grad_()
    ref, lse_ref = ref_attn(q, k, v, heads)
    d_o = q16(rnd(B, N, C, seed=14))
    gq, gk, gv = torch.autograd.grad(ref, [q, k, v], d_o)
    qb, kb, vb = (t.detach().to(torch.bfloat16) for t in (q, k, v))
    o, lse = K.attn_fwd(qb, kb, vb, heads)
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all()
    assert_close(o.float(), ref, BF, "attn o (wide)")
    assert_close(lse, lse_ref, 2e-3, "lse (wide)", atol_frac=2e-3)
    dq, dk, dv = K.attn_bwd(qb, kb, vb, o, lse, d_o.to(torch.bfloat16), heads)
    assert_close(dv.float(), gv, 2e-2, "dv (wide)")
    assert_close(dq.float(), gq, 2e-2, "dq (wide)")
    assert_close(dk.float(), gk, 2e-2, "dk (wide)")


def test_attention_fused_qkv_views(K):
    B, N, C, heads = 2, 128, 128, 8
    qkv = q16(rnd(B, N, 3 * C, seed=1)).to(torch.bfloat16)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    ref, _ = ref_attn(q.float(), k.float(), v.float(), heads)
    o, _ = K.attn_fwd(q, k, v, heads)
    assert_close(o.float(), ref, BF, "attn views")


@pytest.mark.parametrize("N,Nk,C", [(256, 256, 128), (1024, 8, 128), (1024, 1, 128),   # flash / few-token / one-token
                                    (256, 256, 512), (512, 1, 1024)])                   # wide heads
def test_attention_dropout_consistency(K, N, Nk, C):
    """Dropout masks come from a counter RNG; check keep-rate and that the backward uses the forward's mask
    (dV of a one-hot d_o equals P_drop^T restricted to that column)."""
    B, heads, p = 1, 8, 0.25
    q = q16(rnd(B, N, C, seed=1)).to(torch.bfloat16)
    k = q16(rnd(B, Nk, C, seed=2)).to(torch.bfloat16)
    v = torch.ones(B, Nk, C, device=dev(), dtype=torch.bfloat16)
    o, lse = K.attn_fwd(q, k, v, heads, drop_p=p, drop_seed=1234)
    # with V == 1: o = sum_k P_drop = (kept mass)/(1-p); mean over queries ~ 1
    assert abs(float(o.float().mean()) - 1.0) < 0.02
    o0, _ = K.attn_fwd(q, k, v, heads, drop_p=0.0)
    assert float((o0.float() - 1).abs().max()) < 2e-2
    d_o = torch.ones_like(o)
    dq, dk, dv = K.attn_bwd(q, k, v, o, lse, d_o, heads, drop_p=p, drop_seed=1234)
    # sum_keys dV[key] = sum_q sum_k P_drop[q,k] = sum_q o[q]
    assert_close(dv.float().sum(1), o.float().sum(1), 2e-2, "dropout mass")


@pytest.mark.parametrize("B,p", [(1, 0.0), (2, 0.1)])
def test_attention_single_pass_backward_matches_two_kernel_backward(K, B, p):
    """hd 16, 2304 keys (the UNet's 48x48 self-attention): dm_attn_bwd with the workspace takes attn_bwd_fused (the
    probabilities are computed once for dq, dk, dv); without it, attn_bwd_kv + attn_bwd_q.  Same dropout words, same
    arithmetic up to summation order: the two must agree far inside the bf16 output rounding, and the fused path must
    be run-to-run bit-identical.  (Against torch autograd: test_attention_fwd_bwd[1-2304-2304-128].)"""
    from diffusion_model_amd import _lib
    heads, N, C = 8, 2304, 128
    q = q16(rnd(B, N, C, seed=1)).to(torch.bfloat16)
    k = q16(rnd(B, N, C, seed=2)).to(torch.bfloat16)
    v = q16(rnd(B, N, C, seed=3)).to(torch.bfloat16)
    d_o = q16(rnd(B, N, C, seed=4)).to(torch.bfloat16)
    o, lse = K.attn_fwd(q, k, v, heads, drop_p=p, drop_seed=77)
    a = K.attn_bwd(q, k, v, o, lse, d_o, heads, drop_p=p, drop_seed=77)
    assert _lib.lib().dm_last_variant().decode().startswith("attn_bwd_fused")
    b = K.attn_bwd(q, k, v, o, lse, d_o, heads, drop_p=p, drop_seed=77, single_pass=False)
    assert _lib.lib().dm_last_variant().decode().startswith("attn_flash_bwd")
    for name, x, y in zip(("dq", "dk", "dv"), a, b):
        assert_close(x.float(), y.float(), 8e-3, name, atol_frac=4e-3)
        assert rel_l2(x, y) < 3e-3, (name, rel_l2(x, y))
    a2 = K.attn_bwd(q, k, v, o, lse, d_o, heads, drop_p=p, drop_seed=77)
    for x, y in zip(a, a2):
        assert torch.equal(x, y)


@pytest.mark.parametrize("B,N,C,p", [(2, 2304, 128, 0.0), (2, 2304, 128, 0.1), (3, 1000, 64, 0.25), (2, 576, 256, 0.1),
                                     (2, 70, 512, 0.1), (2, 33, 1024, 0.1)])
def test_cross1_matches_onekey_attention_plus_to_out(K, B, N, C, p):
    """dm_cross1_fwd / dm_cross1_bwd (one context token: dropout(v) through to_out in one streaming pass) against the
    formulation they replace, run with the SAME seed (the dropout words are shared): attn_onekey_fwd -> gemm(+res) in the
    forward; to_out data gradient -> attn_onekey_bwd, and the to_out weight gradient, in the backward.  Also against plain
    fp32 torch through the extracted mask."""
    heads, seed = 8, 4242
    v = q16(rnd(B, 1, C, seed=1)).to(torch.bfloat16)
    w = rnd(C, C, seed=2, scale=C ** -0.5).contiguous()
    res = q16(rnd(B, N, C, seed=3)).to(torch.bfloat16)
    dy = q16(rnd(B, N, C, seed=4)).to(torch.bfloat16)
    # the replaced path
    o = K.attn_onekey_fwd(v, N, heads, drop_p=p, drop_seed=seed)
    y_ref = K.gemm(o, K.pack_weight(w), C, res=res)
    d_o = K.gemm(dy, K.pack_weight(w, transpose=True), C)
    dv_ref = K.attn_onekey_bwd(v, o, d_o, heads, drop_p=p, drop_seed=seed)
    dw_ref = torch.zeros(C, C, device=dev())
    K.wgrad(o, dy, dw_ref, None, taps=1)
    # the fused path
    y = K.cross1_fwd(v, w, N, heads, res=res, drop_p=p, drop_seed=seed)
    assert K.last_variant().startswith("cross1_fwd")
    dw = torch.zeros(C, C, device=dev())
    dv = K.cross1_bwd(dy, v, w, heads, dw_out=dw, drop_p=p, drop_seed=seed)
    assert_close(y.float(), y_ref.float(), 1.5e-2, "y vs onekey + to_out")
    assert_close(dv.float(), dv_ref.float(), 1.5e-2, "dv vs onekey bwd")
    assert_close(dw, dw_ref, 1.5e-2, "dWout vs wgrad")
    # fp32 torch with the mask the kernels drew (o is mask * v / keep, exactly representable pattern: o != 0 <=> kept)
    hd = C // heads
    z = (o.float().view(B, N, heads, hd).abs().sum(-1) > 0).float() if p > 0 else torch.ones(B, N, heads, device=dev())
    keep = 1.0 - round(p * 32768) / 32768.0
    of = (z / keep).unsqueeze(-1) * v.float().view(B, 1, heads, hd)
    y32 = res.float() + of.reshape(B, N, C) @ w.t()
    assert_close(y.float(), y32, BF, "y vs fp32")
    do32 = dy.float() @ w
    dv32 = ((z / keep).unsqueeze(-1) * do32.view(B, N, heads, hd)).sum(1).reshape(B, 1, C)
    dw32 = torch.einsum("bno,bni->oi", dy.float(), of.reshape(B, N, C))
    assert_close(dv.float(), dv32, BF, "dv vs fp32")
    assert_close(dw, dw32, 2e-3, "dWout vs fp32")
    # accumulation (+=), bit-reproducibility, and the no-residual form
    dw2 = dw.clone()
    dvb = K.cross1_bwd(dy, v, w, heads, dw_out=dw2, drop_p=p, drop_seed=seed)
    assert torch.equal(dvb, dv) and torch.equal(dw2, 2 * dw)
    y0 = K.cross1_fwd(v, w, N, heads, drop_p=p, drop_seed=seed)
    assert_close(y0.float(), (y32 - res.float()), BF, "y without residual")
    # the log-sum-exp-only form of the one-key forward (o declined) returns the same lse as the full call
    qq = q16(rnd(B, N, C, seed=5)).to(torch.bfloat16)
    kk = q16(rnd(B, 1, C, seed=6)).to(torch.bfloat16)
    o_full, lse_full = K.attn_fwd(qq, kk, v, heads, drop_p=p, drop_seed=seed)
    o_none, lse_only = K.attn_fwd(qq, kk, v, heads, drop_p=p, drop_seed=seed, want_o=False)
    assert o_none is None and torch.equal(lse_only, lse_full) and torch.equal(o_full, o)


# ------------------------------------------------------------------ layout / elementwise
def test_layout_and_elementwise(K):
    x = rnd(2, 3, 16, 16, seed=1)
    y = K.nchw_to_nhwc(x, 64)
    assert tuple(y.shape) == (2, 16, 16, 64)
    assert torch.equal(y[..., :3].float(), q16(x).permute(0, 2, 3, 1))
    assert float(y[..., 3:].float().abs().max()) == 0.0
    back = K.nhwc_to_nchw(y, 3)
    assert torch.equal(back, q16(x))
    f = rnd(2, 16, 16, 32, seed=2)
    assert torch.equal(K.nhwc_to_nchw(f, 3), f[..., :3].permute(0, 3, 1, 2))
    a = q16(rnd(2, 8, 8, 64, seed=3)).to(torch.bfloat16)
    b = q16(rnd(2, 8, 8, 64, seed=4)).to(torch.bfloat16)
    buf = torch.zeros(2, 8, 8, 128, dtype=torch.bfloat16, device=dev())
    K.add(a, b, out=buf[..., 64:])
    assert_close(buf[..., 64:].float(), a.float() + b.float(), BF, "add")
    K.add(a, None, out=buf[..., :64])
    assert torch.equal(buf[..., :64], a)
    xs = a.float().requires_grad_()
    (g,) = torch.autograd.grad(F.silu(xs), xs, b.float())
    assert_close(K.silu_bwd(b, a).float(), g, BF, "silu bwd")
    cs = torch.empty(2, 64, device=dev())
    K.colsum(a, cs, 2)
    assert_close(cs, a.float().view(2, -1, 64).sum(1), 1e-4, "colsum")


def test_qsample_mse_and_sampler_step(K):
    from oracle import diffusion_ref as dref
    B = 4
    x, e = rnd(B, 3, 96, 96, seed=1), rnd(B, 3, 96, 96, seed=2)
    beta, alpha, abar = (t.to(dev()) for t in dref.cosine_schedule(500))
    t = torch.tensor([0, 17, 250, 499], device=dev())
    xt, v = K.qsample_vtarget(x, e, abar[t].contiguous())
    rxt, rv = dref.q_sample_v_target(x, e, abar[t])
    assert_close(xt, rxt, 1e-6, "x_t")
    assert_close(v, rv, 1e-6, "v")
    out = rnd(B, 3, 96, 96, seed=3).requires_grad_()
    ref_loss = dref.v_loss(out, rv, t, 0.01)
    (ref_g,) = torch.autograd.grad(ref_loss / 2, out)
    w = torch.exp(-0.01 * t.float()) / (B * 3 * 96 * 96)
    loss, dout = K.mse_loss_fwd_bwd(out.detach(), rv, w.contiguous(), grad_scale=0.5)
    assert_close(loss, ref_loss.reshape(1), 1e-5, "loss")
    assert_close(dout, ref_g, 1e-5, "dloss")
    # sampler step vs the oracle at t=499, t=2, t=1 and the final clamp step
    z, vu, vc, n = (rnd(B, 3, 96, 96, seed=s) for s in (5, 6, 7, 8))
    bc, ac = beta.cpu(), abar.cpu()
    for tt, g, eta in ((499, 6.0, 1.0), (2, 6.0, 1.0), (1, 6.0, 1.0), (300, 0.0, 0.5), (0, 6.0, 1.0)):
        tt_t = torch.full((B,), tt)
        eps = dref.cfg_eps_from_v(z.cpu(), vu.cpu(), vc.cpu(), g, ac[tt_t])
        if tt > 0:
            ref = dref.ddpm_step(z.cpu(), tt, eps, bc, ac, eta, n.cpu())
        else:
            ref = dref.final_step(z.cpu(), eps, bc, ac)
        from diffusion_model_amd.sampler import step_coefficients
        coef = step_coefficients(bc, ac, tt, g, eta, B, final=(tt == 0)).to(dev())
        got = K.cfg_ddpm_step(z, vu, vc, n, coef)
        assert_close(got.cpu(), ref, 2e-5, f"sampler step t={tt}")


def test_adamw_clip_ema_matches_torch(K):
    from oracle import diffusion_ref as dref
    n = 4096 * 5 + 8
    p0, g0 = rnd(n, seed=1), rnd(n, seed=2, scale=3.0)
    # torch reference: clip_grad_norm_ + AdamW + EMA, two steps
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pr], lr=5e-5, weight_decay=1e-4, betas=(0.9, 0.99))
    ema_r = p0.clone()
    p, m, v, ema = p0.clone(), torch.zeros(n, device=dev()), torch.zeros(n, device=dev()), p0.clone()
    for step in (1, 2):
        g = g0 * step
        pr.grad = g.clone()
        tn = torch.nn.utils.clip_grad_norm_([pr], 1.0)
        opt.step()
        ema_r = 0.995 * ema_r + 0.005 * pr.detach()
        nc = K.grad_norm_clipcoef(g, 1.0)
        K.adamw_ema_step(p, g, m, v, ema, nc, lr=5e-5, beta1=0.9, beta2=0.99, eps=1e-8, wd=1e-4, step=step,
                         ema_decay=0.995)
        assert_close(nc[0:1], tn.reshape(1), 1e-5, "grad norm")
        assert_close(p, pr.detach(), 1e-6, "param", atol_frac=1e-7)
        assert_close(ema, ema_r, 1e-6, "ema", atol_frac=1e-7)
    # and the oracle restatement agrees with torch on CPU
    po, mo, vo = p0.cpu().clone(), torch.zeros(n), torch.zeros(n)
    go = [g0.cpu().clone()]
    dref.clip_grad_norm(go, 1.0)
    dref.adamw_step(po, go[0], mo, vo, 1, 5e-5)
    pr2 = torch.nn.Parameter(p0.cpu().clone())
    opt2 = torch.optim.AdamW([pr2], lr=5e-5, weight_decay=1e-4, betas=(0.9, 0.99))
    pr2.grad = g0.cpu().clone()
    torch.nn.utils.clip_grad_norm_([pr2], 1.0)
    opt2.step()
    assert_close(po, pr2.detach(), 1e-6, "oracle adamw", atol_frac=1e-7)


def test_gn_onepass_forward_full_size_and_dispatch(K, monkeypatch):
    """GroupNorm forward in ONE pass (gn_onepass_fwd_kernel: a (sample, group) slab in the registers of one workgroup)
    at a slab of the maximum size (256 channels @ 48x48: nine pieces per thread), with more than eight samples (two
    rounds of the sample -> XCD placement), a channel-add and an output that is a slice of a wider buffer; 64-channel
    tensors (8-channel groups) must keep the two-pass kernels.  Both against fp32 torch, and against each other."""
    B, H, C = 9, 48, 256
    x = q16(rnd(B, C, H, H, seed=1) * 1.3 - 0.2)
    gamma, beta, ca = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3), rnd(B, C, seed=4)
    ref = F.silu(F.group_norm(x + ca[:, :, None, None], 8, gamma, beta, 1e-5))
    xg = (x + ca[:, :, None, None]).view(B, 8, -1)
    monkeypatch.setenv("DM_GN_ONEPASS", "0")   # the two-pass pair
    y2, stats2 = K.gn_silu_fwd(nhwc(x), gamma, beta, 8, chan_add=ca)
    assert K.last_variant() == "gn_stats+gn_apply"
    # mode 1: the whole-CU workgroup (opt-in); default (unset) = mode 2: the half-CU workgroup, 18 / 9 / 5 pieces per thread
    for mode, variant in (("1", "gn_onepass_fwd"), (None, "gn_onepass2_fwd<18>")):
        if mode is None:
            monkeypatch.delenv("DM_GN_ONEPASS")
        else:
            monkeypatch.setenv("DM_GN_ONEPASS", mode)
        buf = torch.full((B, H, H, 2 * C), 3.0, dtype=torch.bfloat16, device=dev())
        y, stats = K.gn_silu_fwd(nhwc(x), gamma, beta, 8, chan_add=ca, out=buf[..., C:])
        assert K.last_variant() == variant
        assert_close(nchw(buf[..., C:].contiguous()), ref, BF, "one-pass gn fwd " + variant)
        assert float((buf[..., :C].float() - 3.0).abs().max()) == 0.0
        assert_close(stats[..., 0], xg.mean(-1), 1e-4, "mean", atol_frac=1e-4)
        assert_close(stats[..., 1], torch.rsqrt(xg.var(-1, unbiased=False) + 1e-5), 1e-4, "rstd")
        assert_close(y2.float(), buf[..., C:].float(), BF, "two-pass vs one-pass")
        assert_close(stats2, stats, 1e-5, "two-pass vs one-pass statistics", atol_frac=1e-6)
        K.gn_silu_fwd(nhwc(q16(rnd(2, 64, 16, 16, seed=5))), torch.ones(64, device=dev()), torch.zeros(64, device=dev()), 8)
        assert K.last_variant() == "gn_stats+gn_apply"      # 8-channel groups keep the two passes
    # the smaller instantiations of the default form, a ragged pixel count (piece validity), no activation
    for (Bs, Hs, Ws, Cs), variant in (((3, 24, 24, 256), "gn_onepass2_fwd<5>"), ((2, 48, 48, 128), "gn_onepass2_fwd<9>"),
                                      ((2, 5, 7, 128), "gn_onepass2_fwd<5>"), ((2, 40, 33, 256), "gn_onepass2_fwd<18>")):
        xs = q16(rnd(Bs, Cs, Hs, Ws, seed=Cs + Hs) * 0.8 + 0.1)
        gs, bs = 1 + 0.1 * rnd(Cs, seed=6), 0.1 * rnd(Cs, seed=7)
        ys, _ = K.gn_silu_fwd(nhwc(xs), gs, bs, 8, act=(Hs != 5))
        assert K.last_variant() == variant
        r = F.group_norm(xs, 8, gs, bs, 1e-5)
        assert_close(nchw(ys), F.silu(r) if Hs != 5 else r, BF, variant)


@pytest.mark.parametrize("shape,variant", [((3, 24, 24, 256), "gn_onepass2_bwd<5>"), ((2, 48, 48, 128), "gn_onepass2_bwd<9>"),
                                           ((9, 48, 48, 256), "gn_onepass2_bwd<18>"), ((2, 40, 33, 256), "gn_onepass2_bwd<18>"),
                                           ((2, 5, 7, 128), "gn_onepass2_bwd<5>"), ((2, 8, 8, 1024), "gn_onepass2_bwd<5>"),
                                           ((2, 47, 48, 128), "gn_onepass2_bwd<9>")])
@pytest.mark.parametrize("full", [False, True])
def test_gn_onepass_backward(K, monkeypatch, shape, variant, full):
    """GroupNorm(+SiLU) backward in ONE pass (gn_onepass2_bwd_kernel: the x slab of a (sample, group) in LDS, the dy slab in
    registers) against the two-pass kernels and against torch autograd: every instantiation, ragged pixel counts, more than
    eight samples; `full` adds what the ResBlock backward passes -- channel add, residual add_in, dropout mask, fused
    column sums, strided x / dy / dx views of wider buffers -- and drops the activation on one shape."""
    B, H, W, C = shape
    act = not (full and H == 5)
    xw = q16(rnd(B, H, W, 2 * C, seed=1) * 1.2 + 0.2).to(torch.bfloat16)
    dyw = q16(rnd(B, H, W, C + 64, seed=2)).to(torch.bfloat16)
    x, dy = (xw[..., C:], dyw[..., :C]) if full else (xw[..., C:].contiguous(), dyw[..., :C].contiguous())
    gamma, beta = 1 + 0.1 * rnd(C, seed=3), 0.1 * rnd(C, seed=4)
    ca = rnd(B, C, seed=5) if full else None
    add_in = q16(rnd(B, H, W, C, seed=6)).to(torch.bfloat16) if full else None
    p, seed = (0.1, 1234) if full else (0.0, 0)
    _, stats = K.gn_silu_fwd(x, gamma, beta, 8, chan_add=ca, act=act)

    def run(env):
        if env is None:
            monkeypatch.delenv("DM_GN_ONEPASS_BWD", raising=False)
        else:
            monkeypatch.setenv("DM_GN_ONEPASS_BWD", env)
        dg, db = torch.full((C,), 0.25, device=dev()), torch.full((C,), -0.5, device=dev())
        cs = torch.full((B, C), 0.5, device=dev()) if full else None
        buf = torch.full((B, H, W, 2 * C), 3.0, dtype=torch.bfloat16, device=dev())
        out = buf[..., :C] if full else None
        dx = K.gn_silu_bwd(x, dy, gamma, beta, stats, 8, dg, db, chan_add=ca, add_in=add_in, act=act, drop_p=p, drop_seed=seed,
                           out=out, colsum_out=cs)
        v = K.last_variant()
        if full:
            assert float((buf[..., C:].float() - 3.0).abs().max()) == 0.0
        return dx.float().clone(), dg, db, cs, v

    dx2, dg2, db2, cs2, v2 = run("0")
    assert v2 == "gn_bwd_reduce+gn_bwd_apply"
    big = variant.endswith("<18>")       # the 18-piece slabs are opt-in (slower than the two passes when run alone)
    dx1, dg1, db1, cs1, v1 = run("18" if big else None)
    assert v1 == variant
    if big:
        assert run(None)[4] == "gn_bwd_reduce+gn_bwd_apply"
    assert_close(dx1, dx2, BF, "one-pass vs two-pass dx")
    assert float((dx1 != dx2).float().mean()) < 0.02     # same expressions; only the group means are summed in another order
    assert_close(dg1, dg2, 1e-4, "dgamma", atol_frac=1e-5)
    assert_close(db1, db2, 1e-4, "dbeta", atol_frac=1e-5)
    if full:
        assert_close(cs1, cs2, 1e-4, "column sums", atol_frac=1e-4)
        assert torch.allclose(cs1, dx1.sum(dim=(1, 2)) + 0.5, rtol=1e-4, atol=2e-3)
        assert float(((dx1 - add_in.float()) == 0).float().mean()) > 0.05     # the dropout mask is applied
    else:
        xr = nchw(x).requires_grad_()
        g_, b_ = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
        ref = F.silu(F.group_norm(xr, 8, g_, b_, 1e-5))
        gx, gg, gb = torch.autograd.grad(ref, [xr, g_, b_], nchw(dy))
        assert_close(nchw(dx1), gx, BF, "one-pass dx vs autograd")
        assert_close(dg1 - 0.25, gg, 5e-3, "dgamma vs autograd")
        assert_close(db1 + 0.5, gb, 5e-3, "dbeta vs autograd")
    # bit-reproducible: a second run gives the same bits
    dx1b, dg1b, db1b, _, _ = run("18" if big else None)
    assert torch.equal(dx1, dx1b) and torch.equal(dg1, dg1b) and torch.equal(db1, db1b)


def test_gn_silu_bwd_fused_column_sums():
    """dm_gn_silu_bwd(dx_colsum=...): per-sample channel sums of the written dx, accumulated (+=) in the same pass."""
    from diffusion_model_amd import kernels as K
    torch.manual_seed(3)
    for B, H, W, C, G in ((3, 12, 10, 64, 8), (2, 8, 8, 1024, 8)):   # the second: latent-model width
        x = torch.randn(B, H, W, C, device="cuda").to(torch.bfloat16)
        dy = torch.randn(B, H, W, C, device="cuda").to(torch.bfloat16)
        gam, bet = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
        _, stats = K.gn_silu_fwd(x, gam, bet, G)
        dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        cs = torch.full((B, C), 0.5, device="cuda")
        dx = K.gn_silu_bwd(x, dy, gam, bet, stats, G, dg, db, drop_p=0.1, drop_seed=7, colsum_out=cs)
        ref = dx.float().sum(dim=(1, 2)) + 0.5
        assert torch.allclose(cs, ref, rtol=1e-4, atol=1e-3), float((cs - ref).abs().max())


def test_film_fwd_bwd_and_dropout_words(K):
    """dm_film_fwd / dm_film_bwd (FiLMLayer.forward, reference unet.py:32-42, + the Dropout of unet.py:105) against the
    torch statement; the dropout word of an element is the one dm_conv3x3's epilogue draws for (seed, pixel, channel) --
    which is also what dm_gn_silu_bwd regenerates in the backward."""
    for B, H, W, C in ((2, 12, 12, 64), (3, 72, 72, 128), (2, 8, 8, 1024), (1, 20, 36, 192)):
        h0 = rnd(B, H, W, C, seed=1).to(torch.bfloat16)
        gb = rnd(B, 2 * C, seed=2)
        gamma, beta = gb[:, None, None, :C], gb[:, None, None, C:]
        out = K.film_fwd(h0, gb)
        ref = gamma.double() * h0.double() + beta.double()   # (the kernel contracts to one fma: compare to one bf16 ulp)
        assert float(((out.double() - ref).abs() / (ref.abs() + 1e-2)).max()) < 2 ** -8
        dh = rnd(B, H, W, C, seed=3).to(torch.bfloat16)
        dh0, dgamma = K.film_bwd(dh, h0, gb)
        assert torch.equal(dh0, (gamma * dh.float()).to(torch.bfloat16))
        assert_close(dgamma, (dh.float() * h0.float()).sum(dim=(1, 2)), 1e-5, "dgamma", atol_frac=1e-5)
        d2 = K.film_bwd(dh, h0, gb)[1]
        assert torch.equal(d2, dgamma), "dgamma is not run-to-run deterministic"
    # dropout: the same elements are dropped as by the conv epilogue with the same seed, survivors scaled by 1/keep
    B, H, W, C = 2, 16, 16, 64
    x = nhwc(q16(rnd(B, C, H, W, seed=4)))
    wp = K.pack_weight(rnd(C, C, 3, 3, seed=5, scale=0.05))
    y0 = K.conv3x3(x, wp, C)
    y1 = K.conv3x3(x, wp, C, drop_p=0.3, drop_seed=99)
    unit = torch.cat([torch.ones(B, C), torch.zeros(B, C)], 1).to(dev())
    f1 = K.film_fwd(y0, unit, 0.3, 99)
    live = y0 != 0
    assert torch.equal((y1 == 0) & live, (f1 == 0) & live)
    keep = float(((f1 != 0) & live).sum()) / float(live.sum())
    assert abs(keep - 0.7) < 0.02, keep
    kept = (f1 != 0) & live
    assert_close(f1[kept].float(), y0[kept].float() / 0.7, 1e-2, "kept values")


def test_batchsum2_cast(K):
    """One pass for a ResBlock's per-(sample, channel) sums: both bias gradients (+=) and the bf16 copy."""
    for B, C in ((64, 64), (5, 1024), (3, 72)):
        x = rnd(B, C, seed=B + C)
        o1, o2 = torch.full((C,), 0.25, device=dev()), torch.full((C,), -1.0, device=dev())
        c16 = K.batchsum2_cast(x, o1, o2)
        assert_close(o1, x.sum(0) + 0.25, 1e-5, "out1")
        assert_close(o2, x.sum(0) - 1.0, 1e-5, "out2")
        assert torch.equal(c16, x.to(torch.bfloat16))
        o3 = torch.zeros(C, device=dev())
        assert K.batchsum2_cast(x, o3, None, want_bf16=False) is None
        assert_close(o3, x.sum(0), 1e-5, "out1 alone")


# ------------------------------------------------------------------ GroupNorm folded into the consuming conv
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 16, 16, 64, 64), (1, 20, 36, 128, 64), (2, 12, 12, 64, 128),
                                            (3, 24, 24, 256, 256), (1, 48, 48, 128, 128), (5, 8, 8, 64, 64)])
def test_groupnorm_folded_into_conv_and_wgrad(K, B, H, W, Cin, Cout, monkeypatch):
    """dm_gn_prepare + dm_conv_args.norm_scale/shift: the conv stages silu(x*scale + shift) itself.  Must equal the
    materialised path (dm_gn_silu_fwd, then the conv on its output) bit for bit -- same expression, same rounding --
    zero padding included, and the torch statement of GroupNorm -> SiLU -> conv within the bf16 tolerance.
    (Bit equality is against the TWO-PASS dm_gn_silu_fwd, whose statistics kernel dm_gn_prepare shares; the one-pass
    forward sums in another order and has its own test.)"""
    monkeypatch.setenv("DM_GN_ONEPASS", "0")
    x = q16(rnd(B, Cin, H, W, seed=1) * 1.5 + 0.3)
    gamma, beta = 1 + 0.1 * rnd(Cin, seed=2), 0.1 * rnd(Cin, seed=3)
    w = rnd(Cout, Cin, 3, 3, seed=4, scale=1.0 / math.sqrt(9 * Cin))
    bias = rnd(Cout, seed=5, scale=0.1)
    xq, wp = nhwc(x), K.pack_weight(w)
    g_mat, stats_mat = K.gn_silu_fwd(xq, gamma, beta, 8)
    y_mat = K.conv3x3(g_mat, wp, Cout, bias=bias)
    stats, nin = K.gn_prepare(xq, gamma, beta, 8)
    y_fold = K.conv3x3(xq, wp, Cout, bias=bias, norm_in=nin)
    assert "NORM" in K.last_variant()
    assert torch.equal(stats, stats_mat)
    assert torch.equal(y_fold, y_mat)
    ref = F.conv2d(q16(F.silu(F.group_norm(x, 8, gamma, beta, 1e-5))), q16(w), bias, padding=1)
    assert_close(nchw(y_fold), ref, BF, "folded GroupNorm conv vs torch")
    dy = nhwc(q16(rnd(B, Cout, H, W, seed=6)))
    dw_mat, db_mat = torch.zeros(Cout, Cin, 3, 3, device=dev()), torch.zeros(Cout, device=dev())
    K.wgrad(g_mat, dy, dw_mat, db_mat)
    dw_fold, db_fold = torch.zeros_like(dw_mat), torch.zeros_like(db_mat)
    K.wgrad(xq, dy, dw_fold, db_fold, norm_in=nin)
    assert "NORM" in K.last_variant()
    assert torch.equal(dw_fold, dw_mat) and torch.equal(db_fold, db_mat)
    # with a per-(sample, channel) pre-add (the time-embedding form of dm_gn_silu_fwd)
    ca = rnd(B, Cin, seed=7)
    g2, _ = K.gn_silu_fwd(xq, gamma, beta, 8, chan_add=ca)
    _, nin2 = K.gn_prepare(xq, gamma, beta, 8, chan_add=ca)
    assert torch.equal(K.conv3x3(xq, wp, Cout, norm_in=nin2), K.conv3x3(g2, wp, Cout))
