"""The fp32 COMPUTE MODE (kernels.set_compute_dtype(torch.float32); csrc/dm_f32.hip) against the real reference's fp32
results: the north star's second tolerance, rtol 1e-4.  Kernel-level checks against torch fp32 ops on the GPU, module
goldens and the full UNets (STL10 with / without context, FiLM, MNIST, the latent configuration) against
tests/golden/*.npz, which oracle/gen_golden.py produced by running the reference's own unet.py.

Tolerance: |a - b| <= 1e-4 |b| + 1e-4 max|b| (conftest.assert_close), and rel-L2 <= 1e-4 where stated.  Gradients of the
provably dead parameters (one-token context: to_q / to_k / the self-attention in front of a cross-attention)
are round-off in the reference and are only required to stay at round-off level."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close, load_golden, rel_l2
from oracle import unet_ref as oref

pytestmark = pytest.mark.gpu
TOL = 1e-4


def dev():
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def fp32_mode():
    from diffusion_model_amd import kernels as K
    with K.compute_dtype(torch.float32):
        yield


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


# ------------------------------------------------------------------ kernels against torch fp32
@pytest.mark.parametrize("mode,cin,cout,H", [(0, 64, 64, 12), (0, 3, 64, 9), (0, 128, 3, 8), (1, 64, 64, 12), (2, 64, 128, 6),
                                             (3, 64, 64, 6), (0, 96, 80, 7)])
def test_conv3x3_fp32_all_modes(mode, cin, cout, H):
    from diffusion_model_amd import kernels as K
    g = torch.Generator().manual_seed(mode * 100 + cin)
    B = 3
    x = torch.randn(B, cin, H, H + 1 if mode in (0, 2, 3) else H, generator=g).to(dev())
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1).to(dev())
    bias = torch.randn(cout, generator=g).to(dev())
    if mode == 0:
        ref = F.conv2d(x, w, bias, padding=1)
    elif mode == 1:
        ref = F.conv2d(x, w, bias, stride=2, padding=1)
    elif mode == 2:
        ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, bias, padding=1)
    else:   # zero-insert x2 then conv: the data-gradient form of a stride-2 convolution
        z = torch.zeros(B, cin, 2 * x.shape[2], 2 * x.shape[3], device=dev())
        z[:, :, ::2, ::2] = x
        ref = F.conv2d(z, w, bias, padding=1)
    y = K.conv3x3(nhwc(x), K.pack_weight(w, dtype=torch.float32), cout, bias=bias, mode=mode)
    assert y.dtype == torch.float32 and K.last_variant().startswith("conv_f32")
    assert_close(nchw(y), ref, 1e-5, f"conv mode {mode}")


def test_conv3x3_fp32_epilogue_and_pooled_dgrad():
    from diffusion_model_amd import kernels as K
    g = torch.Generator().manual_seed(5)
    B, C, H = 2, 64, 8
    x = torch.randn(B, C, H, H, generator=g).to(dev())
    w = (torch.randn(C, C, 3, 3, generator=g) * 0.1).to(dev())
    bias, ca = torch.randn(C, generator=g).to(dev()), torch.randn(B, C, generator=g).to(dev())
    res = torch.randn(B, C, H, H, generator=g).to(dev())
    y = K.conv3x3(nhwc(x), K.pack_weight(w, dtype=torch.float32), C, bias=bias, chan_add=ca, res=nhwc(res), act_silu_in=True)
    ref = F.conv2d(F.silu(x), w, bias, padding=1) + ca[:, :, None, None] + res
    assert_close(nchw(y), ref, 1e-5, "bias + chan_add + residual + SiLU input")
    # dropout: the kept elements are scaled by 1/keep, the rest are exactly zero, and the keep rate is ~0.9
    yd = K.conv3x3(nhwc(x), K.pack_weight(w, dtype=torch.float32), C, bias=bias, drop_p=0.1, drop_seed=77)
    full = nhwc(F.conv2d(x, w, bias, padding=1))
    kept = yd != 0
    assert 0.85 < float(kept.float().mean()) < 0.95
    big = kept & (full.abs() > 0.5)
    inv_keep = float((yd[big] / full[big]).mean())
    assert abs(inv_keep - 1 / 0.9) < 2e-3
    assert_close(yd, torch.where(kept, full * inv_keep, torch.zeros_like(full)), 1e-5, "dropout scaling")
    # data-gradient of nearest-x2 + conv: conv with the transposed / flipped weights at full size, then 2x2 sums
    xs = torch.randn(B, C, H // 2, H // 2, generator=g).to(dev()).requires_grad_()
    dy = torch.randn(B, C, H, H, generator=g).to(dev())
    F.conv2d(F.interpolate(xs, scale_factor=2, mode="nearest"), w, None, padding=1).backward(dy)
    dx = K.conv3x3(nhwc(dy), K.pack_weight(w, True, dtype=torch.float32), C, pool2=True)
    assert_close(nchw(dx), xs.grad, 1e-5, "pooled data-gradient")


@pytest.mark.parametrize("M,Kd,N", [(300, 64, 64), (1000, 256, 384), (77, 512, 100), (5, 256, 256)])
def test_gemm_fp32(M, Kd, N):
    from diffusion_model_amd import kernels as K
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, Kd, generator=g).to(dev())
    w = (torch.randn(N, Kd, generator=g) * 0.05).to(dev())
    b = torch.randn(N, generator=g).to(dev())
    y = K.gemm(x, K.pack_weight(w, dtype=torch.float32), N, bias=b)
    assert_close(y, x @ w.t() + b, 1e-5, "gemm")
    dx = K.gemm(y, K.pack_weight(w, True, dtype=torch.float32), Kd)     # data-gradient operand
    assert_close(dx, y @ w, 1e-5, "gemm dgrad")


@pytest.mark.parametrize("mode,cin,cout,H", [(0, 64, 64, 12), (1, 64, 128, 12), (2, 128, 64, 6), (0, 3, 64, 10)])
def test_wgrad_fp32(mode, cin, cout, H):
    from diffusion_model_amd import kernels as K
    g = torch.Generator().manual_seed(cin + mode)
    B = 3
    x = torch.randn(B, cin, H, H, generator=g).to(dev())
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1).to(dev()).requires_grad_()
    b = torch.zeros(cout, device=dev(), requires_grad=True)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if mode == 2 else x
    y = F.conv2d(xin, w, b, stride=2 if mode == 1 else 1, padding=1)
    dy = torch.randn(y.shape, generator=g).to(dev())
    y.backward(dy)
    dw, db = torch.zeros_like(w), torch.zeros(cout, device=dev())
    K.wgrad(nhwc(x), nhwc(dy), dw, db, mode=mode)
    assert_close(dw, w.grad, 2e-5, "dw")
    assert_close(db, b.grad, 2e-5, "dbias")
    dw2, db2 = torch.zeros_like(w), torch.zeros(cout, device=dev())
    K.wgrad(nhwc(x), nhwc(dy), dw2, db2, mode=mode)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)      # ordered split-K: bit-reproducible


@pytest.mark.parametrize("C,H", [(64, 12), (128, 9), (256, 6)])
def test_groupnorm_silu_fp32(C, H):
    from diffusion_model_amd import kernels as K
    g = torch.Generator().manual_seed(C)
    B = 3
    x = (torch.randn(B, C, H, H, generator=g) * 2 + 0.7).to(dev()).requires_grad_()
    ca = torch.randn(B, C, generator=g).to(dev())
    gam = (torch.rand(C, generator=g) + 0.5).to(dev()).requires_grad_()
    bet = torch.randn(C, generator=g).to(dev()).requires_grad_()
    ref = F.silu(F.group_norm(x + ca[:, :, None, None], 8, gam, bet, 1e-5))
    dy = torch.randn(ref.shape, generator=g).to(dev())
    ref.backward(dy)
    y, stats = K.gn_silu_fwd(nhwc(x.detach()), gam.detach(), bet.detach(), 8, chan_add=ca)
    assert_close(nchw(y), ref, 1e-5, "gn+silu")
    dg, db = torch.zeros(C, device=dev()), torch.zeros(C, device=dev())
    cs = torch.zeros(B, C, device=dev())
    dx = K.gn_silu_bwd(nhwc(x.detach()), nhwc(dy), gam.detach(), bet.detach(), stats, 8, dg, db, chan_add=ca, colsum_out=cs)
    assert_close(nchw(dx), x.grad, 2e-5, "dx")
    assert_close(dg, gam.grad, 2e-5, "dgamma")
    assert_close(db, bet.grad, 2e-5, "dbeta")
    assert_close(cs, x.grad.sum((2, 3)), 2e-5, "dx column sums", atol_frac=1e-4)


@pytest.mark.parametrize("hd,Nq,Nk", [(8, 100, 1), (16, 300, 300), (32, 130, 70), (64, 96, 33), (128, 40, 64), (16, 64, 5)])
def test_attention_fp32_vs_torch(hd, Nq, Nk):
    from diffusion_model_amd import kernels as K
    heads, B = 8, 2
    C = heads * hd
    g = torch.Generator().manual_seed(hd + Nk)
    q, k, v = (torch.randn(B, n, C, generator=g).to(dev()).requires_grad_() for n in (Nq, Nk, Nk))
    d_o = torch.randn(B, Nq, C, generator=g).to(dev())

    def split(t):
        return t.view(B, -1, heads, hd).transpose(1, 2)
    ref = F.scaled_dot_product_attention(split(q), split(k), split(v)).transpose(1, 2).reshape(B, Nq, C)
    ref.backward(d_o)
    o, lse = K.attn_fwd(q.detach(), k.detach(), v.detach(), heads)
    assert_close(o, ref, 2e-5, "o")
    s = torch.einsum("bhqd,bhkd->bhqk", split(q), split(k)) / hd ** 0.5
    assert_close(lse, torch.logsumexp(s, -1), 2e-5, "lse")
    dq, dk, dv = K.attn_bwd(q.detach(), k.detach(), v.detach(), o, lse, d_o, heads)
    if Nk == 1:   # softmax over one key: dq = dk = 0 up to round-off, here and in torch
        assert float(dq.abs().max()) < 1e-5 and float(dk.abs().max()) < 1e-4 * float(d_o.abs().max()) * Nq
    else:
        assert_close(dq, q.grad, 5e-5, "dq")
        assert_close(dk, k.grad, 5e-5, "dk")
    assert_close(dv, v.grad, 5e-5, "dv")


def test_attention_fp32_dropout_is_consistent():
    """The mask is extracted from the forward itself (uniform scores, identity values), then the dropped-out forward and
    its backward are checked against torch autograd through that explicit mask."""
    from diffusion_model_amd import kernels as K
    heads, hd, B, Nq, Nk = 8, 16, 2, 70, 16
    C = heads * hd
    p, seed = 0.25, 1234
    zq, zk = torch.zeros(B, Nq, C, device=dev()), torch.zeros(B, Nk, C, device=dev())
    eye = torch.eye(Nk, hd, device=dev()).repeat(1, heads).unsqueeze(0).repeat(B, 1, 1)      # v[b, k, h*hd + d] = (k == d)
    o, _ = K.attn_fwd(zq, zk, eye, heads, drop_p=p, drop_seed=seed)
    keep = o.view(B, Nq, heads, hd).permute(0, 2, 1, 3)[..., :Nk] > 0                          # [B, heads, Nq, Nk]
    rate = float(keep.float().mean())
    assert 0.70 < rate < 0.80
    inv_keep = float(o.max()) * Nk
    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(B, n, C, generator=g).to(dev()).requires_grad_() for n in (Nq, Nk, Nk))
    d_o = torch.randn(B, Nq, C, generator=g).to(dev())

    def split(t):
        return t.view(B, -1, heads, hd).transpose(1, 2)
    pr = torch.softmax(torch.einsum("bhqd,bhkd->bhqk", split(q), split(k)) / hd ** 0.5, -1) * keep * inv_keep
    ref = torch.einsum("bhqk,bhkd->bhqd", pr, split(v)).transpose(1, 2).reshape(B, Nq, C)
    ref.backward(d_o)
    o, lse = K.attn_fwd(q.detach(), k.detach(), v.detach(), heads, drop_p=p, drop_seed=seed)
    assert_close(o, ref, 5e-5, "o with dropout")
    dq, dk, dv = K.attn_bwd(q.detach(), k.detach(), v.detach(), o, lse, d_o, heads, drop_p=p, drop_seed=seed)
    assert_close(dq, q.grad, 1e-4, "dq")
    assert_close(dk, k.grad, 1e-4, "dk")
    assert_close(dv, v.grad, 1e-4, "dv")


# ------------------------------------------------------------------ modules against the reference's goldens
def fill(mod, gain=1.0):
    shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
    mod.load_state_dict(oref.seeded_fill(shapes, 1234, gain), strict=True)
    return mod.to(dev()).eval()


def run(mod, g, in_names, G="G"):
    ins = [g[n].to(dev()).requires_grad_() for n in in_names]
    y = mod(*ins)
    (y * g[G].to(dev())).sum().backward()
    return y, ins


def check_param_grads(mod, g, tol=TOL, skip=()):
    for n, p in mod.named_parameters():
        if n in skip:
            continue
        assert p.grad is not None, n
        assert_close(p.grad, g["g." + n], tol, n)


@pytest.mark.parametrize("tag,cin,cout", [("id64", 64, 64), ("skip64_128", 64, 128), ("film_id64", 64, 64),
                                          ("film_skip64_128", 64, 128)])
def test_resblock_fp32_vs_reference(tag, cin, cout):
    from diffusion_model_amd.unet import ResBlock
    g = load_golden(f"resblock_{tag}")
    m = fill(ResBlock(cin, cout, 256, 8, 0.0, use_film=tag.startswith("film")))
    y, (x, te) = run(m, g, ["x", "temb"])
    assert_close(y, g["y"], TOL, "y")
    assert_close(x.grad, g["gx"], TOL, "gx")
    assert_close(te.grad, g["gtemb"], TOL, "gtemb")
    check_param_grads(m, g)


@pytest.mark.parametrize("tag,C", [("c128_hd16", 128), ("c256_hd32", 256)])
def test_self_attention_fp32_vs_reference(tag, C):
    from diffusion_model_amd.unet import Attention
    g = load_golden(f"attention_{tag}")
    m = fill(Attention(C, 8, 0.0), 1.0)
    y, (x,) = run(m, g, ["x"])
    assert_close(y, g["y"], TOL, "y")
    assert_close(x.grad, g["gx"], TOL, "gx")
    check_param_grads(m, g)


@pytest.mark.parametrize("tag,C", [("c64_hd8_L1", 64), ("c128_hd16_L4", 128), ("c256_hd32_L1", 256)])
def test_cross_attention_fp32_vs_reference(tag, C):
    from diffusion_model_amd.unet import CrossAttention
    g = load_golden(f"cross_{tag}")
    m = fill(CrossAttention(C, 512, 8, 0.0), 1.0)
    y, (x, ctx) = run(m, g, ["x", "ctx"])
    assert_close(y, g["y"], TOL, "y")
    assert_close(ctx.grad, g["gctx"], TOL, "gctx")
    one_token = g["ctx"].dim() == 2 or g["ctx"].shape[1] == 1
    live = float(g["gctx"].abs().max())
    if one_token:  # softmax over one key == 1: dx, dWq, dWk are round-off in the reference and here
        assert float(x.grad.abs().max()) < 1e-4 * live
        check_param_grads(m, g, skip=("to_q.weight", "to_k.weight"))
        for n in ("to_q.weight", "to_k.weight"):
            assert float(getattr(m, n.split(".")[0]).weight.grad.abs().max()) < 1e-4 * float(g["g.to_v.weight"].abs().max())
    else:
        assert_close(x.grad, g["gx"], TOL, "gx")
        check_param_grads(m, g)


@pytest.mark.parametrize("name", ["upsample", "downsample"])
def test_updown_fp32_vs_reference(name):
    from diffusion_model_amd import unet
    g = load_golden(name)
    m = fill(getattr(unet, name.capitalize())(64))
    y, (x,) = run(m, g, ["x"])
    assert_close(y, g["y"], TOL, "y")
    assert_close(x.grad, g["gx"], TOL, "gx")
    check_param_grads(m, g)


STL = dict(input_channels=3, output_channels=3, context_dim=512, dropout_prob=0.0)
LDM_KW = dict(Channels=[256, 256, 512, 1024, 512], Attentions=[False, True, True, False], input_channels=4,
              output_channels=4, context_dim=512, dropout_prob=0.0)
INIT_CASES = [("stl10_init", STL, "clip"), ("stl10_init_noctx", STL, None),
              ("stl10_film_init", dict(STL, use_film=True), "clip"),
              ("mnist_init", dict(Attentions=[False] * 6, dropout_prob=0.0), "labels"),
              ("ldm_init", LDM_KW, "clip"), ("ldm_init_noctx", LDM_KW, None)]


@pytest.mark.parametrize("tag,spec_kw,ctx_mode", INIT_CASES)
def test_full_unet_fp32_vs_reference(tag, spec_kw, ctx_mode):
    """test_full_unet_realistic_init_vs_reference (tests/test_gpu_modules.py) in the fp32 compute mode: reference default
    init + non-zero output_conv; the output, dL/dx and EVERY parameter gradient the fixture holds (norms, 64 fixed-index
    samples, full tensors where stored) against the real reference's fp32 results at rtol 1e-4."""
    from diffusion_model_amd.unet import UNET
    g = load_golden(f"unet_{tag}")
    torch.manual_seed(0)
    m = UNET(**spec_kw)
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        m.output_conv.weight.copy_(torch.randn(m.output_conv.weight.shape, generator=gen) * 0.05)
    m = m.to(dev()).eval()
    x = g["x"].to(dev()).requires_grad_()
    ctx = g["ctx"].to(dev()) if "ctx" in g else None
    y = m(x, g["t"], ctx)
    assert_close(y, g["y"], TOL, "y")
    assert rel_l2(y, g["y"]) < TOL
    (y * g["G"].to(dev())).sum().backward()
    assert_close(x.grad, g["gx"], TOL, "gx")
    assert rel_l2(x.grad, g["gx"]) < TOL
    gmax = max(float(v) for k, v in g.items() if k.startswith("gn."))
    worst = (0.0, "")
    for n, p in m.named_parameters():
        if "gn." + n not in g:
            continue
        ref_n = float(g["gn." + n])
        # the fixture's norms are fp32 torch-CPU vector_norm values: over the multi-million-element tensors of the latent
        # configuration that accumulation alone is off by up to 4e-4 (the float64 norm of the SAME reference gradient is
        # 47.0939 where the fixture holds 47.0857), so the norm is taken with the same CPU kernel here; the exact checks
        # are the 64 sampled values and the full tensors below
        got_n = float(torch.linalg.vector_norm(p.grad.float().cpu()))
        if ref_n < 1e-6 * gmax:  # dead path: round-off in the reference, round-off here
            assert got_n < 1e-5 * gmax, n
            continue
        idx = (torch.arange(64) * 7919) % p.numel()
        assert abs(got_n - ref_n) <= TOL * ref_n, f"{n}: |g| {got_n} vs {ref_n}"
        err = rel_l2(p.grad.reshape(-1)[idx.to(dev())], g["gs." + n])
        assert err <= TOL, f"{n}: sampled rel-L2 {err:.3e}"
        assert_close(p.grad.reshape(-1)[idx.to(dev())], g["gs." + n], TOL, f"{n} (64 samples)")
        worst = max(worst, (err, n))
        if "gf." + n in g:
            assert_close(p.grad, g["gf." + n], TOL, n)
    print(f"[{tag}, fp32] worst sampled gradient rel-L2 {worst[0]:.3e} ({worst[1]})")


def test_train_step_fp32_matches_oracle():
    """One micro-step of the Trainer (q-sample, forward, low-t-weighted loss, backward into the flat gradient) in the fp32
    mode against the oracle's fp32 autograd on the same batch: loss and every parameter gradient at 1e-4; then the fused
    clip + AdamW + EMA step runs on it (its own parity: tests/test_gpu_train_sample.py)."""
    from diffusion_model_amd.ema import ModelEmaV2
    from diffusion_model_amd.trainer import Trainer
    from diffusion_model_amd.unet import UNET
    from diffusion_model_amd.utils import DDPM_beta_t_cosine_scheduler
    from oracle import diffusion_ref as dref
    torch.manual_seed(0)
    m = UNET(**STL)
    with torch.no_grad():
        m.output_conv.weight.normal_(0, 0.05, generator=torch.Generator().manual_seed(1))
    sd = {k: v.detach().clone().requires_grad_() for k, v in m.state_dict().items()}
    names = list(sd)
    m = m.to(dev()).train()
    tr = Trainer(m, DDPM_beta_t_cosine_scheduler(500), lr=1e-3, clip=1.0, accum=1, total_steps=0, low_t_weight=0.01,
                 ema=ModelEmaV2(m, 0.995))
    tr.defer_optimizer = True
    spec = oref.UNetSpec(**STL)
    _, _, abar = dref.cosine_schedule(500)
    g = torch.Generator().manual_seed(11)
    x = torch.clamp(torch.randn(2, 3, 32, 32, generator=g) * 0.5, -1, 1)
    e = torch.randn(2, 3, 32, 32, generator=g)
    t = torch.randint(0, 500, (2,), generator=g)
    ctx = F.normalize(torch.randn(2, 512, generator=g), dim=-1)
    loss = tr.micro_step(x.to(dev()), ctx.to(dev()), t.to(dev()), e.to(dev()))
    xt, v = dref.q_sample_v_target(x, e, abar[t])
    ref_loss = dref.v_loss(oref.unet_forward(sd, spec, xt, t, ctx), v, t, 0.01)
    grads = torch.autograd.grad(ref_loss, [sd[k] for k in names], allow_unused=True)
    assert abs(float(loss) - float(ref_loss)) <= 1e-5 * abs(float(ref_loss)), (float(loss), float(ref_loss))
    gmax = max(float(gg.norm()) for gg in grads if gg is not None)
    worst = (0.0, "")
    for p, gg, n in zip(m.parameters(), grads, names):
        got = tr.fp.grad_views[id(p)].detach().cpu()
        if gg is None or float(gg.norm()) < 1e-6 * gmax:   # dead under the one-token context
            assert float(got.norm()) < 1e-5 * gmax, n
            continue
        err = rel_l2(got, gg)
        assert err < TOL, f"{n}: gradient rel-L2 {err:.3e}"
        worst = max(worst, (err, n))
    print(f"[train step, fp32] worst gradient rel-L2 {worst[0]:.3e} ({worst[1]})")
    tr.optimizer_step()
    torch.cuda.synchronize()
    assert tr.opt_step == 1 and bool(torch.isfinite(tr.fp.flat).all())
