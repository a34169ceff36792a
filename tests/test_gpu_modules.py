"""GPU parity proper: the drop-in modules (HIP path, through the C ABI) against golden vectors
produced by the REAL reference (tests/golden, oracle/gen_golden.py) and against the oracle on
fresh seeded inputs.  Tolerance = the north-star's bf16 bound: rtol 2e-2 with an absolute floor
of 2e-2*max|ref| (near-zero elements / dead-path gradients)."""
import pytest
import torch

from conftest import assert_close, load_golden, rel_l2
from oracle import unet_ref as oref

pytestmark = pytest.mark.gpu
TOL = 2e-2


def dev():
    return torch.device("cuda:0")


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


@pytest.mark.parametrize("tag,cin,cout", [("id64", 64, 64), ("skip64_128", 64, 128),
                                          ("film_id64", 64, 64), ("film_skip64_128", 64, 128)])
def test_resblock_vs_reference(tag, cin, cout):
    from diffusion_model_amd.unet import ResBlock
    g = load_golden(f"resblock_{tag}")
    m = fill(ResBlock(cin, cout, 256, 8, 0.0, use_film=tag.startswith("film")))
    y, (x, te) = run(m, g, ["x", "temb"])
    assert_close(y, g["y"], TOL, "y")
    assert_close(x.grad, g["gx"], TOL, "gx")
    assert_close(te.grad, g["gtemb"], TOL, "gtemb")
    check_param_grads(m, g)


@pytest.mark.parametrize("tag,C", [("c128_hd16", 128), ("c256_hd32", 256)])
def test_self_attention_vs_reference(tag, C):
    from diffusion_model_amd.unet import Attention
    g = load_golden(f"attention_{tag}")
    m = fill(Attention(C, 8, 0.0), 1.0)
    y, (x,) = run(m, g, ["x"])
    assert_close(y, g["y"], TOL, "y")
    assert_close(x.grad, g["gx"], TOL, "gx")
    check_param_grads(m, g)


@pytest.mark.parametrize("tag,C", [("c64_hd8_L1", 64), ("c128_hd16_L4", 128), ("c256_hd32_L1", 256)])
def test_cross_attention_vs_reference(tag, C):
    from diffusion_model_amd.unet import CrossAttention
    g = load_golden(f"cross_{tag}")
    m = fill(CrossAttention(C, 512, 8, 0.0), 1.0)
    y, (x, ctx) = run(m, g, ["x", "ctx"])
    assert_close(y, g["y"], TOL, "y")
    assert_close(ctx.grad, g["gctx"], TOL, "gctx")
    one_token = g["ctx"].dim() == 2 or g["ctx"].shape[1] == 1
    live = float(g["gctx"].abs().max())
    if one_token:  # softmax over one key == 1: dx, dWq, dWk are identically zero (reference: round-off)
        assert float(x.grad.abs().max()) < 1e-2 * live
        check_param_grads(m, g, skip=("to_q.weight", "to_k.weight"))
        for n in ("to_q.weight", "to_k.weight"):
            assert float(getattr(m, n.split(".")[0]).weight.grad.abs().max()) < 1e-2 * float(g["g.to_v.weight"].abs().max())
    else:
        assert_close(x.grad, g["gx"], TOL, "gx")
        check_param_grads(m, g)


@pytest.mark.parametrize("name", ["upsample", "downsample"])
def test_updown_vs_reference(name):
    from diffusion_model_amd import unet
    g = load_golden(name)
    m = fill(getattr(unet, name.capitalize())(64))
    y, (x,) = run(m, g, ["x"])
    assert_close(y, g["y"], TOL, "y")
    assert_close(x.grad, g["gx"], TOL, "gx")
    check_param_grads(m, g)


def _unet(spec_kw):
    from diffusion_model_amd.unet import UNET
    m = UNET(**spec_kw)
    spec = oref.UNetSpec(**{k: v for k, v in spec_kw.items() if k in oref.UNetSpec.__dataclass_fields__})
    m.load_state_dict(oref.closed_form_weights(spec), strict=True)
    return m.to(dev()).eval()


STL = dict(input_channels=3, output_channels=3, context_dim=512, dropout_prob=0.0)


@pytest.mark.parametrize("tag,spec_kw", [("stl10_small", STL), ("stl10_clip", STL), ("stl10_noctx", STL),
                                         ("mnist", dict(Attentions=[False] * 6, dropout_prob=0.0))])
def test_full_unet_vs_reference(tag, spec_kw):
    g = load_golden(f"unet_{tag}")
    m = _unet(spec_kw)
    x = g["x"].to(dev()).requires_grad_()
    ctx = g["ctx"].to(dev()) if "ctx" in g else None
    y = m(x, g["t"], ctx)
    assert tuple(y.shape) == tuple(g["y"].shape) and y.dtype == torch.float32
    assert_close(y, g["y"], TOL, "y")
    # Forward parity only: the gradients of these CLOSED-FORM (sin-pattern) weights are cancellation-
    # dominated (the oracle's own bf16-storage noise model, oracle.unet_ref.EMULATE_BF16, is 5-50 % off on
    # them), so they cannot discriminate a correct bf16 pipeline from a wrong one.  Gradient parity is
    # asserted on realistic weights in test_full_unet_realistic_init_vs_reference below.


LDM_KW = dict(Channels=[256, 256, 512, 1024, 512], Attentions=[False, True, True, False], input_channels=4,
              output_channels=4, context_dim=512, dropout_prob=0.0)   # reference utils.py:145-156
INIT_CASES = [("stl10_init", STL, "clip"), ("stl10_init_noctx", STL, None),
              ("stl10_film_init", dict(STL, use_film=True), "clip"),   # reference unet.py:9-42,82-84,96-98
              ("mnist_init", dict(Attentions=[False] * 6, dropout_prob=0.0), "labels"),
              ("ldm_init", LDM_KW, "clip"), ("ldm_init_noctx", LDM_KW, None)]


@pytest.mark.parametrize("tag,spec_kw,ctx_mode", INIT_CASES)
def test_full_unet_realistic_init_vs_reference(tag, spec_kw, ctx_mode):
    """Reference default init (torch.manual_seed(0)) + non-zero output_conv: outputs and ALL gradients
    within the north-star bf16 bound against the real reference's fp32 results."""
    from diffusion_model_amd.unet import UNET
    g = load_golden(f"unet_{tag}")
    torch.manual_seed(0)
    m = UNET(**spec_kw)
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        m.output_conv.weight.copy_(torch.randn(m.output_conv.weight.shape, generator=gen) * 0.05)
    for k, v in m.state_dict().items():  # same registration order => same init as the reference
        assert float(torch.linalg.vector_norm(v)) == pytest.approx(float(g["wn." + k]), rel=1e-6), k
    m = m.to(dev()).eval()
    x = g["x"].to(dev()).requires_grad_()
    ctx = g["ctx"].to(dev()) if "ctx" in g else None
    y = m(x, g["t"], ctx)
    assert_close(y, g["y"], TOL, "y")
    assert rel_l2(y, g["y"]) < TOL
    (y * g["G"].to(dev())).sum().backward()
    assert_close(x.grad, g["gx"], TOL, "gx")
    assert rel_l2(x.grad, g["gx"]) < TOL
    # bf16-storage noise floor of every gradient, measured by the oracle's noise model on the same
    # weights/inputs (CPU): bound = max(2e-2, 1.5 x floor).  For the STL10 flagship the floor is at or
    # below 2e-2 (worst 64-sample slice: 2.01e-2), so the north-star bound is what applies there.
    spec = oref.UNetSpec(**{k: v for k, v in spec_kw.items() if k in oref.UNetSpec.__dataclass_fields__})
    sd = {k: v.detach().cpu().clone().requires_grad_() for k, v in m.state_dict().items()}
    xe = g["x"].clone().requires_grad_()
    oref.EMULATE_BF16 = True
    try:
        ye = oref.unet_forward(sd, spec, xe, g["t"], g.get("ctx"))
        names = list(sd)
        ge = dict(zip(names, torch.autograd.grad((ye * g["G"]).sum(), [sd[k] for k in names], allow_unused=True)))
    finally:
        oref.EMULATE_BF16 = False
    gmax = max(float(v) for k, v in g.items() if k.startswith("gn."))
    worst = (0.0, "")
    for n, p in m.named_parameters():
        if "gn." + n not in g:
            continue
        ref_n = float(g["gn." + n])
        got_n = float(torch.linalg.vector_norm(p.grad.float()))
        if ref_n < 1e-6 * gmax:  # dead path: must stay (near) zero
            assert got_n < 1e-3 * gmax, n
            continue
        idx = (torch.arange(64) * 7919) % p.numel()
        ref_s = g["gs." + n]
        floor_n = abs(float(torch.linalg.vector_norm(ge[n])) - ref_n) / ref_n
        floor_s = rel_l2(ge[n].reshape(-1)[idx], ref_s)
        assert abs(got_n - ref_n) <= max(TOL, 1.5 * floor_n) * ref_n, f"{n}: |g| {got_n} vs {ref_n}"
        err = rel_l2(p.grad.reshape(-1)[idx.to(dev())], ref_s)
        assert err <= max(TOL, 1.5 * floor_s), f"{n}: sampled rel-L2 {err:.3e} (bf16 floor {floor_s:.3e})"
        assert_close(p.grad.reshape(-1)[idx.to(dev())], ref_s, TOL, f"{n} (64 samples)", atol_frac=2 * TOL)
        worst = max(worst, (err, n))
        if "gf." + n in g:
            assert_close(p.grad, g["gf." + n], TOL, n, atol_frac=2 * TOL)
    print(f"[{tag}] worst sampled gradient rel-L2 {worst[0]:.3e} ({worst[1]})")


def test_full_unet_vs_oracle_fresh_inputs():
    """Same comparison on seeded random inputs/weights that no fixture holds (oracle = checker)."""
    torch.manual_seed(7)
    spec = oref.UNetSpec(**STL)
    sd = {k: v for k, v in oref.closed_form_weights(spec).items()}
    m = _unet(STL)
    x = torch.randn(2, 3, 32, 32)
    t = torch.tensor([3, 411])
    ctx = torch.nn.functional.normalize(torch.randn(2, 512), dim=-1)
    ref = oref.unet_forward(sd, spec, x, t, ctx)
    y = m(x.to(dev()), t, ctx.to(dev()))
    assert_close(y, ref, TOL, "y vs oracle")
    ctx3 = torch.randn(2, 5, 512) * 0.2  # multi-token context exercises the general-L attention path
    ref3 = oref.unet_forward(sd, spec, x, t, ctx3)
    y3 = m(x.to(dev()), t, ctx3.to(dev()))
    assert_close(y3, ref3, TOL, "y vs oracle, L=5")


@pytest.mark.parametrize("ctx_mode", ["clip", None])
def test_full_backward_is_bit_reproducible(ctx_mode):
    """Every gradient reduction on the path is ORDERED (split-K weight-gradient slabs, GroupNorm dgamma / dbeta / column
    sums, one-token cross-attention partials, the attention dq splits): the same inputs give the same bits.  Full STL10
    UNet at 96x96 (the kernel variants of the real shapes), forward + backward twice, with dropout (the masks are a
    function of the seed) and across the three streams."""
    import itertools
    from diffusion_model_amd import unet as U
    torch.manual_seed(0)
    m = U.UNET(**dict(STL, dropout_prob=0.1))
    with torch.no_grad():
        m.output_conv.weight.normal_(0, 0.05, generator=torch.Generator().manual_seed(1))
    m = m.to(dev()).train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 96, 96, generator=g).to(dev())
    t = torch.tensor([17, 433])
    ctx = torch.nn.functional.normalize(torch.randn(2, 512, generator=g), dim=-1).to(dev()) if ctx_mode else None
    G = torch.randn(2, 3, 96, 96, generator=g).to(dev())
    runs = []
    saved = U._SEED
    try:
        for _ in range(2):
            U._SEED = itertools.count(1000)      # the same dropout seeds in both runs
            m.zero_grad(set_to_none=True)
            xx = x.clone().requires_grad_()
            y = m(xx, t, ctx)
            (y * G).sum().backward()
            torch.cuda.synchronize()
            runs.append((y.detach().clone(), xx.grad.clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
    finally:
        U._SEED = saved
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    for n in runs[0][2]:
        assert torch.equal(runs[0][2][n], runs[1][2][n]), f"{n}: gradient differs between two identical runs"


def test_state_dict_is_reference_compatible():
    from diffusion_model_amd.unet import UNET
    m = UNET(**STL)
    spec = oref.UNetSpec(**STL)
    want = oref.param_shapes(spec)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert list(got) == list(want) and got == want
    assert "time_embeddings.embeddings" not in got  # non-persistent buffer (reference unet.py:54)


@pytest.mark.parametrize("self_attention", [True, False])
def test_overlapped_attention_backward_is_exact(self_attention):
    """One context token: UnetLayer._bwd runs the cross-attention and self-attention backward on the auxiliary stream and
    does not add what comes back through attention(r) -- it must be an exact zero, and input / parameter gradients must
    equal the single-stream order's."""
    from diffusion_model_amd import unet as U
    torch.manual_seed(3)
    layer = U.UnetLayer(False, self_attention, 8, 0.1, 64, 64, 256, num_heads=8, use_cross_attention=True,
                        context_dim=512).to(dev())   # (False: the layers with a cross-attention block only, L1 / L4)
    layer.train()
    B, H = 4, 24
    x = torch.randn(B, H, H, 64, device=dev()).to(torch.bfloat16)
    temb = torch.randn(B, 256, device=dev()).to(torch.bfloat16)
    ctx = torch.randn(B, 1, 512, device=dev()).to(torch.bfloat16)
    dy = torch.randn(B, H // 2, H // 2, 64, device=dev()).to(torch.bfloat16)
    dsk = torch.randn(B, H, H, 64, device=dev()).to(torch.bfloat16)
    outs = {}
    for flag in (True, False):
        U.OVERLAP_DEAD_ATTENTION[0] = flag
        U.KEEP_DEAD_DX[0] = flag
        try:
            U.reseed_dropout(77, 0, 0)
            y, xo, saved = layer._fwd(x, temb, ctx)
            g = U.Grads()
            dx, dts = layer._bwd(dy, dsk, saved, g)
            K_ = U.K
            K_.join_aux(dev()); K_.join_side(dev())
            torch.cuda.synchronize()
            outs[flag] = (dx.float().clone(), dts.float().clone() if torch.is_tensor(dts) else None,
                          {n: g.peek(p).clone() for n, p in layer.named_parameters() if g.peek(p) is not None})
            if flag:
                dead = layer._dead_dx
                assert dead is not None and float(dead.float().abs().max()) == 0.0, "the skipped add is not an exact zero"
        finally:
            U.OVERLAP_DEAD_ATTENTION[0] = True
            U.KEEP_DEAD_DX[0] = False
    a, b = outs[True], outs[False]
    assert torch.equal(a[0], b[0]), "dx differs between the overlapped and the single-stream backward"
    for n in b[2]:
        ref = b[2][n]
        err = float((a[2][n] - ref).abs().max())
        assert err <= 1e-5 * max(1.0, float(ref.abs().max())), (n, err)   # (GroupNorm partial merges use fp32 atomics)
    dead = ("cross_attention.to_q.weight", "cross_attention.to_k.weight")
    if self_attention:
        dead += ("attention.linear_qkv.weight", "attention.linear_out.weight")
    for n in dead:
        assert float(a[2][n].abs().max()) == 0.0, f"{n}: dead parameter with a non-zero gradient"


@pytest.mark.parametrize("L", [0, 1, 5])
def test_unet_context_gradient_vs_oracle(L):
    """d(loss)/d(context) through all four cross-attention layers (reference unet.py:197-201 differentiated by autograd
    there): one token as [B, 512] (L=0) or [B, 1, 512] -- the gradient arrives through to_v alone, dk being exactly zero --
    and five tokens (general attention backward, dk . Wk + dv . Wv).  Checker: the oracle's fp32 autograd; bound = the
    north-star's 2e-2 or 1.5x the oracle's own bf16-storage noise floor on the same inputs."""
    from diffusion_model_amd.unet import UNET
    torch.manual_seed(0)
    m = UNET(**STL)
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        m.output_conv.weight.copy_(torch.randn(m.output_conv.weight.shape, generator=gen) * 0.05)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    spec = oref.UNetSpec(**STL)
    m = m.to(dev()).eval()
    x = torch.randn(2, 3, 32, 32, generator=gen)
    t = torch.tensor([17, 402])
    G = torch.randn(2, 3, 32, 32, generator=gen)
    shape = (2, 512) if L == 0 else (2, L, 512)
    ctx = torch.nn.functional.normalize(torch.randn(shape, generator=gen), dim=-1)

    def oracle_grad(emulate):
        c = ctx.clone().requires_grad_()
        oref.EMULATE_BF16 = emulate
        try:
            y = oref.unet_forward(sd, spec, x, t, c)
            (gc,) = torch.autograd.grad((y * G).sum(), [c])
        finally:
            oref.EMULATE_BF16 = False
        return y.detach(), gc

    y_ref, g_ref = oracle_grad(False)
    _, g_emu = oracle_grad(True)
    floor = rel_l2(g_emu, g_ref)
    cd = ctx.to(dev()).requires_grad_()
    y = m(x.to(dev()), t, cd)
    assert_close(y, y_ref, TOL, "y")
    (y * G.to(dev())).sum().backward()
    assert cd.grad is not None and tuple(cd.grad.shape) == shape and cd.grad.dtype == torch.float32
    err = rel_l2(cd.grad, g_ref)
    print(f"[L={L}] context gradient rel-L2 {err:.3e} (oracle bf16 floor {floor:.3e})")
    assert err <= max(TOL, 1.5 * floor)
    # parameter gradients are unchanged by asking for the context gradient
    m2_grads = {n: p.grad.clone() for n, p in m.named_parameters()}
    for p in m.parameters():
        p.grad = None
    y2 = m(x.to(dev()), t, ctx.to(dev()))
    (y2 * G.to(dev())).sum().backward()
    for n, p in m.named_parameters():
        ref = p.grad
        assert float((m2_grads[n] - ref).abs().max()) <= 1e-4 * max(1e-6, float(ref.abs().max())), n
