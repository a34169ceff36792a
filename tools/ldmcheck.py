"""The latent-diffusion UNet variant (reference utils.py:145-156) against the CPU fp32 oracle:
forward, dL/dx and every parameter gradient.  --no-attn: the conv / GroupNorm / GEMM stack alone."""
import sys, torch
sys.path.insert(0, ".")
from diffusion_model_amd.unet import UNET
from oracle import unet_ref as oref
torch.manual_seed(0)
WITH_ATTN = "--no-attn" not in sys.argv
kw = dict(Channels=[256, 256, 512, 1024, 512], Attentions=[False, True, True, False] if WITH_ATTN else [False] * 4,
          input_channels=4, output_channels=4, dropout_prob=0.0, context_dim=512 if WITH_ATTN else None,
          use_cross_attention=WITH_ATTN, time_steps=1000)
m = UNET(**kw)
with torch.no_grad():
    m.output_conv.weight.normal_(0, 0.05)
sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
print("params", sum(v.numel() for v in sd.values()))
m = m.cuda()
x = torch.randn(2, 4, 32, 32)
t = torch.tensor([10, 700])
xg = x.cuda().requires_grad_()
ctx = torch.randn(2, 512) if WITH_ATTN else None
y = m(xg, t.cuda(), ctx.cuda() if WITH_ATTN else None)
print("fwd ok", tuple(y.shape), float(y.abs().max()))
spec = oref.UNetSpec(**{k: v for k, v in kw.items() if k in oref.UNetSpec.__dataclass_fields__})
xr = x.clone().requires_grad_()
p = {k: v.clone().requires_grad_() for k, v in sd.items()}
yr = oref.unet_forward(p, spec, xr, t, ctx)
print("fwd rel l2", float((y.detach().cpu() - yr).norm() / yr.norm()))
w = torch.randn_like(yr)
(y * w.cuda()).sum().backward()
(yr * w).sum().backward()
print("dx rel l2", float((xg.grad.cpu() - xr.grad).norm() / xr.grad.norm()))
worst = 0
for k, v in m.named_parameters():
    r = float((v.grad.cpu() - p[k].grad).norm() / (p[k].grad.norm() + 1e-20))
    worst = max(worst, r)
    if r > 5e-2: print("  grad", k, r)
print("worst param-grad rel l2", worst)
