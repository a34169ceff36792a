#!/usr/bin/env python3
"""End-to-end functional check on real kernels: train the STL10-configuration UNet (at 32x32) on a synthetic four-class
dataset (each class a flat colour + a class-specific horizontal/vertical gradient + noise) for a few hundred optimizer steps
through Trainer.micro_step, then sample with the CFG sampler and check that generated images carry their class's colour.
Developer tool behind tests/test_gpu_learn.py: prints the loss curve and the per-class hit rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def make_data(n, H, gen):
    cols = torch.tensor([[0.8, -0.6, -0.6], [-0.6, 0.8, -0.6], [-0.6, -0.6, 0.8], [0.7, 0.7, -0.7]])
    labels = torch.randint(0, 4, (n,), generator=gen)
    ramp = torch.linspace(-0.2, 0.2, H)
    x = cols[labels][:, :, None, None].expand(n, 3, H, H).clone()
    x += torch.where((labels % 2 == 0)[:, None, None, None], ramp[None, None, :, None], ramp[None, None, None, :])
    x += 0.05 * torch.randn(x.shape, generator=gen)
    return x.clamp(-1, 1), labels, cols


def run(steps=300, B=32, H=32, lr=2e-4, seed=0, sample_steps=100, guidance=3.0, verbose=True):
    from diffusion_model_amd.unet import UNET
    from diffusion_model_amd.trainer import Trainer
    from diffusion_model_amd.sampler import CFGSampler
    from diffusion_model_amd.ema import ModelEmaV2
    from diffusion_model_amd.utils import DDPM_beta_t_cosine_scheduler
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    model = UNET(time_steps=500, input_channels=3, output_channels=3, label_embedding=True, dropout_prob=0.1, context_dim=512).to(dev).train()
    sched = DDPM_beta_t_cosine_scheduler(500)
    ema = ModelEmaV2(model, decay=0.98)
    tr = Trainer(model, sched, lr=lr, weight_decay=1e-4, betas=(0.9, 0.99), clip=1.0, ema_decay=0.98, accum=1,
                 total_steps=steps, low_t_weight=0.01, ema=ema)
    table = torch.randn(5, 512, generator=gen)          # four classes + the unconditional row (id 4)
    x_all, y_all, cols = make_data(512, H, gen)
    losses = []
    t0 = time.time()
    for s in range(steps):
        idx = torch.randint(0, x_all.shape[0], (B,), generator=gen)
        x, y = x_all[idx], y_all[idx].clone()
        y[torch.rand(B, generator=gen) < 0.1] = 4     # label drop -> unconditional row
        t = torch.randint(0, 500, (B,), generator=gen)
        e = torch.randn(x.shape, generator=gen)
        loss = tr.micro_step(x.to(dev), table[y].to(dev), t.to(dev), e.to(dev))
        losses.append(float(loss))
    torch.cuda.synchronize()
    train_s = time.time() - t0
    smp = CFGSampler(ema.module, sched, table.to(dev), uncond_id=4)
    labels = torch.arange(4).repeat_interleave(8)
    z = torch.randn(labels.numel(), 3, H, H, generator=gen).to(dev)
    g2 = torch.Generator(device=dev).manual_seed(seed + 1)
    img = smp.sample(z, labels, guidance, 1.0, generator=g2).cpu()
    mean_col = img.mean(dim=(2, 3))                                  # [S, 3]
    pred = ((mean_col[:, None, :] - cols[None]) ** 2).sum(-1).argmin(1)
    hit = float((pred == labels).float().mean())
    first, last = sum(losses[:10]) / 10, sum(losses[-10:]) / 10
    if verbose:
        print(f"loss first10 {first:.4f} last10 {last:.4f}  train {train_s:.1f}s  hit rate {hit:.3f}  finite {bool(torch.isfinite(img).all())}")
        print("per-class mean colours:", [[round(float(v), 2) for v in mean_col[labels == c].mean(0)] for c in range(4)])
    return first, last, hit, img


if __name__ == "__main__":
    run(steps=int(sys.argv[1]) if len(sys.argv) > 1 else 300, lr=float(sys.argv[2]) if len(sys.argv) > 2 else 2e-4,
        guidance=float(sys.argv[3]) if len(sys.argv) > 3 else 3.0)
