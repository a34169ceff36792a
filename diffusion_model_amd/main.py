"""Drop-in for the reference's MNIST baseline ``main.train()`` (main.py:142-197) on the HIP hot path:
32x32x1 (28x28 zero-padded by 2), attention-free UNet, epsilon target, Adam at a constant 5e-5, gradient
clipping at 0.5, EMA 0.999, label drop p = 0.1 -> 10, one optimizer step per batch.  The integer labels the
reference passes as ``context`` are ignored by a UNet without cross-attention (unet.py:271), so they are
not sent to the device path.  The torchvision MNIST reader and the matplotlib collages are outside the hot
path: batches come from ``train_loader`` (x in [-1,1] of shape [B,1,28,28] or [B,1,32,32])."""
from __future__ import annotations

from typing import Callable, Iterable, Optional

import torch
import torch.nn.functional as F

from .ema import ModelEmaV2
from .trainer import Trainer
from .unet import UNET
from .utils import DDPM_beta_t_cosine_scheduler, device

NUM_EPOCHS = 50          # main.py:14
BATCH_SIZE = 64          # main.py:15
LR = 5e-5                # main.py:16
EMA_DECAY = 0.999        # main.py:17
NUM_TIME_STEPS = 1000    # main.py:18
GRAD_CLIP = 0.5          # main.py:176
P_UNCOND = 0.1           # main.py:163-164


class SyntheticDigits:
    """Stand-in for the MNIST DataLoader: seeded [B,1,28,28] batches in [-1,1], labels 0..9."""

    def __init__(self, n_batches=4, batch_size=BATCH_SIZE, seed=1234):
        self.n, self.bs, self.seed = n_batches, batch_size, seed

    def __len__(self):
        return self.n

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        for _ in range(self.n):
            yield (torch.rand(self.bs, 1, 28, 28, generator=g) * 2 - 1), torch.randint(0, 10, (self.bs,), generator=g)


def build_model() -> UNET:
    """main.py:148-149."""
    return UNET(time_steps=NUM_TIME_STEPS, label_embedding=True, Attentions=[False] * 6, use_film=False)


def train(checkpoint_path=None, *, train_loader: Optional[Iterable] = None, num_epochs: int = NUM_EPOCHS,
          log: Callable = print):
    """Reference main.py:142-197.  Returns ``(model, ema, trainer)``; the checkpoint dict the reference builds
    (and never saves, main.py:192-197) is available as ``main_stl10._checkpoint_dict(model, trainer, ema)``."""
    scheduler = DDPM_beta_t_cosine_scheduler(NUM_TIME_STEPS, beta_min=0.0)   # main.py:56 clamps at 0
    if train_loader is None:
        log("[warn] no train_loader given: the MNIST reader is outside this package's scope; using SyntheticDigits")
        train_loader = SyntheticDigits()
    model = build_model().to(device)
    if checkpoint_path is not None:
        model.load_state_dict(torch.load(checkpoint_path, map_location="cpu", weights_only=True)["weights"])
    ema = ModelEmaV2(model, decay=EMA_DECAY)
    trainer = Trainer(model, scheduler, lr=LR, weight_decay=0.0, betas=(0.9, 0.999), clip=GRAD_CLIP,
                      ema_decay=EMA_DECAY, accum=1, total_steps=0, low_t_weight=None, target="e", adamw=False,
                      ema=ema)
    for epoch in range(num_epochs):
        sum_loss = torch.zeros(1, device=device)
        model.train()
        for x, labels in train_loader:
            x = x.to(device).float()
            _ = torch.rand(x.shape[0], device=device)            # the label-drop draw (labels unused downstream)
            t = torch.randint(0, NUM_TIME_STEPS, (x.shape[0],)).to(device)
            if x.shape[-1] == 28:
                x = F.pad(x, (2, 2, 2, 2))                       # main.py:169
            e = torch.randn_like(x)
            sum_loss += trainer.micro_step(x, None, t, e).detach().reshape(-1)[:1]
        log(f"Epoch {epoch + 1} | Loss {float(sum_loss.item()) / max(1, len(train_loader)):.5f}")
    return model, ema, trainer
