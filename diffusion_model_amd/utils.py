"""Drop-in for the hot-path part of the reference's ``utils.py``: schedulers, model factory,
CLIP context table, CFG epsilon predictor, checkpoint loader.  (Dataset / VAE / LPIPS / CLIP image
scorer of the reference are outside the hot path -- SURVEY.md section 2 rows 9,12,13,17.)

The schedule tables are built on the CPU with the *same torch fp32 op sequence* as the reference
(utils.py:274-287) so they are bit-identical, then moved to the device."""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import config
from .unet import UNET

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")

CONTEXT_TABLE: Optional[torch.Tensor] = None
CONTEXT_TABLE_DIM = 512
CONTEXT_TABLE_LABELS = 1001 if config.IMAGE_NET else 11
_PKG = os.path.dirname(os.path.abspath(__file__))
CLIP_TABLE_PATH = os.path.join(_PKG, "artifacts", "clip",
                               "clip_text_emb_imagenet_vitb32.npz" if config.IMAGE_NET
                               else "clip_text_emb_stl10_vitb32.npz")


class DDPM_beta_t_linear_scheduler(nn.Module):
    """Reference utils.py:254-263 (``alpha_t`` stores alpha-bar)."""

    def __init__(self, num_steps: int = 1000):
        super().__init__()
        beta = torch.linspace(1e-4, 0.02, num_steps, requires_grad=False)
        self.T = num_steps
        self.beta_cpu, self.alpha_bar_cpu = beta, torch.cumprod(1 - beta, dim=0)   # fp32 host copies (sampler scalars)
        self.beta_t = beta.to(device)
        self.alpha_t = self.alpha_bar_cpu.requires_grad_(False).to(device)

    def call(self, t):
        return self.alpha_t[t], self.beta_t[t]


class DDPM_beta_t_cosine_scheduler(nn.Module):
    """Reference utils.py:265-289; ``beta_min=0`` gives the MNIST variant (main.py:56).
    ``.alpha_t`` is alpha-bar (cumulative), ``.alphas`` is 1-beta, as in the reference."""

    def __init__(self, num_steps=1000, s=0.008, beta_min: float = 1e-4):
        super().__init__()
        self.T = num_steps
        self.s = s
        self.beta_min = beta_min
        b, a, ab = self.compute_betas_and_alphas()
        self.beta_cpu, self.alphas_cpu, self.alpha_bar_cpu = b, a, ab   # fp32 host copies (sampler scalars)
        self.beta_t, self.alphas, self.alpha_t = b.to(device), a.to(device), ab.to(device)

    def compute_betas_and_alphas(self):
        t = torch.arange(0, self.T + 1).requires_grad_(False)
        alpha_bar = torch.cos(((t / self.T + self.s) / (1 + self.s)) * torch.pi / 2) ** 2
        betas = torch.clamp(1 - (alpha_bar[1:] / alpha_bar[:-1]), min=self.beta_min, max=0.999)
        alphas = 1 - betas
        return betas, alphas, torch.cumprod(alphas, dim=0)


def load_model(LDM=False, NUM_TIME_STEPS=1000, DROPOUT=0.1, use_flash_attention: bool = True):
    """Reference utils.py:141-168.  Cross-attention is enabled iff a CLIP table is loaded."""
    context_dim = CONTEXT_TABLE_DIM if CONTEXT_TABLE is not None else None
    if LDM:
        model = UNET(time_steps=NUM_TIME_STEPS, Channels=[256, 256, 512, 1024, 512],
                     Attentions=[False, True, True, False], input_channels=4, output_channels=4,
                     label_embedding=True, use_film=False, dropout_prob=DROPOUT, use_cross_attention=True,
                     context_dim=context_dim, use_flash_attention=use_flash_attention)
    else:
        model = UNET(time_steps=NUM_TIME_STEPS, input_channels=3, output_channels=3, label_embedding=True,
                     use_film=False, dropout_prob=DROPOUT, use_cross_attention=True, context_dim=context_dim,
                     use_flash_attention=use_flash_attention)
    return model.to(device)


def _load_clip_context_table(path: str = CLIP_TABLE_PATH) -> torch.Tensor:
    """Reference utils.py:170-181, accepting the ``.npz`` the reference actually ships (its code looks
    for a ``.pt`` that ``.gitignore`` excludes) as well as a ``.pt`` dict; prefers ``embeddings_norm``."""
    global CONTEXT_TABLE
    if path.endswith(".npz"):
        d = np.load(path, allow_pickle=False)
        key = "embeddings_norm" if "embeddings_norm" in d.files else "embeddings"
        table = torch.from_numpy(d[key].astype(np.float32))
    else:
        obj = torch.load(path, map_location="cpu", weights_only=True)
        emb = obj.get("embeddings_norm", obj.get("embeddings"))
        table = emb.float() if isinstance(emb, torch.Tensor) else torch.tensor(emb, dtype=torch.float32)
    CONTEXT_TABLE = table
    return table


def _labels_to_context(labels: torch.Tensor) -> torch.Tensor:
    """Reference utils.py:183-186."""
    assert CONTEXT_TABLE is not None, "CONTEXT_TABLE not loaded"
    return CONTEXT_TABLE.to(labels.device)[labels.long()]


def _cfg_predict_eps_from_v(model, scheduler, z, t_tensor, labels, guidance_scale: float):
    """Reference utils.py:19-41 / main_stl10.py:117-137 (the ``predict_eps_fn`` callback contract of
    visualization.run_reverse_process).  cond+uncond are batched into ONE UNet call (numerically
    identical per sample in eval mode: GroupNorm is per-sample)."""
    t_tensor = t_tensor.to(z.device)
    B = z.shape[0]
    null = torch.full_like(labels, config.UNCOND_ID)
    ctx = _labels_to_context(torch.cat([null, labels]).to(z.device))
    tt = t_tensor if t_tensor.numel() == B else t_tensor.reshape(-1)[:1].expand(B)
    v = model(torch.cat([z, z]), torch.cat([tt, tt]), ctx)
    v_uncond, v_cond = v[:B], v[B:]
    v_hat = v_uncond + guidance_scale * (v_cond - v_uncond)
    alpha_bar_t = scheduler.alpha_t[tt]
    alpha_t_sqrt = torch.sqrt(alpha_bar_t).view(-1, 1, 1, 1)
    sigma_t = torch.sqrt(1.0 - alpha_bar_t).view(-1, 1, 1, 1)
    return sigma_t * z + alpha_t_sqrt * v_hat


def load_checkpoint(checkpoint_path: str, clip_table_path: str = CLIP_TABLE_PATH, LDM=config.LDM,
                    NUM_TIME_STEPS=1000, DROPOUT=config.DROPOUT, EMA_DECAY=config.EMA_DECAY):
    """Reference utils.py:291-315.  Loads the reference's ``{'weights','optimizer','ema'}`` dict
    (tensor-only load: ``weights_only=True``)."""
    from .ema import ModelEmaV2
    checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    global CONTEXT_TABLE
    if CONTEXT_TABLE is None and os.path.exists(clip_table_path):
        _load_clip_context_table(clip_table_path)
    model = load_model(LDM=LDM, NUM_TIME_STEPS=NUM_TIME_STEPS, DROPOUT=DROPOUT)
    if isinstance(checkpoint, dict) and "weights" in checkpoint:
        model.load_state_dict(checkpoint["weights"])
    else:
        model.load_state_dict(checkpoint)
    ema = ModelEmaV2(model, decay=EMA_DECAY)
    if isinstance(checkpoint, dict) and "ema" in checkpoint:
        ema.load_state_dict(checkpoint["ema"])
    return model, ema
