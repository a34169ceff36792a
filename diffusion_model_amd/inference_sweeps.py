"""Scheduler / CFG / eta sweeps -- SURVEY.md section 8(f) item 4; drop-in for the reference's ``inference_sweeps.run_sweep``
(inference_sweeps.py:101-138) on the fused sampler.

The reference walks ``len(schedulers) x len(etas) x len(cfg_scales) x len(labels)`` serial B=1 reverse processes (2 x 2 x 5
x 11 = 220 runs of 500 steps with the default arguments).  Here every (eta, cfg, label) combination of one noise schedule
is a SAMPLE of one batched job: guidance and eta are per-sample entries of the step kernel's coefficient rows
(``sampler.step_coefficients``), the beta / alpha-bar tables are shared by the batch, so a whole schedule costs 500
batched forwards (chunked by ``max_batch``).  The start noise of a grid cell is the reference's: ``torch.manual_seed(base_seed
+ idx)`` then ``randn(1,3,96,96)`` for label index ``idx`` (:64-66,131), so the same label starts from the same noise in
every grid; the per-step ancestral noise comes from one generator per chunk (the reference's device RNG stream is not
reproducible across backends).

Arithmetic per step is the reference's ``_sample_from_noise`` (:63-87) = ``visualization._ddpm_step``: with ``eta = 0`` the
update is the deterministic DDPM mean (what the reference's sweep calls its eta-0 variant), on either the cosine or the
linear (``linspace(1e-4, 0.02, T)``, :20-26) schedule."""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import torch

from . import config, utils
from .sampler import CFGSampler
from .utils import DDPM_beta_t_cosine_scheduler, DDPM_beta_t_linear_scheduler

UNCOND_ID = config.UNCOND_ID
device = utils.device


def sweep_grids(model, sched_name: str, cfg_scales: Sequence[float], etas: Sequence[float], labels: Sequence[int],
                base_seed: int = 42, num_steps: int = 500, size: int = 96, max_batch: int = 176) -> Dict[str, torch.Tensor]:
    """All grids of one noise schedule -> {"<sched>_cfg<cfg>_eta<eta>": [len(labels),3,size,size] in [-1,1]}."""
    sched = DDPM_beta_t_linear_scheduler(num_steps) if sched_name == "linear" else DDPM_beta_t_cosine_scheduler(num_steps)
    z_label = []
    for idx in range(len(labels)):
        torch.manual_seed(base_seed + idx)
        z_label.append(torch.randn(1, 3, size, size, device=device))
    cells = [(eta, cfg, idx) for eta in etas for cfg in cfg_scales for idx in range(len(labels))]
    sampler = CFGSampler(model, sched, utils.CONTEXT_TABLE.to(device), UNCOND_ID)
    outs: List[torch.Tensor] = []
    for c0 in range(0, len(cells), max_batch):
        chunk = cells[c0:c0 + max_batch]
        z = torch.cat([z_label[idx] for _, _, idx in chunk])
        lab = torch.tensor([int(labels[idx]) for _, _, idx in chunk], device=device, dtype=torch.long)
        g = torch.tensor([0.0 if int(labels[idx]) == UNCOND_ID else float(cfg) for _, cfg, idx in chunk])
        eta = torch.tensor([float(e) for e, _, _ in chunk])
        gen = torch.Generator(device=device).manual_seed(base_seed + c0)
        outs.append(sampler.sample(z, lab, guidance_scale=g, eta=eta, num_time_steps=num_steps, generator=gen))
    x0 = torch.cat(outs)
    grids, k = {}, 0
    for eta in etas:
        for cfg in cfg_scales:
            grids[f"{sched_name}_cfg{cfg}_eta{eta}"] = x0[k:k + len(labels)]
            k += len(labels)
    return grids


def _save_grid(images: torch.Tensor, out_path: str, nrow: int = 6) -> None:
    """Reference inference_sweeps.py:90-98 (make_grid, value_range (-1,1)) with Pillow: rows of ``nrow`` tiles, 2 px pad."""
    import numpy as np
    from PIL import Image
    x = torch.clamp((images.detach().float().cpu() + 1) / 2, 0, 1)
    n, c, h, w = x.shape
    rows = (n + nrow - 1) // nrow
    sheet = np.zeros((rows * (h + 2) + 2, nrow * (w + 2) + 2, 3), dtype=np.uint8)
    for i in range(n):
        r, q = divmod(i, nrow)
        tile = (x[i].permute(1, 2, 0).numpy() * 255.0 + 0.5).astype(np.uint8)
        sheet[2 + r * (h + 2):2 + r * (h + 2) + h, 2 + q * (w + 2):2 + q * (w + 2) + w] = tile if c == 3 else tile.repeat(3, -1)
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    Image.fromarray(sheet).save(out_path)


def run_sweep(checkpoint: Optional[str], schedulers: List[str], cfg_scales: List[float], etas: List[float],
              labels: List[int], base_seed: int, out_dir: str, *, model=None, num_steps: int = 500, size: int = 96,
              max_batch: int = 176) -> Dict[str, torch.Tensor]:
    """Reference inference_sweeps.py:101-138: same positional arguments and output files
    (``out_dir/<sched>/grid_<sched>_cfg<cfg>_eta<eta>.png``).  ``model``: an already loaded (EMA) model to use instead of
    reading ``checkpoint``.  Returns every grid as a tensor as well."""
    if model is None:
        _, ema = utils.load_checkpoint(checkpoint, NUM_TIME_STEPS=num_steps)
        model = ema.module
    model.eval()
    all_grids: Dict[str, torch.Tensor] = {}
    for sched_name in schedulers:
        grids = sweep_grids(model, sched_name, cfg_scales, etas, labels, base_seed, num_steps, size, max_batch)
        for tag, imgs in grids.items():
            _save_grid(imgs, os.path.join(out_dir, sched_name, f"grid_{tag}.png"), nrow=6)
        all_grids.update(grids)
    return all_grids
