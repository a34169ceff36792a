"""Monitoring images of the training loop -- SURVEY.md section 8(f) item 2: the callers on the output side of the
sampler.  Same function names, arguments and output paths as the reference's ``visualization.py``
(generate_denoise_collage :183-222, generate_random_sample_image :225-261, reconstruct_from_dataset_sample :264-330,
reconstruct_sweep :333-401, sample_dpm_solver :480-534, generate_fixed_seed_class_grid_dpm :537-617;
generate_fixed_seed_class_grid lives in main_stl10.py), so ``main_stl10.train``'s epoch hook
can call them unchanged.

What differs, on purpose:
* every function is ONE batched job on the fused sampler (``sampler.CFGSampler``): the reconstruction sweep runs all of
  its start times as one batch (``CFGSampler.sample_from``), where the reference walks 1 + len(t_list) serial B=1
  reverse processes;
* ``predict_eps_fn`` is accepted for signature compatibility; when it is anything other than this package's
  ``utils._cfg_predict_eps_from_v`` (or None) the reference's own per-step loop (``sampler.run_reverse_process``) is used
  with that callback, so custom predictors keep working;
* panels are written as PNG strips with Pillow (a caption row above the images) instead of matplotlib figures: the
  image content is the reference's (``tensor_to_image``: clamp((x+1)/2, 0, 1)), the typography is not."""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import config, utils
from .sampler import CFGSampler, run_reverse_process

UNCOND_ID = config.UNCOND_ID
STL10_CLASS_NAMES = {0: "airplane", 1: "bird", 2: "car", 3: "cat", 4: "deer", 5: "dog", 6: "horse", 7: "monkey",
                     8: "ship", 9: "truck", 10: "uncond"}


def sample_noise(LDM: bool = False, W: int = 96, H: int = 96, batch_size: int = 1,
                 device: torch.device = torch.device("cuda")) -> torch.Tensor:
    """Reference visualization.py:31-38 (image-space branch; the latent branch needs the un-fetchable VAE)."""
    if LDM:
        raise NotImplementedError("latent-diffusion sampling needs the VAE, which is out of scope (SURVEY 8f item 3)")
    return torch.randn(batch_size, 3, W, H, device=device)


def tensor_to_image(img_tensor: torch.Tensor) -> np.ndarray:
    """Reference visualization.py:40-69: [N,C,H,W] / [C,H,W] in [-1,1] -> HxWxC float array in [0,1] (first of a batch)."""
    img = img_tensor.detach().float().cpu()
    if img.ndim == 4:
        img = img[0]
    if img.ndim == 2:
        img = img.unsqueeze(0)
    img = torch.clamp((img + 1) / 2, 0, 1)
    return img.permute(1, 2, 0).numpy()


def save_image_strip(images: Sequence, titles: Sequence[str], out_path: str, scale: int = 2) -> None:
    """One row of panels with a caption above each (stands in for the reference's matplotlib subplots)."""
    from PIL import Image, ImageDraw
    tiles = []
    for im in images:
        a = tensor_to_image(im) if isinstance(im, torch.Tensor) else np.asarray(im)
        if a.shape[-1] == 1:
            a = np.repeat(a, 3, axis=-1)
        t = Image.fromarray((a * 255.0 + 0.5).astype(np.uint8))
        tiles.append(t.resize((t.width * scale, t.height * scale), Image.NEAREST))
    cap = 14
    w, h = tiles[0].width, tiles[0].height
    sheet = Image.new("RGB", (w * len(tiles) + 4 * (len(tiles) + 1), h + cap + 8), (255, 255, 255))
    draw = ImageDraw.Draw(sheet)
    for i, (t, title) in enumerate(zip(tiles, titles)):
        x = 4 + i * (w + 4)
        draw.text((x, 2), str(title)[: max(4, w // 6)], fill=(0, 0, 0))
        sheet.paste(t, (x, cap + 4))
    out_dir = os.path.dirname(out_path)
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    sheet.save(out_path)


def save_denoising_collage(images: Sequence[torch.Tensor], timesteps: Sequence[int], out_path: str) -> None:
    """Reference visualization.py:72-89."""
    save_image_strip(images, [f"t={t}" for t in timesteps], out_path)


def _fast(predict_eps_fn) -> bool:
    return predict_eps_fn is None or predict_eps_fn is utils._cfg_predict_eps_from_v


def _sampler(model, scheduler) -> CFGSampler:
    assert utils.CONTEXT_TABLE is not None, "CONTEXT_TABLE not loaded"
    return CFGSampler(model, scheduler, utils.CONTEXT_TABLE, UNCOND_ID)


def _reverse(model, scheduler, z, labels, guidance_scale, eta, num_time_steps, predict_eps_fn, capture):
    """-> (x0 clamped, {t: clamped latent}) like run_reverse_process, on the fused sampler when possible."""
    if not _fast(predict_eps_fn):
        return run_reverse_process(model, scheduler, z, labels, guidance_scale, eta, num_time_steps, predict_eps_fn,
                                   capture_timesteps=list(capture))
    snaps: Dict[int, torch.Tensor] = {int(t): None for t in capture}
    snaps[0] = None
    x0 = _sampler(model, scheduler).sample(z, labels, guidance_scale, eta, num_time_steps=num_time_steps, snapshots=snaps)
    return x0, {t: v for t, v in snaps.items() if v is not None}


def generate_denoise_collage(model, scheduler, epoch: int, step: int, timesteps: Optional[Sequence[int]],
                             device: torch.device, num_time_steps: int, sampling_eta: float, predict_eps_fn: Callable,
                             output_dir: str, labels: Optional[torch.Tensor] = None, guidance_scale: float = 0.0) -> None:
    """Reference visualization.py:183-222: one reverse process, snapshots at ``timesteps``, one PNG row."""
    if timesteps is None:
        timesteps = [999, 700, 550, 400, 300, 200, 100, 50, 15, 0]
    with torch.no_grad():
        model.eval()
        z = sample_noise(False).to(device)
        if labels is None:
            labels = torch.randint(0, 10, (1,), device=device)
        _, snapshots = _reverse(model, scheduler, z, labels, guidance_scale, sampling_eta, num_time_steps,
                                predict_eps_fn, timesteps)
        kept = [t for t in timesteps if t in snapshots]
        if not kept:
            return
        out_dir = os.path.join(output_dir, f"epoch_{epoch}")
        save_denoising_collage([snapshots[t] for t in kept], kept,
                               os.path.join(out_dir, f"epoch_{epoch}_step_{step}_stl10_collage.png"))


def generate_random_sample_image(model, scheduler, epoch: int, device: torch.device, num_time_steps: int,
                                 sampling_eta: float, predict_eps_fn: Callable, output_dir: str,
                                 labels: Optional[torch.Tensor] = None, guidance_scale: float = 0.0) -> None:
    """Reference visualization.py:225-261."""
    with torch.no_grad():
        model.eval()
        z = sample_noise(False).to(device)
        if labels is None:
            labels = torch.randint(0, 10, (1,), device=device)
        x0, _ = _reverse(model, scheduler, z, labels, guidance_scale, sampling_eta, num_time_steps, predict_eps_fn, [])
        save_image_strip([x0], [f"STL10 sample - Epoch {epoch}"],
                         os.path.join(output_dir, f"epoch_{epoch}", f"epoch_{epoch}_stl10_sample.png"), scale=3)


def _noised(scheduler, x: torch.Tensor, t0: int) -> torch.Tensor:
    a_bar = scheduler.alpha_t[t0].to(x.device).view(1, 1, 1, 1)
    return torch.sqrt(a_bar) * x + torch.sqrt(1 - a_bar) * torch.randn_like(x)


def _reconstruct(model, scheduler, x, labels, t0s: List[int], guidance_scale, eta, predict_eps_fn, num_time_steps):
    """x: [1,C,H,W]; -> ([x_t for each t0], [x0_hat for each t0]); all start times in one batch on the fast path."""
    t0s = [int(max(1, min(num_time_steps - 1, t))) for t in t0s]
    x_ts = [_noised(scheduler, x, t0) for t0 in t0s]
    if _fast(predict_eps_fn):
        recon = _sampler(model, scheduler).sample_from(torch.cat(x_ts), labels.expand(len(t0s)).contiguous(), t0s,
                                                       guidance_scale=guidance_scale, eta=eta)
        return t0s, x_ts, [recon[i:i + 1] for i in range(len(t0s))]
    from .sampler import _ddpm_step
    outs = []
    for t0, x_t in zip(t0s, x_ts):  # the reference's serial walk (visualization.py:296-309), custom predictor
        z = x_t.clone()
        for t in reversed(range(1, t0 + 1)):
            eps = predict_eps_fn(model, scheduler, z, torch.tensor([t], device=z.device), labels, guidance_scale)
            z = _ddpm_step(z, t, eps, scheduler, eta)
        eps = predict_eps_fn(model, scheduler, z, torch.tensor([0], device=z.device), labels, guidance_scale)
        beta_0, abar_0 = scheduler.beta_t[0], scheduler.alpha_t[0]
        outs.append(torch.clamp((1.0 / torch.sqrt(1.0 - beta_0)) * (z - (beta_0 / torch.sqrt(1.0 - abar_0)) * eps), -1, 1))
    return t0s, x_ts, outs


def reconstruct_from_dataset_sample(model, scheduler, dataset, epoch: int, device: torch.device, num_time_steps: int,
                                    sampling_eta: float, predict_eps_fn: Callable, output_dir: str, t_start: int = 350,
                                    guidance_scale: float = 0.0) -> None:
    """Reference visualization.py:264-330: a random dataset image, noised to ``t_start``, walked back to 0."""
    with torch.no_grad():
        model.eval()
        idx = int(torch.randint(0, len(dataset), (1,)).item())
        x, label = dataset[idx]
        x = x.unsqueeze(0).to(device).float()
        label = int(label)
        labels = torch.tensor([label if guidance_scale != 0.0 else UNCOND_ID], device=device, dtype=torch.long)
        (t0,), (x_t,), (recon,) = _reconstruct(model, scheduler, x, labels, [t_start], guidance_scale, sampling_eta,
                                               predict_eps_fn, num_time_steps)
        save_image_strip([torch.clamp(x, -1, 1), torch.clamp(x_t, -1, 1), recon],
                         [STL10_CLASS_NAMES.get(label, str(label)), f"x_t (t={t0})", "recon"],
                         os.path.join(output_dir, f"epoch_{epoch}", f"epoch_{epoch}_reconstruction_t{t0}.png"))


def reconstruct_sweep(model, scheduler, loader, epoch: int, device: torch.device, num_time_steps: int,
                      sampling_eta: float, predict_eps_fn: Callable, output_dir: str,
                      t_list: Optional[Sequence[int]] = None, guidance_scale: float = 0.0) -> None:
    """Reference visualization.py:333-401: the first image of the loader, reconstructed from several noise levels."""
    if t_list is None:
        t_list = [400, 300, 200, 100]
    with torch.no_grad():
        model.eval()
        xb, lb = next(iter(loader))
        x, label = xb[0].unsqueeze(0).to(device).float(), int(lb[0])
        labels = torch.tensor([label if guidance_scale != 0.0 else UNCOND_ID], device=device, dtype=torch.long)
        t0s, _, recons = _reconstruct(model, scheduler, x, labels, list(t_list), guidance_scale, sampling_eta,
                                      predict_eps_fn, num_time_steps)
        name = STL10_CLASS_NAMES.get(label, str(label))
        save_image_strip([torch.clamp(x, -1, 1)] + recons, [name] + [f"recon t={t} {name}" for t in t0s],
                         os.path.join(output_dir, f"epoch_{epoch}", f"epoch_{epoch}_reconstruction_sweep.png"))


def sample_dpm_solver(model, scheduler, label_id: int, cfg_scale: float, num_inference_steps: int,
                      device: torch.device, latent_shape: tuple = (1, 3, 96, 96)) -> torch.Tensor:
    """Reference visualization.py:480-534 (DPM-Solver++ with CFG for v-prediction): same arguments and return value (the
    final latents, un-clamped).  ``scheduler`` is a ``dpm_solver.DPMSolverMultistepScheduler``; guidance applies only when
    ``cfg_scale > 1.0`` (:513-526).  ``latent_shape`` defaults to the image-space model's (the reference's default is the
    latent model's (1,4,32,32), whose VAE is out of scope).  Runs on the fused sampler."""
    latents = torch.randn(latent_shape, device=device)
    labels = torch.full((latent_shape[0],), int(label_id), device=device, dtype=torch.long)
    return _sampler(model, scheduler).sample_dpm(latents, labels, scheduler, cfg_scale=cfg_scale,
                                                 num_inference_steps=num_inference_steps)


def generate_fixed_seed_class_grid_dpm(model, scheduler, epoch: int, step: int, cfg_scale: float,
                                       num_inference_steps: int, label_ids: List[int], label_names: dict, base_seed: int,
                                       device: torch.device, output_dir: str,
                                       latent_shape: tuple = (1, 3, 96, 96)) -> torch.Tensor:
    """Reference visualization.py:537-617: one sample per label, start noise of label i drawn after
    ``torch.manual_seed(base_seed + i)``, guidance 0.0 (= none) for the unconditional label (:563), one PNG under
    ``output_dir/epoch_N``.  All labels run as ONE batched DPM-Solver++ job instead of len(label_ids) serial B=1 runs.
    (The reference's optional CLIP top-1 caption needs the un-fetchable CLIP model and is left out.)  Also returns the
    batch of final latents."""
    zs = []
    for i in range(len(label_ids)):
        torch.manual_seed(base_seed + i)
        zs.append(torch.randn(latent_shape, device=device))
    labels = torch.tensor(list(label_ids), device=device, dtype=torch.long)
    g = torch.tensor([0.0 if int(l) == UNCOND_ID else float(cfg_scale) for l in label_ids])
    out = _sampler(model, scheduler).sample_dpm(torch.cat(zs), labels, scheduler, cfg_scale=g,
                                                num_inference_steps=num_inference_steps)
    path = os.path.join(output_dir, f"epoch_{epoch}",
                        f"epoch_{epoch}_step_{step}_dpm_solver_cfg{cfg_scale}_steps{num_inference_steps}.png")
    save_image_strip([out[i:i + 1] for i in range(out.shape[0])], [label_names.get(int(l), str(l)) for l in label_ids],
                     path)
    print(f"Saved: {path}")
    return out
