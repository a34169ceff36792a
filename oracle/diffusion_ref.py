"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU fp32 restatement of the schedule / train-step / sampler / optimizer arithmetic
around the UNet.  utils.py, main_stl10.py and visualization.py of the reference are
NOT importable in this image (ModuleNotFoundError: torchvision, timm, lpips,
diffusers), so these functions are restated from the reference text, line by line:

  * cosine schedule ......... utils.py:265-289   (clamp [1e-4, 0.999])
                              main.py:39-64      (clamp [0, 0.999], MNIST variant)
  * linear schedule ......... utils.py:254-263
  * q-sample + v-target ..... main_stl10.py:456-465
  * loss (low-t weighting) .. main_stl10.py:477-482, config.py:42-44
  * CFG combine + v->eps .... main_stl10.py:117-137 (dup utils.py:19-41)
  * DDPM/eta step ........... visualization.py:92-109
  * reverse loop + final .... visualization.py:127-152
  * clip_grad_norm_ ......... main_stl10.py:508 (torch.nn.utils semantics)
  * AdamW ................... main_stl10.py:396,509 (torch.optim.AdamW semantics)
  * EMA ..................... main_stl10.py:395,510 (timm ModelEmaV2.update; timm is an
                              un-pinned, un-vendored dependency: restated from its
                              published algorithm ema = d*ema + (1-d)*model over
                              state_dict values)
  * LR lambda ............... main_stl10.py:405-414

Pinned by tests/golden/schedule_*.npz (sha256 + notebook KATs);
the optimizer/EMA/sampler pieces are checked against torch.optim.AdamW /
clip_grad_norm_ on CPU in tests (torch is the reference's own dependency).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor


# ------------------------------------------------------------------ schedules
def cosine_schedule(T: int = 1000, s: float = 0.008, beta_min: float = 1e-4
                    ) -> Tuple[Tensor, Tensor, Tensor]:
    """Returns (beta, alpha, alpha_bar) fp32 [T]; same op sequence as utils.py:274-287 so
    the result is bit-identical with torch CPU fp32."""
    t = torch.arange(0, T + 1)
    abar = torch.cos(((t / T + s) / (1 + s)) * torch.pi / 2) ** 2
    beta = torch.clamp(1 - (abar[1:] / abar[:-1]), min=beta_min, max=0.999)
    alpha = 1 - beta
    return beta, alpha, torch.cumprod(alpha, dim=0)


def linear_schedule(T: int = 1000) -> Tuple[Tensor, Tensor]:
    """utils.py:257-260. Returns (beta, alpha_bar)."""
    beta = torch.linspace(1e-4, 0.02, T)
    return beta, torch.cumprod(1 - beta, dim=0)


# ------------------------------------------------------------------ train-step math
def q_sample_v_target(x: Tensor, e: Tensor, abar_t: Tensor) -> Tuple[Tensor, Tensor]:
    """main_stl10.py:457-465: x_t = sqrt(a) x + sqrt(1-a) e ; v = sqrt(a) e - sqrt(1-a) x."""
    a = abar_t.view(-1, 1, 1, 1)
    sa, so = torch.sqrt(a), torch.sqrt(1 - a)
    return sa * x + so * e, sa * e - so * x


def v_loss(out: Tensor, v: Tensor, t: Tensor, low_t_weight: Optional[float]) -> Tensor:
    """main_stl10.py:477-482. low_t_weight=None -> plain mean MSE; else per-sample
    mean * exp(-w t), then batch mean."""
    se = (out - v) ** 2
    if low_t_weight is None:
        return se.mean()
    per = se.mean(dim=(1, 2, 3)) * torch.exp(-low_t_weight * t.float())
    return per.mean()


def lr_lambda(step: int, total_steps: int) -> float:
    """main_stl10.py:405-412."""
    warm = max(1, int(0.05 * total_steps))
    if step < warm:
        return float(step + 1) / float(max(1, warm))
    prog = (step - warm) / float(max(1, total_steps - warm))
    return 0.5 * (1.0 + math.cos(math.pi * prog))


def clip_grad_norm(grads: Sequence[Tensor], max_norm: float) -> Tensor:
    """torch.nn.utils.clip_grad_norm_: total = ||(||g_i||_2)_i||_2 ;
    coef = clamp(max_norm/(total+1e-6), max=1); g *= coef.  In-place. Returns total."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
               beta1: float = 0.9, beta2: float = 0.99, eps: float = 1e-8,
               wd: float = 1e-4) -> None:
    """torch.optim.AdamW single-tensor math (decoupled decay; step counts from 1). In-place."""
    p.mul_(1 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def ema_update(ema: Dict[str, Tensor], model: Dict[str, Tensor], decay: float) -> None:
    """timm ModelEmaV2.update: for every state_dict value e.copy_(d*e + (1-d)*m)."""
    for k in ema:
        ema[k].copy_(decay * ema[k] + (1.0 - decay) * model[k])


# ------------------------------------------------------------------ sampler
def cfg_eps_from_v(z: Tensor, v_u: Tensor, v_c: Tensor, g: float, abar_t: Tensor) -> Tensor:
    """main_stl10.py:130-136."""
    v_hat = v_u + g * (v_c - v_u)
    a = abar_t.view(-1, 1, 1, 1)
    return torch.sqrt(1.0 - a) * z + torch.sqrt(a) * v_hat


def ddpm_step(z: Tensor, t: int, eps: Tensor, beta: Tensor, abar: Tensor, eta: float,
              noise: Optional[Tensor]) -> Tensor:
    """visualization.py:92-109 (noise supplied by the caller instead of randn_like)."""
    abar_t, beta_t = abar[t], beta[t]
    alpha_t = 1.0 - beta_t
    abar_prev = abar[t - 1] if t > 0 else torch.tensor(1.0)
    z = (1.0 / torch.sqrt(alpha_t)) * (z - (beta_t / torch.sqrt(1.0 - abar_t)) * eps)
    if t > 1 and eta > 0.0:
        var = beta_t * (1.0 - abar_prev) / (1.0 - abar_t)
        z = z + eta * torch.sqrt(var) * noise
    return z


def final_step(z: Tensor, eps: Tensor, beta: Tensor, abar: Tensor) -> Tensor:
    """visualization.py:145-152: t=0 mean step, no noise, clamp to [-1,1]."""
    a0 = 1.0 - beta[0]
    x0 = (1.0 / torch.sqrt(a0)) * (z - (beta[0] / torch.sqrt(1.0 - abar[0])) * eps)
    return torch.clamp(x0, -1, 1)


def reverse_process(v_fn: Callable[[Tensor, Tensor, Tensor], Tensor], z: Tensor,
                    ctx_c: Tensor, ctx_u: Tensor, g: float, eta: float, T: int,
                    beta: Tensor, abar: Tensor, noises: Optional[Dict[int, Tensor]] = None,
                    gen: Optional[torch.Generator] = None, t_stop: int = 0) -> Tensor:
    """visualization.py:127-152 with the CFG callback of main_stl10.py:117-137 inlined.
    v_fn(z, t[B] long, ctx) -> v.  ``noises[t]`` overrides the RNG draw for step t.
    t_stop>0 runs only the first (T-1-t_stop+1) noisy steps (used for bounded timing)."""
    B = z.shape[0]
    z = z.clone()
    for t in range(T - 1, max(0, t_stop - 1), -1):
        if t == 0:
            break
        tt = torch.full((B,), t, dtype=torch.long)
        eps = cfg_eps_from_v(z, v_fn(z, tt, ctx_u), v_fn(z, tt, ctx_c), g, abar[tt])
        n = None
        if t > 1 and eta > 0:
            n = noises[t] if noises is not None and t in noises else torch.randn(z.shape, generator=gen)
        z = ddpm_step(z, t, eps, beta, abar, eta, n)
    if t_stop > 0:
        return z
    tt = torch.zeros((B,), dtype=torch.long)
    eps = cfg_eps_from_v(z, v_fn(z, tt, ctx_u), v_fn(z, tt, ctx_c), g, abar[tt])
    return final_step(z, eps, beta, abar)
