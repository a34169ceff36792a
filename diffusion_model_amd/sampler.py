"""CFG + DDPM/eta sampler on the fused HIP step kernel.

Mirrors visualization.py:92-152 (`_ddpm_step`, `run_reverse_process`) and
main_stl10.py:117-137 (`_cfg_predict_eps_from_v`) of the reference; the per-step scalar
coefficients are computed on the host in fp32 with the same torch ops as the reference
(so they are bit-identical), the per-element work is ONE kernel (dm_cfg_ddpm_step)."""
from __future__ import annotations

import torch


def step_coefficients(beta: torch.Tensor, abar: torch.Tensor, t: int, guidance, eta: float, batch: int,
                      final: bool = False) -> torch.Tensor:
    """[batch, 8] fp32 rows {sqrt(1-abar_t), sqrt(abar_t), 1/sqrt(alpha_t), beta_t/sqrt(1-abar_t),
    eta*sqrt(posterior_var) or 0, guidance, clamp flag, 0}.  `beta`/`abar` are the fp32 CPU tables;
    `guidance` is a float or a per-sample [batch] tensor (0.0 for the unconditional label,
    visualization.py:424)."""
    beta, abar = beta.detach().cpu(), abar.detach().cpu()
    abar_t, beta_t = abar[t], beta[t]
    alpha_t = 1.0 - beta_t
    sig = torch.sqrt(1.0 - abar_t)
    sa = torch.sqrt(abar_t)
    ria = 1.0 / torch.sqrt(alpha_t)
    bs = beta_t / torch.sqrt(1.0 - abar_t)
    ns = torch.tensor(0.0)
    if (not final) and t > 1 and eta > 0.0:
        abar_prev = abar[t - 1]
        ns = eta * torch.sqrt(beta_t * (1.0 - abar_prev) / (1.0 - abar_t))
    row = torch.stack([sig, sa, ria, bs, ns.float(), torch.tensor(0.0), torch.tensor(1.0 if final else 0.0),
                       torch.tensor(0.0)]).float()
    coef = row[None, :].repeat(batch, 1)
    coef[:, 5] = torch.as_tensor(guidance, dtype=torch.float32)
    return coef.contiguous()
