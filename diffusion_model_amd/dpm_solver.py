"""DPM-Solver++(2M) sampling for the v-prediction UNet -- SURVEY.md section 8(f) item 4.

The reference samples with ``diffusers.DPMSolverMultistepScheduler`` (main_stl10.py:674-684: ``num_train_timesteps =
NUM_TIME_STEPS, beta_start 0.00085, beta_end 0.012, beta_schedule "scaled_linear", prediction_type "v_prediction",
algorithm_type "dpmsolver++", solver_order 2``) driven by ``visualization.sample_dpm_solver`` (:480-534):
``scheduler.set_timesteps(n, device)``, then per timestep two B=1 forwards (uncond, cond), the CFG combine and
``scheduler.step(v_pred, t, latents).prev_sample``.

diffusers is a third-party dependency that is neither vendored in the reference nor installed in this image, and the
reference pins no version: the class below restates the published algorithm of that scheduler for exactly the
configuration the reference constructs (multistep DPM-Solver++ with the midpoint rule, "linspace" timestep spacing,
final sigma 0, first-order update on the first and the last step), in fp32 torch ops in the order diffusers applies them.
PARITY UNPINNED: no reference fixture or test covers this path; ``oracle/dpm_solver_ref.py`` is an independent numpy
restatement used as the checker.

Two ways in:
* ``DPMSolverMultistepScheduler.step(model_output, t, sample).prev_sample`` keeps diffusers' calling convention (plain
  torch ops, any device), so ``visualization.sample_dpm_solver`` reads like the reference's;
* ``CFGSampler.sample_dpm`` (sampler.py) is the fast path: one UNet forward over [uncond; cond] and ONE fused kernel
  (``dm_cfg_dpmpp2m_step``: CFG combine, v -> x0, multistep update, x0 history) per step, for a whole batch of samples with
  per-sample guidance, using ``step_coefficients`` below."""
from __future__ import annotations

from types import SimpleNamespace
from typing import List, Optional

import numpy as np
import torch


class DPMSolverMultistepScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02,
                 beta_schedule: str = "linear", prediction_type: str = "epsilon", algorithm_type: str = "dpmsolver++",
                 solver_order: int = 2, **other):
        if algorithm_type != "dpmsolver++" or solver_order not in (1, 2):
            raise ValueError("only algorithm_type='dpmsolver++' with solver_order 1 or 2 is built (main_stl10.py:682-683)")
        if prediction_type != "v_prediction":
            raise ValueError("only prediction_type='v_prediction' is built (main_stl10.py:681)")
        for k, v in other.items():  # diffusers options the reference leaves at their defaults
            defaults = dict(solver_type="midpoint", lower_order_final=True, euler_at_final=False, use_karras_sigmas=False,
                            final_sigmas_type="zero", timestep_spacing="linspace", steps_offset=0, thresholding=False)
            if k not in defaults or defaults[k] != v:
                raise ValueError(f"option {k}={v!r} is not supported by this restatement")
        if beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise ValueError(f"beta_schedule {beta_schedule!r} is not supported")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, prediction_type=prediction_type,
                                      algorithm_type=algorithm_type, solver_order=solver_order)
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1, dtype=torch.int64)
        self.sigmas = torch.flip(((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5, [0])
        self._reset()

    def _reset(self) -> None:
        self.model_outputs: List[Optional[torch.Tensor]] = [None] * self.config.solver_order
        self.lower_order_nums = 0
        self._step_index = 0

    @property
    def step_index(self) -> int:
        return self._step_index

    def set_timesteps(self, num_inference_steps: int, device=None) -> None:
        T = self.config.num_train_timesteps
        timesteps = np.linspace(0, T - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sig = np.interp(timesteps, np.arange(0, len(sig)), sig)
        sig = np.concatenate([sig, [0.0]]).astype(np.float32)          # final sigma = 0
        self.sigmas = torch.from_numpy(sig)
        self.timesteps = torch.from_numpy(timesteps).to(device=device, dtype=torch.int64)
        self.num_inference_steps = len(timesteps)
        self._reset()

    @staticmethod
    def _alpha_sigma(sigma: torch.Tensor):
        alpha_t = 1 / ((sigma ** 2 + 1) ** 0.5)
        return alpha_t, sigma * alpha_t

    def step_coefficients(self, i: int, have_history: bool):
        """fp32 scalars of step i: (alpha_s, sigma_s, sigma_t/sigma_s, alpha_t*(exp(-h)-1), 1/r0, second-order?).
        ``have_history``: an x0 of the previous step exists (False on the first step)."""
        n = len(self.sigmas) - 1
        a_s, s_s = self._alpha_sigma(self.sigmas[i])
        a_t, s_t = self._alpha_sigma(self.sigmas[i + 1])
        lam_t = torch.log(a_t) - torch.log(s_t)
        lam_s = torch.log(a_s) - torch.log(s_s)
        h = lam_t - lam_s
        ratio = s_t / s_s
        cc = a_t * (torch.exp(-h) - 1.0)
        second = self.config.solver_order == 2 and have_history and i != n - 1   # the final step is first-order
        inv_r0 = torch.tensor(0.0)
        if second:
            a_p, s_p = self._alpha_sigma(self.sigmas[i - 1])
            lam_p = torch.log(a_p) - torch.log(s_p)
            r0 = (lam_s - lam_p) / h
            inv_r0 = 1.0 / r0
        return a_s, s_s, ratio, cc, inv_r0, second

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor):
        """diffusers' calling convention: -> object with ``.prev_sample``.  ``timestep`` is not used to find the position:
        like diffusers after its first call, the scheduler counts its own steps."""
        i = self._step_index
        if i >= len(self.sigmas) - 1:
            raise IndexError("step() called more often than set_timesteps() prepared")
        a_s, s_s, ratio, cc, inv_r0, second = self.step_coefficients(i, self.lower_order_nums >= 1)
        dev = sample.device
        a_s, s_s, ratio, cc, inv_r0 = (c.to(dev) for c in (a_s, s_s, ratio, cc, inv_r0))
        sample = sample.to(torch.float32)
        x0 = a_s * sample - s_s * model_output
        for k in range(self.config.solver_order - 1):
            self.model_outputs[k] = self.model_outputs[k + 1]
        self.model_outputs[-1] = x0
        if not second:
            prev = ratio * sample - cc * x0
        else:
            m0, m1 = self.model_outputs[-1], self.model_outputs[-2]
            d1 = inv_r0 * (m0 - m1)
            prev = ratio * sample - cc * m0 - 0.5 * cc * d1
        if self.lower_order_nums < self.config.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        return SimpleNamespace(prev_sample=prev.to(model_output.dtype))
