"""TEST INFRASTRUCTURE ONLY (oracle) -- never imported by the product path.

Numpy restatement of the sampler the reference runs in ``visualization.sample_dpm_solver`` (visualization.py:480-534) with
the scheduler built at main_stl10.py:674-684: ``diffusers.DPMSolverMultistepScheduler(num_train_timesteps, beta_start
0.00085, beta_end 0.012, "scaled_linear", prediction_type "v_prediction", algorithm_type "dpmsolver++", solver_order 2)``.

diffusers is a third-party dependency, absent from /root/reference and from this image, and the reference pins no version
of it.  What follows restates the algorithm that scheduler publishes (Lu et al., DPM-Solver++, multistep 2M with the
midpoint rule; diffusers' defaults: timestep_spacing "linspace", final_sigmas_type "zero", lower_order_final) from the
formulas, written independently of ``diffusion_model_amd/dpm_solver.py`` (numpy, explicit loops over the history).
PARITY UNPINNED: the reference holds no fixture, golden vector or test for this path.

    sigma_k   = sqrt((1 - abar_k) / abar_k)                  on the training grid, interpolated at the chosen timesteps
    alpha(s)  = 1 / sqrt(s^2 + 1),  sig(s) = s * alpha(s),   lambda(s) = log alpha(s) - log sig(s)
    x0_i      = alpha_i * x - sig_i * v                      (v-prediction -> data prediction)
    1st order:  x' = (sig_{i+1} / sig_i) x - alpha_{i+1} (exp(-h) - 1) x0_i,            h = lambda_{i+1} - lambda_i
    2nd order:  x' = ... - 0.5 alpha_{i+1} (exp(-h) - 1) (x0_i - x0_{i-1}) / r0,       r0 = (lambda_i - lambda_{i-1}) / h
    first step and last step (sigma -> 0) are first order."""
from __future__ import annotations

import numpy as np

f32 = np.float32


def training_sigmas(num_train_timesteps: int, beta_start: float, beta_end: float, schedule: str = "scaled_linear"):
    if schedule == "scaled_linear":
        betas = np.linspace(f32(beta_start ** 0.5), f32(beta_end ** 0.5), num_train_timesteps, dtype=np.float64).astype(f32) ** 2
    elif schedule == "linear":
        betas = np.linspace(beta_start, beta_end, num_train_timesteps, dtype=np.float64).astype(f32)
    else:
        raise ValueError(schedule)
    abar = np.cumprod((f32(1.0) - betas).astype(f32), dtype=f32)
    return np.sqrt((f32(1.0) - abar) / abar).astype(f32), abar


def inference_grid(num_train_timesteps: int, n: int, sig_train: np.ndarray):
    """-> (timesteps [n] int64 descending, sigmas [n+1] fp32 with a trailing 0)."""
    ts = np.linspace(0, num_train_timesteps - 1, n + 1).round()[::-1][:-1].astype(np.int64)
    sig = np.interp(ts, np.arange(len(sig_train)), sig_train.astype(np.float64))
    return ts, np.concatenate([sig, [0.0]]).astype(f32)


def _als(s):
    a = f32(1.0) / np.sqrt(f32(s) * f32(s) + f32(1.0), dtype=f32)
    return f32(a), f32(f32(s) * a)


def _lam(s):
    a, sg = _als(s)
    with np.errstate(divide="ignore"):
        return f32(np.log(a, dtype=f32) - np.log(sg, dtype=f32))


def sample(v_fn, x: np.ndarray, timesteps: np.ndarray, sigmas: np.ndarray, order: int = 2) -> np.ndarray:
    """``v_fn(x, t) -> v`` (already CFG-combined).  Walks all steps; returns the final sample (fp32)."""
    x = x.astype(f32)
    hist = []
    n = len(timesteps)
    for i, t in enumerate(timesteps):
        v = v_fn(x, int(t)).astype(f32)
        a_i, s_i = _als(sigmas[i])
        x0 = a_i * x - s_i * v
        hist.append(x0)
        a_n, s_n = _als(sigmas[i + 1])
        h = f32(_lam(sigmas[i + 1]) - _lam(sigmas[i]))
        c = f32(a_n * f32(np.exp(-h, dtype=f32) - f32(1.0)))
        nxt = f32(s_n / s_i) * x - c * x0
        if order == 2 and i >= 1 and i != n - 1:
            r0 = f32(f32(_lam(sigmas[i]) - _lam(sigmas[i - 1])) / h)
            nxt = nxt - f32(0.5) * c * ((hist[-1] - hist[-2]) * f32(f32(1.0) / r0))
        x = nxt.astype(f32)
    return x
