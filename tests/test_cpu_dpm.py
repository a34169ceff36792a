"""DPM-Solver++(2M) host logic against the numpy oracle -- CPU only.

PARITY UNPINNED: the reference drives diffusers' DPMSolverMultistepScheduler (absent here, version un-pinned) and holds
no fixture for this path; the checks below compare two independent restatements of the published algorithm."""
import numpy as np
import pytest
import torch

from oracle import dpm_solver_ref as R


def _sched(T=500, order=2):
    from diffusion_model_amd.dpm_solver import DPMSolverMultistepScheduler
    return DPMSolverMultistepScheduler(num_train_timesteps=T, beta_start=0.00085, beta_end=0.012,
                                       beta_schedule="scaled_linear", prediction_type="v_prediction",
                                       algorithm_type="dpmsolver++", solver_order=order)


def test_timestep_grid_and_sigmas():
    s = _sched()
    s.set_timesteps(5)
    assert s.timesteps.tolist() == [499, 399, 299, 200, 100]          # linspace(0, 499, 6).round(), reversed, last dropped
    sig_tr, abar = R.training_sigmas(500, 0.00085, 0.012)
    ts, sig = R.inference_grid(500, 5, sig_tr)
    assert ts.tolist() == s.timesteps.tolist()
    assert float(s.sigmas[-1]) == 0.0 and len(s.sigmas) == 6
    np.testing.assert_allclose(s.sigmas.numpy()[:-1], sig[:-1], rtol=1e-5)
    # sigma is increasing in t, abar decreasing
    assert np.all(np.diff(sig_tr) > 0) and np.all(np.diff(abar) < 0)


@pytest.mark.parametrize("n,order", [(1, 2), (2, 2), (5, 2), (20, 2), (50, 2), (10, 1)])
def test_step_matches_oracle_walk(n, order):
    rng = np.random.default_rng(n)
    W = (rng.standard_normal((16, 16)) * 0.2).astype(np.float32)

    def v_fn(x, t):
        return np.tanh(x @ W + 0.001 * t).astype(np.float32)

    s = _sched(order=order)
    s.set_timesteps(n)
    sig_tr, _ = R.training_sigmas(500, 0.00085, 0.012)
    ts, sig = R.inference_grid(500, n, sig_tr)
    x0 = rng.standard_normal((4, 16)).astype(np.float32)
    ref = R.sample(v_fn, x0, ts, sig, order=order)
    x = torch.from_numpy(x0)
    last_x0 = None
    for t in s.timesteps:
        v = torch.from_numpy(v_fn(x.numpy(), int(t)))
        x = s.step(v, t, x).prev_sample
        last_x0 = s.model_outputs[-1]
    np.testing.assert_allclose(x.numpy(), ref, rtol=0, atol=2e-5 * np.abs(ref).max())
    # the last step lands on sigma = 0: the sample IS the last data prediction
    assert torch.equal(x, last_x0)
    with pytest.raises(IndexError):
        s.step(v, 0, x)


def test_unsupported_configurations_fail_loudly():
    from diffusion_model_amd.dpm_solver import DPMSolverMultistepScheduler as S
    with pytest.raises(ValueError):
        S(prediction_type="epsilon")
    with pytest.raises(ValueError):
        S(prediction_type="v_prediction", algorithm_type="dpmsolver")
    with pytest.raises(ValueError):
        S(prediction_type="v_prediction", solver_order=3)
    with pytest.raises(ValueError):
        S(prediction_type="v_prediction", use_karras_sigmas=True)
    S(prediction_type="v_prediction", solver_type="midpoint")   # a default spelled out is fine


def test_per_sample_eta_and_guidance_rows():
    """inference_sweeps batches grid cells with different (cfg, eta): rows must equal the scalar-argument rows."""
    from diffusion_model_amd.sampler import step_coefficients
    from diffusion_model_amd.utils import DDPM_beta_t_cosine_scheduler, DDPM_beta_t_linear_scheduler
    for sch in (DDPM_beta_t_cosine_scheduler(500), DDPM_beta_t_linear_scheduler(500)):
        beta, abar = sch.beta_cpu, sch.alpha_bar_cpu
        for t in (499, 2, 1, 0):
            etas, gs = [0.0, 1.0, 0.5], [6.0, 0.0, 1.5]
            rows = step_coefficients(beta, abar, t, torch.tensor(gs), torch.tensor(etas), 3, final=(t == 0))
            for i in range(3):
                one = step_coefficients(beta, abar, t, gs[i], etas[i], 1, final=(t == 0))
                assert torch.equal(rows[i], one[0]), (t, i)
    lin = DDPM_beta_t_linear_scheduler(500)
    assert torch.equal(lin.beta_cpu, torch.linspace(1e-4, 0.02, 500)) and lin.T == 500   # inference_sweeps.py:22
