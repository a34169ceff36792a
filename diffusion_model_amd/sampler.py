"""CFG + DDPM/eta sampler on the fused HIP step kernel.

Mirrors visualization.py:92-152 (`_ddpm_step`, `run_reverse_process`) and
main_stl10.py:117-137 (`_cfg_predict_eps_from_v`) of the reference; the per-step scalar
coefficients are computed on the host in fp32 with the same torch ops as the reference
(so they are bit-identical), the per-element work is ONE kernel (dm_cfg_ddpm_step)."""
from __future__ import annotations

from typing import Optional

import torch


def step_coefficients(beta: torch.Tensor, abar: torch.Tensor, t: int, guidance, eta: float, batch: int,
                      final: bool = False) -> torch.Tensor:
    """[batch, 8] fp32 rows {sqrt(1-abar_t), sqrt(abar_t), 1/sqrt(alpha_t), beta_t/sqrt(1-abar_t),
    eta*sqrt(posterior_var) or 0, guidance, clamp flag, 0}.  `beta`/`abar` are the fp32 CPU tables;
    `guidance` and `eta` are floats or per-sample [batch] tensors (guidance 0.0 for the
    unconditional label, visualization.py:424)."""
    beta, abar = beta.detach().cpu(), abar.detach().cpu()
    abar_t, beta_t = abar[t], beta[t]
    alpha_t = 1.0 - beta_t
    sig = torch.sqrt(1.0 - abar_t)
    sa = torch.sqrt(abar_t)
    ria = 1.0 / torch.sqrt(alpha_t)
    bs = beta_t / torch.sqrt(1.0 - abar_t)
    eta = torch.as_tensor(eta, dtype=torch.float32)   # a float, or one eta per sample (inference_sweeps.py:127-133)
    ns = torch.zeros(())
    if (not final) and t > 1 and bool((eta > 0.0).any()):
        abar_prev = abar[t - 1]
        ns = eta * torch.sqrt(beta_t * (1.0 - abar_prev) / (1.0 - abar_t))
    row = torch.stack([sig, sa, ria, bs, torch.tensor(0.0), torch.tensor(0.0), torch.tensor(1.0 if final else 0.0),
                       torch.tensor(0.0)]).float()
    coef = row[None, :].repeat(batch, 1)
    coef[:, 4] = ns.float()
    coef[:, 5] = torch.as_tensor(guidance, dtype=torch.float32)
    return coef.contiguous()


def _eps_step(z: torch.Tensor, t: int, predicted_noise: torch.Tensor, scheduler, eta: float, final: bool) -> torch.Tensor:
    """One DDPM / eta step from a GIVEN epsilon on the fused kernel: with the coefficient row
    {sqrt(1-abar) := 0, sqrt(abar) := 1, guidance := 0} and v_uncond = v_cond = eps, dm_cfg_ddpm_step's
    ``eps = sig z + sa (v_u + g (v_c - v_u))`` returns the caller's epsilon unchanged, and the rest of the kernel is
    the reference's mean step, eta noise and (final) clamp -- no second statement of that arithmetic in torch."""
    from . import kernels as K
    if not z.is_cuda:
        raise RuntimeError("diffusion_model_amd samples only on an AMD GPU (HIP kernels); there is no CPU fallback")
    B = z.shape[0]
    beta = getattr(scheduler, "beta_cpu", None)
    abar = getattr(scheduler, "alpha_bar_cpu", None)
    if beta is None or abar is None:   # a reference-style scheduler object: tables on some device
        beta, abar = scheduler.beta_t.detach().float().cpu(), scheduler.alpha_t.detach().float().cpu()
    coef = step_coefficients(beta, abar, t, 0.0, eta, B, final=final)
    coef[:, 0], coef[:, 1] = 0.0, 1.0
    zc = z.contiguous().float()
    eps = predicted_noise.contiguous().float()
    noise = torch.randn_like(zc) if float(coef[0, 4]) != 0.0 else zc
    return K.cfg_ddpm_step(zc, eps, eps, noise, coef.to(z.device))


def _ddpm_step(z: torch.Tensor, t: int, predicted_noise: torch.Tensor, scheduler, eta: float) -> torch.Tensor:
    """Reference visualization.py:92-109, for callers that use the reference's ``predict_eps_fn`` callback contract."""
    return _eps_step(z, t, predicted_noise, scheduler, eta, final=False)


def run_reverse_process(model, scheduler, z_init, labels, guidance_scale, eta, num_time_steps, predict_eps_fn,
                        capture_timesteps=None):
    """Reference visualization.py:112-152, same signature and return value.  The model evaluations are the
    caller's ``predict_eps_fn`` (callback contract); every step's arithmetic runs on ``dm_cfg_ddpm_step``."""
    with torch.no_grad():
        model.eval()
        capture_timesteps = capture_timesteps or []
        snapshots = {}
        z = z_init.clone()
        for t in reversed(range(1, num_time_steps)):
            t_tensor = torch.tensor([t], device=z.device)
            eps = predict_eps_fn(model, scheduler, z, t_tensor, labels, guidance_scale)
            z = _ddpm_step(z, t, eps, scheduler, eta)
            if t in capture_timesteps:
                snapshots[t] = torch.clamp(z, -1, 1)
        t_tensor = torch.tensor([0], device=z.device)
        eps = predict_eps_fn(model, scheduler, z, t_tensor, labels, guidance_scale)
        x0 = _eps_step(z, 0, eps, scheduler, eta, final=True)   # t = 0: mean step, no noise, clamp (:144-152)
        snapshots[0] = x0
        return x0, snapshots


class CFGSampler:
    """Batched 500-step CFG sampler: per step ONE UNet forward over [uncond; cond] (batch 2S) and ONE
    fused elementwise kernel (CFG combine, v->eps, DDPM mean, eta noise, final clamp).  No collective
    in the loop: under data parallelism every rank samples its own shard of the labels."""

    def __init__(self, model, scheduler, context_table: torch.Tensor, uncond_id: int):
        self.model, self.sched = model, scheduler
        self.table = context_table
        self.uncond_id = uncond_id

    @torch.no_grad()
    def sample(self, z: torch.Tensor, labels: torch.Tensor, guidance_scale=6.0, eta: float = 1.0,
               num_time_steps: Optional[int] = None, generator: Optional[torch.Generator] = None,
               noises=None, t_stop: int = 0, snapshots: Optional[dict] = None) -> torch.Tensor:
        """z: [S,C,H,W] fp32 start noise; labels: [S] long.  ``guidance_scale`` / ``eta``: a float or one value per
        sample.  Guidance is 0.0 for samples whose label is the unconditional id (reference visualization.py:424).  ``noises[t]`` overrides the RNG (tests);
        ``t_stop>0`` stops after step t_stop (bounded timing runs) and returns the un-clamped latent.
        ``snapshots``: a dict whose keys are the timesteps to capture (visualization.py:144-146); it is filled
        with the clamped latent after those steps (key 0 = the final image)."""
        from . import kernels as K
        model = self.model
        model.eval()
        S = z.shape[0]
        T = num_time_steps or self.sched.T
        dev = z.device
        beta, abar = self.sched.beta_cpu, self.sched.alpha_bar_cpu
        labels = labels.to(dev).long()
        g = torch.as_tensor(guidance_scale, dtype=torch.float32).expand(S).clone()
        g[labels.cpu() == self.uncond_id] = 0.0
        table = self.table.to(dev)
        ctx = torch.cat([table[torch.full_like(labels, self.uncond_id)], table[labels]]).contiguous()
        z = z.contiguous().clone()
        zz = torch.empty((2 * S, *z.shape[1:]), dtype=torch.float32, device=dev)
        tt = torch.empty((2 * S,), dtype=torch.long, device=dev)
        steps = [t for t in range(T - 1, -1, -1) if t >= t_stop]
        any_eta = bool((torch.as_tensor(eta) > 0.0).any())
        # every step's coefficient rows in one upload (host fp32 math identical to the reference's)
        coefs = torch.stack([step_coefficients(beta, abar, t, g, eta, S, final=(t == 0)) for t in steps]).to(dev)
        # (Replaying the forward as a captured HIP graph was measured: 7.56 vs 7.28 ms/step eager at B=176 -- the
        # loop is GPU-bound, so the launches stay eager.)
        for i, t in enumerate(steps):
            final = (t == 0)
            zz[:S].copy_(z)
            zz[S:].copy_(z)
            tt.fill_(t)
            v, _ = model._fwd(zz, tt, ctx)
            noise = None
            if (not final) and t > 1 and any_eta:
                if noises is not None and t in noises:
                    noise = noises[t].to(dev)
                else:
                    noise = torch.randn(z.shape, dtype=torch.float32, device=dev, generator=generator)
            else:
                noise = z  # unused (coef[4] == 0)
            z = K.cfg_ddpm_step(z, v[:S], v[S:], noise, coefs[i])
            if snapshots is not None and t in snapshots:
                snapshots[t] = torch.clamp(z, -1, 1)
        return z

    @torch.no_grad()
    def sample_from(self, x_t: torch.Tensor, labels: torch.Tensor, t_start, guidance_scale=0.0, eta: float = 1.0,
                    generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """Partial reverse process (reference visualization.py:264-330, 333-401: the reconstruction panels):
        sample i starts at timestep ``t_start[i]`` from the already-noised ``x_t[i]`` and walks t_start[i]..0.
        All samples run as ONE batch: they are ordered by start time and join the UNet batch when the global
        timestep reaches their own, so a sweep over several start times costs max(t_start)+1 forwards, not the
        sum.  Returns the clamped x0 estimates in the caller's order."""
        from . import kernels as K
        model = self.model
        model.eval()
        S, dev = x_t.shape[0], x_t.device
        starts = [int(t) for t in (t_start.tolist() if torch.is_tensor(t_start) else t_start)]
        assert len(starts) == S and min(starts) >= 0 and max(starts) < self.sched.T
        order = sorted(range(S), key=lambda i: -starts[i])
        inv = torch.empty(S, dtype=torch.long)
        inv[torch.tensor(order)] = torch.arange(S)
        oi = torch.tensor(order, device=dev)
        z = x_t.contiguous().index_select(0, oi).clone()
        labels = labels.to(dev).long().index_select(0, oi)
        starts = [starts[i] for i in order]
        g = torch.full((S,), float(guidance_scale), dtype=torch.float32)
        g[labels.cpu() == self.uncond_id] = 0.0
        beta, abar = self.sched.beta_cpu, self.sched.alpha_bar_cpu
        table = self.table.to(dev)
        ctx_u, ctx_c = table[torch.full_like(labels, self.uncond_id)], table[labels]
        for t in range(starts[0], -1, -1):
            n = sum(1 for s0 in starts if s0 >= t)          # the active samples are a prefix
            zz = torch.cat([z[:n], z[:n]])
            tt = torch.full((2 * n,), t, dtype=torch.long, device=dev)
            v, _ = model._fwd(zz, tt, torch.cat([ctx_u[:n], ctx_c[:n]]).contiguous())
            coef = step_coefficients(beta, abar, t, g[:n], eta, n, final=(t == 0)).to(dev)
            if t > 1 and eta > 0.0:
                noise = torch.randn(z[:n].shape, dtype=torch.float32, device=dev, generator=generator)
            else:
                noise = zz[:n]  # unused (coef[4] == 0)
            z[:n] = K.cfg_ddpm_step(z[:n].contiguous(), v[:n], v[n:], noise, coef)
        return z.index_select(0, inv.to(dev))

    @torch.no_grad()
    def sample_dpm(self, z: torch.Tensor, labels: torch.Tensor, dpm, cfg_scale=6.0,
                   num_inference_steps: int = 50) -> torch.Tensor:
        """DPM-Solver++(2M) sampling (reference visualization.py:480-534) for a whole batch: per step ONE UNet forward
        over [uncond; cond] and ONE fused kernel (``dm_cfg_dpmpp2m_step``).  ``dpm`` is a
        ``dpm_solver.DPMSolverMultistepScheduler``; ``cfg_scale`` a float or one value per sample.  As in the reference
        (:513-526) guidance is applied only where cfg_scale > 1.0: the other samples take the conditional branch alone
        (callers pass 0.0 for the unconditional label, whose context row is the null embedding).  When no sample is
        guided the uncond half of the batch is not run at all.  Returns the final latents, un-clamped, like the reference."""
        from . import kernels as K
        model = self.model
        model.eval()
        S, dev = z.shape[0], z.device
        labels = labels.to(dev).long()
        g = torch.as_tensor(cfg_scale, dtype=torch.float32).expand(S).clone()
        g[g <= 1.0] = 1.0                                   # 1.0 = "conditional branch only" for the kernel
        guided = bool((g != 1.0).any())
        table = self.table.to(dev)
        ctx_c = table[labels]
        ctx = torch.cat([table[torch.full_like(labels, self.uncond_id)], ctx_c]).contiguous() if guided else ctx_c.contiguous()
        dpm.set_timesteps(num_inference_steps, device=dev)
        steps = [int(t) for t in dpm.timesteps.tolist()]
        rows = []
        for i in range(len(steps)):
            a_s, s_s, ratio, cc, inv_r0, second = dpm.step_coefficients(i, have_history=(i >= 1))
            row = torch.stack([torch.tensor(0.0), a_s, s_s, ratio, cc, inv_r0, torch.tensor(1.0 if second else 0.0),
                               torch.tensor(0.0)]).float()
            rows.append(row[None, :].repeat(S, 1))
        coefs = torch.stack(rows)
        coefs[:, :, 0] = g
        coefs = coefs.to(dev)
        z = z.contiguous().float().clone()
        m_prev = z                                           # not read on the first (first-order) step
        n_in = 2 * S if guided else S
        zz = torch.empty((n_in, *z.shape[1:]), dtype=torch.float32, device=dev)
        tt = torch.empty((n_in,), dtype=torch.long, device=dev)
        for i, t in enumerate(steps):
            zz[:S].copy_(z)
            if guided:
                zz[S:].copy_(z)
            tt.fill_(t)
            v, _ = model._fwd(zz, tt, ctx)
            v_u, v_c = (v[:S], v[S:]) if guided else (v, v)
            z, m_prev = K.cfg_dpmpp2m_step(z, v_u, v_c, m_prev, coefs[i])
        return z


def shard_bounds(n: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of n samples for ``rank`` of ``world`` (sizes differ by at most one)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def sample_sharded(sample_fn, z: torch.Tensor, labels: torch.Tensor, *, group=None, **kw) -> torch.Tensor:
    """Data-parallel sampling (config C4 = 704 images over the ranks): every rank runs
    ``sample_fn(z[lo:hi], labels[lo:hi], **kw)`` on its own contiguous shard -- NO collective inside the step loop --
    and ONE all-gather at the end returns the full batch, in the caller's order, on every rank.  ``sample_fn`` is e.g.
    ``CFGSampler(...).sample`` or ``.sample_dpm`` bound with its arguments; ``z`` / ``labels`` are the same on all ranks
    (seeded start noise), the per-step ancestral noise is rank-local."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return sample_fn(z, labels, **kw)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = z.shape[0]
    lo, hi = shard_bounds(n, rank, world)
    out = sample_fn(z[lo:hi], labels[lo:hi], **kw) if hi > lo else z[:0].clone()
    width = (n + world - 1) // world                       # all_gather wants equal shapes: pad the short shards
    pad = torch.zeros((width, *z.shape[1:]), dtype=out.dtype if hi > lo else z.dtype, device=z.device)
    pad[: hi - lo] = out
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    sizes = [shard_bounds(n, r, world) for r in range(world)]
    return torch.cat([p[: b - a] for p, (a, b) in zip(parts, sizes)])
