#!/usr/bin/env python3
"""Headline benchmark: STL10 96x96 v-prediction UNet training (B=64 per GPU, accumulation 2, AdamW +
clip + EMA, dropout 0.1, low-t weighting -- the reference's shipped train() configuration,
main_stl10.py:352-516) on synthetic data, plus the 500-step CFG sampling rate.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE OPTIMIZER STEP = 2 micro-batches of 64 images per GPU (forward + backward each), the
gradient exchange (N>1), clip + AdamW + EMA and the repack of the bf16 weight shadows.  Rank 0
prints one JSON line (contract in the task statement)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16 = 2.5e15       # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md, chip-level parameters)
PEAK_F32 = 157.3e12      # fp32-input MFMA peak = the fp32 vector rate (MI355X_MICROARCH.md, matrix cores table)
TRAIN_FLOP_PER_IMG = 55.7e9   # algorithmic fwd+dgrad+wgrad FLOPs / image (reference op count)
FWD_FLOP_PER_IMG = 18.08e9


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--sample-images", type=int, default=-1,
                    help="CFG samples per GPU; default 704 / N = configs[3] (11 labels x 64 samples, batch-sharded); 0 = skip")
    ap.add_argument("--sample-chunk", type=int, default=88, help="images per sampler launch (UNet batch = 2 x this)")
    ap.add_argument("--sample-steps", type=int, default=500)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-latent", action="store_true", help="skip the secondary latent-diffusion UNet leg (1 GPU only)")
    ap.add_argument("--no-elided", action="store_true", help="skip the secondary legs (dropout 0, dead-attention elided)")
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32-compute-mode leg (1 GPU only)")
    ap.add_argument("--no-mnist", action="store_true", help="skip the MNIST small-shape leg (configs[4]; 1 GPU only)")
    ap.add_argument("--graph", action="store_true", help="run the headline steps as HIP-graph replays (single GPU)")
    ap.add_argument("--rehearse-dp", action="store_true",
                    help="1 GPU: run the data-parallel code path (RCCL process group of ONE rank, bucket all-reduces "
                         "issued from the backward hooks) -- a rehearsal of the N>1 path, not a scaling number")
    return ap.parse_args()


def synth_batch(B, rank, dev, table, scheduler):
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    x = torch.clamp(torch.randn(B, 3, 96, 96, generator=g) * 0.5, -1, 1).to(dev)
    labels = torch.randint(0, 10, (B,), generator=g)
    labels[torch.rand(B, generator=g) < 0.2] = 10
    t = torch.randint(0, 500, (B,), generator=g).to(dev)
    e = torch.randn(B, 3, 96, 96, generator=g).to(dev)
    ctx = table[labels].to(dev).contiguous()
    return x, ctx, t, e


class StepData:
    """The per-micro-step input statements of the reference loop (main_stl10.py:453-469), run INSIDE the timed steps:
    label drop w.p. 0.2 -> UNCOND_ID, t ~ randint(0, 500) drawn on the host as in the reference (:458), e ~ randn_like(x)
    on the device, context = CONTEXT_TABLE[labels_dropped].  The images and labels of the rank's two batches are resident
    in HBM (the DataLoader's job, outside the hot path)."""

    def __init__(self, B, rank, dev, table):
        self.dev, self.table = dev, table.to(dev)
        self.batches = []
        for i in range(2):
            g = torch.Generator(device="cpu").manual_seed(1234 + 2 * rank + i)
            x = torch.clamp(torch.randn(B, 3, 96, 96, generator=g) * 0.5, -1, 1).to(dev)
            self.batches.append((x, torch.randint(0, 10, (B,), generator=g).to(dev)))
        self.g_cpu = torch.Generator().manual_seed(777 + rank)
        self.g_dev = torch.Generator(device=dev).manual_seed(778 + rank)

    def draw(self, i):
        x, labels = self.batches[i]
        drop = torch.rand(labels.shape[0], device=self.dev, generator=self.g_dev) < 0.2
        labels_dropped = torch.where(drop, torch.full_like(labels, 10), labels)
        t = torch.randint(0, 500, (x.shape[0],), generator=self.g_cpu).to(self.dev, non_blocking=True)
        e = torch.randn(x.shape, device=self.dev, dtype=x.dtype, generator=self.g_dev)
        return x, self.table[labels_dropped], t, e


def build_model(dropout, dev):
    from diffusion_model_amd import utils
    from diffusion_model_amd.unet import UNET
    table = utils._load_clip_context_table()
    torch.manual_seed(0)
    model = UNET(time_steps=1000, input_channels=3, output_channels=3, dropout_prob=dropout, context_dim=512)
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():  # zero-init output_conv would make every other gradient exactly 0
        model.output_conv.weight.copy_(torch.randn(model.output_conv.weight.shape, generator=gen) * 0.05)
    return model.to(dev), table


def latent_leg(a, dev, table, sched):
    """Secondary leg: the reference's latent-diffusion UNet configuration (utils.py:145-156: Channels
    [256,256,512,1024,512], 4 -> 4 channels, 55.9 M parameters) on synthetic 4x32x32 latents -- the same optimizer step
    (B per micro-batch, accumulation 2, dropout, AdamW+EMA+clip) and a 50-step CFG DPM-Solver++ sampling run.  Latents are
    N(0,1) (scaled VAE latents have about unit variance); the VAE itself is not part of the path."""
    from diffusion_model_amd.dpm_solver import DPMSolverMultistepScheduler
    from diffusion_model_amd.ema import ModelEmaV2
    from diffusion_model_amd.sampler import CFGSampler
    from diffusion_model_amd.trainer import Trainer
    from diffusion_model_amd.unet import UNET
    torch.manual_seed(0)
    model = UNET(time_steps=1000, Channels=[256, 256, 512, 1024, 512], Attentions=[False, True, True, False],
                 input_channels=4, output_channels=4, dropout_prob=a.dropout, context_dim=512)
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        model.output_conv.weight.copy_(torch.randn(model.output_conv.weight.shape, generator=gen) * 0.05)
    n_params = sum(p.numel() for p in model.parameters())
    model = model.to(dev).train()
    ema = ModelEmaV2(model, decay=0.995)
    tr = Trainer(model, sched, lr=5e-5, weight_decay=1e-4, betas=(0.9, 0.99), clip=1.0, ema_decay=0.995, accum=2,
                 total_steps=100 * 79, start_step=20 * 40, low_t_weight=0.01, ema=ema, distributed=False)
    batches = []
    for i in range(2):
        g = torch.Generator(device="cpu").manual_seed(4321 + i)
        labels = torch.randint(0, 10, (a.batch,), generator=g)
        labels[torch.rand(a.batch, generator=g) < 0.2] = 10
        batches.append((torch.randn(a.batch, 4, 32, 32, generator=g).to(dev), table[labels].to(dev).contiguous(),
                        torch.randint(0, 500, (a.batch,), generator=g).to(dev),
                        torch.randn(a.batch, 4, 32, 32, generator=g).to(dev)))

    def opt_step():
        for b in batches:
            loss = tr.micro_step(*b)
        return loss

    for _ in range(2):
        opt_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = opt_step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"what": "reference LDM UNet configuration (utils.py:145-156) on synthetic 4x32x32 latents, same optimizer step",
           "params": n_params, "value": a.steps * 2 * a.batch / dt, "unit": "latents/s", "ms_per_step": dt / a.steps * 1e3,
           "final_loss": float(loss.item())}
    S = 88
    smp = CFGSampler(ema.module, sched, table, uncond_id=10)
    dpm = DPMSolverMultistepScheduler(num_train_timesteps=500, beta_start=0.00085, beta_end=0.012,
                                      beta_schedule="scaled_linear", prediction_type="v_prediction",
                                      algorithm_type="dpmsolver++", solver_order=2)
    z = torch.randn(S, 4, 32, 32, generator=torch.Generator().manual_seed(42)).to(dev)
    labels = torch.arange(S) % 11
    gd = torch.where(labels == 10, 0.0, 6.0)
    smp.sample_dpm(z, labels, dpm, cfg_scale=gd, num_inference_steps=3)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    xd = smp.sample_dpm(z, labels, dpm, cfg_scale=gd, num_inference_steps=50)
    torch.cuda.synchronize()
    dd = time.perf_counter() - t1
    out["dpm_solver_pp_2m_50_steps"] = {"value": S / dd, "unit": "latents/s", "finite": bool(torch.isfinite(xd).all().item())}
    return out


def mnist_leg(a, dev):
    """configs[4]: the reference's MNIST baseline (main.py:142-197: attention-free UNet, 32x32x1 = 28x28 padded by 2,
    B=64, epsilon target, Adam 5e-5, clip 0.5, EMA 0.999, dropout 0.1, one optimizer step per batch) on synthetic
    digits.  Every kernel of this configuration is tiny, so the step is bound by launch latency: timed once as eager
    launches and once with the micro-step replayed as ONE HIP graph (Trainer.enable_graph)."""
    from diffusion_model_amd import main as M
    from diffusion_model_amd.ema import ModelEmaV2
    from diffusion_model_amd.trainer import Trainer
    from diffusion_model_amd.utils import DDPM_beta_t_cosine_scheduler
    torch.manual_seed(0)
    model = M.build_model()
    with torch.no_grad():
        model.output_conv.weight.normal_(0, 0.05, generator=torch.Generator().manual_seed(1))
    model = model.to(dev).train()
    sched = DDPM_beta_t_cosine_scheduler(M.NUM_TIME_STEPS, beta_min=0.0)
    tr = Trainer(model, sched, lr=M.LR, weight_decay=0.0, betas=(0.9, 0.999), clip=M.GRAD_CLIP, ema_decay=M.EMA_DECAY,
                 accum=1, total_steps=0, low_t_weight=None, target="e", adamw=False, ema=ModelEmaV2(model, M.EMA_DECAY))
    g = torch.Generator().manual_seed(99)
    B = M.BATCH_SIZE
    batches = [((torch.nn.functional.pad(torch.rand(B, 1, 28, 28, generator=g) * 2 - 1, (2, 2, 2, 2))).to(dev), None,
                torch.randint(0, M.NUM_TIME_STEPS, (B,), generator=g).to(dev),
                torch.randn(B, 1, 32, 32, generator=g).to(dev)) for _ in range(4)]
    n = max(20, 5 * a.steps)

    def run(k):
        loss = None
        for i in range(k):
            loss = tr.micro_step(*batches[i % 4])
        return loss

    def timed():
        run(5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = run(n)
        t_host = time.perf_counter() - t0     # host time to ISSUE n steps (no sync inside)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n, t_host / n, float(loss.item())

    dt_e, host_e, loss_e = timed()
    tr.enable_graph(*batches[0])
    dt_g, host_g, loss_g = timed()
    tr.disable_graph()
    # the WHOLE optimizer step as one graph: micro-step + clip + Adam + EMA + gradient clear + repack, inputs from a
    # device ring of the four batches, step count / rate / bias corrections advanced on the device
    tr.enable_step_graph(batches)

    def run(k):   # noqa: F811  (same timing harness, one replay per step)
        loss = None
        for _ in range(k):
            loss = tr.graph_step()
        return loss

    dt_w, host_w, loss_w = timed()
    tr.disable_graph()
    return {"what": "configs[4]: reference main.py MNIST baseline UNet (3.05 M parameters, no attention), 32x32x1, "
                    "B=64, dropout 0.1, Adam + clip 0.5 + EMA, synthetic digits; step = one optimizer step",
            "steps": n, "value": B / dt_w, "unit": "img/s",
            "eager": {"ms_per_step": dt_e * 1e3, "host_issue_ms_per_step": host_e * 1e3, "img_per_s": B / dt_e,
                      "final_loss": loss_e},
            "hip_graph": {"ms_per_step": dt_g * 1e3, "host_issue_ms_per_step": host_g * 1e3, "img_per_s": B / dt_g,
                          "final_loss": loss_g, "what": "micro-step replayed, optimizer eager"},
            "hip_graph_whole_step": {"ms_per_step": dt_w * 1e3, "host_issue_ms_per_step": host_w * 1e3,
                                     "img_per_s": B / dt_w, "final_loss": loss_w,
                                     "what": "one replay per optimizer step (Trainer.enable_step_graph): inputs from a "
                                             "device ring, schedule on the device"}}


def time_conv_kernel(B, dev, H=48, cin=128, cout=128, mode=2, reps=20):
    """Live HIP-event timing of one conv launch shape, back to back on the stream it is launched on.  Default: the
    dominant MFMA kernel, nearest-x2 + conv3x3 at the L4.upsample.conv shape (2.72 of the 18.08 forward GFLOP per
    image).  Returns (avg_seconds, flops_per_launch, kernel variant name)."""
    from diffusion_model_amd import kernels as K
    x = torch.randn(B, H, H, cin, device=dev).to(torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.03
    wp = K.pack_weight(w)
    bias = torch.zeros(cout, device=dev)
    Ho = H // 2 if mode == 1 else (2 * H if mode in (2, 3) else H)
    y = torch.empty(B, Ho, Ho, cout, dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        K.conv3x3(x, wp, cout, bias=bias, mode=mode, out=y)
    variant = K.last_variant()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        K.conv3x3(x, wp, cout, bias=bias, mode=mode, out=y)
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e-3 / reps, 2.0 * B * Ho * Ho * cout * cin * 9, variant


PEAK_HBM = 8.0e12        # MI355X HBM3E peak (MI355X_MICROARCH.md; ~6.3 TB/s is what a streaming kernel reaches)
# VALU issue peak: 256 CUs x 4 SIMDs x 32 lanes per cycle at 2.4 GHz (MI355X_MICROARCH.md, cycle constants: a wave64 VALU
# instruction issues over 2 cycles when >= 2 waves per SIMD supply instructions; a lone wave's stream issues at 4 cycles)
PEAK_VALU_LANE_OPS = 256 * 4 * 32 * 2.4e9


def _timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e-3 / reps


def hbm_rooflines(dev, B):
    """Live HIP-event timing (back to back, current stream) of the HBM-bound kernels of the step at their B-per-GPU shapes,
    against ALGORITHMIC bytes (GroupNorm+SiLU 4 B/elem forward, 6 B/elem backward at bf16; fused optimizer 32
    B/param moved by adamw_ema_step itself (+4 for the norm pass = the survey's 36); sampler step 20 B/elem) and the
    8 TB/s HBM peak."""
    from diffusion_model_amd import kernels as K
    out = []

    def row(name, sec, nbytes):
        out.append({"kernel": name, "avg_launch_us": sec * 1e6, "algorithmic_bytes": nbytes,
                    "achieved": nbytes / sec / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": nbytes / sec / PEAK_HBM})

    for H, C in ((96, 64), (48, 128), (24, 256), (48, 256)):   # the ResBlock norm shapes of the STL10 UNet
        x = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
        dy = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
        gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        y, dx = torch.empty_like(x), torch.empty_like(x)
        _, stats = K.gn_silu_fwd(x, gam, bet, 8, out=y)
        n = x.numel()
        row(f"gn_silu_fwd {C}ch@{H}x{H}", _timeit(lambda: K.gn_silu_fwd(x, gam, bet, 8, out=y)), 4 * n)
        row(f"gn_silu_bwd {C}ch@{H}x{H}", _timeit(lambda: K.gn_silu_bwd(x, dy, gam, bet, stats, 8, dg, db, out=dx)), 6 * n)
        del x, dy, y, dx
    N = 4177539 // 4 * 4 + 4
    p, g, m, v, ema = (torch.randn(N, device=dev) * 0.01 for _ in range(5))
    v = v.abs()
    norm = torch.zeros(2, device=dev)
    row("grad_norm_clipcoef 4.18 M params", _timeit(lambda: K.grad_norm_clipcoef(g, 1.0, out=norm)), 4 * N)
    row("adamw_ema_step 4.18 M params",
        _timeit(lambda: K.adamw_ema_step(p, g, m, v, ema, norm, lr=5e-5, beta1=0.9, beta2=0.99, eps=1e-8, wd=1e-4, step=3,
                                         ema_decay=0.995)), 32 * N)
    S = 88
    z, vu, vc, nz = (torch.randn(S, 3, 96, 96, device=dev) for _ in range(4))
    coef = torch.rand(S, 8, device=dev)
    zo = torch.empty_like(z)
    row(f"cfg_ddpm_step S={S}", _timeit(lambda: K.cfg_ddpm_step(z, vu, vc, nz, coef, out=zo)), 20 * z.numel())
    x, e = torch.randn(B, 3, 96, 96, device=dev), torch.randn(B, 3, 96, 96, device=dev)
    ab = torch.rand(B, device=dev)
    row(f"qsample_vtarget B={B}", _timeit(lambda: K.qsample_vtarget(x, e, ab)), 16 * x.numel())
    return out


def attention_roofline(dev, B, p):
    """The L2 self-attention (N = 2304 tokens, 8 heads x 16) forward and backward: with 16-dimensional heads the flash
    loop spends 64 MFMA flop per score element but ~5.8 VALU instructions (one of them v_exp_f32 at quarter rate: 8.8 issue
    slots), so the kernel's roofline is VALU ISSUE, not the MFMA pipe (profiles/r02_pmc_attention.txt: 380 M VALU against
    32 M MFMA instructions).  Both fractions are reported; `bound` names the one that binds."""
    from diffusion_model_amd import kernels as K
    N, C, heads = 2304, 128, 8
    g = torch.Generator().manual_seed(0)
    q, k, v, d_o = ((torch.randn(B, N, C, generator=g) * 0.7).to(dev).to(torch.bfloat16) for _ in range(4))
    o, lse = K.attn_fwd(q, k, v, heads, drop_p=p, drop_seed=5)
    tf = _timeit(lambda: K.attn_fwd(q, k, v, heads, drop_p=p, drop_seed=5), reps=10, warm=2)
    tb = _timeit(lambda: K.attn_bwd(q, k, v, o, lse, d_o, heads, drop_p=p, drop_seed=5), reps=10, warm=2)
    scores = float(B) * heads * N * N
    flop_f, flop_b = 4.0 * scores * (C // heads), 10.0 * scores * (C // heads)
    # issue slots per score element: forward 4.8 plain VALU + one quarter-rate v_exp_f32; backward 28 VALU
    # instructions per 16x16 tile of which 4 are v_exp_f32: (24 + 4 * 4) wave instructions x 64 lanes / 256 scores
    slots_f, slots_b = 8.8, (24 + 4 * 4) * 64 / 256.0
    return [{"kernel": "attn_fwd L2 self-attention N=2304 hd=16, B=%d, dropout %.1f" % (B, p), "bound": "valu",
             "avg_launch_us": tf * 1e6, "valu_issue_slots_per_score": slots_f,
             "achieved": scores * slots_f / tf / 1e12, "peak": PEAK_VALU_LANE_OPS / 1e12, "unit": "T lane-ops/s",
             "frac": scores * slots_f / tf / PEAK_VALU_LANE_OPS,
             "frac_of_single_wave_issue_rate": scores * slots_f / tf / (PEAK_VALU_LANE_OPS / 2),
             "mfma_tflops": flop_f / tf / 1e12, "mfma_frac": flop_f / tf / PEAK_BF16},
            {"kernel": "attn_bwd (single pass) same shape", "bound": "valu", "avg_launch_us": tb * 1e6,
             "valu_issue_slots_per_score": slots_b, "achieved": scores * slots_b / tb / 1e12,
             "peak": PEAK_VALU_LANE_OPS / 1e12, "unit": "T lane-ops/s", "frac": scores * slots_b / tb / PEAK_VALU_LANE_OPS,
             "frac_of_single_wave_issue_rate": scores * slots_b / tb / (PEAK_VALU_LANE_OPS / 2),
             "mfma_tflops": flop_b / tb / 1e12, "mfma_frac": flop_b / tb / PEAK_BF16,
             "note": "28 VALU instructions (4 v_exp_f32) + 5 MFMAs per 16x16 score tile through one issue port"}]


def cpu_baseline():
    """The oracle (CPU fp32 restatement of the reference, oracle/) timed on this box's host cores, SURVEY.md 8(d):
    config C1 = B=8, 96x96, CLIP context, v-loss with low-t weighting, clip 1.0, AdamW, same synthetic recipe --
    (a) dropout 0.1, the configuration the reference ships (main_stl10.py:370) and the headline uses: 1 warm-up + 2
        timed optimizer steps (the attention-probability dropout makes torch materialise the [8,8,2304,2304] matrix:
        seconds per step);
    (b) dropout 0.0 ("parity mode"): 2 warm-up + 5 timed steps;
    (c) a 20-step slice of the B=1 CFG sampler (2 forwards per step, main_stl10.py:125-129), extrapolated to 500.
    About 20-30 s of CPU work in all.  A reported baseline, not a target."""
    from oracle import diffusion_ref as dref
    from oracle import unet_ref as oref
    import numpy as np
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    tab = torch.from_numpy(np.load(os.path.join(ROOT, "diffusion_model_amd", "artifacts", "clip",
                                                "clip_text_emb_stl10_vitb32.npz"),
                                   allow_pickle=False)["embeddings_norm"].astype(np.float32))
    beta, alpha, abar = dref.cosine_schedule(500)
    B = 8

    def train_leg(dropout, warm, n):
        spec = oref.UNetSpec(input_channels=3, output_channels=3, context_dim=512, dropout_prob=dropout)
        g = torch.Generator().manual_seed(0)
        torch.manual_seed(0)   # F.dropout draws from the global generator
        sd = {k: v.requires_grad_() for k, v in oref.seeded_fill(oref.param_shapes(spec), 0, 1.0).items()}
        names = list(sd)
        m = {k: torch.zeros_like(v) for k, v in sd.items()}
        vv = {k: torch.zeros_like(v) for k, v in sd.items()}

        def step(i):
            x = torch.clamp(torch.randn(B, 3, 96, 96, generator=g) * 0.5, -1, 1)
            t = torch.randint(0, 500, (B,), generator=g)
            e = torch.randn(B, 3, 96, 96, generator=g)
            ctx = tab[torch.randint(0, 11, (B,), generator=g)]
            xt, v = dref.q_sample_v_target(x, e, abar[t])
            out = oref.unet_forward(sd, spec, xt, t, ctx, train=dropout > 0)
            loss = dref.v_loss(out, v, t, 0.01)
            grads = list(torch.autograd.grad(loss, [sd[k] for k in names], allow_unused=True))
            grads = [gg if gg is not None else torch.zeros_like(sd[k]) for gg, k in zip(grads, names)]
            dref.clip_grad_norm(grads, 1.0)
            with torch.no_grad():
                for k, gg in zip(names, grads):
                    dref.adamw_step(sd[k], gg, m[k], vv[k], i + 1, 5e-5)

        for i in range(warm):
            step(i)
        t0 = time.perf_counter()
        for i in range(n):
            step(warm + i)
        return B / ((time.perf_counter() - t0) / n), sd

    v01, _ = train_leg(0.1, 1, 2)
    v00, sd = train_leg(0.0, 2, 5)
    # (c) CFG sampling slice: the reference's own shape -- one image, uncond then cond forward per step
    spec = oref.UNetSpec(input_channels=3, output_channels=3, context_dim=512, dropout_prob=0.0)
    sdd = {k: v.detach() for k, v in sd.items()}
    z = torch.randn(1, 3, 96, 96, generator=torch.Generator().manual_seed(42))
    with torch.no_grad():
        v_fn = lambda zz, tt, cc: oref.unet_forward(sdd, spec, zz, tt, cc)  # noqa: E731
        dref.reverse_process(v_fn, z, tab[3:4], tab[10:11], 6.0, 1.0, 500, beta, abar, t_stop=498)   # 2 warm-up steps
        t0 = time.perf_counter()
        dref.reverse_process(v_fn, z, tab[3:4], tab[10:11], 6.0, 1.0, 500, beta, abar, t_stop=480)   # steps 499..480
        per_step = (time.perf_counter() - t0) / 20
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:
        pass
    return {"value": v01, "unit": "img/s", "cores": threads, "kind": "port", "cpu_model": cpu_model,
            "sample": f"oracle (fp32 CPU restatement of the reference), B={B} 96x96, fwd+bwd+clip+AdamW, no accumulation, "
                      f"dropout 0.1 as shipped: 1 warm-up + 2 timed optimizer steps; torch {torch.__version__} CPU, "
                      f"{threads} threads",
            "dropout_0": {"value": v00, "unit": "img/s", "sample": "same, dropout 0.0: 2 warm-up + 5 timed steps"},
            "sampling": {"value": 1.0 / (500 * per_step), "unit": "img/s", "seconds_per_step": per_step,
                         "sample": "B=1 CFG sampler, 2 forwards per step, 20 timed steps (t = 499..480) extrapolated "
                                   "to 500 steps"}}


def launch_ranks(n: int) -> int:
    """``python bench.py --gpus N`` without a launcher: start N rank processes (torch.distributed.run, one per GPU,
    rendezvous on 127.0.0.1) from THIS process, which never touches a GPU itself, and return their exit code.  Rank 0
    of the children prints the JSON line on the inherited stdout."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a.gpus))   # before anything initialises a GPU in this process
    # stdout carries exactly ONE line, the JSON: libraries print banners there (RCCL writes its version block to stdout
    # when a communicator is created), so file descriptor 1 points at stderr for the whole run and the result goes to
    # the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        _run(a, real_stdout)
    finally:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        os.close(real_stdout)


def _run(a, real_stdout):
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: one rank per GPU is the contract")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if (world > 1 or a.rehearse_dp) and os.environ.get("DM_RESERVE_STREAMS", "1") != "0":
        from diffusion_model_amd import kernels as K0
        K0.reserve_streams(dev)   # the step's own streams take their hardware queues before RCCL creates its streams
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
        assert dist.get_world_size() == a.gpus, (dist.get_world_size(), a.gpus)
    elif a.rehearse_dp and os.environ.get("DM_REHEARSE_MODE", "") != "hooksonly":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        if os.environ.get("DM_PG_LAZY", "") == "1":
            dist.init_process_group("nccl", rank=0, world_size=1)
        else:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    from diffusion_model_amd.ema import ModelEmaV2
    from diffusion_model_amd.sampler import CFGSampler
    from diffusion_model_amd.trainer import Trainer
    from diffusion_model_amd.utils import DDPM_beta_t_cosine_scheduler

    model, table = build_model(a.dropout, dev)
    model.train()
    sched = DDPM_beta_t_cosine_scheduler(500)
    ema = ModelEmaV2(model, decay=0.995)
    tr = Trainer(model, sched, lr=5e-5, weight_decay=1e-4, betas=(0.9, 0.99), clip=1.0, ema_decay=0.995,
                 accum=2, total_steps=100 * 79, start_step=20 * 40, low_t_weight=0.01, ema=ema,
                 distributed=world > 1 or (a.rehearse_dp and os.environ.get("DM_REHEARSE_MODE", "") != "pgonly"))
    if a.rehearse_dp and tr.buckets is not None:
        tr.buckets.force = os.environ.get("DM_REHEARSE_NOFORCE", "") != "1"   # (=1: process group and hooks, no collective)
    data = StepData(a.batch, rank, dev, table)
    batches = [synth_batch(a.batch, rank * 2 + i, dev, table, sched) for i in range(2)]   # (--graph: static inputs)

    def opt_step():
        loss = None
        for i in range(2):
            loss = tr.micro_step(*data.draw(i))   # t, e, label drop and context drawn per micro-step, as the reference does
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        opt_step()
    barrier()
    if a.graph and world == 1:
        tr.enable_graph(*batches[0])
        for _ in range(2):
            opt_step()
        barrier()
    # roofline leg: HIP events around every launch of the dominant MFMA kernel (nearest-x2 + conv3x3 at the
    # L4.upsample.conv shape) INSIDE the timed steps, on the stream it is launched on
    from diffusion_model_amd import kernels as K
    probe = None
    if not a.no_roofline and rank == 0:
        probe = {"mode": 2, "H": 48, "cin": 128, "cout": 128, "pairs": []}
        K.PROBE = probe
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = opt_step()
    barrier()
    dt = time.perf_counter() - t0
    K.PROBE = None
    if a.graph and world == 1:
        tr.disable_graph()
    if world > 1:
        tt = torch.tensor([dt], device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    imgs = a.steps * 2 * a.batch * world
    value = imgs / dt
    final_loss = float(loss.item())

    # ---------------- secondary leg: dropout 0.0 ("parity mode", SURVEY 8d asks for both); the headline keeps 0.1
    dropout0 = None
    if not a.no_elided and a.dropout > 0:
        saved_p = []
        for mod in model.modules():
            for attr in ("p", "dropout_prob"):
                if isinstance(getattr(mod, attr, None), float):
                    saved_p.append((mod, attr, getattr(mod, attr)))
                    setattr(mod, attr, 0.0)
        try:
            for _ in range(2):
                opt_step()
            barrier()
            t0d = time.perf_counter()
            for _ in range(a.steps):
                opt_step()
            barrier()
            dtd = time.perf_counter() - t0d
        finally:
            for mod, attr, val in saved_p:
                setattr(mod, attr, val)
        if world > 1:
            tt = torch.tensor([dtd], device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dtd = float(tt.item())
        dropout0 = {"value": imgs / dtd, "unit": "img/s", "ms_per_step": dtd / a.steps * 1e3}

    # ---------------- secondary leg: the same step with context=None (reference inference_sweeps.py:110-115 /
    # Classifier_Guidance.py:177-181 style models): no cross-attention runs, so the self-attention blocks are LIVE --
    # their output reaches the loss (under the one-token CLIP context they are provably dead)
    train_live_attention = None
    if not a.no_elided:
        def opt_step_noctx():
            loss = None
            for i in range(2):
                x_, _, t_, e_ = data.draw(i)
                loss = tr.micro_step(x_, None, t_, e_)
            return loss
        for _ in range(2):
            opt_step_noctx()
        barrier()
        t0n = time.perf_counter()
        for _ in range(a.steps):
            lossn = opt_step_noctx()
        barrier()
        dtn = time.perf_counter() - t0n
        if world > 1:
            tt = torch.tensor([dtn], device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dtn = float(tt.item())
        train_live_attention = {"value": imgs / dtn, "unit": "img/s", "ms_per_step": dtn / a.steps * 1e3,
                                "final_loss": float(lossn.item()),
                                "what": "same optimizer step with context=None: cross-attention absent, the 48x48 and 24x24 "
                                        "self-attention blocks live (their gradients reach the loss)"}

    # ---------------- secondary leg: the SAME optimizer step in the fp32 compute mode (reference precision: fp32 NHWC
    # activations, fp32 MFMA contractions, fp32 attention; parity 1e-4, tests/test_gpu_fp32.py).  1 GPU only.
    train_fp32 = None
    if world == 1 and not a.no_fp32:
        with K.compute_dtype(torch.float32):
            opt_step()
            barrier()
            n32 = max(2, min(3, a.steps))
            t0f = time.perf_counter()
            for _ in range(n32):
                loss32 = opt_step()
            barrier()
            dtf = time.perf_counter() - t0f
        v32 = n32 * 2 * a.batch / dtf
        train_fp32 = {"value": v32, "unit": "img/s", "ms_per_step": dtf / n32 * 1e3, "steps": n32, "dtype": "f32",
                      "final_loss": float(loss32.item()),
                      "mfma_f32_frac_algorithmic": v32 * TRAIN_FLOP_PER_IMG / PEAK_F32,
                      "what": "same config and step in the fp32 compute mode (kernels.set_compute_dtype(torch.float32)): "
                              "v_mfma_f32_16x16x4_f32 contractions (157.3 TFLOP/s peak), general untuned kernels -- the "
                              "reference-precision / parity mode, not the headline"}

    # ---------------- secondary leg (never the headline): the same steps with unet.ELIDE_DEAD_ATTENTION on.  With the
    # one-token CLIP context the self-attention blocks and to_q / to_k cannot reach the output; skipping
    # them leaves outputs and live gradients unchanged (tests/test_gpu_train_sample.py::test_elide_dead_attention_is_exact)
    elided = None
    if not a.no_elided:
        from diffusion_model_amd import unet as U
        U.ELIDE_DEAD_ATTENTION[0] = True
        try:
            for _ in range(2):
                opt_step()
            barrier()
            t0e = time.perf_counter()
            for _ in range(a.steps):
                opt_step()
            barrier()
            dte = time.perf_counter() - t0e
        finally:
            U.ELIDE_DEAD_ATTENTION[0] = False
        if world > 1:
            tt = torch.tensor([dte], device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dte = float(tt.item())
        elided = {"value": imgs / dte, "unit": "img/s", "ms_per_step": dte / a.steps * 1e3,
                  "what": "same config with the provably dead self-attention / to_q / to_k work skipped (opt-in, "
                          "unet.ELIDE_DEAD_ATTENTION); identical outputs and live gradients; NOT the headline value"}

    # ---------------- sampling leg (batch-sharded, no collective in the loop)
    sampling = None
    from diffusion_model_amd.sampler import shard_bounds
    TOTAL = 704                                    # configs[3]: 11 labels x 64 samples
    if a.sample_images < 0:
        lo, hi = shard_bounds(TOTAL, rank, world)  # batch-sharded over the ranks, contiguous shards (sampler.sample_sharded)
    else:
        lo, hi = rank * a.sample_images, (rank + 1) * a.sample_images
    S = hi - lo
    n_images = TOTAL if a.sample_images < 0 else a.sample_images * world
    if n_images > 0:
        smp = CFGSampler(ema.module, sched, table, uncond_id=10)
        gz = torch.Generator(device="cpu").manual_seed(42 + rank)
        z = torch.randn(S, 3, 96, 96, generator=gz).to(dev)
        labels = torch.arange(lo, hi) % 11
        chunk = max(1, a.sample_chunk)

        def chunks(fn):   # the shard in launches of `chunk` images (UNet batch 2 x chunk); no collective anywhere
            return torch.cat([fn(z[c:c + chunk], labels[c:c + chunk]) for c in range(0, S, chunk)]) if S else z

        wl = min(S, chunk)
        if S:
            smp.sample(z[:wl], labels[:wl], 6.0, 1.0, num_time_steps=500, t_stop=497)  # warm-up: 3 steps
        barrier()
        t1 = time.perf_counter()
        x0 = chunks(lambda zz, ll: smp.sample(zz, ll, 6.0, 1.0, num_time_steps=a.sample_steps))
        barrier()
        ds = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([ds], device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ds = float(tt.item())
        ok = bool(torch.isfinite(x0).all().item())
        sampling = {"metric": "500-step CFG sample img/s", "value": n_images / ds, "unit": "img/s",
                    "workload": "configs[3]: 500-step cosine DDPM, eta 1.0, CFG 6.0 (cond+uncond batched), "
                                "11 labels x 64 samples = 704 images, batch-sharded over the ranks"
                                if a.sample_images < 0 else "%d images per GPU (reduced: NOT configs[3])" % a.sample_images,
                    "images": n_images, "images_per_launch": chunk, "steps": a.sample_steps, "seconds": ds,
                    "guidance": 6.0, "eta": 1.0, "cond_uncond_batched": True, "finite": ok,
                    "mfma_frac_algorithmic": (n_images / ds) * FWD_FLOP_PER_IMG * 2 * a.sample_steps / (PEAK_BF16 * world)}
        # DPM-Solver++(2M) leg (reference main_stl10.py:634-703, its default 50 steps): the same
        # images on the fused multistep kernel.  Secondary: fewer UNet evaluations per image, not a faster kernel.
        from diffusion_model_amd.dpm_solver import DPMSolverMultistepScheduler
        dpm = DPMSolverMultistepScheduler(num_train_timesteps=500, beta_start=0.00085, beta_end=0.012,
                                          beta_schedule="scaled_linear", prediction_type="v_prediction",
                                          algorithm_type="dpmsolver++", solver_order=2)
        gd = torch.where(labels == 10, 0.0, 6.0)
        if S:
            smp.sample_dpm(z[:wl], labels[:wl], dpm, cfg_scale=gd[:wl], num_inference_steps=3)
        barrier()
        t1 = time.perf_counter()
        xd = torch.cat([smp.sample_dpm(z[c:c + chunk], labels[c:c + chunk], dpm, cfg_scale=gd[c:c + chunk],
                                       num_inference_steps=50) for c in range(0, S, chunk)]) if S else z
        barrier()
        dd = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([dd], device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dd = float(tt.item())
        sampling["dpm_solver_pp_2m"] = {"value": n_images / dd, "unit": "img/s", "steps": 50, "seconds": dd,
                                        "guidance": 6.0, "finite": bool(torch.isfinite(xd).all().item())}
        if elided is not None and S:  # one launch of the same sampling run with the dead attention work skipped (secondary)
            from diffusion_model_amd import unet as U
            U.ELIDE_DEAD_ATTENTION[0] = True
            try:
                smp.sample(z[:wl], labels[:wl], 6.0, 1.0, num_time_steps=500, t_stop=497)
                barrier()
                t1 = time.perf_counter()
                x0e = smp.sample(z[:wl], labels[:wl], 6.0, 1.0, num_time_steps=a.sample_steps)
                barrier()
                dse = time.perf_counter() - t1
            finally:
                U.ELIDE_DEAD_ATTENTION[0] = False
            if world > 1:
                tt = torch.tensor([dse], device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dse = float(tt.item())
            elided["sampling_value"] = wl * world / dse
            elided["sampling_images"] = wl * world
            elided["sampling_finite"] = bool(torch.isfinite(x0e).all().item())

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    out = {
        "metric": "train img/s (node) STL10 96x96 UNet B=64/GPU", "value": value, "unit": "img/s",
        "n_gpus": world, "world_size": dist.get_world_size() if world > 1 else 1,
        "collective": "RCCL all-reduce (torch.distributed nccl backend), 6 buckets, only on accumulation boundaries"
                      if world > 1 else None,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
        # (the compute dtype of the headline legs: bf16 unless the process was started with DM_COMPUTE_DTYPE=fp32)
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if K.ACT[0] == torch.float32 else "bf16",
        "data": "synthetic", "final_loss": final_loss,
        "config": {"workload": "configs[1]: main_stl10 UNet v-pred train step 96x96x3, B=64/GPU "
                               + ("f32" if K.ACT[0] == torch.float32 else "bf16") + ", AdamW+EMA+clip, "
                               "accumulation 2 (step = 1 optimizer step = 128 img/GPU), CLIP ctx, low-t weighting",
                   "global_batch": 2 * a.batch * world, "micro_batch": a.batch, "dropout": a.dropout,
                   "parallelism": f"dp{world}", "weights": "reference init (seed 0), output_conv N(0,0.05^2)"},
        "step_mfma_frac_algorithmic": value * TRAIN_FLOP_PER_IMG / (PEAK_BF16 * world),
    }
    if dropout0:
        out["train_dropout_0"] = dropout0
    if train_fp32:
        out["train_fp32"] = train_fp32
    if train_live_attention:
        out["train_live_attention"] = train_live_attention
    if elided:
        out["dead_attention_elided"] = elided
    if sampling:
        out["sampling"] = sampling
    if world == 1 and not a.no_latent and not a.no_elided:
        out["latent_unet"] = latent_leg(a, dev, table, sched)
    if world == 1 and not a.no_mnist:
        out["mnist"] = mnist_leg(a, dev)
    if a.graph:
        out["config"]["launch"] = "micro-step replayed as one HIP graph"
    if not a.no_roofline:
        iso_sec, fl, variant = time_conv_kernel(a.batch, dev)
        sec = iso_sec
        if probe and probe["pairs"]:
            sec = sum(e0.elapsed_time(e1) for e0, e1 in probe["pairs"]) * 1e-3 / len(probe["pairs"])
        # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE,
        # collected in separate rocprofv3 --pmc runs of the same kernel and shape; bench.py cannot run PMC itself)
        traffic = traffic_src = None
        for name in ("r03_pmc_dominant_conv.json", "r02_pmc_dominant_conv.json"):
            pmc = os.path.join(ROOT, "profiles", name)
            if a.batch == 64 and os.path.exists(pmc):
                traffic, traffic_src = json.load(open(pmc))["traffic_bytes"], "profiles/" + name
                break
        out["roofline"] = {"bound": "mfma", "achieved": fl / sec / 1e12, "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                           "frac": fl / sec / PEAK_BF16, "traffic": traffic,
                           "traffic_source": (traffic_src or "none") + " (separate rocprofv3 --pmc passes of this kernel and "
                                             "shape, FETCH_SIZE x2-corrected + WRITE_SIZE; not re-measured by this run)",
                           "kernel": "%s: nearest-x2 + conv3x3 128->128 48->96 (L4.upsample.conv), B=%d" % (variant, a.batch),
                           "avg_launch_us": sec * 1e6, "flop_per_launch": fl,
                           "launches_timed": len(probe["pairs"]) if probe else 0,
                           "timed_where": "inside the timed training steps (HIP events around each launch)",
                           "isolated_back_to_back_us": iso_sec * 1e6}
        # the other conv instantiations of the step, back to back after the legs (the 64-cout tile is the top line of
        # the rocprof table by total time; stride 2 is the slowest per flop)
        others = []
        for what, H, cin, cout, mode in (("L1.conv / late_conv2 64->64 @96x96", 96, 64, 64, 0),
                                         ("late_conv1 128->64 @96x96", 96, 128, 64, 0),
                                         ("L2.conv2 128->128 @48x48", 48, 128, 128, 0),
                                         ("L1.downsample stride 2 64->64 96->48", 96, 64, 64, 1)):
            sec_o, fl_o, var_o = time_conv_kernel(a.batch, dev, H, cin, cout, mode)
            others.append({"kernel": f"{var_o}: {what}", "avg_launch_us": sec_o * 1e6,
                           "achieved": fl_o / sec_o / 1e12, "frac": fl_o / sec_o / PEAK_BF16})
        out["roofline"]["other_conv_kernels_back_to_back"] = others
        out["roofline_hbm"] = hbm_rooflines(dev, a.batch)
        out["roofline_attention"] = attention_roofline(dev, a.batch, a.dropout)
    if not a.no_cpu_baseline and world == 1:   # the CPU leg is reported at N = 1 only
        out["cpu_baseline"] = cpu_baseline()
    os.write(real_stdout, (json.dumps(out) + "\n").encode())   # the one line of stdout; written before the process group is torn down
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
