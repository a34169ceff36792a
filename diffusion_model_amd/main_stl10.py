"""Drop-in for the reference's ``main_stl10.train()`` / ``inference()`` (main_stl10.py:352-545, 547-566) on
the HIP hot path.  Same call signatures, same arithmetic per batch, same checkpoint dict
(``{'weights', 'optimizer', 'ema'}``, main_stl10.py:533-539).

``train`` takes its batches from a ``train_loader`` argument (any iterable of ``(x[B,3,96,96] in [-1,1],
labels[B])`` with ``len()``).  When none is given it reads the STL10 binary files under ``stl10_data`` with
``data.STL10Loader`` (device-resident uint8 split, the reference's ToTensor + Normalize restated; torchvision
itself is not used) and falls back to a seeded synthetic loader of the same shape when the files are absent
(nothing is ever downloaded).  Outside the hot path, on purpose: the matplotlib
monitoring images -- ``inference`` returns the fixed-seed class grid as a tensor instead of writing PNGs, and
monitoring hooks can be attached through ``on_epoch_end``."""
from __future__ import annotations

import os
from typing import Callable, Iterable, List, Optional

import torch

from . import config, utils
from .ema import ModelEmaV2
from .sampler import CFGSampler
from .trainer import Trainer
from .utils import DDPM_beta_t_cosine_scheduler, load_model

BATCH_SIZE = config.BATCH_SIZE
LR = config.LR
EMA_DECAY = config.EMA_DECAY
NUM_TIME_STEPS = config.NUM_TIME_STEPS
CFG_SCALE = config.CFG_SCALE
CFG_P_UNCOND = config.CFG_P_UNCOND
SAMPLING_ETA = config.SAMPLING_ETA
UNCOND_ID = config.UNCOND_ID
VIRTUAL_BATCH_SIZE_MULTIPLIER = config.VIRTUAL_BATCH_SIZE_MULTIPLIER
TOTAL_EPOCHS_NEW = config.TOTAL_EPOCHS_NEW
START_EPOCH = config.START_EPOCH
CLIP_TABLE_PATH = utils.CLIP_TABLE_PATH
device = utils.device


class SyntheticLoader:
    """Stand-in for the STL10 DataLoader: ``n_batches`` seeded batches of clamp(N(0, 0.5^2), -1, 1) images
    and uniform labels 0..9 (the bench's recipe)."""

    def __init__(self, n_batches=8, batch_size=BATCH_SIZE, seed=1234, size=96):
        self.n, self.bs, self.seed, self.size = n_batches, batch_size, seed, size

    def __len__(self):
        return self.n

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        for _ in range(self.n):
            x = (torch.randn(self.bs, 3, self.size, self.size, generator=g) * 0.5).clamp_(-1, 1)
            yield x, torch.randint(0, 10, (self.bs,), generator=g)


def _checkpoint_dict(model, trainer: Trainer, ema: ModelEmaV2) -> dict:
    return {"weights": {k: v.detach().clone() for k, v in model.state_dict().items()},
            "optimizer": trainer.optimizer_state_dict(),
            "ema": {k: v.detach().clone() for k, v in ema.state_dict().items()}}


def train(checkpoint_path=None, clip_table_path: str = CLIP_TABLE_PATH, *, train_loader: Optional[Iterable] = None,
          total_epochs: int = TOTAL_EPOCHS_NEW, start_epoch: int = START_EPOCH, checkpoint_dir: str = ".",
          checkpoint_every: int = 5, on_epoch_end: Optional[Callable] = None, check_finite: bool = True,
          distributed: Optional[bool] = None, log: Callable = print, monitor_dir: Optional[str] = None,
          monitor_every: int = 1, seed: Optional[int] = None):
    """Reference main_stl10.py:352-545.  Returns ``(model, ema, trainer)``.  ``monitor_dir``: when given, the
    reference's end-of-epoch images (:521-524: reconstruction sweep, denoising collage, fixed-seed class grid) are
    written under it every ``monitor_every`` epochs from the EMA weights, each as one batched sampling job.

    Data parallel (``distributed``, default: on when a process group is initialised; one process per GPU under
    torchrun): every rank reads ITS shard of the epoch order (the loaders' ``rank``/``world_size``), continues from
    rank 0's parameters / EMA / moments (``Trainer.broadcast_state``), draws label drops, timesteps, noise and dropout
    masks from rank-offset seeds (``seed``, default ``torch.initial_seed()``), skips a non-finite batch on all ranks
    or on none, and only rank 0 logs, writes checkpoints and monitoring images."""
    import torch.distributed as dist
    if distributed is None:
        distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if distributed else 0
    world = dist.get_world_size() if distributed else 1
    if rank != 0:
        log = lambda *a, **k: None   # noqa: E731  (one voice per job)
    scheduler = DDPM_beta_t_cosine_scheduler(NUM_TIME_STEPS)
    if train_loader is None and config.LDM:
        # reference utils.py:127-132: pre-computed VAE latents (the latent model, utils.py:145-156); no synthetic
        # stand-in here -- without the file there is nothing to train the latent model on
        from .data import LatentLoader
        train_loader = LatentLoader(os.path.join("docs", "data", "imagenet_latents.pt" if config.IMAGE_NET
                                                 else "stl10_latents.pt"), batch_size=BATCH_SIZE, device=device,
                                   rank=rank, world_size=world)
    if train_loader is None:
        try:  # reference: torchvision STL10(root='stl10_data', split='train'), shuffled, batch_size BATCH_SIZE
            from .data import STL10Loader
            train_loader = STL10Loader("stl10_data", "train", batch_size=BATCH_SIZE, shuffle=True, device=device,
                                       rank=rank, world_size=world)
        except FileNotFoundError as exc:
            log(f"[warn] {exc}; using SyntheticLoader")
            train_loader = SyntheticLoader(seed=1234 + rank)
    if utils.CONTEXT_TABLE is None and os.path.exists(clip_table_path):
        utils._load_clip_context_table(clip_table_path)
        log(f"[info] Loaded CLIP context table from {clip_table_path} with dim {utils.CONTEXT_TABLE.shape[-1]}")
    elif utils.CONTEXT_TABLE is None:
        log(f"[warn] CLIP table not found at {clip_table_path}; training without cross-attention context")
    model = load_model(LDM=config.LDM, NUM_TIME_STEPS=1000, DROPOUT=config.DROPOUT).to(device)
    ema = ModelEmaV2(model, decay=EMA_DECAY)

    if checkpoint_path is not None:
        if os.path.exists(checkpoint_path):
            ck = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
            model.load_state_dict(ck["weights"])
            try:
                ema.load_state_dict(ck["ema"])
            except Exception as exc:  # reference: re-initialise the EMA from the model
                log(f"[warn] EMA state_dict load failed ({exc}); initializing EMA from model.")
                ema.module.load_state_dict(model.state_dict())
            log(f"Loaded checkpoint from {checkpoint_path}")
        else:
            log(f"Checkpoint file not found: {checkpoint_path}")

    n_batches = len(train_loader)
    if distributed and world > 1:   # a custom train_loader may be ragged across ranks: every rank runs the common prefix
        nb = torch.tensor([n_batches], device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(nb, op=dist.ReduceOp.MIN)
        n_batches = int(nb.item())
    accum = max(1, VIRTUAL_BATCH_SIZE_MULTIPLIER)
    total_steps_new = total_epochs * n_batches                      # main_stl10.py:402 (counts batches)
    steps_so_far = start_epoch * ((n_batches + accum - 1) // accum)  # main_stl10.py:441-446
    trainer = Trainer(model, scheduler, lr=LR, weight_decay=config.WEIGHT_DECAY, betas=config.BETAS,
                      clip=config.GRAD_CLIP, ema_decay=EMA_DECAY, accum=accum, total_steps=total_steps_new,
                      start_step=steps_so_far, ema=ema, distributed=distributed)
    table = utils.CONTEXT_TABLE.to(device) if utils.CONTEXT_TABLE is not None else None
    # per-batch randomness (label drop, t, noise): the process-global generators, as in the reference -- unless the job is
    # data parallel or a seed is given: every process starts from the same default seed, so ranks would draw the SAME
    # timesteps and noise; they get generators seeded (seed, rank) instead
    g_cpu = g_dev = None
    if distributed or seed is not None:
        base = (torch.initial_seed() if seed is None else int(seed)) + 7919 * rank
        g_cpu = torch.Generator().manual_seed(base)
        g_dev = torch.Generator(device=device).manual_seed(base + 1)

    for epoch in range(start_epoch, total_epochs):
        low_t = config.USING_LOW_T_REWARDING and epoch >= config.LOW_T_REWARDING_WARMUP_EPOCHS
        trainer.low_t_weight = config.LOW_T_REWARDING_WARMUP_WEIGHT if low_t else None
        sum_loss = torch.zeros(1, device=device)
        model.train()
        for i, (x, labels) in enumerate(train_loader):
            if i >= n_batches:
                break
            x = x.to(device, non_blocking=True).float()
            labels = labels.to(device)
            drop_mask = torch.rand(labels.shape[0], device=device, generator=g_dev) < CFG_P_UNCOND   # main_stl10.py:455-457
            labels_dropped = labels.clone()
            labels_dropped[drop_mask] = UNCOND_ID
            t = torch.randint(0, NUM_TIME_STEPS, (x.shape[0],), generator=g_cpu).to(device)  # drawn on the host, as :458
            e = torch.randn(x.shape, device=device, dtype=x.dtype, generator=g_dev)
            context = table[labels_dropped.long()] if table is not None else None
            do_step = ((i + 1) % accum == 0) or (i + 1 == n_batches)                # main_stl10.py:506
            loss = trainer.micro_step(x, context, t, e, step_now=do_step, check_finite=check_finite)
            if not trainer.last_skipped:   # reference :484-486 `continue`s before `sum_loss += loss.item()` (:504)
                sum_loss += loss.detach().reshape(-1)[:1]
        log(f"Epoch {epoch + 1} | Loss {float(sum_loss.item()) / max(1, n_batches):.5f}")
        if monitor_dir is not None and (epoch + 1) % max(1, monitor_every) == 0 and utils.CONTEXT_TABLE is not None:
            from . import visualization as V
            # EVERY rank draws the monitoring batch: a shuffling loader advances its epoch-order generator per
            # iter(), and the ranks' generators must stay in lockstep or the next epoch's order[rank::world] shards
            # no longer partition the epoch (tests/test_cpu_data.py).  Only rank 0 samples from it and writes files.
            mon_batch = next(iter(train_loader))
            if rank == 0:   # single-rank jobs: the other ranks would write the same files
                V.reconstruct_sweep(ema.module, scheduler, [mon_batch], epoch + 1, device, NUM_TIME_STEPS, SAMPLING_ETA,
                                    None, monitor_dir, t_list=[700, 500, 300, 200], guidance_scale=CFG_SCALE)
                V.generate_denoise_collage(ema.module, scheduler, epoch + 1, 1, [499, 400, 300, 200, 100, 50, 15, 0],
                                           device, NUM_TIME_STEPS, SAMPLING_ETA, None, monitor_dir,
                                           guidance_scale=CFG_SCALE)
            # the class grid is batch-sharded over the ranks (one gather at the end): every rank takes part
            grid = generate_fixed_seed_class_grid(ema.module, scheduler, num_time_steps=NUM_TIME_STEPS)
            if rank == 0:
                V.save_image_strip([g for g in grid], [V.STL10_CLASS_NAMES[i] for i in range(grid.shape[0])],
                                   os.path.join(monitor_dir, f"epoch_{epoch + 1}", f"epoch_{epoch + 1}_fixed_seed_grid.png"))
        if on_epoch_end is not None:
            on_epoch_end(epoch + 1, model, ema, scheduler)
        if checkpoint_every and (epoch + 1) % checkpoint_every == 0 and rank == 0:   # replicas are identical: one writer
            torch.save(_checkpoint_dict(model, trainer, ema),
                       os.path.join(checkpoint_dir, f"stl10_checkpoint_epoch_{epoch + 1}.pt"))
    return model, ema, trainer


def generate_fixed_seed_class_grid(model, scheduler, cfg_scale: float = CFG_SCALE, eta: float = SAMPLING_ETA,
                                   label_ids=None, base_seed: int = 42, num_time_steps: int = NUM_TIME_STEPS,
                                   size: int = 96) -> torch.Tensor:
    """Reference visualization.py:404-461 without the plotting: one sample per label, start noise of label i
    drawn from seed ``base_seed + i`` (the reference seeds the global generator per label), guidance 0.0 for
    the unconditional label (:424).  All labels run as ONE batch through the fused sampler; the per-step
    ancestral noise comes from one device generator seeded ``base_seed`` (the reference's device RNG stream is
    not reproducible across backends anyway).  Returns ``[len(label_ids), 3, size, size]`` in [-1, 1]."""
    if label_ids is None:
        label_ids = list(range(10)) + [UNCOND_ID]
    z = torch.cat([torch.randn(1, 3, size, size, generator=torch.Generator().manual_seed(base_seed + i))
                   for i in range(len(label_ids))]).to(device)
    labels = torch.tensor(label_ids, device=device, dtype=torch.long)
    import torch.distributed as dist
    from .sampler import sample_sharded
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    gen = torch.Generator(device=device).manual_seed(base_seed + rank)
    sampler = CFGSampler(model, scheduler, utils.CONTEXT_TABLE.to(device), UNCOND_ID)
    model.eval()
    with torch.no_grad():  # under torchrun: every rank samples its shard of the labels, one gather at the end
        return sample_sharded(sampler.sample, z, labels, guidance_scale=cfg_scale, eta=eta,
                              num_time_steps=num_time_steps, generator=gen)


def inference(checkpoint_path, label: Optional[int] = None, guidance_scale: float = CFG_SCALE,
              clip_table_path: str = CLIP_TABLE_PATH, *, num_time_steps: int = NUM_TIME_STEPS, size: int = 96):
    """Reference main_stl10.py:547-566: load ``weights`` and ``ema`` from the checkpoint, sample with the EMA
    weights.  Returns the fixed-seed class grid tensor (all 11 labels, or only ``label``)."""
    model, ema = utils.load_checkpoint(checkpoint_path, clip_table_path)
    scheduler = DDPM_beta_t_cosine_scheduler(NUM_TIME_STEPS)
    ids = None if label is None else [int(label)]
    return generate_fixed_seed_class_grid(ema.module, scheduler, cfg_scale=guidance_scale, eta=SAMPLING_ETA,
                                          label_ids=ids, num_time_steps=num_time_steps, size=size)


def generate_cfg_grids_dpm_solver(checkpoint_path: str, cfg_scales: Optional[List[float]] = None,
                                  num_inference_steps: int = 50, clip_table_path: str = CLIP_TABLE_PATH, *,
                                  output_dir: str = os.path.join("intermiate imagees", "dpm"),
                                  latent_shape: tuple = (1, 3, 96, 96)):
    """Reference main_stl10.py:634-703: load a checkpoint and write one fixed-seed class grid per CFG scale, sampled
    with DPM-Solver++(2M) under the scheduler configuration the reference constructs (:676-684).  Each grid is one
    batched job on the fused sampler.  Returns the list of latent batches (one per scale)."""
    from . import visualization as V
    from .dpm_solver import DPMSolverMultistepScheduler
    if cfg_scales is None:
        cfg_scales = [3.0, 5.0, 7.0]
    model, ema = utils.load_checkpoint(checkpoint_path, clip_table_path)
    scheduler = DPMSolverMultistepScheduler(num_train_timesteps=NUM_TIME_STEPS, beta_start=0.00085, beta_end=0.012,
                                            beta_schedule="scaled_linear", prediction_type="v_prediction",
                                            algorithm_type="dpmsolver++", solver_order=2)
    import re
    m = re.search(r"epoch_(\d+)", os.path.basename(str(checkpoint_path)))    # main_stl10.py:567-580; 0 when absent
    epoch = int(m.group(1)) if m else 0
    outs = []
    for cfg in cfg_scales:
        outs.append(V.generate_fixed_seed_class_grid_dpm(ema.module, scheduler, epoch, 1, cfg_scale=cfg,
                                                         num_inference_steps=num_inference_steps,
                                                         label_ids=list(V.STL10_CLASS_NAMES.keys()),
                                                         label_names=V.STL10_CLASS_NAMES, base_seed=42, device=device,
                                                         output_dir=output_dir, latent_shape=latent_shape))
    return outs
