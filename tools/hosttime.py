"""Host issue time vs GPU time of one training optimizer step (2 micro-steps of B=64): time until the Python calls
return (all launches enqueued) against the time until the device is idle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as Bn  # noqa: E402
from diffusion_model_amd.ema import ModelEmaV2
from diffusion_model_amd.trainer import Trainer
from diffusion_model_amd.utils import DDPM_beta_t_cosine_scheduler

dev = torch.device("cuda:0")
model, table = Bn.build_model(0.1, dev)
model.train()
sched = DDPM_beta_t_cosine_scheduler(500)
ema = ModelEmaV2(model, decay=0.995)
tr = Trainer(model, sched, lr=5e-5, weight_decay=1e-4, betas=(0.9, 0.99), clip=1.0, ema_decay=0.995, accum=2,
             total_steps=7900, start_step=800, low_t_weight=0.01, ema=ema, distributed=False)
batches = [Bn.synth_batch(64, i, dev, table, sched) for i in range(2)]
for _ in range(4):
    for b in batches:
        tr.micro_step(*b)
torch.cuda.synchronize()
host, total = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches:
        tr.micro_step(*b)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(t1 - t0); total.append(t2 - t0)
print(f"host issue {1e3 * sum(host) / len(host):.2f} ms   until idle {1e3 * sum(total) / len(total):.2f} ms per optimizer step (from an idle device)")
