"""cProfile of the host side of the training step (which Python functions the launch path spends its time in)."""
import cProfile, pstats, os, sys, io
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
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    for b in batches:
        tr.micro_step(*b)
    torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue())
