"""8x8-stage convolutions of the latent model (B=64): time per call and TFLOP/s."""
import sys, torch
sys.path.insert(0, ".")
from diffusion_model_amd import kernels as K
dev = torch.device("cuda:0")
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for B, H, cin, cout in [(64, 8, 512, 1024), (64, 8, 1024, 1024), (64, 8, 1024, 512), (3, 8, 1024, 1024), (64, 16, 512, 512), (64, 16, 1024, 512)]:
    x = torch.randn(B, H, H, cin, device=dev).to(torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.02
    wp = K.pack_weight(w)
    us = timeit(lambda: K.conv3x3(x, wp, cout))
    print(f"B={B:3d} {H}x{H} {cin:5d}->{cout:5d}  {us:8.1f} us  {2*B*H*H*cin*cout*9/us/1e6:8.1f} TF/s")
