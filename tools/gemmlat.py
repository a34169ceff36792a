import sys, torch
sys.path.insert(0, ".")
from diffusion_model_amd import kernels as K
dev = torch.device("cuda:0")
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for M, Kd, N in [(16384, 512, 512), (16384, 512, 1536), (65536, 256, 256), (4096, 1024, 1024), (4096, 1024, 3072),
                 (16384, 1024, 512), (16384, 256, 512), (65536, 512, 256), (147456, 128, 384), (147456, 128, 128), (36864, 256, 768), (36864, 256, 256)]:
    x = torch.randn(M, Kd, device=dev).to(torch.bfloat16)
    w = torch.randn(N, Kd, device=dev) * 0.05
    wp = K.pack_weight(w)
    us = timeit(lambda: K.gemm(x, wp, N))
    print(f"M={M:7d} K={Kd:5d} N={N:5d}  {us:8.1f} us  {2*M*Kd*N/us/1e6:8.1f} TF/s  {(M*(Kd+N)*2)/us/1e3:8.1f} GB/s")
