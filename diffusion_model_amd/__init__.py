"""MI355X-native UNet denoiser hot path (drop-in for barebucan/Diffusion_model's unet.py /
utils.py / main_stl10.py API).  Compute runs in hand-written gfx950 HIP kernels loaded from
libdm_hip.so through a C ABI (include/dm_hip.h); there is no CPU / eager fallback."""
__version__ = "0.1.0"
