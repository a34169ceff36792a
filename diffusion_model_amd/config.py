"""Constants of the reference's intended STL10 image-space configuration (reference config.py:1-44 and
main_stl10.py:35-107).  The reference ships LDM=True / IMAGE_NET=True, which needs an un-fetchable
pretrained VAE; the hot path this package replaces is the image-space STL10
model, so those two toggles default to False here and UNCOND_ID is 10 (visualization.py:11-14)."""
LDM = False
IMAGE_NET = False
DROPOUT = 0.1
EMA_DECAY = 0.995            # main_stl10.py:39 (config.py:4 says 0.999; train() uses main_stl10's)
NUM_TIME_STEPS = 500
USING_LPIPS = False
UNCOND_ID = 1000 if IMAGE_NET else 10
UNCOND_LABEL = "uncond"

BATCH_SIZE = 64              # main_stl10.py:37
LR = 5e-5                    # main_stl10.py:38
WEIGHT_DECAY = 1e-4          # main_stl10.py:396
BETAS = (0.9, 0.99)
GRAD_CLIP = 1.0              # main_stl10.py:508
CFG_SCALE = 8.0              # main_stl10.py:69 (README best setting: 6.0)
CFG_P_UNCOND = 0.2           # main_stl10.py:70
SAMPLING_ETA = 1.0           # main_stl10.py:71
VIRTUAL_BATCH_SIZE_MULTIPLIER = 2   # main_stl10.py:107
TOTAL_EPOCHS_NEW = 100       # main_stl10.py:50
START_EPOCH = 20             # main_stl10.py:51

LOW_T_REWARDING_WARMUP_EPOCHS = 20      # config.py:42
LOW_T_REWARDING_WARMUP_WEIGHT = 0.01    # config.py:43
USING_LOW_T_REWARDING = True            # config.py:44
