"""End to end on real kernels, no oracle involved: the STL10-configuration UNet (cross-attention on a one-token context, at
32x32) is TRAINED through Trainer.micro_step (q-sample, v-target, loss, hand-written backward, clip + AdamW + EMA, LR
schedule, label drop) on a synthetic four-class dataset and then SAMPLED with the 500-step CFG sampler from the EMA weights:
the loss must fall and every generated image must carry its class's colour.  Parity tests pin each step against the
reference; this one checks that the steps compose into a system that learns (tools/learn_check.py is the same run as a
script)."""
import os
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["bf16", "fp32"])
def test_training_learns_class_colours_and_cfg_sampling_reproduces_them(mode):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import learn_check
    from diffusion_model_amd import kernels as K
    with K.compute_dtype(torch.float32 if mode == "fp32" else torch.bfloat16):
        first, last, hit, img = learn_check.run(steps=300, B=32, H=32, lr=2e-4, seed=0, verbose=False)
    assert torch.isfinite(img).all()
    assert last < 0.85 * first, (first, last)                      # measured 0.166 -> 0.118 after 300 steps
    assert hit >= 0.9, hit                                          # measured 1.0: 32 of 32 images nearest to their own class colour
    cols = torch.tensor([[0.8, -0.6, -0.6], [-0.6, 0.8, -0.6], [-0.6, -0.6, 0.8], [0.7, 0.7, -0.7]])
    labels = torch.arange(4).repeat_interleave(8)
    for c in range(4):
        mean_c = img[labels == c].mean(dim=(0, 2, 3))
        assert float((mean_c - cols[c]).abs().max()) < 0.4, (c, mean_c)
