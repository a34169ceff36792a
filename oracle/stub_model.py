"""ORACLE (test infrastructure): a deterministic stand-in for the UNet used by oracle/gen_golden_diffusion.py when it
runs the reference's sampler / train-step code, and by the tests that replay those fixtures.  No RNG; depends on every
argument (image, timestep, context).  Not reference code."""
import torch
import torch.nn as nn


def stub_model(z, t, ctx):
    tt = t.to(z.device).float().view(-1, 1, 1, 1)
    if torch.is_floating_point(ctx):
        c = ctx.float().mean(dim=-1).view(-1, 1, 1, 1)
    else:
        c = ctx.float().view(-1, 1, 1, 1) * 0.01
    return torch.tanh(0.7 * z) * 0.8 + 0.3 * torch.sin(tt / 50.0) + 4.0 * c


class StubNet(nn.Module):   # the reference's run_reverse_process calls model.eval()
    def forward(self, z, t, ctx):
        return stub_model(z, t, ctx)
