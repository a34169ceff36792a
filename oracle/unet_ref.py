"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU fp32 restatement of the reference UNet denoiser, written functionally over a
flat ``state_dict`` (no nn.Module), so that it can be driven with the exact
parameter names/shapes of the reference checkpoint format.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file.  The product package ``diffusion_model_amd`` must
never import it.

Pinned by: ``tests/golden/*.npz`` (generated in the build container by
``oracle/gen_golden.py``, which imports the real ``/root/reference/unet.py``) and
the two notebook known-answer vectors.  The reference has
no test-suite of its own, so there are no reference-held fixtures to pin to.

Reference statements followed (file:line into /root/reference):
  * sinusoidal table ............ unet.py:44-66
  * ResBlock .................... unet.py:70-108
  * self-attention .............. unet.py:110-139
  * cross-attention ............. unet.py:141-220
  * up/down-sample .............. unet.py:222-237
  * UnetLayer ................... unet.py:239-285
  * UNET ctor / forward ......... unet.py:287-356
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass
class UNetSpec:
    """Mirror of the reference constructor arguments (unet.py:288-302)."""
    Channels: List[int] = field(default_factory=lambda: [64, 64, 128, 256, 128])
    Upscales: List[bool] = field(default_factory=lambda: [False, False, True, True])
    Attentions: List[bool] = field(default_factory=lambda: [False, True, True, False])
    num_groups: int = 8
    dropout_prob: float = 0.1
    num_heads: int = 8
    time_steps: int = 1000
    input_channels: int = 1
    output_channels: int = 1
    use_cross_attention: bool = True
    context_dim: Optional[int] = None
    time_emb_dim: int = 256
    use_film: bool = False

    @property
    def num_layers(self) -> int:
        return len(self.Channels) - 1

    def has_cross(self) -> bool:
        return bool(self.use_cross_attention and self.context_dim is not None)


def sinusoid_table(time_steps: int, dim: int) -> Tensor:
    """unet.py:49-53: E[p,2i]=sin(p*w_i), E[p,2i+1]=cos(p*w_i), w_i=exp(-2i ln(1e4)/dim)."""
    p = torch.arange(0, time_steps, dtype=torch.float32)[:, None]
    w = torch.exp(torch.arange(0, dim, 2).float() * (-math.log(10000.0) / dim))
    tab = torch.zeros(time_steps, dim)
    tab[:, 0::2] = torch.sin(p * w)
    tab[:, 1::2] = torch.cos(p * w)
    return tab


def param_shapes(spec: UNetSpec) -> "Dict[str, tuple]":
    """Names and shapes of every state_dict entry, in reference registration order."""
    out: Dict[str, tuple] = {}
    temb = spec.time_emb_dim
    for i in range(spec.num_layers):
        cin, cout = spec.Channels[i], spec.Channels[i + 1]
        p = f"Layer{i + 1}."
        out[p + "res_block.conv1.weight"] = (cout, cin, 3, 3)
        out[p + "res_block.conv1.bias"] = (cout,)
        out[p + "res_block.conv2.weight"] = (cout, cout, 3, 3)
        out[p + "res_block.conv2.bias"] = (cout,)
        out[p + "res_block.norm1.weight"] = (cin,)
        out[p + "res_block.norm1.bias"] = (cin,)
        out[p + "res_block.norm2.weight"] = (cout,)
        out[p + "res_block.norm2.bias"] = (cout,)
        if cin != cout:
            out[p + "res_block.skip.weight"] = (cout, cin, 1, 1)
            out[p + "res_block.skip.bias"] = (cout,)
        if spec.use_film:   # unet.py:82-84 (FiLMLayer, unet.py:18-21)
            out[p + "res_block.film_layer.film_mlp.1.weight"] = (2 * cout, temb)
            out[p + "res_block.film_layer.film_mlp.1.bias"] = (2 * cout,)
        else:
            out[p + "res_block.time_mlp.1.weight"] = (cout, temb)
            out[p + "res_block.time_mlp.1.bias"] = (cout,)
        if spec.Attentions[i]:
            out[p + "attention.linear_qkv.weight"] = (3 * cout, cout)
            out[p + "attention.linear_out.weight"] = (cout, cout)
        if spec.has_cross():
            out[p + "cross_attention.to_q.weight"] = (cout, cout)
            out[p + "cross_attention.to_k.weight"] = (cout, spec.context_dim)
            out[p + "cross_attention.to_v.weight"] = (cout, spec.context_dim)
            out[p + "cross_attention.to_out.weight"] = (cout, cout)
        if spec.Upscales[i]:
            out[p + "upsample.conv.weight"] = (cout, cout, 3, 3)
            out[p + "upsample.conv.bias"] = (cout,)
            out[p + "conv.weight"] = (cout // 2, cout, 1, 1)
            out[p + "conv.bias"] = (cout // 2,)
        else:
            out[p + "downsample.conv.weight"] = (cout, cout, 3, 3)
            out[p + "downsample.conv.bias"] = (cout,)
    cat = spec.Channels[-1] // 2 + spec.Channels[1]
    out["input_conv.weight"] = (spec.Channels[0], spec.input_channels, 3, 3)
    out["input_conv.bias"] = (spec.Channels[0],)
    out["late_conv1.weight"] = (cat // 2, cat, 3, 3)
    out["late_conv1.bias"] = (cat // 2,)
    out["late_conv2.weight"] = (cat // 2, cat // 2, 3, 3)
    out["late_conv2.bias"] = (cat // 2,)
    out["output_conv.weight"] = (spec.output_channels, cat // 2, 1, 1)
    out["output_conv.bias"] = (spec.output_channels,)
    out["time_mlp.0.weight"] = (temb, temb)
    out["time_mlp.0.bias"] = (temb,)
    out["time_mlp.2.weight"] = (temb, temb)
    out["time_mlp.2.bias"] = (temb,)
    return out


def closed_form_fill(shapes: "Dict[str, tuple]", gain: float = 1.0) -> Dict[str, Tensor]:
    """Deterministic, RNG-free fill (reproducible on the GPU box without the reference):
    value[i] = s * sin(0.37*i*(1+0.01*j) + 1.3*j) for the j-th tensor, with s chosen per
    tensor kind so activations stay O(1)."""
    import numpy as np

    sd: Dict[str, Tensor] = {}
    for j, (name, shape) in enumerate(shapes.items()):
        n = 1
        for s in shape:
            n *= s
        i = np.arange(n, dtype=np.float64)
        base = np.sin(0.37 * i * (1.0 + 0.01 * j) + 1.3 * j)
        if re.search(r"norm\d?\.weight$", name):
            v = 1.0 + 0.1 * base
        elif name.endswith(".bias"):
            v = 0.05 * base
        else:
            fan_in = n // shape[0]
            v = gain * math.sqrt(3.0 / fan_in) * base
        sd[name] = torch.from_numpy(v.astype(np.float32)).reshape(shape).clone()
    return sd


def closed_form_weights(spec: UNetSpec, gain: float = 1.0) -> Dict[str, Tensor]:
    """closed_form_fill over the whole UNet. output_conv is deliberately NON-zero (the
    reference zero-inits it, unet.py:324-326, which makes every other gradient exactly
    0 - useless for parity)."""
    return closed_form_fill(param_shapes(spec), gain)


def det_tensor(shape, seed: float = 0.0, scale: float = 1.0) -> Tensor:
    """Closed-form pseudo-data: scale * sin(0.7311*i + 0.013*i*i mod 7 + seed)."""
    import numpy as np

    n = 1
    for s in shape:
        n *= s
    i = np.arange(n, dtype=np.float64)
    v = np.sin(0.7311 * i + np.mod(0.013 * i * i, 7.0) + seed) * scale
    return torch.from_numpy(v.astype(np.float32)).reshape(shape).clone()


def seeded_fill(shapes: "Dict[str, tuple]", seed: int, gain: float = 1.0) -> Dict[str, Tensor]:
    """Well-conditioned random fill from an explicit CPU generator (same torch build => same
    values on the GPU box): weights N(0, gain^2/fan_in), norm weights 1+0.1 N, biases 0.05 N.
    (The closed-form sin patterns above cancel heavily inside long dot products, which makes
    bf16-storage pipelines look 10x worse than they are on realistic weights.)"""
    gen = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}
    for name, shape in shapes.items():
        r = torch.randn(shape, generator=gen)
        if re.search(r"norm\d?\.weight$", name) or (name == "weight" and len(shape) == 1):
            v = 1.0 + 0.1 * r
        elif name.endswith("bias"):
            v = 0.05 * r
        else:
            n = 1
            for s in shape:
                n *= s
            v = r * (gain / math.sqrt(n // shape[0]))
        sd[name] = v
    return sd


def seeded_tensor(shape, seed: int, scale: float = 1.0) -> Tensor:
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


# --------------------------------------------------------------------------- bf16 noise model
class _RoundBoth(torch.autograd.Function):
    """Identity that rounds to bf16 in forward AND rounds the gradient in backward."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).float()


EMULATE_BF16 = False  # tests flip this to estimate the rounding noise a bf16-storage pipeline must show


def _q(x: Tensor) -> Tensor:
    """Activation materialisation point (identity in the fp32 oracle)."""
    return _RoundBoth.apply(x) if EMULATE_BF16 else x


def _w(x: Tensor) -> Tensor:
    """Weight use point: bf16 shadow of the fp32 master (forward value only; gradient stays fp32)."""
    return x + (x.to(torch.bfloat16).float() - x).detach() if EMULATE_BF16 else x


# --------------------------------------------------------------------------- blocks
def _gn_silu(x: Tensor, w: Tensor, b: Tensor, groups: int) -> Tensor:
    return _q(F.silu(F.group_norm(x, groups, w, b, eps=1e-5)))


def res_block(sd, p: str, x: Tensor, temb: Tensor, groups: int, drop_p: float = 0.0,
              train: bool = False) -> Tensor:
    """unet.py:92-108; the FiLM branch (unet.py:96-98, FiLMLayer.forward unet.py:23-42) is taken when the state dict
    holds a film_layer."""
    h = F.conv2d(_gn_silu(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], groups),
                 _w(sd[p + "conv1.weight"]), sd[p + "conv1.bias"], padding=1)
    if (p + "film_layer.film_mlp.1.weight") in sd:
        gb = F.linear(F.silu(temb), _w(sd[p + "film_layer.film_mlp.1.weight"]), sd[p + "film_layer.film_mlp.1.bias"])
        gamma, beta = gb.chunk(2, dim=1)
        h = gamma[:, :, None, None] * _q(h) + beta[:, :, None, None]
    else:
        tp = F.linear(F.silu(temb), _w(sd[p + "time_mlp.1.weight"]), sd[p + "time_mlp.1.bias"])
        h = h + tp[:, :, None, None]
    h = _q(F.dropout(h, drop_p, training=train))
    h = F.conv2d(_gn_silu(h, sd[p + "norm2.weight"], sd[p + "norm2.bias"], groups),
                 _w(sd[p + "conv2.weight"]), sd[p + "conv2.bias"], padding=1)
    if (p + "skip.weight") in sd:
        x = _q(F.conv2d(x, _w(sd[p + "skip.weight"]), sd[p + "skip.bias"]))
    return _q(x + h)


def softmax_attention(q: Tensor, k: Tensor, v: Tensor, drop_p: float = 0.0,
                      train: bool = False) -> Tensor:
    """[B,H,N,d] x [B,H,L,d] -> [B,H,N,d]; explicit softmax(QK^T/sqrt(d))V
    (the SDPA semantics called at unet.py:123/126 and :169/172)."""
    s = torch.matmul(q, k.transpose(-1, -2)) * (1.0 / math.sqrt(q.shape[-1]))
    pr = torch.softmax(s, dim=-1)
    pr = F.dropout(pr, drop_p, training=train)
    return torch.matmul(pr, v)


def self_attention(sd, p: str, x: Tensor, heads: int, drop_p=0.0, train=False) -> Tensor:
    """unet.py:128-139."""
    B, C, H, W = x.shape
    tok = x.permute(0, 2, 3, 1).reshape(B, H * W, C)
    qkv = _q(F.linear(tok, _w(sd[p + "linear_qkv.weight"]))).reshape(B, H * W, 3, heads, C // heads)
    q, k, v = (qkv[:, :, j].permute(0, 2, 1, 3) for j in range(3))
    o = _q(softmax_attention(q, k, v, drop_p, train)).permute(0, 2, 1, 3).reshape(B, H * W, C)
    o = _q(F.linear(o, _w(sd[p + "linear_out.weight"])))
    return o.reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()


def cross_attention(sd, p: str, x: Tensor, ctx: Tensor, heads: int, drop_p=0.0,
                    train=False) -> Tensor:
    """unet.py:189-220; ctx is [B,D] (-> one token) or [B,L,D]."""
    B, C, H, W = x.shape
    tok = x.permute(0, 2, 3, 1).reshape(B, H * W, C)
    if ctx.dim() == 2:
        ctx = ctx[:, None, :]
    L = ctx.shape[1]
    d = C // heads
    ctx = _q(ctx)
    q = _q(F.linear(tok, _w(sd[p + "to_q.weight"]))).reshape(B, H * W, heads, d).permute(0, 2, 1, 3)
    k = _q(F.linear(ctx, _w(sd[p + "to_k.weight"]))).reshape(B, L, heads, d).permute(0, 2, 1, 3)
    v = _q(F.linear(ctx, _w(sd[p + "to_v.weight"]))).reshape(B, L, heads, d).permute(0, 2, 1, 3)
    o = _q(softmax_attention(q, k, v, drop_p, train)).permute(0, 2, 1, 3).reshape(B, H * W, C)
    o = _q(F.linear(o, _w(sd[p + "to_out.weight"])))
    return o.reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()


def unet_layer(sd, spec: UNetSpec, i: int, x: Tensor, temb: Tensor, ctx, train=False):
    """unet.py:265-285. Returns (x_out, skip)."""
    p = f"Layer{i + 1}."
    dp = spec.dropout_prob if train else 0.0
    x = res_block(sd, p + "res_block.", x, temb, spec.num_groups, dp, train)
    res = x
    if spec.Attentions[i]:
        x = self_attention(sd, p + "attention.", x, spec.num_heads, dp, train)
    if spec.has_cross() and ctx is not None:
        x = cross_attention(sd, p + "cross_attention.", x, ctx, spec.num_heads, dp, train)
    x = _q(x + res)
    if spec.Upscales[i]:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        x = _q(F.conv2d(x, _w(sd[p + "upsample.conv.weight"]), sd[p + "upsample.conv.bias"], padding=1))
        x = _q(F.conv2d(x, _w(sd[p + "conv.weight"]), sd[p + "conv.bias"]))
        return x, x
    skip = x
    x = _q(F.conv2d(x, _w(sd[p + "downsample.conv.weight"]), sd[p + "downsample.conv.bias"],
                    stride=2, padding=1))
    return x, skip


def time_embedding(sd, spec: UNetSpec, t) -> Tensor:
    """unet.py:59-66 + 339-340."""
    tab = sinusoid_table(spec.time_steps, spec.time_emb_dim)
    t = torch.as_tensor(t, dtype=torch.long).reshape(-1)
    e = _q(tab[t])
    e = _q(F.linear(e, _w(sd["time_mlp.0.weight"]), sd["time_mlp.0.bias"]))
    return _q(F.linear(F.silu(e), _w(sd["time_mlp.2.weight"]), sd["time_mlp.2.bias"]))


def unet_forward(sd: Dict[str, Tensor], spec: UNetSpec, x: Tensor, t, context=None,
                 train: bool = False) -> Tensor:
    """unet.py:337-356."""
    temb = time_embedding(sd, spec, t)
    if context is not None and not torch.is_floating_point(context):
        # legacy integer labels: only meaningful when no cross-attention exists (unet.py:271)
        if spec.has_cross():
            raise TypeError("integer context with cross-attention enabled")
        context = None
    skips = []
    x = _q(F.conv2d(_q(x), _w(sd["input_conv.weight"]), sd["input_conv.bias"], padding=1))
    half = spec.num_layers // 2
    for i in range(half):
        x, r = unet_layer(sd, spec, i, x, temb, context, train)
        skips.append(r)
    for i in range(half, spec.num_layers):
        y, _ = unet_layer(sd, spec, i, x, temb, context, train)
        x = torch.cat((y, skips.pop()), dim=1)
    x = _q(F.silu(_q(F.conv2d(x, _w(sd["late_conv1.weight"]), sd["late_conv1.bias"], padding=1))))
    x = _q(F.silu(_q(F.conv2d(x, _w(sd["late_conv2.weight"]), sd["late_conv2.bias"], padding=1))))
    return F.conv2d(x, _w(sd["output_conv.weight"]), sd["output_conv.bias"])
