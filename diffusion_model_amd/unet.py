"""Drop-in for the reference's ``unet.py``: same class names, constructor signatures,
``forward`` signatures and ``state_dict`` keys/shapes/order (reference unet.py:44-356), so a
reference checkpoint loads with ``strict=True`` -- but every op runs in the gfx950 HIP kernels
of libdm_hip.so (NHWC bf16 activations, fp32 master weights + packed bf16 shadows, fp32
accumulation).  Forward AND backward are scheduled by hand (no autograd graph inside the
network); one ``torch.autograd.Function`` per public module keeps ``loss.backward()`` working
for callers (main_stl10.py:502-503).

Parameter containers are plain ``nn.Conv2d`` / ``nn.Linear`` / ``nn.GroupNorm`` objects created
in the reference's registration order, so ``torch.manual_seed(s); UNET(...)`` yields the
reference's initial weights bit-for-bit; their ``forward`` is never called.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import kernels as K

BF16 = torch.bfloat16
device = torch.device("cuda" if torch.cuda.is_available() else "cpu")

_EPOCH = [0]


class DropoutSeeds:
    """Iterator of dropout seeds (one per dropout site and forward).  The sequence is a pure function of ``base``:
    ``reseed`` derives the base from the process seed, the data-parallel rank and the optimizer step, so ranks draw
    different masks, a resumed run does not replay the masks of step 0, and ``torch.manual_seed`` controls it.  Tests
    pin a sequence by assigning ``unet._SEED = itertools.count(k)`` (any iterator of ints works)."""

    def __init__(self, base: int = 0x5EED):
        self.base, self.n = int(base), 0

    def __iter__(self):
        return self

    def __next__(self) -> int:
        self.n += 1
        return (self.base + self.n * 0x9E3779B97F4A7C15) & 0x7FFFFFFFFFFFFFFF

    def reseed(self, seed: int, rank: int = 0, step: int = 0) -> None:
        x = (int(seed) * 0xBF58476D1CE4E5B9 + (int(rank) + 1) * 0x94D049BB133111EB + int(step) * 0xD6E8FEB86659FD93)
        x &= 0xFFFFFFFFFFFFFFFF
        x ^= x >> 31
        self.base, self.n = x & 0x7FFFFFFFFFFFFFFF, 0


_SEED = DropoutSeeds()


def reseed_dropout(seed: Optional[int] = None, rank: int = 0, step: int = 0) -> None:
    """(Re)start the dropout seed sequence from (seed or torch.initial_seed(), rank, step)."""
    global _SEED
    if not isinstance(_SEED, DropoutSeeds):
        _SEED = DropoutSeeds()
    _SEED.reseed(torch.initial_seed() if seed is None else seed, rank, step)


def invalidate_packed_weights() -> None:
    """Call after updating parameters with raw kernels (no torch version bump)."""
    _EPOCH[0] += 1


class PW:
    """Packed bf16 shadows (forward operand, data-gradient operand) of one conv / linear."""

    def __init__(self, c: nn.Module):
        w = c.weight
        self.c = c
        self.taps = 9 if (w.dim() == 4 and w.shape[-1] == 3) else 1
        self.n, self.k = w.shape[0], w.shape[1]
        self._fwd = self._bwd = None
        self._kf = self._kb = None
        self._fwd32 = self._bwd32 = None     # fp32 compute mode: fp32 repacks (kernels.pack_weight(dtype=float32)),
        self._kf32 = self._kb32 = None       # kept apart from the bf16 shadows that repack_all() rebuilds in one launch

    def _key(self):
        w = self.c.weight
        return (w.data_ptr(), w._version, _EPOCH[0])

    @staticmethod
    def of(c: nn.Module) -> "PW":
        pw = c.__dict__.get("_dm_pw")
        if pw is None:
            pw = PW(c)
            c.__dict__["_dm_pw"] = pw
        return pw

    @property
    def fwd(self):
        key = self._key()
        if K.ACT[0] == torch.float32:
            if self._kf32 != key:
                self._fwd32 = K.pack_weight(self.c.weight.detach(), False, self._fwd32, dtype=torch.float32)
                self._kf32 = key
            return self._fwd32
        if self._kf != key:
            self._fwd = K.pack_weight(self.c.weight.detach(), False, self._fwd)
            self._kf = key
        return self._fwd

    @property
    def bwd(self):
        key = self._key()
        if K.ACT[0] == torch.float32:
            if self._kb32 != key:
                self._bwd32 = K.pack_weight(self.c.weight.detach(), True, self._bwd32, dtype=torch.float32)
                self._kb32 = key
            return self._bwd32
        if self._kb != key:
            self._bwd = K.pack_weight(self.c.weight.detach(), True, self._bwd)
            self._kb = key
        return self._bwd

    def bias_pad(self, npad: int):
        """The bias zero-padded to ``npad`` entries (the output conv's 16-wide channel tile), rebuilt only when the bias
        changed (same staleness rule as the packed weights)."""
        b = self.c.bias
        if b is None:
            return None
        key = (b.data_ptr(), b._version, _EPOCH[0], npad)
        if getattr(self, "_kbias", None) != key:
            self._bias = _pad_bias(self.c, npad)
            self._kbias = key
        return self._bias


def repack_all(model: nn.Module, force: bool = False) -> None:
    """Rebuild every packed shadow the model has used so far in ONE launch (dm_pack_weights_batch) and mark
    them current.  Called by the trainer right after the fused optimizer step; shadows that do not exist yet
    are still packed lazily on first use.  ``force``: (whole-step graph capture) bump the staleness epoch first, so the
    marks written here are the current ones."""
    if force:
        invalidate_packed_weights()
    entries = []
    for m in model.modules():
        pw = m.__dict__.get("_dm_pw")
        if pw is None:
            continue
        if pw._fwd is not None:
            entries.append((pw, False, pw._fwd))
        if pw._bwd is not None:
            entries.append((pw, True, pw._bwd))
    if not entries:
        return
    sig = tuple((pw.c.weight.data_ptr(), buf.data_ptr(), tr) for pw, tr, buf in entries)
    cached = model.__dict__.get("_dm_pack_table")
    if cached is None or cached[0] != sig:
        import numpy as np
        dt = np.dtype([("w", "<u8"), ("dst", "<u8"), ("N", "<i4"), ("K", "<i4"), ("taps", "<i4"), ("tr", "<i4"),
                       ("bb", "<i4"), ("nb", "<i4")])
        tab = np.zeros(len(entries), dtype=dt)
        blk = 0
        for i, (pw, tr, buf) in enumerate(entries):
            N, Kd = (pw.k, pw.n) if tr else (pw.n, pw.k)
            total = ((N + 15) // 16 * 16) * ((Kd + 63) // 64 * 64) * pw.taps
            nb = min((total + 255) // 256, 1024)
            tab[i] = (pw.c.weight.data_ptr(), buf.data_ptr(), N, Kd, pw.taps, int(tr), blk, nb)
            blk += nb
        dev = entries[0][2].device
        table = torch.from_numpy(tab.view(np.uint8).copy()).to(dev)
        cached = (sig, table, len(entries), blk)
        model.__dict__["_dm_pack_table"] = cached
    K.pack_weights_batch(cached[1], cached[2], cached[3])
    for pw, tr, _ in entries:
        if tr:
            pw._kb = pw._key()
        else:
            pw._kf = pw._key()


class Grads:
    """Parameter -> fp32 gradient buffer that the wgrad kernels accumulate (+=) into."""

    def __init__(self, views: Optional[Dict[int, torch.Tensor]] = None):
        self.views = views
        self.bufs: Dict[int, torch.Tensor] = {}
        # gradient wrt the cross-attention context (fp32 [B, L, Dc]); None = not asked for.  Every CrossAttention of the
        # network adds its share (UNET._bwd with need_dctx=True)
        self.dctx: Optional[torch.Tensor] = None

    def of(self, p: Optional[torch.Tensor]):
        if p is None:
            return None
        if self.views is not None:
            return self.views[id(p)]
        g = self.bufs.get(id(p))
        if g is None:
            g = torch.zeros(p.shape, dtype=torch.float32, device=p.device)
            self.bufs[id(p)] = g
        return g

    def peek(self, p):
        return self.bufs.get(id(p)) if self.views is None else self.views.get(id(p))


def _b(c):
    return None if c.bias is None else c.bias.detach()


_ZEROS_RO: Dict[tuple, torch.Tensor] = {}


def _zeros_ro(shape, dtype, device) -> torch.Tensor:
    """A cached all-zeros tensor that callers only READ (never an output, never written in place)."""
    if torch.cuda.is_current_stream_capturing():
        return torch.zeros(shape, dtype=dtype, device=device)   # (a first use inside a capture stays in the graph's pool)
    key = (tuple(shape), dtype, device.index if device.type == "cuda" else -1)
    z = _ZEROS_RO.get(key)
    if z is None:
        z = _ZEROS_RO[key] = torch.zeros(shape, dtype=dtype, device=device)
        torch.cuda.current_stream(device).synchronize()   # once per shape: readers on ANY stream may follow without an event
    return z


def _nhwc_in(x: torch.Tensor, cpad: int = 0) -> torch.Tensor:
    return K.nchw_to_nhwc(x.detach().float().contiguous(), cpad)


class _Fn(torch.autograd.Function):
    """Generic adapter: runs a module's hand-written forward/backward under autograd."""

    @staticmethod
    def forward(ctx, mod, n_in, *args):
        ins = args[:n_in]
        out, saved = mod._run_fwd(*ins)
        ctx.mod, ctx.saved, ctx.n_in = mod, saved, n_in
        ctx.params = args[n_in:]
        ctx.need_in = [torch.is_tensor(a) and a.requires_grad for a in ins]
        return out

    @staticmethod
    def backward(ctx, dout):
        g = Grads()
        dins = ctx.mod._run_bwd(dout.contiguous(), ctx.saved, g, ctx.need_in)
        K.join_aux(dout.device)   # overlapped attention backward branches (UnetLayer._bwd)
        K.join_side(dout.device)  # weight gradients were enqueued on the side stream
        dins = list(dins) + [None] * (ctx.n_in - len(dins))
        pg = []
        for p in ctx.params:
            t = g.peek(p)
            pg.append(t if (t is not None and p.requires_grad) else None)
        return (None, None, *dins, *pg)


class HipModule(nn.Module):
    def _apply_fn(self, *ins):
        if not ins[0].is_cuda:
            raise RuntimeError("diffusion_model_amd runs only on an AMD GPU (HIP kernels); there is no CPU fallback")
        return _Fn.apply(self, len(ins), *ins, *list(self.parameters()))


# ============================================================================ building blocks
class SinusoidalEmbeddings(nn.Module):
    """Reference unet.py:44-66 (table built with the same torch CPU ops -> bit-identical)."""

    def __init__(self, time_steps: int, embed_dim: int):
        super().__init__()
        pos = torch.arange(0, time_steps, dtype=torch.float).unsqueeze(1)
        div = torch.exp(torch.arange(0, embed_dim, 2).float() * (-math.log(10000.0) / embed_dim))
        emb = torch.zeros(time_steps, embed_dim, requires_grad=False)
        emb[:, 0::2] = torch.sin(pos * div)
        emb[:, 1::2] = torch.cos(pos * div)
        self.register_buffer("embeddings", emb, persistent=False)

    @property
    def embedings(self):  # reference keeps this misspelt alias
        return self.embeddings

    def forward(self, t):
        if not torch.is_tensor(t):
            t = torch.as_tensor(t, dtype=torch.long, device=self.embeddings.device)
        else:
            t = t.to(self.embeddings.device).long()
        return self.embeddings[t]


# GroupNorm + SiLU folded into the consuming convolution (dm_gn_prepare + dm_conv_args.norm_scale/shift; bit-identical
# values: the conv prologue evaluates the very expression the apply kernel evaluates).  OFF by default -- measured
# (tools/normfold_bench.py, B=64, profiles/r02_normfold.txt): the forward pair gains (L1: 109.7 -> 102.7 us, no
# normalised tensor written) but every consumer re-evaluates exp + rcp for its halo and once per 64-cout block, which
# costs the conv +12 us and the weight gradient +19 us at that shape: a net loss for training.  DM_FOLD_GROUPNORM=1
# turns it on (less activation memory, one pass over x in the forward).
FOLD_GROUPNORM = [os.environ.get("DM_FOLD_GROUPNORM", "0") == "1"]


class FiLMLayer(HipModule):
    """Reference unet.py:9-42: features * gamma + beta with [gamma | beta] = Linear(SiLU(conditioning))."""

    def __init__(self, conditioning_dim: int, feature_dim: int):
        super().__init__()
        self.film_mlp = nn.Sequential(nn.SiLU(), nn.Linear(conditioning_dim, feature_dim * 2))
        self.feature_dim = feature_dim

    # cond: bf16 [B, conditioning_dim] (pre-SiLU); h0: NHWC bf16
    def _fwd(self, h0, cond, drop_p=0.0, drop_seed=0):
        lin = self.film_mlp[1]
        gb = torch.empty((h0.shape[0], 2 * self.feature_dim), dtype=torch.float32, device=h0.device)
        K.gemm(cond, PW.of(lin).fwd, 2 * self.feature_dim, bias=_b(lin), act_silu_in=True, out_f32=gb)
        return K.film_fwd(h0, gb, drop_p, drop_seed), (h0, cond, gb)

    def _bwd(self, dh, dbeta, saved, g: Grads):
        """dh: gradient wrt gamma*h0+beta (dropout mask already applied); dbeta: its per-(sample, channel) sums, fp32
        [B, C] (None: computed here).  Returns (dh0, gradient wrt silu(cond) as fp32 [B, conditioning_dim])."""
        h0, cond, gb = saved
        lin = self.film_mlp[1]
        B, Cf = h0.shape[0], self.feature_dim
        if dbeta is None:
            dbeta = torch.empty((B, Cf), dtype=torch.float32, device=h0.device)
            K.colsum(dh.view(-1, Cf), dbeta, B)
        dh0, dgamma = K.film_bwd(dh, h0, gb)
        dgb = torch.cat([dgamma, dbeta], dim=1)
        dgb16 = K.batchsum2_cast(dgb, g.of(lin.bias))
        K.wgrad_side(cond, dgb16, g.of(lin.weight), None, taps=1, act_silu_in=True)
        dcond = torch.empty((B, cond.shape[-1]), dtype=torch.float32, device=h0.device)
        K.gemm(dgb16, PW.of(lin).bwd, cond.shape[-1], out_f32=dcond)
        return dh0, dcond

    def forward(self, features, conditioning):
        return self._apply_fn(features, conditioning)

    def _run_fwd(self, features, conditioning):
        y, saved = self._fwd(_nhwc_in(features), K.to_bf16(conditioning.detach().float().contiguous()))
        return K.nhwc_to_nchw(y), saved

    def _run_bwd(self, dout, saved, g, need):
        dh0, dcond = self._bwd(_nhwc_in(dout), None, saved, g)
        return K.nhwc_to_nchw(dh0), K.to_f32(K.silu_bwd(K.to_bf16(dcond), saved[1]))


class ResBlock(HipModule):
    """Reference unet.py:70-108."""

    def __init__(self, in_channels: int, out_channels: int, time_emb_dim: int, num_groups: int,
                 dropout_prob: float, use_film: bool = False):
        super().__init__()
        self.relu = nn.SiLU()
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1)
        self.dropout = nn.Dropout(dropout_prob)
        self.norm1 = nn.GroupNorm(num_groups, in_channels)
        self.norm2 = nn.GroupNorm(num_groups, out_channels)
        self.skip = nn.Conv2d(in_channels, out_channels, kernel_size=1) if in_channels != out_channels else nn.Identity()
        self.use_film = use_film
        if use_film:   # unet.py:82-84
            self.film_layer = FiLMLayer(time_emb_dim, out_channels)
        else:          # unet.py:85-90
            self.time_mlp = nn.Sequential(nn.SiLU(), nn.Linear(time_emb_dim, out_channels))
        self.cin, self.cout, self.groups, self.p = in_channels, out_channels, num_groups, dropout_prob

    # temb: bf16 [B, time_emb_dim] (pre-SiLU)
    def _fwd(self, x, temb, out=None):
        if self.use_film:
            return self._fwd_film(x, temb, out)
        lin = self.time_mlp[1]
        B = x.shape[0]
        tp = torch.empty((B, self.cout), dtype=torch.float32, device=x.device)
        K.gemm(temb, PW.of(lin).fwd, self.cout, bias=_b(lin), act_silu_in=True, out_f32=tp)
        p = self.p if self.training else 0.0
        seed = next(_SEED) if p > 0 else 0
        # GroupNorm + SiLU are folded into the consuming conv (FOLD_GROUPNORM): one statistics pass over the input,
        # then the conv normalises while staging its patch; the normalised tensor is never written.  (8x8 images of
        # a batch >= 4 keep the materialised form: their four-images-per-workgroup conv tile has no folded variant.)
        fold = FOLD_GROUPNORM[0] and x.dtype == BF16 and not (x.shape[1] == 8 and x.shape[2] == 8 and B >= 4)
        if fold:
            st1, n1 = K.gn_prepare(x, self.norm1.weight.detach(), self.norm1.bias.detach(), self.groups)
            h = K.conv3x3(x, PW.of(self.conv1).fwd, self.cout, bias=_b(self.conv1), chan_add=tp, drop_p=p,
                          drop_seed=seed, norm_in=n1)
            st2, n2 = K.gn_prepare(h, self.norm2.weight.detach(), self.norm2.bias.detach(), self.groups)
            g1 = g2 = None
        else:
            n1 = n2 = None
            g1, st1 = K.gn_silu_fwd(x, self.norm1.weight.detach(), self.norm1.bias.detach(), self.groups)
            h = K.conv3x3(g1, PW.of(self.conv1).fwd, self.cout, bias=_b(self.conv1), chan_add=tp, drop_p=p, drop_seed=seed)
            g2, st2 = K.gn_silu_fwd(h, self.norm2.weight.detach(), self.norm2.bias.detach(), self.groups)
        if isinstance(self.skip, nn.Identity):
            sk = x
        else:
            sk = K.gemm(x, PW.of(self.skip).fwd, self.cout, bias=_b(self.skip))
        if fold:
            y = K.conv3x3(h, PW.of(self.conv2).fwd, self.cout, bias=_b(self.conv2), res=sk, out=out, norm_in=n2)
        else:
            y = K.conv3x3(g2, PW.of(self.conv2).fwd, self.cout, bias=_b(self.conv2), res=sk, out=out)
        return y, (x, temb, (g1, n1), st1, h, (g2, n2), st2, p, seed, None)

    def _fwd_film(self, x, temb, out=None):
        """use_film=True (unet.py:96-98): h = dropout(gamma * conv1(..) + beta) instead of conv1(..) + time projection."""
        p = self.p if self.training else 0.0
        seed = next(_SEED) if p > 0 else 0
        g1, st1 = K.gn_silu_fwd(x, self.norm1.weight.detach(), self.norm1.bias.detach(), self.groups)
        h0 = K.conv3x3(g1, PW.of(self.conv1).fwd, self.cout, bias=_b(self.conv1))
        h, s_film = self.film_layer._fwd(h0, temb, p, seed)
        g2, st2 = K.gn_silu_fwd(h, self.norm2.weight.detach(), self.norm2.bias.detach(), self.groups)
        sk = x if isinstance(self.skip, nn.Identity) else K.gemm(x, PW.of(self.skip).fwd, self.cout, bias=_b(self.skip))
        y = K.conv3x3(g2, PW.of(self.conv2).fwd, self.cout, bias=_b(self.conv2), res=sk, out=out)
        return y, (x, temb, (g1, None), st1, h, (g2, None), st2, p, seed, s_film)

    def _bwd(self, dy, saved, g: Grads, dts=None, need_dx=True):
        """Returns (dx, dts): dts = accumulated gradient wrt silu(temb), bf16 [B, temb_dim]."""
        x, temb, (g1, n1), st1, h, (g2, n2), st2, p, seed, s_film = saved
        B = x.shape[0]
        lin = None if self.use_film else self.time_mlp[1]
        d_g2 = K.conv3x3(dy, PW.of(self.conv2).bwd, self.cout)
        if g2 is None:   # folded GroupNorm: the weight gradient re-normalises h while staging, as the forward conv did
            K.wgrad_side(h, dy, g.of(self.conv2.weight), g.of(self.conv2.bias), norm_in=n2)
        else:
            K.wgrad_side(g2, dy, g.of(self.conv2.weight), g.of(self.conv2.bias))
        if isinstance(self.skip, nn.Identity):
            dsk = dy
        else:
            dsk = K.gemm(dy, PW.of(self.skip).bwd, self.cin)
            K.wgrad_side(x, dy, g.of(self.skip.weight), g.of(self.skip.bias), taps=1)
        dtp = torch.zeros((B, self.cout), dtype=torch.float32, device=x.device)
        dh = K.gn_silu_bwd(h, d_g2, self.norm2.weight.detach(), self.norm2.bias.detach(), st2, self.groups,
                           g.of(self.norm2.weight), g.of(self.norm2.bias), drop_p=p, drop_seed=seed, colsum_out=dtp)
        if self.use_film:   # dh is the gradient of the FiLM output; dtp = its per-(sample, channel) sums = dbeta
            dh, d_this = self.film_layer._bwd(dh, dtp, s_film, g)
            conv1_bias_grad = g.of(self.conv1.bias)
        else:
            dtp16 = K.batchsum2_cast(dtp, g.of(self.conv1.bias), g.of(lin.bias))   # both bias gradients + the bf16 operand
            K.wgrad_side(temb, dtp16, g.of(lin.weight), None, taps=1, act_silu_in=True)
            d_this = torch.empty((B, temb.shape[-1]), dtype=torch.float32, device=x.device)
            K.gemm(dtp16, PW.of(lin).bwd, temb.shape[-1], out_f32=d_this)
            conv1_bias_grad = None
        if dts is None:
            dts = d_this
        else:  # fp32 accumulation across the ResBlocks (batchsum with one "batch" row == axpy)
            K.batchsum(d_this.view(1, -1), dts.view(-1))
        if g1 is None:
            K.wgrad_side(x, dh, g.of(self.conv1.weight), conv1_bias_grad, norm_in=n1)
        else:
            K.wgrad_side(g1, dh, g.of(self.conv1.weight), conv1_bias_grad)
        dx = None
        if need_dx:
            d_g1 = K.conv3x3(dh, PW.of(self.conv1).bwd, self.cin)
            dx = K.gn_silu_bwd(x, d_g1, self.norm1.weight.detach(), self.norm1.bias.detach(), st1, self.groups,
                               g.of(self.norm1.weight), g.of(self.norm1.bias), add_in=dsk)
        else:  # norm1 affine grads are still needed
            d_g1 = K.conv3x3(dh, PW.of(self.conv1).bwd, self.cin)
            K.gn_silu_bwd(x, d_g1, self.norm1.weight.detach(), self.norm1.bias.detach(), st1, self.groups,
                          g.of(self.norm1.weight), g.of(self.norm1.bias), add_in=dsk)
        return dx, dts

    # ---- public NCHW fp32 interface
    def forward(self, x, time_emb):
        return self._apply_fn(x, time_emb)

    def _run_fwd(self, x, time_emb):
        y, saved = self._fwd(_nhwc_in(x), K.to_bf16(time_emb.detach().float().contiguous()))
        return K.nhwc_to_nchw(y), saved

    def _run_bwd(self, dout, saved, g, need):
        temb = saved[1]
        dx, dts = self._bwd(_nhwc_in(dout), saved, g)
        dtemb = K.silu_bwd(K.to_bf16(dts), temb)
        return K.nhwc_to_nchw(dx), K.to_f32(dtemb)


class Attention(HipModule):
    """Reference unet.py:110-139 (self-attention over H*W tokens; no norm, no residual inside)."""

    def __init__(self, C: int, num_heads: int, dropout_prob: float, use_flash_attention: bool = True):
        super().__init__()
        self.linear_qkv = nn.Linear(C, C * 3, bias=False)
        self.linear_out = nn.Linear(C, C, bias=False)
        self.dropout_prob = dropout_prob
        self.num_heads = num_heads
        self.use_flash_attention = use_flash_attention
        self.C = C

    def _fwd(self, x, res=None, out=None, seed=None):
        B, H, W, C = x.shape
        tok = x.view(B, H * W, C)
        qkv = K.gemm(tok, PW.of(self.linear_qkv).fwd, 3 * C)
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        p = self.dropout_prob if self.training else 0.0
        if seed is None:
            seed = next(_SEED) if p > 0 else 0
        o, lse = K.attn_fwd(q, k, v, self.num_heads, drop_p=p, drop_seed=seed)
        y = K.gemm(o, PW.of(self.linear_out).fwd, C, res=None if res is None else res.view(B, H * W, C),
                   out=None if out is None else out.view(B, H * W, C))
        return y.view(B, H, W, C), (tok, qkv, o, lse, p, seed)

    def _bwd(self, dy, saved, g: Grads, add=None):
        tok, qkv, o, lse, p, seed = saved
        B, N, C = tok.shape
        dyt = dy.view(B, N, C)
        d_o = K.gemm(dyt, PW.of(self.linear_out).bwd, C)
        K.wgrad_side(o, dyt, g.of(self.linear_out.weight), None, taps=1)
        dqkv = torch.empty_like(qkv)
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        K.attn_bwd(q, k, v, o, lse, d_o, self.num_heads, dq=dqkv[..., :C], dk=dqkv[..., C:2 * C],
                   dv=dqkv[..., 2 * C:], drop_p=p, drop_seed=seed)
        K.wgrad_side(tok, dqkv, g.of(self.linear_qkv.weight), None, taps=1)
        dx = K.gemm(dqkv, PW.of(self.linear_qkv).bwd, C, res=None if add is None else add.view(B, N, C))
        return dx.view(dy.shape)

    def forward(self, x):
        return self._apply_fn(x)

    def _run_fwd(self, x):
        y, saved = self._fwd(_nhwc_in(x))
        return K.nhwc_to_nchw(y), saved

    def _run_bwd(self, dout, saved, g, need):
        return (K.nhwc_to_nchw(self._bwd(_nhwc_in(dout), saved, g)),)


class CrossAttention(HipModule):
    """Reference unet.py:141-220 (queries from pixels, keys/values from the context tokens)."""

    def __init__(self, query_dim: int, context_dim: int, num_heads: int = 8, dropout_prob: float = 0.0,
                 use_flash_attention: bool = True):
        super().__init__()
        assert query_dim % num_heads == 0, "query_dim must be divisible by num_heads"
        self.query_dim, self.context_dim, self.num_heads = query_dim, context_dim, num_heads
        self.dropout_prob, self.use_flash_attention = dropout_prob, use_flash_attention
        self.to_q = nn.Linear(query_dim, query_dim, bias=False)
        self.to_k = nn.Linear(context_dim, query_dim, bias=False)
        self.to_v = nn.Linear(context_dim, query_dim, bias=False)
        self.to_out = nn.Linear(query_dim, query_dim, bias=False)

    def _onekey_out(self, vc, B, N, C, res, out, p, seed):
        """y = res + to_out(dropout(vc)) for a ONE-token context (the attention output does not depend on the
        queries: softmax over one key).  Fused (dm_cross1_fwd: one pass res -> y, the attention output never reaches
        memory; returns o = None) or, with FUSE_CROSS1 off / an unsupported width, attn_onekey_fwd + the to_out GEMM."""
        res = None if res is None else res.view(B, N, C)
        out = None if out is None else out.view(B, N, C)
        if FUSE_CROSS1[0] and K.cross1_ok(self.num_heads, C):
            return K.cross1_fwd(vc, self.to_out.weight.detach(), N, self.num_heads, res=res, out=out, drop_p=p,
                                drop_seed=seed), None
        o = K.attn_onekey_fwd(vc, N, self.num_heads, drop_p=p, drop_seed=seed)
        return K.gemm(o, PW.of(self.to_out).fwd, C, res=res, out=out), o

    # ctx: bf16 [B, L, Dc]
    def _fwd(self, x, ctx, res=None, out=None, elide=False):
        B, H, W, C = x.shape
        N = H * W
        if elide:  # one context token: the output does not depend on x (see ELIDE_DEAD_ATTENTION)
            vc = K.gemm(ctx, PW.of(self.to_v).fwd, C)
            p = self.dropout_prob if self.training else 0.0
            seed = next(_SEED) if p > 0 else 0
            y, o = self._onekey_out(vc, B, N, C, res, out, p, seed)
            return y.view(B, H, W, C), (None, ctx, None, None, vc, o, None, p, seed)
        tok = x.view(B, N, C)
        q = K.gemm(tok, PW.of(self.to_q).fwd, C)
        kc = K.gemm(ctx, PW.of(self.to_k).fwd, C)
        vc = K.gemm(ctx, PW.of(self.to_v).fwd, C)
        p = self.dropout_prob if self.training else 0.0
        seed = next(_SEED) if p > 0 else 0
        one = ctx.shape[1] == 1 and FUSE_CROSS1[0] and K.cross1_ok(self.num_heads, C) and x.dtype == BF16
        o, lse = K.attn_fwd(q, kc, vc, self.num_heads, drop_p=p, drop_seed=seed, want_o=not one)
        if one:
            # one token: same output path as the two-stream schedule (UnetLayer._fwd_attn_overlapped), so the two
            # schedules stay bit-identical; q / k only feed the log-sum-exp
            y, o = self._onekey_out(vc, B, N, C, res, out, p, seed)
            return y.view(B, H, W, C), (tok, ctx, q, kc, vc, o, lse, p, seed)
        y = K.gemm(o, PW.of(self.to_out).fwd, C, res=None if res is None else res.view(B, N, C),
                   out=None if out is None else out.view(B, N, C))
        return y.view(B, H, W, C), (tok, ctx, q, kc, vc, o, lse, p, seed)

    def _bwd(self, dy, saved, g: Grads, add=None, need_dctx=False):
        tok, ctx, q, kc, vc, o, lse, p, seed = saved
        B, H, W, C = dy.shape
        N = H * W
        dyt = dy.view(B, N, C)
        fused = o is None     # _onekey_out took dm_cross1_fwd: one pass over dy gives dv and the to_out weight gradient
        if fused:
            dv = K.cross1_bwd(dyt, vc, self.to_out.weight.detach(), self.num_heads, dw_out=g.of(self.to_out.weight),
                              drop_p=p, drop_seed=seed)
        else:
            d_o = K.gemm(dyt, PW.of(self.to_out).bwd, C)
            K.wgrad_side(o, dyt, g.of(self.to_out.weight), None, taps=1)
        if tok is None:  # elided form: dq = dk = 0 exactly, so to_q / to_k get no gradient and dx is just `add`
            if not fused:
                dv = K.attn_onekey_bwd(vc, o, d_o, self.num_heads, drop_p=p, drop_seed=seed)
            K.wgrad_side(ctx, dv, g.of(self.to_v.weight), None, taps=1)
            g.of(self.to_q.weight), g.of(self.to_k.weight)  # (allocate the zero gradients for autograd callers)
            if g.dctx is not None:   # dk = 0: the context gradient comes through to_v alone
                self._add_dctx(g, None, dv)
            return add, None
        if fused:   # one key: dq and dk are exact zeros (dm_attn_bwd writes the same zeros); the rest runs as always
            dq = _zeros_ro((B, N, C), dy.dtype, dy.device)    # (read-only operands of the GEMMs below: one cached buffer
            dk = _zeros_ro(tuple(dv.shape), dv.dtype, dv.device)  #  per shape instead of a 75 MB memset per layer and step)
        else:
            dq, dk, dv = K.attn_bwd(q, kc, vc, o, lse, d_o, self.num_heads, drop_p=p, drop_seed=seed)
        K.wgrad_side(tok, dq, g.of(self.to_q.weight), None, taps=1)
        K.wgrad_side(ctx, dk, g.of(self.to_k.weight), None, taps=1)
        K.wgrad_side(ctx, dv, g.of(self.to_v.weight), None, taps=1)
        dx = K.gemm(dq, PW.of(self.to_q).bwd, C, res=None if add is None else add.view(B, N, C))
        dctx = None
        if need_dctx:
            dctx = K.gemm(dk, PW.of(self.to_k).bwd, self.context_dim)
            dctx = K.gemm(dv, PW.of(self.to_v).bwd, self.context_dim, res=dctx)
        if g.dctx is not None:
            self._add_dctx(g, dk, dv)
        return dx.view(dy.shape), dctx

    def _add_dctx(self, g: Grads, dk, dv):
        """g.dctx += dk . Wk + dv . Wv (reference unet.py:200-201 differentiated; fp32 accumulation across layers)."""
        B, L, _ = dv.shape
        for d, lin in ((dk, self.to_k), (dv, self.to_v)):
            if d is None:
                continue
            part = torch.empty((B, L, self.context_dim), dtype=torch.float32, device=dv.device)
            K.gemm(d, PW.of(lin).bwd, self.context_dim, out_f32=part)
            K.batchsum(part.view(1, -1), g.dctx.view(-1))

    def forward(self, x, context):
        return self._apply_fn(x, context)

    def _run_fwd(self, x, context):
        c = context.detach().float()
        self._ctx2d = c.dim() == 2
        if c.dim() == 2:
            c = c.unsqueeze(1)
        y, saved = self._fwd(_nhwc_in(x), K.to_bf16(c.contiguous()))
        return K.nhwc_to_nchw(y), saved

    def _run_bwd(self, dout, saved, g, need):
        dx, dctx = self._bwd(_nhwc_in(dout), saved, g, need_dctx=True)
        dctx = K.to_f32(dctx)
        if self._ctx2d:
            dctx = dctx.squeeze(1)
        return K.nhwc_to_nchw(dx), dctx


class Upsample(HipModule):
    """Reference unet.py:222-229: nearest x2 then conv3x3 (the x2 tensor is never materialised)."""

    def __init__(self, ch: int):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, kernel_size=3, padding=1)
        self.ch = ch

    def _fwd(self, x):
        return K.conv3x3(x, PW.of(self.conv).fwd, self.ch, bias=_b(self.conv), mode=2), (x,)

    def _bwd(self, dy, saved, g: Grads, add=None):
        (x,) = saved
        K.wgrad_side(x, dy, g.of(self.conv.weight), g.of(self.conv.bias), mode=2)
        dx = K.conv3x3(dy, PW.of(self.conv).bwd, self.ch, pool2=True)
        return dx if add is None else K.add(dx, add)

    def forward(self, x):
        return self._apply_fn(x)

    def _run_fwd(self, x):
        y, saved = self._fwd(_nhwc_in(x))
        return K.nhwc_to_nchw(y), saved

    def _run_bwd(self, dout, saved, g, need):
        return (K.nhwc_to_nchw(self._bwd(_nhwc_in(dout), saved, g)),)


class Downsample(HipModule):
    """Reference unet.py:231-237: conv3x3 stride 2."""

    def __init__(self, ch: int):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, kernel_size=3, stride=2, padding=1)
        self.ch = ch

    def _fwd(self, x):
        return K.conv3x3(x, PW.of(self.conv).fwd, self.ch, bias=_b(self.conv), mode=1), (x,)

    def _bwd(self, dy, saved, g: Grads, add=None):
        (x,) = saved
        K.wgrad_side(x, dy, g.of(self.conv.weight), g.of(self.conv.bias), mode=1)
        return K.conv3x3(dy, PW.of(self.conv).bwd, self.ch, mode=3, res=add)

    def forward(self, x):
        return self._apply_fn(x)

    def _run_fwd(self, x):
        y, saved = self._fwd(_nhwc_in(x))
        return K.nhwc_to_nchw(y), saved

    def _run_bwd(self, dout, saved, g, need):
        return (K.nhwc_to_nchw(self._bwd(_nhwc_in(dout), saved, g)),)


# With ONE context token the cross-attention softmax is identically 1, so its output does not depend on its query
# input: in `x = attention(x); x = cross_attention(x, ctx); x = x + res` (unet.py:269-276) the self-attention block
# and the to_q / to_k projections cannot influence the output, and their gradients are exactly zero.
# ELIDE_DEAD_ATTENTION[0] = True skips that provably dead work (outputs, live gradients and the dropout seed
# sequence are unchanged bit for bit).  Off by default: the default network runs everything the reference runs.
ELIDE_DEAD_ATTENTION = [os.environ.get("DM_ELIDE_DEAD_ATTENTION", "0") == "1"]


# The same data-flow fact used for SCHEDULING instead of elision (on by default; DM_OVERLAP_DEAD_ATTENTION=0 turns it
# off): everything still runs, but the branch nothing on the critical path waits for -- self-attention, to_q, to_k and
# the attention call that yields the log-sum-exp -- is enqueued on an auxiliary HIP stream, where its VALU-bound
# kernels run beside the MFMA-bound convolutions of the main chain.  The main stream joins it at the end of the forward.
OVERLAP_DEAD_ATTENTION = [os.environ.get("DM_OVERLAP_DEAD_ATTENTION", "1") != "0"]
# ... also in the layers that have a cross-attention block but no self-attention block (L1 / L4 of the STL10 network):
# there the auxiliary stream takes to_q, to_k, the log-sum-exp and, in the backward, the whole cross-attention backward
# (DM_OVERLAP_CROSS_ONLY=0: only layers with a self-attention block are scheduled this way).
OVERLAP_CROSS_ONLY = [os.environ.get("DM_OVERLAP_CROSS_ONLY", "1") != "0"]
# One-token context: attention output + to_out + residual as ONE streaming pass (dm_cross1_fwd / dm_cross1_bwd,
# include/dm_hip.h) instead of attn_onekey_fwd -> GEMM and, backward, GEMM -> weight gradient -> attn_onekey_bwd.
# Same dropout words, same results up to bf16 rounding of the (no longer materialised) attention output.
# DM_FUSE_CROSS1=0 restores the three-kernel form.
FUSE_CROSS1 = [os.environ.get("DM_FUSE_CROSS1", "1") != "0"]
KEEP_DEAD_DX = [False]   # tests: keep the (zero) input gradient of the overlapped backward branch in UnetLayer._dead_dx


class UnetLayer(HipModule):
    """Reference unet.py:239-285."""

    def __init__(self, upscale: bool, attention: bool, num_groups: int, dropout_prob: float, in_channels: int,
                 out_channels: int, time_emb_dim: int, num_heads: int = 4, use_film: bool = False,
                 use_cross_attention: bool = False, context_dim: int = None, use_flash_attention: bool = True):
        super().__init__()
        self.res_block = ResBlock(in_channels, out_channels, time_emb_dim, num_groups, dropout_prob, use_film)
        if attention:
            self.attention = Attention(out_channels, num_heads, dropout_prob, use_flash_attention=use_flash_attention)
        if use_cross_attention and context_dim is not None:
            self.cross_attention = CrossAttention(query_dim=out_channels, context_dim=context_dim,
                                                  num_heads=num_heads, dropout_prob=dropout_prob,
                                                  use_flash_attention=use_flash_attention)
        if upscale:
            self.upsample = Upsample(out_channels)
            self.conv = nn.Conv2d(out_channels, out_channels // 2, kernel_size=1)
        else:
            self.downsample = Downsample(out_channels)
        self.cout = out_channels

    def _fwd(self, x, temb, ctx, skip_out=None, y_out=None):
        """Down layers write the pre-downsample tensor (the skip) into ``skip_out``; up layers write
        their output into ``y_out`` (both are channel slices of the decoder's concat buffer)."""
        has_a = hasattr(self, "attention")
        has_c = hasattr(self, "cross_attention") and ctx is not None
        up = hasattr(self, "upsample")
        r, s_res = self.res_block._fwd(x, temb)
        dst = None if up else skip_out
        s_a = s_c = None
        bf = r.dtype == BF16   # the one-token specialisations (elision, two-stream schedule, dm_cross1) are bf16-mode kernels
        elide = has_c and bf and ELIDE_DEAD_ATTENTION[0] and ctx.shape[1] == 1
        if elide:
            if has_a and self.attention.training and self.attention.dropout_prob > 0:
                next(_SEED)  # keep the dropout seed sequence of the un-elided network
            xo, s_c = self.cross_attention._fwd(r, ctx, res=r, out=dst, elide=True)
        elif has_c and bf and OVERLAP_DEAD_ATTENTION[0] and ctx.shape[1] == 1 and (has_a or OVERLAP_CROSS_ONLY[0]):
            xo, s_a, s_c = self._fwd_attn_overlapped(r, ctx, dst)
        elif has_a and has_c:
            xa, s_a = self.attention._fwd(r)
            xo, s_c = self.cross_attention._fwd(xa, ctx, res=r, out=dst)
        elif has_a:
            xo, s_a = self.attention._fwd(r, res=r, out=dst)
        elif has_c:
            xo, s_c = self.cross_attention._fwd(r, ctx, res=r, out=dst)
        else:
            xo = K.add(r, r, out=dst)
        if up:
            u, s_u = self.upsample._fwd(xo)
            B, H2, W2, C = u.shape
            y = K.gemm(u.view(B, H2 * W2, C), PW.of(self.conv).fwd, C // 2, bias=_b(self.conv),
                       out=None if y_out is None else y_out.view(B, H2 * W2, C // 2))
            y = y.view(B, H2, W2, C // 2)
            return y, xo, (s_res, s_a, s_c, s_u, u)
        y, s_d = self.downsample._fwd(xo)
        return y, xo, (s_res, s_a, s_c, s_d, None)

    def _fwd_attn_overlapped(self, r, ctx, dst):
        """x = cross_attention(attention(r), ctx) + r for a ONE-token context, scheduled on two streams.  The
        cross-attention output is to_out(dropout(to_v(ctx))) whatever its query input (softmax over one key), so the
        main stream computes exactly that and goes on; self-attention, to_q, to_k and the attention call that returns
        the log-sum-exp run on the auxiliary stream (same kernels, same dropout seeds, same saved tensors as the
        single-stream order)."""
        ca, at = self.cross_attention, getattr(self, "attention", None)   # (no self-attention block: only to_q / to_k
        B, H, W, C = r.shape                                              #  and the log-sum-exp go to the other stream)
        N = H * W
        p_a = at.dropout_prob if (at is not None and at.training) else 0.0
        seed_a = next(_SEED) if p_a > 0 else 0          # the single-stream order draws attention's seed first
        p_c = ca.dropout_prob if ca.training else 0.0
        seed_c = next(_SEED) if p_c > 0 else 0
        vc = K.gemm(ctx, PW.of(ca.to_v).fwd, C)
        with K.on_aux(r.device):
            K.hold_for_aux(r.device, r, vc)
            s_a = None
            xa = r
            if at is not None:
                xa, s_a = at._fwd(r, seed=seed_a)
            tok = xa.view(B, N, C)
            q = K.gemm(tok, PW.of(ca.to_q).fwd, C)
            kc = K.gemm(ctx, PW.of(ca.to_k).fwd, C)
            # (only the log-sum-exp: the attention output itself is formed on the main stream, _onekey_out)
            _, lse = K.attn_fwd(q, kc, vc, ca.num_heads, drop_p=p_c, drop_seed=seed_c, want_o=False)
        y, o = ca._onekey_out(vc, B, N, C, r, dst, p_c, seed_c)
        return y.view(B, H, W, C), s_a, (tok, ctx, q, kc, vc, o, lse, p_c, seed_c)

    def _bwd(self, dy, d_skip, saved, g: Grads, dts=None, need_dx=True):
        s_res, s_a, s_c, s_ud, u = saved
        if hasattr(self, "upsample"):
            B, H2, W2, Ch = dy.shape
            dyt = dy.view(B, H2 * W2, Ch)
            du = K.gemm(dyt, PW.of(self.conv).bwd, 2 * Ch)
            K.wgrad_side(u.view(B, H2 * W2, 2 * Ch), dyt, g.of(self.conv.weight), g.of(self.conv.bias), taps=1)
            dxo = self.upsample._bwd(du.view(B, H2, W2, 2 * Ch), s_ud, g)
        else:
            dxo = self.downsample._bwd(dy, s_ud, g, add=d_skip)
        # xo = attn_path(r) + r
        if s_c is not None and s_c[0] is None:  # elided: the attention path carries no gradient back to r
            self.cross_attention._bwd(dxo, s_c, g)
            if hasattr(self, "attention"):  # zero gradients, exactly: leave the accumulators untouched
                g.of(self.attention.linear_qkv.weight), g.of(self.attention.linear_out.weight)
            dr = dxo  # d(xo)/d(r) is the residual branch alone
        elif (s_c is not None and dxo.dtype == BF16 and OVERLAP_DEAD_ATTENTION[0] and s_c[1].shape[1] == 1
              and (s_a is not None or OVERLAP_CROSS_ONLY[0]) and g.dctx is None):   # (a context gradient is accumulated across layers: keep that on one stream)
            # One context token: d(cross_attention)/d(its query input) is exactly 0 (dq = 0 out of a softmax over one
            # key), so the gradient that comes back through attention(r) is an exact zero and dr = dxo.  Everything
            # still runs -- cross-attention backward (to_out / to_v / to_q / to_k gradients) and the whole self-attention
            # backward -- but on the auxiliary stream, beside the ResBlock backward that no longer waits for it; the
            # zero it returns is not added (tests/test_gpu_modules.py checks that it IS zero, bit for bit).
            with K.on_aux(dxo.device):
                K.hold_for_aux(dxo.device, dxo)
                dxa, _ = self.cross_attention._bwd(dxo, s_c, g)
                if s_a is not None:
                    dxa = self.attention._bwd(dxa, s_a, g)
                self._dead_dx = dxa if KEEP_DEAD_DX[0] else None
            dr = dxo
        elif s_a is not None and s_c is not None:
            dxa, _ = self.cross_attention._bwd(dxo, s_c, g)
            dr = self.attention._bwd(dxa, s_a, g, add=dxo)
        elif s_a is not None:
            dr = self.attention._bwd(dxo, s_a, g, add=dxo)
        elif s_c is not None:
            dr, _ = self.cross_attention._bwd(dxo, s_c, g, add=dxo)
        else:
            dr = K.add(dxo, dxo)
        return self.res_block._bwd(dr, s_res, g, dts, need_dx)


# ============================================================================ the network
class UNET(HipModule):
    """Reference unet.py:287-356."""

    def __init__(self, Channels: List = [64, 64, 128, 256, 128], Upscales: List = [False, False, True, True],
                 Attentions: List = [False, True, True, False], num_groups: int = 8, dropout_prob: float = 0.1,
                 num_heads: int = 8, time_steps: int = 1000, input_channels: int = 1, output_channels: int = 1,
                 label_embedding: bool = True, use_film: bool = False, use_cross_attention: bool = True,
                 context_dim: int = None, use_flash_attention: bool = True):
        super().__init__()
        time_emb_dim = 256
        self.num_layers = len(Channels) - 1
        for i in range(self.num_layers):
            layer = UnetLayer(Upscales[i], Attentions[i], num_groups, dropout_prob, Channels[i], Channels[i + 1],
                              time_emb_dim, num_heads, use_film, use_cross_attention=use_cross_attention,
                              context_dim=context_dim, use_flash_attention=use_flash_attention)
            setattr(self, f"Layer{i + 1}", layer)
        out_channels = (Channels[-1] // 2) + Channels[1]
        self.input_conv = nn.Conv2d(input_channels, Channels[0], kernel_size=3, padding=1)
        self.late_conv1 = nn.Conv2d(out_channels, out_channels // 2, kernel_size=3, padding=1)
        self.late_conv2 = nn.Conv2d(out_channels // 2, out_channels // 2, kernel_size=3, padding=1)
        self.output_conv = nn.Conv2d(out_channels // 2, output_channels, kernel_size=1)
        self.relu = nn.SiLU()
        nn.init.zeros_(self.output_conv.weight)
        if self.output_conv.bias is not None:
            nn.init.zeros_(self.output_conv.bias)
        self.time_embeddings = SinusoidalEmbeddings(time_steps, time_emb_dim)
        self.time_mlp = nn.Sequential(nn.Linear(time_emb_dim, time_emb_dim), nn.SiLU(),
                                      nn.Linear(time_emb_dim, time_emb_dim))
        self.Channels, self.Upscales = list(Channels), list(Upscales)
        self.cin, self.cout_img, self.temb_dim = input_channels, output_channels, time_emb_dim
        half = self.num_layers // 2
        for c in Channels:
            if c % 64:
                raise ValueError("the gfx950 conv kernels tile channels by 64: every entry of Channels must be a multiple of 64")
        if any(Upscales[:half]) or not all(Upscales[half:]):
            raise ValueError("encoder layers must downsample and decoder layers upsample (as in every reference config)")

    def layers(self):
        return [getattr(self, f"Layer{i + 1}") for i in range(self.num_layers)]

    # ------------------------------------------------------------------ hand-scheduled forward
    def _fwd(self, x, t, context):
        B, Cin, H, W = x.shape
        dev_ = x.device
        e16 = K.to_bf16(self.time_embeddings(t).contiguous())
        if e16.shape[0] != B:
            if e16.shape[0] != 1:
                raise ValueError("t must have one entry per sample (or a single entry)")
            e16 = e16.expand(B, -1).contiguous()
        l0, l2 = self.time_mlp[0], self.time_mlp[2]
        h1 = K.gemm(e16, PW.of(l0).fwd, self.temb_dim, bias=_b(l0))
        temb = K.gemm(h1, PW.of(l2).fwd, self.temb_dim, bias=_b(l2), act_silu_in=True)
        ctx = None
        if context is not None and torch.is_floating_point(context):
            c = context.detach().float()
            if c.dim() == 2:
                c = c.unsqueeze(1)
            ctx = K.to_bf16(c.contiguous())
        x0 = K.nchw_to_nhwc(x.detach().float().contiguous(), (Cin + 63) // 64 * 64)
        a = K.conv3x3(x0, PW.of(self.input_conv).fwd, self.Channels[0], bias=_b(self.input_conv))
        L, half = self.num_layers, self.num_layers // 2
        cats: List[Optional[torch.Tensor]] = [None] * L
        saved_layers = []
        for i in range(half):
            j = L - 1 - i
            c_dec, c_skip = self.Channels[j + 1] // 2, self.Channels[i + 1]
            cats[j] = torch.empty((B, a.shape[1], a.shape[2], c_dec + c_skip), dtype=a.dtype, device=dev_)
            a, _, s = self.layers()[i]._fwd(a, temb, ctx, skip_out=cats[j][..., c_dec:])
            saved_layers.append(s)
        for i in range(half, L):
            c_dec = self.Channels[i + 1] // 2
            _, _, s = self.layers()[i]._fwd(a, temb, ctx, y_out=cats[i][..., :c_dec])
            a = cats[i]
            saved_layers.append(s)
        cat = a
        Bc, Hc, Wc, Cc = cat.shape
        a1 = K.conv3x3(cat, PW.of(self.late_conv1).fwd, Cc // 2, bias=_b(self.late_conv1))
        a2 = K.conv3x3(a1, PW.of(self.late_conv2).fwd, Cc // 2, bias=_b(self.late_conv2), act_silu_in=True)
        npad = (self.cout_img + 15) // 16 * 16
        of = torch.empty((Bc, Hc, Wc, npad), dtype=torch.float32, device=dev_)
        K.gemm(a2.view(Bc, Hc * Wc, Cc // 2), PW.of(self.output_conv).fwd, npad, bias=PW.of(self.output_conv).bias_pad(npad),
               act_silu_in=True, out_f32=of.view(Bc, Hc * Wc, npad), n_valid=self.cout_img)
        y = K.nhwc_to_nchw(of, self.cout_img)
        K.join_aux(dev_)   # the overlapped attention branches (UnetLayer._fwd_attn_overlapped) are part of this forward
        return y, (x0, e16, h1, temb, ctx, saved_layers, cats, a1, a2)

    # ------------------------------------------------------------------ hand-scheduled backward
    def _bwd(self, dout, saved, g: Grads, need_dx=False, join=True, need_dctx=False):
        """Returns dx (NCHW fp32 or None); with need_dctx the gradient wrt the context is left in g.dctx
        (fp32 [B, L, Dc])."""
        x0, e16, h1, temb, ctx, saved_layers, cats, a1, a2 = saved
        if need_dctx and ctx is not None:
            g.dctx = torch.zeros(ctx.shape, dtype=torch.float32, device=ctx.device)
        L, half = self.num_layers, self.num_layers // 2
        cat = cats[L - 1]
        B, Hc, Wc, Cc = cat.shape
        Ch = Cc // 2
        M = Hc * Wc
        dy16 = K.nchw_to_nhwc(dout.detach().float().contiguous(), 16 if self.cout_img <= 16 else 0)
        kp = dy16.shape[-1]
        oc = self.output_conv
        K.wgrad_side(a2.view(B, M, Ch), dy16.view(B, M, kp), g.of(oc.weight).view(self.cout_img, Ch), g.of(oc.bias), taps=1,
                act_silu_in=True, n_valid=self.cout_img)
        d_s2 = K.gemm(dy16.view(B, M, kp), PW.of(oc).bwd, Ch)
        d_a2 = K.silu_bwd(d_s2.view(B, Hc, Wc, Ch), a2)
        K.wgrad_side(a1, d_a2, g.of(self.late_conv2.weight), g.of(self.late_conv2.bias), act_silu_in=True)
        d_s1 = K.conv3x3(d_a2, PW.of(self.late_conv2).bwd, Ch)
        d_a1 = K.silu_bwd(d_s1, a1)
        K.wgrad_side(cat, d_a1, g.of(self.late_conv1.weight), g.of(self.late_conv1.bias))
        d_cat = K.conv3x3(d_a1, PW.of(self.late_conv1).bwd, Cc)
        user_hook = self.__dict__.get("_bwd_hook")

        def hook(stage):
            # gradients of a finished stage are about to be consumed by the caller's hook (the data-parallel bucket
            # all-reduce): it is issued from the weight-gradient stream, behind the stage's weight gradients and the
            # main stream's work so far -- the main stream itself does not wait (kernels.call_on_side)
            if user_hook is not None:
                K.call_on_side(da_dev, lambda: user_hook(stage))

        da_dev = dout.device
        hook("head")
        dts = None
        d_skips: List[Optional[torch.Tensor]] = [None] * L
        da = None
        for i in range(L - 1, half - 1, -1):
            c_dec = self.Channels[i + 1] // 2
            d_skips[L - 1 - i] = d_cat[..., c_dec:]
            da, dts = self.layers()[i]._bwd(d_cat[..., :c_dec], None, saved_layers[i], g, dts)
            hook(f"Layer{i + 1}")
            d_cat = da  # gradient wrt this decoder layer's input = previous concat buffer (or encoder output)
        for i in range(half - 1, -1, -1):
            da, dts = self.layers()[i]._bwd(da, d_skips[i], saved_layers[i], g, dts, need_dx=True)
            hook(f"Layer{i + 1}")
        # input conv
        ic = self.input_conv
        K.wgrad_side(x0, da, g.of(ic.weight), g.of(ic.bias), k_valid=self.cin)
        dx = None
        if need_dx:
            # rare path (callers normally do not differentiate wrt the image): the data-gradient
            # operand needs its N (= input channels) padded to the 64-wide tile, so pad on the host
            wpad = torch.zeros((ic.weight.shape[0], x0.shape[-1], 3, 3), dtype=torch.float32, device=da.device)
            wpad[:, : self.cin] = ic.weight.detach()
            dxp = K.conv3x3(da, K.pack_weight(wpad, True, dtype=da.dtype), x0.shape[-1])
            dx = K.nhwc_to_nchw(dxp, self.cin)
        # time MLP
        l0, l2 = self.time_mlp[0], self.time_mlp[2]
        dtemb = K.silu_bwd(K.to_bf16(dts), temb)
        K.wgrad_side(h1, dtemb, g.of(l2.weight), g.of(l2.bias), taps=1, act_silu_in=True)
        dsh1 = K.gemm(dtemb, PW.of(l2).bwd, self.temb_dim)
        dh1 = K.silu_bwd(dsh1, h1)
        K.wgrad_side(e16, dh1, g.of(l0.weight), g.of(l0.bias), taps=1)
        hook("tail")
        K.join_aux(da_dev)   # the overlapped attention backward branches are part of this backward (saved tensors die with it)
        if join:  # join=False: the caller joins before it consumes the gradients (Trainer: before the optimizer step),
            K.join_side(da_dev)  # so the tail of the weight-gradient stream overlaps the next micro-batch's forward
        return dx

    # ------------------------------------------------------------------ public API
    def forward(self, x, t, context=None):
        if not x.is_cuda:
            raise RuntimeError("diffusion_model_amd.UNET runs only on an AMD GPU (HIP kernels); no CPU fallback")
        if context is not None and not torch.is_floating_point(context):
            if any(hasattr(l, "cross_attention") for l in self.layers()):
                raise TypeError("integer labels as context are only accepted when no cross-attention exists "
                                "(reference unet.py:271 would feed them to nn.Linear and fail)")
            context = None  # legacy label tensor: ignored, as in the reference (main.py:177)
        if not torch.is_tensor(t):
            t = torch.as_tensor(t, dtype=torch.long)
        if torch.is_grad_enabled() and (x.requires_grad or (torch.is_tensor(context) and context.requires_grad)
                                        or any(p.requires_grad for p in self.parameters())):
            return _UNetFn.apply(self, x, t, context, *list(self.parameters()))
        y, _ = self._fwd(x, t, context)
        return y


def _pad_bias(c: nn.Module, npad: int):
    if c.bias is None:
        return None
    b = torch.zeros(npad, dtype=torch.float32, device=c.bias.device)
    b[: c.bias.numel()] = c.bias.detach()
    return b


class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, t, context, *params):
        y, saved = mod._fwd(x, t, context)
        ctx.mod, ctx.saved, ctx.params = mod, saved, params
        ctx.need_dx = x.requires_grad
        ctx.need_dctx = torch.is_tensor(context) and torch.is_floating_point(context) and context.requires_grad
        ctx.ctx_shape = tuple(context.shape) if torch.is_tensor(context) else None
        return y

    @staticmethod
    def backward(ctx, dout):
        g = Grads()
        dx = ctx.mod._bwd(dout.contiguous(), ctx.saved, g, need_dx=ctx.need_dx, need_dctx=ctx.need_dctx)
        pg = []
        for p in ctx.params:
            t = g.peek(p)
            pg.append(t if p.requires_grad else None)
        dctx = None
        if ctx.need_dctx:   # zeros when the network has no cross-attention (the context is then unused)
            dctx = (g.dctx if g.dctx is not None else torch.zeros(ctx.ctx_shape, device=dout.device)).view(ctx.ctx_shape)
        return (None, dx, None, dctx, *pg)
