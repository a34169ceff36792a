"""Thin torch-tensor front-ends over the C ABI (include/dm_hip.h).  No autograd here and no
torch math: every function only validates shapes/strides on the host, then launches HIP
kernels on the current stream.  Activations are NHWC bf16 tensors whose last dim is
contiguous; slices along the channel dim are allowed (pixel stride = stride(-2))."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib
from ._lib import AttnArgs, ConvArgs, Cross1Args, GemmArgs, WgradArgs, check

BF16 = torch.bfloat16
F32 = torch.float32
# COMPUTE MODE: the dtype of the NHWC activations the host layer creates at the module boundary (nchw_to_nhwc, to_act).
# bf16 (default) = the MFMA bf16 kernels that carry the headline; fp32 = the reference-precision mode (csrc/dm_f32.hip:
# fp32 MFMA contractions, fp32 attention, double-accumulated GroupNorm statistics; parity bar 1e-4).  Every front-end
# below dispatches on the dtype of the activation it is handed, so the two modes share the host code of unet.py.
ACT = [F32 if os.environ.get("DM_COMPUTE_DTYPE", "bf16").lower() in ("fp32", "float32", "f32") else BF16]


def set_compute_dtype(dtype) -> None:
    """torch.bfloat16 (default) or torch.float32 (reference-precision mode).  Process-wide; switch between forwards."""
    if isinstance(dtype, str):
        dtype = {"bf16": BF16, "bfloat16": BF16, "fp32": F32, "float32": F32, "f32": F32}[dtype.lower()]
    if dtype not in (BF16, F32):
        raise ValueError("compute dtype must be torch.bfloat16 or torch.float32")
    ACT[0] = dtype


class compute_dtype:
    """``with compute_dtype(torch.float32): ...`` -- the mode inside the block, the previous one restored after it."""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        self.prev = ACT[0]
        set_compute_dtype(self.dtype)

    def __exit__(self, *exc):
        ACT[0] = self.prev
        return False


def _is32(t) -> bool:
    return t.dtype == F32


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)
_FORCED_STREAM = [None]  # set by wgrad_side: launch on this raw stream instead of torch's current one


def _stream() -> int:
    """Raw handle of the stream to launch on.  The launch path calls this once per kernel, so it goes through
    torch's C accessors when they exist (torch.cuda.current_stream() builds a Stream object per call)."""
    f = _FORCED_STREAM[0]
    if f is not None:
        return f
    if _RAW_STREAM is not None and _GET_DEVICE is not None:
        return _RAW_STREAM(_GET_DEVICE())
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _act(t: torch.Tensor, what: str = "activation") -> int:
    """Validate an NHWC/[M,C] activation view (bf16, or fp32 in the fp32 compute mode); return the pixel stride."""
    if not (t.is_cuda and t.dtype in (BF16, F32) and t.stride(-1) == 1):
        raise ValueError(f"{what}: need a CUDA bf16 (or fp32-mode float32) tensor with contiguous channels")
    if t.is_contiguous():  # the common case, one C call
        return t.shape[-1]
    ld = t.stride(-2) if t.dim() >= 2 else t.shape[-1]
    n = 1
    for i in range(t.dim() - 2, -1, -1):  # outer dims must be densely packed over ld
        if t.shape[i] != 1 and t.stride(i) != ld * n:
            raise ValueError(f"{what}: unsupported strides {t.stride()} for shape {tuple(t.shape)}")
        n *= t.shape[i]
    return ld


def _f32(t: Optional[torch.Tensor], what: str):
    if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError(f"{what}: need a contiguous CUDA fp32 tensor")
    return t


def last_variant() -> str:
    """Name of the kernel instantiation the last conv3x3 / gemm / wgrad / attention call of this thread launched
    (dm_last_variant, include/dm_hip.h): the parity tests assert the size-dependent dispatch through it."""
    return (_lib.lib().dm_last_variant() or b"").decode()


def set_dropout_counter(counter: Optional[torch.Tensor]) -> None:
    """Register (or, with None, unregister) the device counter every dropout kernel adds to its seed
    (dm_set_dropout_counter): int32/uint32 tensor of one element that the caller keeps alive."""
    if counter is not None and not (counter.is_cuda and counter.numel() == 1 and counter.element_size() == 4):
        raise ValueError("dropout counter: need a one-element 32-bit CUDA tensor")
    if counter is not None and _DROP_COUNTER[0] is not None and _DROP_COUNTER[0] != counter.data_ptr():
        raise RuntimeError("dropout counter: another counter is registered (one graph-replaying Trainer per process; "
                           "call disable_graph() on the other one first)")
    check(_lib.lib().dm_set_dropout_counter(_p(counter)), "dm_set_dropout_counter")
    _DROP_COUNTER[0] = None if counter is None else counter.data_ptr()


_DROP_COUNTER = [None]   # address registered with the library (process-global state of libdm_hip.so: INTEGRATION.md)


def release_dropout_counter(ptr: int) -> None:
    """Unregister the counter at ``ptr`` if it is still the registered one (weakref.finalize of its owner: a Trainer
    dropped without disable_graph() must not leave the library pointing at freed memory)."""
    if _DROP_COUNTER[0] == ptr and _lib is not None:
        _lib.lib().dm_set_dropout_counter(None)
        _DROP_COUNTER[0] = None


def bump_dropout_counter(counter: torch.Tensor) -> None:
    check(_lib.lib().dm_bump_dropout_counter(_p(counter), _stream()), "dm_bump_dropout_counter")


# --------------------------------------------------------------------------- workspace
_WS = {}
_WS_RETIRED = []   # outgrown side-stream buffers, kept alive until join_side() (see workspace())
# HIP-graph replay bakes workspace ADDRESSES into the captured kernels.  While a graph is live (Trainer.enable_graph
# pins, disable_graph unpins) an outgrown buffer is therefore never released: it moves here, so a later eager call with
# a larger shape gets a new buffer while the graph keeps replaying into memory that is still ours.
_WS_PIN_COUNT = [0]
_WS_PINNED_KEEP = []


def pin_workspaces() -> None:
    _WS_PIN_COUNT[0] += 1


def unpin_workspaces() -> None:
    _WS_PIN_COUNT[0] = max(0, _WS_PIN_COUNT[0] - 1)
    if _WS_PIN_COUNT[0] == 0:
        _WS_PINNED_KEEP.clear()


def workspace(nbytes: int, device) -> torch.Tensor:
    """A per-(device, stream) scratch buffer that only ever grows (stable under graph replay once warm).
    Kernels on different streams must not share scratch: the weight-gradient side stream gets its own.

    A buffer is allocated while torch's CURRENT stream is the main one, so the caching allocator would hand an
    outgrown block straight back to the next main-stream ``torch.empty``.  For the side stream's buffer that is a
    race: the split-K weight gradient queued before the growth may still be reading and writing its slabs there.  An
    outgrown buffer of a forced (side) stream is therefore parked in ``_WS_RETIRED`` and released only by
    ``join_side``, after the main stream has been ordered behind the side stream."""
    forced = _FORCED_STREAM[0]
    key = (device.index if isinstance(device, torch.device) else torch.device(device).index, _stream())
    cur = _WS.get(key)
    if cur is None or cur.numel() < nbytes:
        if cur is not None and _WS_PIN_COUNT[0] > 0:
            _WS_PINNED_KEEP.append(cur)
        elif cur is not None and forced is not None:
            _WS_RETIRED.append(cur)
        cur = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = cur
    return cur


def reserve_workspace(nbytes: int, device) -> None:
    workspace(nbytes, device)


# --------------------------------------------------------------------------- weights
def pack_weight(w: torch.Tensor, transpose: bool = False, out: Optional[torch.Tensor] = None, dtype=BF16) -> torch.Tensor:
    """fp32 [N,K,3,3] / [N,K,1,1] / [N,K] -> packed bf16 shadow (uint8 buffer).  transpose=True
    builds the data-gradient operand (roles of N and K swapped, taps flipped).  dtype=float32: the fp32 mode's repack
    (dm_pack_weight_f32: [taps][K][N] fp32, zero padded to multiples of 64)."""
    _f32(w, "weight")
    taps = 9 if (w.dim() == 4 and w.shape[-1] == 3) else 1
    n0, k0 = w.shape[0], w.shape[1]
    N, K = (k0, n0) if transpose else (n0, k0)
    if dtype == F32:
        nbytes = _lib.lib().dm_packed_weight_f32_bytes(N, K, taps)
        if out is None:
            out = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        assert out.numel() >= nbytes
        check(_lib.lib().dm_pack_weight_f32(_p(w), _p(out), N, K, taps, int(transpose), _stream()), "dm_pack_weight_f32")
        return out
    nbytes = _lib.lib().dm_packed_weight_bytes(N, K, taps)
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    assert out.numel() >= nbytes
    check(_lib.lib().dm_pack_weight(_p(w), _p(out), N, K, taps, int(transpose), _stream()), "dm_pack_weight")
    return out


def _check_pack(wp: torch.Tensor, n: int, k: int, taps: int, f32m: bool) -> None:
    """The kernels index the packed weight by (n, k, taps) alone: a buffer packed for the OTHER compute mode (half / twice
    the bytes) or for other sizes would be read out of bounds -- refuse it on the host."""
    L = _lib.lib()
    need = L.dm_packed_weight_f32_bytes(n, k, taps) if f32m else L.dm_packed_weight_bytes(n, k, taps)
    if wp.numel() * wp.element_size() < need or (f32m and wp.numel() * wp.element_size() < need):
        raise ValueError(f"packed weight holds {wp.numel() * wp.element_size()} bytes, the {'fp32' if f32m else 'bf16'} kernel "
                         f"reads {need} (N={n}, K={k}, taps={taps}): packed for another compute mode or shape?")


# --------------------------------------------------------------------------- conv / gemm
# bench.py's roofline leg: when set to {"mode", "H", "cin", "cout", "pairs": []}, matching conv launches are bracketed
# by HIP events on the launch stream, so the kernel is timed inside the timed training steps themselves.
PROBE = None


def conv3x3(x, wp, cout, *, bias=None, chan_add=None, res=None, out=None, mode=0, pool2=False,
            act_silu_in=False, drop_p=0.0, drop_seed=0, out_f32=None, n_valid=0, norm_in=None):
    """norm_in = (scale, shift) from gn_prepare: the conv reads silu(x * scale + shift), i.e. SiLU(GroupNorm(x)),
    while staging its input -- the normalised activation is never materialised."""
    B, H, W, Cin = x.shape
    ldx = _act(x, "x")
    Ho, Wo = (H // 2, W // 2) if mode == 1 else ((2 * H, 2 * W) if mode in (2, 3) else (H, W))
    if pool2:
        Ho, Wo = Ho // 2, Wo // 2
    f32m = _is32(x)
    if out is None and out_f32 is None:
        out = torch.empty((B, Ho, Wo, cout), dtype=x.dtype, device=x.device)
    if f32m and pool2:
        # fp32 mode: the nearest-x2 data-gradient = full-size convolution, then the 2x2 sums (dm_pool2_f32)
        assert out_f32 is None and res is None and chan_add is None and drop_p == 0.0
        full = conv3x3(x, wp, cout, bias=None, mode=mode, act_silu_in=act_silu_in, n_valid=n_valid)
        check(_lib.lib().dm_pool2_f32(_p(full), _p(out), _act(out, "out"), B, 2 * Ho, 2 * Wo, cout, _stream()), "dm_pool2_f32")
        if bias is not None:
            raise ValueError("fp32 pooled data-gradient takes no bias")
        return out
    _check_pack(wp, cout, Cin, 9, f32m)
    a = ConvArgs()
    a.x, a.ldx, a.w = _p(x), ldx, _p(wp)
    a.bias, a.chan_add = _p(_f32(bias, "bias")), _p(_f32(chan_add, "chan_add"))
    if res is not None:
        a.res, a.ldr = _p(res), _act(res, "res")
        assert tuple(res.shape) == (B, Ho, Wo, cout) and res.dtype == x.dtype
    if out is not None:
        assert tuple(out.shape[:3]) == (B, Ho, Wo) and out.shape[3] >= (n_valid or cout)
        a.y, a.ldy = _p(out), _act(out, "out")
    if out_f32 is not None:
        assert out_f32.dtype == torch.float32 and out_f32.stride(-1) == 1
        a.yf, a.ldyf = _p(out_f32), out_f32.stride(-2)
    a.B, a.H, a.W, a.Cin, a.Cout = B, H, W, Cin, cout
    a.mode, a.n_valid, a.pool2, a.act_silu_in = mode, n_valid, int(pool2), int(act_silu_in)
    a.drop_p, a.drop_seed = float(drop_p), int(drop_seed)
    if norm_in is not None:
        sc, sh = norm_in
        assert tuple(sc.shape) == (B, Cin) and tuple(sh.shape) == (B, Cin)
        a.norm_scale, a.norm_shift = _p(_f32(sc, "norm scale")), _p(_f32(sh, "norm shift"))
    if f32m:
        assert norm_in is None, "GroupNorm folding is a bf16-mode option"
        check(_lib.lib().dm_conv3x3_f32(C.byref(a), _stream()), "dm_conv3x3_f32")
        return out if out is not None else out_f32
    pr = PROBE
    if pr is not None and mode == pr["mode"] and H == pr["H"] and Cin == pr["cin"] and cout == pr["cout"]:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.lib().dm_conv3x3(C.byref(a), _stream()), "dm_conv3x3")
        e1.record()
        pr["pairs"].append((e0, e1))
    else:
        check(_lib.lib().dm_conv3x3(C.byref(a), _stream()), "dm_conv3x3")
    return out if out is not None else out_f32


def gemm(x, wp, n, *, bias=None, chan_add=None, rows_per_batch=0, res=None, out=None, out_f32=None,
         act_silu_in=False, n_valid=0):
    """x: [..., K] bf16 -> [..., n]."""
    ldx = _act(x, "x")
    K = x.shape[-1]
    M = x.numel() // K
    if out is None and out_f32 is None:
        out = torch.empty((*x.shape[:-1], n), dtype=x.dtype, device=x.device)
    _check_pack(wp, n, K, 1, _is32(x))
    a = GemmArgs()
    a.x, a.ldx, a.w = _p(x), ldx, _p(wp)
    a.bias, a.chan_add = _p(_f32(bias, "bias")), _p(_f32(chan_add, "chan_add"))
    if res is not None:
        a.res, a.ldr = _p(res), _act(res, "res")
    if out is not None:
        a.y, a.ldy = _p(out), _act(out, "out")
    if out_f32 is not None:
        a.yf, a.ldyf = _p(out_f32), out_f32.stride(-2)
    a.M, a.K, a.N = M, K, n
    a.rows_per_batch, a.n_valid, a.act_silu_in = rows_per_batch, n_valid, int(act_silu_in)
    if _is32(x):
        check(_lib.lib().dm_gemm_f32(C.byref(a), _stream()), "dm_gemm_f32")
    else:
        check(_lib.lib().dm_gemm(C.byref(a), _stream()), "dm_gemm")
    return out if out is not None else out_f32


def wgrad(x, dy, dw, dbias=None, *, taps=9, mode=0, act_silu_in=False, n_valid=0, k_valid=0, norm_in=None):
    """dw (fp32, reference layout) += dy^T * patches(x); dbias += column sums of dy.  norm_in as in conv3x3: the
    forward conv read silu(x * scale + shift)."""
    ldx, lddy = _act(x, "x"), _act(dy, "dy")
    _f32(dw, "dw"), _f32(dbias, "dbias")
    Cin, Cout = x.shape[-1], dy.shape[-1]
    if taps == 9:
        B, H, W, _ = x.shape
    else:
        B, H, W = 1, 1, x.numel() // Cin
    L = _lib.lib()
    f32m = _is32(x)
    nb = (L.dm_wgrad_f32_workspace_bytes if f32m else L.dm_wgrad_workspace_bytes)(B, H, W, Cin, Cout, taps, mode)
    ws = workspace(nb, x.device)
    a = WgradArgs()
    a.x, a.ldx, a.dy, a.lddy, a.dw, a.dbias = _p(x), ldx, _p(dy), lddy, _p(dw), _p(dbias)
    a.B, a.H, a.W, a.Cin, a.Cout = B, H, W, Cin, Cout
    a.taps, a.mode, a.act_silu_in, a.n_valid, a.k_valid = taps, mode, int(act_silu_in), n_valid, k_valid
    a.workspace, a.ws_bytes = _p(ws), ws.numel()
    if norm_in is not None:
        a.norm_scale, a.norm_shift = _p(_f32(norm_in[0], "norm scale")), _p(_f32(norm_in[1], "norm shift"))
    if f32m:
        assert norm_in is None and dy.dtype == F32
        check(L.dm_wgrad_f32(C.byref(a), _stream()), "dm_wgrad_f32")
        return
    check(L.dm_wgrad(C.byref(a), _stream()), "dm_wgrad")


def pack_weights_batch(table: torch.Tensor, n: int, total_blocks: int) -> None:
    """table: device uint8 tensor holding n dm_pack_desc entries (include/dm_hip.h)."""
    check(_lib.lib().dm_pack_weights_batch(_p(table), int(n), int(total_blocks), _stream()), "dm_pack_weights_batch")


# Weight gradients are off the backward critical path (nothing downstream in the same backward reads
# them), so the UNet enqueues them on a side HIP stream: they fill the CUs that the data-gradient chain
# (dgrad convs, GroupNorm / attention backward: HBM- or VALU-bound) leaves idle.  join_side() makes the
# current stream wait for them; it is called before gradients are consumed (bucket all-reduce, optimizer,
# autograd hand-back).
SIDE_WGRAD = os.environ.get("DM_SIDE_WGRAD", "1") != "0"
_SIDE = {}
_SIDE_DIRTY = set()


def _side_stream(device):
    key = device.index or 0
    st = _SIDE.get(key)
    if st is None:
        prio = os.environ.get("DM_SIDE_PRIORITY", "")   # developer switch (tools/prio_ab.sh)
        stream = _stream_with_priority(device, int(prio)) if prio not in ("", "0") else torch.cuda.Stream(device=device)
        st = (stream, torch.cuda.Event())
        _SIDE[key] = st
    return st


_PENDING = []     # weight gradients waiting for the next fork: (x, dy, dw, dbias, kw)
CAPTURE_KEEPALIVE = None   # a list while a HIP graph is being captured (Trainer.capture), else None
SIDE_BATCH = 8    # hipStreamWaitEvent costs the host ~14 us: fork once per few weight gradients, not per call (8 measured best of 1..16)


def wgrad_side(x, dy, dw, dbias=None, **kw):
    """Queue a weight gradient for the side stream.  x and dy must not be written again by the caller (the UNet's
    backward never does: the side stream reads them concurrently with whatever the main stream runs next)."""
    if not SIDE_WGRAD:
        return wgrad(x, dy, dw, dbias, **kw)
    _PENDING.append((x, dy, dw, dbias, kw))
    if len(_PENDING) >= SIDE_BATCH:
        flush_side()


def flush_side() -> None:
    if not _PENDING:
        return
    dev = _PENDING[0][0].device
    side, ev = _side_stream(dev)
    ev.record()                 # on the current stream: every queued x and dy is complete here
    side.wait_event(ev)         # (one event object per device is enough: waits are enqueued in host order)
    _FORCED_STREAM[0] = side.cuda_stream
    try:
        for x, dy, dw, dbias, kw in _PENDING:
            wgrad(x, dy, dw, dbias, **kw)
    finally:
        _FORCED_STREAM[0] = None
    for x, dy, _, _, kw in _PENDING:
        held = (x, dy, *(kw.get("norm_in") or ()))
        if CAPTURE_KEEPALIVE is not None:
            # under graph capture the operands live in the graph's private pool: keeping them referenced until the
            # capture ends stops the allocator from handing their blocks out again inside the same graph
            CAPTURE_KEEPALIVE.append(held)
        else:
            for t in held:
                t.record_stream(side)   # the caching allocator must not recycle these before the side stream is done
    _PENDING.clear()
    _SIDE_DIRTY.add(dev.index or 0)


def join_side(device=None) -> None:
    flush_side()
    key = torch.device(device).index or 0 if device is not None else torch.cuda.current_device()
    if key in _SIDE_DIRTY:
        torch.cuda.current_stream(key).wait_stream(_SIDE[key][0])
        _SIDE_DIRTY.discard(key)
    if _WS_RETIRED and not _SIDE_DIRTY:
        _WS_RETIRED.clear()   # every stream that could still touch them is ordered before the current one now


# A second auxiliary stream for forward work that nothing on the critical path waits for (unet.UnetLayer: with a
# one-token context the cross-attention output does not depend on its query input, so the self-attention block, the
# to_q / to_k projections and the log-sum-exp feed nothing the main chain needs before the backward pass).
_AUX = {}
_AUX_DIRTY = set()
_AUX_JOIN_EV = {}   # per device: the event call_on_side records on the auxiliary stream (aux -> side join)


def _stream_with_priority(device, priority: int):
    """A HIP stream of the given priority (hipStreamCreateWithPriority: -1 high, 0 normal, 1 low on this runtime; torch's
    own constructor only offers 0 and negative values) wrapped for torch.  The stream lives until the process ends."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    handle = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipStreamCreateWithPriority(ctypes.byref(handle), ctypes.c_uint(1), ctypes.c_int(priority))   # 1 = non-blocking
    if rc != 0 or not handle.value:
        raise RuntimeError(f"hipStreamCreateWithPriority failed ({rc})")
    return torch.cuda.ExternalStream(handle.value, device=device)


def aux_stream(device):
    key = torch.device(device).index or 0
    st = _AUX.get(key)
    if st is None:
        # Priorities were measured and left off (DM_AUX_PRIORITY / DM_SIDE_PRIORITY stay as developer switches, tools/prio_ab.sh):
        # this stream LOW gave 14.751 ms per step against 14.783 at equal priority (-0.2 %), HIGH 14.894 (+0.8 %), the
        # weight-gradient stream low 14.837 -- the step is bound by total throughput, not by its critical path -- and a
        # priority stream in the process made every HIP-graph replay 2.5x slower (MNIST leg 1.75 -> 4.4 ms per step,
        # tools/mnist_ab.sh), whether or not the graph uses it.
        prio = os.environ.get("DM_AUX_PRIORITY", "")
        dev = torch.device(device)
        stream = _stream_with_priority(dev, int(prio)) if prio not in ("", "0") else torch.cuda.Stream(device=dev)
        st = (stream, torch.cuda.Event())
        _AUX[key] = st
    return st


class on_aux:
    """``with on_aux(device):`` -- launches (and torch allocations) inside go to the auxiliary stream, ordered after
    everything enqueued so far on the current stream.  ``join_aux`` makes the current stream wait for it."""

    def __init__(self, device):
        self.dev = torch.device(device)

    def __enter__(self):
        flush_side()   # weight gradients queued so far fork from the CURRENT stream, not from the auxiliary one
        aux, ev = aux_stream(self.dev)
        ev.record()
        aux.wait_event(ev)
        self.ctx = torch.cuda.stream(aux)
        self.ctx.__enter__()
        _AUX_DIRTY.add(self.dev.index or 0)
        return aux

    def __exit__(self, *exc):
        flush_side()   # weight gradients queued inside fork from the auxiliary stream (their operands are produced there)
        return self.ctx.__exit__(*exc)


def hold_for_aux(device, *tensors) -> None:
    """Tensors of the current stream that kernels on the auxiliary stream read, and that the caller may free before
    ``join_aux``: keep the caching allocator from recycling them early."""
    aux, _ = aux_stream(device)
    for t in tensors:
        if CAPTURE_KEEPALIVE is not None:
            CAPTURE_KEEPALIVE.append((t,))
        else:
            t.record_stream(aux)


def join_aux(device) -> None:
    key = torch.device(device).index or 0
    if key in _AUX_DIRTY:
        torch.cuda.current_stream(key).wait_stream(_AUX[key][0])
        _AUX_DIRTY.discard(key)


def reserve_streams(device) -> None:
    """Create the weight-gradient and auxiliary streams NOW and run one tiny operation on each, so that they hold
    hardware queues of their own before anything else in the process creates streams.  A process has few hardware
    queues (GPU_MAX_HW_QUEUES, 4 by default); HIP maps streams onto them as they are created, and RCCL creates several
    when a communicator is set up.  With the communicator first, two of {main, weight-gradient, auxiliary} ended up on
    ONE hardware queue and the step lost its stream overlap: +0.9 ms (6 %) per optimizer step for an IDLE communicator
    (tools/dp_ab.sh).  Call before torch.distributed.init_process_group(..., device_id=...) / the first collective."""
    dev = torch.device(device)
    if os.environ.get("DM_RESERVE_MODE", "") != "nomain":   # (measurement switch, tools/hwq_ab2.sh)
        torch.zeros(1, device=dev)          # the main (current) stream first: streams get their queue at first USE
        torch.cuda.current_stream(dev).synchronize()
    for st, _ in (_side_stream(dev), aux_stream(dev)):
        with torch.cuda.stream(st):
            torch.zeros(1, device=dev)
        st.synchronize()


def call_on_side(device, fn) -> None:
    """Run ``fn()`` with the weight-gradient side stream as torch's CURRENT stream, ordered after everything enqueued
    so far on the current stream and on the side stream.  The data-parallel bucket hook issues its asynchronous
    all-reduce this way: RCCL's internal stream then waits for the side stream (whose fork point covers the main
    stream's work up to here), so the collective sees the finished gradients of the stage while the MAIN stream is
    never made to drain the weight-gradient queue (one event per stage instead of a join per bucket)."""
    dev = torch.device(device)
    flush_side()
    side, ev = _side_stream(dev)
    ev.record()
    side.wait_event(ev)
    # ORDERING RULE: what fn() consumes may have been produced on THREE streams -- the current one
    # (fork above), the weight-gradient stream itself (in order) and the auxiliary stream, which carries the one-token
    # attention backward branch (to_out / to_v / to_q / to_k / qkv gradients, written inline there when DM_SIDE_WGRAD=0).
    # The auxiliary stream is joined explicitly: the side stream waits for everything enqueued on it so far, whether
    # or not a weight gradient happened to fork from it.
    key = dev.index or 0
    if key in _AUX_DIRTY:
        aux, _ = aux_stream(dev)
        jev = _AUX_JOIN_EV.get(key)
        if jev is None:
            jev = _AUX_JOIN_EV[key] = torch.cuda.Event()
        jev.record(aux)
        side.wait_event(jev)
    with torch.cuda.stream(side):
        fn()
    _SIDE_DIRTY.add(key)


# --------------------------------------------------------------------------- norm
def gn_silu_fwd(x, gamma, beta, groups, *, chan_add=None, eps=1e-5, act=True, out=None):
    ldx = _act(x, "x")
    Cc = x.shape[-1]
    B = x.shape[0]
    HW = x.numel() // (B * Cc)
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    stats = torch.empty((B, groups, 2), dtype=torch.float32, device=x.device)
    L = _lib.lib()
    if _is32(x):
        check(L.dm_gn_silu_fwd_f32(_p(x), ldx, _p(_f32(chan_add, "chan_add")), _p(_f32(gamma, "gamma")),
                                   _p(_f32(beta, "beta")), _p(out), _act(out, "out"), _p(stats), B, HW, Cc, groups,
                                   eps, int(act), None, 0, _stream()), "dm_gn_silu_fwd_f32")
        return out, stats
    ws = workspace(L.dm_gn_workspace_bytes(B, HW, Cc, groups), x.device)
    check(L.dm_gn_silu_fwd(_p(x), ldx, _p(_f32(chan_add, "chan_add")), _p(_f32(gamma, "gamma")),
                           _p(_f32(beta, "beta")), _p(out), _act(out, "out"), _p(stats), B, HW, Cc, groups,
                           eps, int(act), _p(ws), ws.numel(), _stream()), "dm_gn_silu_fwd")
    return out, stats


def gn_prepare(x, gamma, beta, groups, *, chan_add=None, eps=1e-5):
    """Statistics of GroupNorm(x) in one pass over x -> (stats [B,G,2] = {mean, rstd}, scale [B,C], shift [B,C]);
    pass (scale, shift) as ``norm_in`` to conv3x3 / wgrad (dm_gn_prepare, include/dm_hip.h)."""
    ldx = _act(x, "x")
    Cc = x.shape[-1]
    B = x.shape[0]
    HW = x.numel() // (B * Cc)
    stats = torch.empty((B, groups, 2), dtype=torch.float32, device=x.device)
    ss = torch.empty((2, B, Cc), dtype=torch.float32, device=x.device)
    L = _lib.lib()
    ws = workspace(L.dm_gn_workspace_bytes(B, HW, Cc, groups), x.device)
    check(L.dm_gn_prepare(_p(x), ldx, _p(_f32(chan_add, "chan_add")), _p(_f32(gamma, "gamma")), _p(_f32(beta, "beta")),
                          _p(stats), _p(ss[0]), _p(ss[1]), B, HW, Cc, groups, eps, _p(ws), ws.numel(), _stream()),
          "dm_gn_prepare")
    return stats, (ss[0], ss[1])


def gn_silu_bwd(x, dy, gamma, beta, stats, groups, dgamma, dbeta, *, chan_add=None, add_in=None, act=True,
                drop_p=0.0, drop_seed=0, out=None, colsum_out=None):
    """colsum_out: optional fp32 [B, C], += per-sample channel sums of the returned dx (same pass)."""
    ldx, lddy = _act(x, "x"), _act(dy, "dy")
    Cc = x.shape[-1]
    B = x.shape[0]
    HW = x.numel() // (B * Cc)
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    L = _lib.lib()
    if _is32(x):
        ws = workspace(L.dm_gn_f32_workspace_bytes(B, Cc), x.device)
        check(L.dm_gn_silu_bwd_f32(_p(x), ldx, _p(_f32(chan_add, "chan_add")), _p(dy), lddy, _p(gamma), _p(beta),
                                   _p(_f32(stats, "stats")), _p(add_in), 0 if add_in is None else _act(add_in, "add_in"),
                                   _p(out), _act(out, "out"), _p(_f32(dgamma, "dgamma")), _p(_f32(dbeta, "dbeta")),
                                   B, HW, Cc, groups, int(act), float(drop_p), int(drop_seed),
                                   _p(_f32(colsum_out, "colsum_out")), _p(ws), ws.numel(), _stream()), "dm_gn_silu_bwd_f32")
        return out
    ws = workspace(L.dm_gn_workspace_bytes(B, HW, Cc, groups), x.device)
    check(L.dm_gn_silu_bwd(_p(x), ldx, _p(_f32(chan_add, "chan_add")), _p(dy), lddy, _p(gamma), _p(beta),
                           _p(_f32(stats, "stats")), _p(add_in), 0 if add_in is None else _act(add_in, "add_in"),
                           _p(out), _act(out, "out"), _p(_f32(dgamma, "dgamma")), _p(_f32(dbeta, "dbeta")),
                           B, HW, Cc, groups, int(act), float(drop_p), int(drop_seed), _p(_f32(colsum_out, "colsum_out")),
                           _p(ws), ws.numel(),
                           _stream()), "dm_gn_silu_bwd")
    return out


# --------------------------------------------------------------------------- attention
def _attn_args(q, k, v, o, lse, heads, drop_p, drop_seed):
    B, Nq, Cc = q.shape
    Nk = k.shape[1]
    hd = Cc // heads
    a = AttnArgs()
    a.q, a.ldq, a.k, a.ldk, a.v, a.ldv = _p(q), _act(q, "q"), _p(k), _act(k, "k"), _p(v), _act(v, "v")
    a.o, a.ldo, a.lse = _p(o), (_act(o, "o") if o is not None else 0), _p(_f32(lse, "lse"))
    a.B, a.heads, a.Nq, a.Nk, a.hd = B, heads, Nq, Nk, hd
    a.scale = 1.0 / (hd ** 0.5)
    a.drop_p, a.drop_seed = float(drop_p), int(drop_seed)
    return a


def attn_fwd(q, k, v, heads, *, drop_p=0.0, drop_seed=0, want_o=True):
    """q: [B,Nq,C], k/v: [B,Nk,C] (views into a fused qkv buffer are fine) -> (o [B,Nq,C], lse [B,heads,Nq]).
    ``want_o=False`` (one key only): just the log-sum-exp, o is returned as None (dm_hip.h)."""
    B, Nq, Cc = q.shape
    o = torch.empty((B, Nq, Cc), dtype=q.dtype, device=q.device) if want_o else None
    lse = torch.empty((B, heads, Nq), dtype=torch.float32, device=q.device)
    a = _attn_args(q, k, v, o, lse, heads, drop_p, drop_seed)
    if _is32(q):
        check(_lib.lib().dm_attn_fwd_f32(C.byref(a), _stream()), "dm_attn_fwd_f32")
        return o, lse
    check(_lib.lib().dm_attn_fwd(C.byref(a), _stream()), "dm_attn_fwd")
    return o, lse


def attn_bwd(q, k, v, o, lse, d_o, heads, *, dq=None, dk=None, dv=None, drop_p=0.0, drop_seed=0, single_pass=True):
    B, Nq, Cc = q.shape
    Nk = k.shape[1]
    dq = torch.empty((B, Nq, Cc), dtype=q.dtype, device=q.device) if dq is None else dq
    dk = torch.empty((B, Nk, Cc), dtype=q.dtype, device=q.device) if dk is None else dk
    dv = torch.empty((B, Nk, Cc), dtype=q.dtype, device=q.device) if dv is None else dv
    # scratch: rowsum(dO*O) [B,heads,Nq]; the few-token path reuses it as the [2,B,Nk,C] fp32 dK/dV reduction
    delta = torch.empty(max(B * heads * Nq, 2 * B * Nk * Cc, 32 * B * Cc if Nk == 1 else 0), dtype=torch.float32,
                        device=q.device)
    a = _attn_args(q, k, v, o, lse, heads, drop_p, drop_seed)
    a.d_o, a.lddo = _p(d_o), _act(d_o, "d_o")
    a.dq, a.lddq, a.dk, a.lddk, a.dv, a.lddv = _p(dq), _act(dq, "dq"), _p(dk), _act(dk, "dk"), _p(dv), _act(dv, "dv")
    a.delta = _p(delta)
    if _is32(q):
        check(_lib.lib().dm_attn_bwd_f32(C.byref(a), _stream()), "dm_attn_bwd_f32")
        return dq, dk, dv
    nb = _lib.lib().dm_attn_bwd_workspace_bytes(B, heads, Nq, Nk, Cc // heads)
    if nb > 0 and single_pass:   # single-pass backward: one fp32 dq partial per key split
        ws = workspace(nb, q.device)
        a.workspace, a.ws_bytes = _p(ws), ws.numel()
    check(_lib.lib().dm_attn_bwd(C.byref(a), _stream()), "dm_attn_bwd")
    return dq, dk, dv


def attn_onekey_fwd(v, Nq, heads, *, drop_p=0.0, drop_seed=0):
    """One context token, no log-sum-exp wanted: o[b, q] = dropout(v[b, 0]) for every query (the q-less form of
    dm_attn_fwd, include/dm_hip.h).  v: [B,1,C] -> o [B,Nq,C]."""
    B, Nk, Cc = v.shape
    assert Nk == 1
    o = torch.empty((B, Nq, Cc), dtype=BF16, device=v.device)
    a = AttnArgs()
    a.v, a.ldv, a.o, a.ldo = _p(v), _act(v, "v"), _p(o), Cc
    a.B, a.heads, a.Nq, a.Nk, a.hd = B, heads, Nq, 1, Cc // heads
    a.scale = 1.0 / ((Cc // heads) ** 0.5)
    a.drop_p, a.drop_seed = float(drop_p), int(drop_seed)
    check(_lib.lib().dm_attn_fwd(C.byref(a), _stream()), "dm_attn_fwd")
    return o


def attn_onekey_bwd(v, o, d_o, heads, *, drop_p=0.0, drop_seed=0):
    """Backward of attn_onekey_fwd: dv [B,1,C] = sum_q dropout-mask * dO (dq and dk are identically zero)."""
    B, Nk, Cc = v.shape
    Nq = o.shape[1]
    dv = torch.empty((B, 1, Cc), dtype=BF16, device=v.device)
    delta = torch.empty(32 * B * Cc, dtype=torch.float32, device=v.device)   # per-block partials of dV (dm_hip.h)
    a = AttnArgs()
    a.v, a.ldv, a.o, a.ldo = _p(v), _act(v, "v"), _p(o), _act(o, "o")
    a.d_o, a.lddo, a.dv, a.lddv, a.delta = _p(d_o), _act(d_o, "d_o"), _p(dv), Cc, _p(delta)
    a.B, a.heads, a.Nq, a.Nk, a.hd = B, heads, Nq, 1, Cc // heads
    a.scale = 1.0 / ((Cc // heads) ** 0.5)
    a.drop_p, a.drop_seed = float(drop_p), int(drop_seed)
    check(_lib.lib().dm_attn_bwd(C.byref(a), _stream()), "dm_attn_bwd")
    return dv


def cross1_ok(heads: int, Cc: int) -> bool:
    """Shapes dm_cross1_fwd / dm_cross1_bwd take (include/dm_hip.h)."""
    return heads == 8 and Cc in (64, 128, 256, 512, 1024)


def _cross1_args(v, w_out, N, heads, drop_p, drop_seed):
    B, Nk, Cc = v.shape
    assert Nk == 1 and tuple(w_out.shape) == (Cc, Cc)
    a = Cross1Args()
    a.v, a.ldv, a.w_out = _p(v), _act(v, "v"), _p(_f32(w_out, "w_out"))
    a.B, a.N, a.heads, a.C = B, N, heads, Cc
    a.drop_p, a.drop_seed = float(drop_p), int(drop_seed)
    return a


def cross1_fwd(v, w_out, N, heads, *, res=None, out=None, drop_p=0.0, drop_seed=0):
    """One context token: y[b, n] = res[b, n] + to_out(dropout(v[b]))  in one pass (dm_cross1_fwd, include/dm_hip.h).
    v: [B,1,C] bf16 (to_v of the token), w_out: the fp32 [C,C] to_out weight, res / out: [B,N,C] bf16."""
    B, _, Cc = v.shape
    y = torch.empty((B, N, Cc), dtype=BF16, device=v.device) if out is None else out
    u = torch.empty((B, heads, Cc), dtype=torch.float32, device=v.device)
    a = _cross1_args(v, w_out, N, heads, drop_p, drop_seed)
    if res is not None:
        a.res, a.ldr = _p(res), _act(res, "res")
    a.y, a.ldy, a.u = _p(y), _act(y, "y"), _p(u)
    check(_lib.lib().dm_cross1_fwd(C.byref(a), _stream()), "dm_cross1_fwd")
    return y


def cross1_bwd(dy, v, w_out, heads, *, dw_out=None, drop_p=0.0, drop_seed=0):
    """Backward of cross1_fwd: returns dv [B,1,C] bf16 and accumulates the to_out weight gradient into ``dw_out``
    (fp32 [C,C]); the gradient with respect to ``res`` is ``dy`` itself."""
    B, _, Cc = v.shape
    N = dy.shape[1]
    dv = torch.empty((B, 1, Cc), dtype=BF16, device=v.device)
    u = torch.empty((B, heads, Cc), dtype=torch.float32, device=v.device)
    ws = workspace(_lib.lib().dm_cross1_bwd_workspace_bytes(B, heads, Cc), v.device)
    a = _cross1_args(v, w_out, N, heads, drop_p, drop_seed)
    a.dy, a.lddy, a.u, a.dv, a.lddv = _p(dy), _act(dy, "dy"), _p(u), _p(dv), Cc
    a.dw_out = _p(_f32(dw_out, "dw_out"))
    a.workspace, a.ws_bytes = _p(ws), ws.numel()
    check(_lib.lib().dm_cross1_bwd(C.byref(a), _stream()), "dm_cross1_bwd")
    return dv


# --------------------------------------------------------------------------- layout / elementwise
def nchw_to_nhwc(x: torch.Tensor, cpad: int = 0, out=None) -> torch.Tensor:
    """NCHW fp32 -> NHWC in the compute mode's activation dtype (ACT), channels zero-padded to ``cpad``."""
    B, Cc, H, W = x.shape
    x = _f32(x.contiguous(), "x")
    cpad = cpad or ((Cc + 7) // 8 * 8)
    if out is None:
        out = torch.empty((B, H, W, cpad), dtype=ACT[0], device=x.device)
    if _is32(out):
        check(_lib.lib().dm_nchw_to_nhwc_f32(_p(x), _p(out), B, Cc, H * W, _act(out, "out"), cpad, _stream()),
              "dm_nchw_to_nhwc_f32")
        return out
    check(_lib.lib().dm_nchw_f32_to_nhwc_bf16(_p(x), _p(out), B, Cc, H * W, _act(out, "out"), cpad, _stream()),
          "dm_nchw_f32_to_nhwc_bf16")
    return out


def nhwc_to_nchw(x: torch.Tensor, channels: int = 0) -> torch.Tensor:
    B, H, W, Cc = x.shape
    channels = channels or Cc
    assert x.stride(-1) == 1
    out = torch.empty((B, channels, H, W), dtype=torch.float32, device=x.device)
    check(_lib.lib().dm_nhwc_to_nchw_f32(_p(x), int(x.dtype == torch.float32), x.stride(-2), _p(out), B, channels,
                                         H * W, _stream()), "dm_nhwc_to_nchw_f32")
    return out


def to_bf16(x: torch.Tensor) -> torch.Tensor:
    """fp32 -> the compute mode's activation dtype (a cast to bf16; the tensor itself in the fp32 mode)."""
    x = _f32(x.contiguous(), "x")
    if ACT[0] == F32:
        return x
    out = torch.empty(x.shape, dtype=BF16, device=x.device)
    check(_lib.lib().dm_cast_f32_to_bf16(_p(x), _p(out), x.numel(), _stream()), "dm_cast_f32_to_bf16")
    return out


def to_f32(x: torch.Tensor) -> torch.Tensor:
    if x.dtype == F32:
        return x
    assert x.dtype == BF16 and x.is_contiguous()
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(_lib.lib().dm_cast_bf16_to_f32(_p(x), _p(out), x.numel(), _stream()), "dm_cast_bf16_to_f32")
    return out


def add(a, b, out=None):
    lda = _act(a, "a")
    if out is None:
        out = torch.empty(a.shape, dtype=a.dtype, device=a.device)
    Cc = a.shape[-1]
    if _is32(a):
        check(_lib.lib().dm_add_f32(_p(a), lda, _p(b), 0 if b is None else _act(b, "b"), _p(out), _act(out, "out"),
                                    a.numel() // Cc, Cc, _stream()), "dm_add_f32")
        return out
    check(_lib.lib().dm_add_bf16(_p(a), lda, _p(b), 0 if b is None else _act(b, "b"), _p(out), _act(out, "out"),
                                 a.numel() // Cc, Cc, _stream()), "dm_add_bf16")
    return out


def silu_bwd(dy, x, out=None):
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    Cc = x.shape[-1]
    if _is32(x):
        check(_lib.lib().dm_silu_bwd_f32(_p(dy), _act(dy, "dy"), _p(x), _act(x, "x"), _p(out), _act(out, "out"),
                                         x.numel() // Cc, Cc, _stream()), "dm_silu_bwd_f32")
        return out
    check(_lib.lib().dm_silu_bwd_bf16(_p(dy), _act(dy, "dy"), _p(x), _act(x, "x"), _p(out), _act(out, "out"),
                                      x.numel() // Cc, Cc, _stream()), "dm_silu_bwd_bf16")
    return out


def film_fwd(h, gamma_beta, drop_p=0.0, drop_seed=0):
    """dropout(gamma * h + beta) for NHWC bf16 h [B, H, W, C] and fp32 rows gamma_beta [B, 2C] (unet.py:32-42,105)."""
    B, Cc = h.shape[0], h.shape[-1]
    HW = h.numel() // Cc // B
    out = torch.empty(h.shape, dtype=h.dtype, device=h.device)
    gb = _f32(gamma_beta, "gamma_beta")
    assert gb.shape == (B, 2 * Cc) and gb.is_contiguous()
    if _is32(h):
        check(_lib.lib().dm_film_fwd_f32(_p(h), _act(h, "h"), _p(gb), _p(out), _act(out, "out"), B, HW, Cc, float(drop_p),
                                         int(drop_seed), _stream()), "dm_film_fwd_f32")
        return out
    check(_lib.lib().dm_film_fwd(_p(h), _act(h, "h"), _p(gb), _p(out), _act(out, "out"), B, HW, Cc, float(drop_p),
                                 int(drop_seed), _stream()), "dm_film_fwd")
    return out


def film_bwd(dh, h0, gamma_beta):
    """Returns (dh0 = gamma * dh, dgamma [B, C] fp32 = sum over pixels of dh * h0)."""
    B, Cc = h0.shape[0], h0.shape[-1]
    HW = h0.numel() // Cc // B
    dh0 = torch.empty(h0.shape, dtype=h0.dtype, device=h0.device)
    if _is32(h0):
        dgamma = torch.empty((B, Cc), dtype=torch.float32, device=h0.device)
        check(_lib.lib().dm_film_bwd_f32(_p(dh), _act(dh, "dh"), _p(h0), _act(h0, "h0"), _p(_f32(gamma_beta, "gamma_beta")),
                                         _p(dh0), _act(dh0, "dh0"), _p(dgamma), B, HW, Cc, _stream()), "dm_film_bwd_f32")
        return dh0, dgamma
    nch = _lib.lib().dm_film_bwd_chunks(HW)
    part = torch.empty((nch, B * Cc), dtype=torch.float32, device=h0.device)
    gb = _f32(gamma_beta, "gamma_beta")
    check(_lib.lib().dm_film_bwd(_p(dh), _act(dh, "dh"), _p(h0), _act(h0, "h0"), _p(gb), _p(dh0), _act(dh0, "dh0"),
                                 _p(part), B, HW, Cc, _stream()), "dm_film_bwd")
    if nch == 1:
        return dh0, part.view(B, Cc)
    dgamma = torch.zeros((B, Cc), dtype=torch.float32, device=h0.device)
    batchsum(part, dgamma.view(-1))   # chunk partials in fixed order
    return dh0, dgamma


def colsum(x, out, nbatch, accumulate=False):
    """out[b][c] (+)= sum over the batch's rows of x (x: [nbatch*rows, C] view)."""
    Cc = x.shape[-1]
    rows = x.numel() // Cc // nbatch
    if _is32(x):
        check(_lib.lib().dm_colsum_f32(_p(x), _act(x, "x"), _p(_f32(out, "out")), nbatch, rows, Cc, int(accumulate),
                                       _stream()), "dm_colsum_f32")
        return out
    check(_lib.lib().dm_colsum_bf16(_p(x), _act(x, "x"), _p(_f32(out, "out")), nbatch, rows, Cc, int(accumulate),
                                    _stream()), "dm_colsum_bf16")
    return out


def batchsum(inp, out):
    B, Cc = inp.shape
    check(_lib.lib().dm_batchsum_f32(_p(_f32(inp, "in")), _p(_f32(out, "out")), B, Cc, _stream()), "dm_batchsum_f32")


def batchsum2_cast(inp, out1, out2=None, want_bf16=True):
    """out1 += sum_b inp[b]; out2 += the same (optional); returns inp as bf16 (optional) -- one launch."""
    B, Cc = inp.shape
    inp = inp.contiguous()
    f32m = ACT[0] == F32      # fp32 mode: the "cast" operand is the fp32 input itself
    cast = torch.empty((B, Cc), dtype=BF16, device=inp.device) if (want_bf16 and not f32m) else None
    check(_lib.lib().dm_batchsum2_cast_f32(_p(_f32(inp, "in")), _p(_f32(out1, "out1")),
                                           _p(_f32(out2, "out2")), _p(cast), B, Cc, _stream()), "dm_batchsum2_cast_f32")
    return inp if (want_bf16 and f32m) else cast


# --------------------------------------------------------------------------- diffusion math
def qsample_vtarget(x, e, abar_t):
    B = x.shape[0]
    chw = x.numel() // B
    xt, v = torch.empty_like(x), torch.empty_like(x)
    check(_lib.lib().dm_qsample_vtarget(_p(_f32(x, "x")), _p(_f32(e, "e")), _p(_f32(abar_t, "abar_t")), _p(xt), _p(v),
                                        B, chw, _stream()), "dm_qsample_vtarget")
    return xt, v


def mse_loss_fwd_bwd(out, v, w, grad_scale=1.0, want_grad=True):
    """loss = sum_b w[b] * sum (out-v)^2 ; dout = 2 w[b] grad_scale (out-v). Returns (loss[1], dout)."""
    B = out.shape[0]
    chw = out.numel() // B
    L = _lib.lib()
    ws = workspace(L.dm_mse_workspace_bytes(B), out.device)
    loss = torch.empty(1, dtype=torch.float32, device=out.device)
    dout = torch.empty_like(out) if want_grad else None
    check(L.dm_mse_loss_fwd_bwd(_p(_f32(out, "out")), _p(_f32(v, "v")), _p(_f32(w, "w")), float(grad_scale), _p(dout),
                                _p(loss), B, chw, _p(ws), ws.numel(), _stream()), "dm_mse_loss_fwd_bwd")
    return loss, dout


def mse_loss_fwd_bwd_noalloc(out, v, w, loss, dout=None, grad_scale=1.0):
    """dm_mse_loss_fwd_bwd_noalloc (include/dm_hip.h): the caller brings every buffer (``loss`` [1] fp32, optional
    ``dout``); the library's static device scratch holds the partial sums.  Nothing is allocated here."""
    B = out.shape[0]
    chw = out.numel() // B
    check(_lib.lib().dm_mse_loss_fwd_bwd_noalloc(_p(_f32(out, "out")), _p(_f32(v, "v")), _p(_f32(w, "w")),
                                                 float(grad_scale), _p(dout), _p(_f32(loss, "loss")), B, chw,
                                                 _stream()), "dm_mse_loss_fwd_bwd_noalloc")
    return loss, dout


def cfg_ddpm_step(z, v_u, v_c, noise, coef, out=None):
    B = z.shape[0]
    chw = z.numel() // B
    if out is None:
        out = torch.empty_like(z)
    check(_lib.lib().dm_cfg_ddpm_step(_p(_f32(z, "z")), _p(_f32(v_u, "v_u")), _p(_f32(v_c, "v_c")),
                                      _p(_f32(noise, "noise")), _p(_f32(coef, "coef")), _p(out), B, chw, _stream()),
          "dm_cfg_ddpm_step")
    return out


def cfg_dpmpp2m_step(z, v_u, v_c, x0_prev, coef, out=None, x0_out=None):
    """One DPM-Solver++(2M) step; returns (z', x0).  ``x0_prev`` is read only by rows whose second-order flag is set."""
    B = z.shape[0]
    chw = z.numel() // B
    if out is None:
        out = torch.empty_like(z)
    if x0_out is None:
        x0_out = torch.empty_like(z)
    check(_lib.lib().dm_cfg_dpmpp2m_step(_p(_f32(z, "z")), _p(_f32(v_u, "v_u")), _p(_f32(v_c, "v_c")),
                                         _p(_f32(x0_prev, "x0_prev")), _p(_f32(coef, "coef")), _p(out), _p(x0_out), B,
                                         chw, _stream()), "dm_cfg_dpmpp2m_step")
    return out, x0_out


def grad_norm_clipcoef(g, max_norm, pre_scale=1.0, out=None):
    L = _lib.lib()
    ws = workspace(L.dm_sqnorm_workspace_bytes(), g.device)
    if out is None:
        out = torch.empty(2, dtype=torch.float32, device=g.device)
    check(L.dm_grad_norm_clipcoef(_p(_f32(g, "g")), g.numel(), float(max_norm), float(pre_scale), _p(out), _p(ws),
                                  ws.numel(), _stream()), "dm_grad_norm_clipcoef")
    return out


def adamw_ema_step(p, g, m, v, ema, clip, *, lr, beta1, beta2, eps, wd, step, ema_decay, pre_scale=1.0, adamw=True):
    check(_lib.lib().dm_adamw_ema_step(_p(_f32(p, "p")), _p(_f32(g, "g")), _p(_f32(m, "m")), _p(_f32(v, "v")),
                                       _p(_f32(ema, "ema")), _p(clip), p.numel(), float(lr), float(beta1),
                                       float(beta2), float(eps), float(wd), int(step), float(ema_decay),
                                       float(pre_scale), int(adamw), _stream()), "dm_adamw_ema_step")


# --------------------------------------------------------------------------- device-resident optimizer schedule / input ring
def opt_state_new(device, step: int, sched_step: int) -> torch.Tensor:
    """The device struct of dm_opt_state_* (include/dm_hip.h), initialised to (step, sched_step)."""
    L = _lib.lib()
    st = torch.zeros(max(32, L.dm_opt_state_bytes()), dtype=torch.uint8, device=device)
    check(L.dm_opt_state_init(_p(st), int(step), int(sched_step), _stream()), "dm_opt_state_init")
    return st


def opt_state_advance(state, *, base_lr, warmup_steps, total_steps, beta1, beta2) -> None:
    check(_lib.lib().dm_opt_state_advance(_p(state), float(base_lr), int(warmup_steps), int(total_steps), float(beta1),
                                          float(beta2), _stream()), "dm_opt_state_advance")


def adamw_ema_step_dev(p, g, m, v, ema, clip, state, *, beta1, beta2, eps, wd, ema_decay, pre_scale=1.0, adamw=True):
    check(_lib.lib().dm_adamw_ema_step_dev(_p(_f32(p, "p")), _p(_f32(g, "g")), _p(_f32(m, "m")), _p(_f32(v, "v")),
                                           _p(_f32(ema, "ema")), _p(clip), p.numel(), _p(state), float(beta1), float(beta2),
                                           float(eps), float(wd), float(ema_decay), float(pre_scale), int(adamw), _stream()),
          "dm_adamw_ema_step_dev")


def ring_fetch(ring: torch.Tensor, dst: torch.Tensor, counter: torch.Tensor) -> None:
    """dst <- ring[counter % len(ring)] (ring: [nslots, ...] contiguous, one slot shaped like dst; slot bytes % 16 == 0)."""
    nslots = ring.shape[0]
    nbytes = dst.numel() * dst.element_size()
    assert ring.is_contiguous() and dst.is_contiguous() and ring.numel() * ring.element_size() == nslots * nbytes
    check(_lib.lib().dm_ring_fetch(_p(ring), _p(dst), nbytes, nslots, _p(counter), _stream()), "dm_ring_fetch")
