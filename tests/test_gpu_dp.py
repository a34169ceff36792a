"""configs[2] on real kernels: the data-parallel training step with TWO (and FOUR, and configs[2]'s EIGHT) ranks.  One GPU box has one GPU and RCCL refuses
two ranks on one device, so the ranks share cuda:0 and exchange gradients over gloo -- the Trainer, the bucket hook on
the weight-gradient side stream (ordered behind the main AND the auxiliary stream, kernels.call_on_side), the broadcast
of the replicas and the fused clip + AdamW + EMA are exactly the N > 1 code path; only the transport differs from RCCL:
GradBuckets stages each bucket to pinned host memory itself, in stream order (trainer.GradBuckets.host_staged), and
reduces the HOST tensor -- torch's gloo-CUDA shim with its worker-thread stream synchronisation is not involved.
One attempt: a stall fails the test, with the ranks' Python stacks and the buckets' stage-event / stream state."""
import os
import tempfile
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
STL = dict(time_steps=500, input_channels=3, output_channels=3, label_embedding=True)


def _batch(seed, B=4, H=32):
    g = torch.Generator().manual_seed(seed)
    x = torch.clamp(torch.randn(B, 3, H, H, generator=g) * 0.5, -1, 1)
    e = torch.randn(B, 3, H, H, generator=g)
    t = torch.randint(0, 500, (B,), generator=g)
    ctx = torch.randn(B, 512, generator=g)
    return x, ctx, t, e


def _make(seed):
    from diffusion_model_amd.unet import UNET
    torch.manual_seed(seed)
    m = UNET(**STL, dropout_prob=0.0)
    with torch.no_grad():
        m.output_conv.weight.normal_(0, 0.05)
    return m.to("cuda:0").train()


def _worker(rank, world, port, q, side_wgrad):
    try:
        _worker_body(rank, world, port, q, side_wgrad)
    except BaseException:   # report instead of leaving the peer (and the parent) waiting
        import traceback
        q.put({"rank": rank, "error": traceback.format_exc()})
        raise


def _worker_body(rank, world, port, q, side_wgrad):
    import datetime
    import faulthandler
    import threading
    import torch.distributed as dist
    # a rank that hangs says where, before the parent gives up (the parent quotes the file in its message): its Python
    # stack, and -- from a watchdog thread -- whether the GPU work ahead of each bucket has drained (stage events, the
    # three streams) and how many collectives were issued / completed: "the streams never drain" and "the host exchange
    # never completes" are different bugs
    _trace = open(os.path.join(tempfile.gettempdir(), "dm_dp_rank%d_%d.trace" % (rank, port)), "w")
    faulthandler.dump_traceback_later(50, exit=False, file=_trace)
    state = {}

    def _watchdog():
        import time
        time.sleep(48)
        try:
            from diffusion_model_amd import kernels as K
            tr = state.get("tr")
            lines = ["rank %d watchdog" % rank]
            if tr is not None and tr.buckets is not None:
                lines.append(tr.buckets.describe())
            for name, tab in (("side", K._SIDE), ("aux", K._AUX)):
                for key, (st, _) in tab.items():
                    lines.append("%s stream dev %s idle=%s" % (name, key, st.query()))
            lines.append("main stream idle=%s" % torch.cuda.current_stream(0).query())
            _trace.write("\n".join(lines) + "\n")
            _trace.flush()
        except Exception as exc:   # diagnostics must not mask the stall itself
            _trace.write("watchdog failed: %r\n" % (exc,))
            _trace.flush()

    threading.Thread(target=_watchdog, daemon=True).start()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        from diffusion_model_amd import kernels as K
        from diffusion_model_amd.trainer import Trainer
        from diffusion_model_amd.utils import DDPM_beta_t_cosine_scheduler
        K.SIDE_WGRAD = bool(side_wgrad)               # False = DM_SIDE_WGRAD=0: weight gradients inline on the main / auxiliary stream
        model = _make(10 + rank)                      # different replicas on purpose: rank 0's must win
        tr = Trainer(model, DDPM_beta_t_cosine_scheduler(500), lr=1e-3, clip=1.0, accum=1, total_steps=100,
                     low_t_weight=0.01, distributed=True)
        state["tr"] = tr
        assert tr.buckets.host_staged
        p0 = tr.fp.flat.detach().cpu().clone()
        tr.defer_optimizer = True
        dev = torch.device("cuda:0")
        out = {"rank": rank, "p0": p0}
        for step in range(2):
            b = [v.to(dev) for v in _batch(100 + 10 * step + rank)]
            loss = tr.micro_step(*b)
            K.join_side(dev)
            tr.buckets.wait()
            torch.cuda.synchronize()
            out[f"g{step}"] = (tr.fp.grad.detach() / world).cpu().clone()
            out[f"loss{step}"] = float(loss)
            tr.optimizer_step()
            torch.cuda.synchronize()
            out[f"p{step + 1}"] = tr.fp.flat.detach().cpu().clone()
        q.put({k: (v.numpy() if torch.is_tensor(v) else v) for k, v in out.items()})   # by value: the worker exits
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("side_wgrad,world", [(True, 2), (False, 2), (True, 4), (True, 8)],
                         ids=["side_wgrad", "inline_wgrad", "four_ranks", "eight_ranks"])   # eight = configs[2]'s world size
def test_data_parallel_ranks_match_gradient_accumulation(side_wgrad, world):
    import queue as _queue
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sk:          # a port that is free right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, side_wgrad)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=90) for _ in procs]
    except _queue.Empty:
        codes = [p.exitcode for p in procs]
        for r in range(world):
            try:
                with open(os.path.join(tempfile.gettempdir(), "dm_dp_rank%d_%d.trace" % (r, port))) as f:
                    codes.append("rank %d: %s" % (r, f.read()[-2500:]))
            except OSError:
                pass
        for p in procs:
            p.kill()
        for p in procs:
            p.join(20)
        raise AssertionError("a data-parallel rank did not report within 90 s (exit codes %s)" % codes)
    for r in res:
        assert "error" not in r, r["error"]
    res = sorted(res, key=lambda r: r["rank"])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rs = [{k: (torch.from_numpy(v) if hasattr(v, 'dtype') else v) for k, v in r.items()} for r in res]
    r0 = rs[0]
    for r1 in rs[1:]:
        # replicas: the other ranks started from other inits and must hold rank 0's parameters after Trainer.__init__
        assert torch.equal(r0["p0"], r1["p0"])
        # the ranks reduce the same buckets and run the same fused optimizer: identical gradients and parameters, bit for bit
        for k in ("g0", "g1", "p1", "p2"):
            assert torch.equal(r0[k], r1[k]), k
    # ... and the reduced gradient is the gradient-accumulation gradient of one process over both ranks' batches
    from diffusion_model_amd.trainer import Trainer
    from diffusion_model_amd.utils import DDPM_beta_t_cosine_scheduler
    from diffusion_model_amd import kernels as K
    model = _make(10)
    tr = Trainer(model, DDPM_beta_t_cosine_scheduler(500), lr=1e-3, clip=1.0, accum=world, total_steps=100, low_t_weight=0.01)
    assert torch.equal(tr.fp.flat.detach().cpu(), r0["p0"])
    tr.defer_optimizer = True
    dev = torch.device("cuda:0")
    for step in range(2):
        for rank in range(world):
            tr.micro_step(*[v.to(dev) for v in _batch(100 + 10 * step + rank)])
        K.join_side(dev)
        torch.cuda.synchronize()
        g = tr.fp.grad.detach().cpu()
        ref = r0[f"g{step}"]
        rel = float((g - ref).norm() / ref.norm())
        assert rel < 1e-5, f"step {step}: reduced gradient differs from the accumulated one, rel-L2 {rel:.2e}"
        if step == 0:
            tr.optimizer_step()
            # the same gradient into the same fused clip + AdamW step (fp32 summation order is the only difference)
            d = (tr.fp.flat.detach().cpu() - r0["p1"]).abs().max()
            assert float(d) <= 2.5e-3, f"parameters after the first step differ by {float(d):.2e}"   # <= 2.5 lr (a sign flip of a noise-level gradient moves a weight by 2 lr)
            tr.fp.flat.copy_(r0["p1"].to(dev))     # continue from the ranks' parameters: step 2 compares gradients only
            from diffusion_model_amd.unet import invalidate_packed_weights, repack_all
            invalidate_packed_weights(); repack_all(tr.model)
