"""CPU (no GPU): host logic -- C-ABI library loads and exports every declared symbol, schedule tables
bit-exact, state_dict compatibility, sampler coefficients, LR schedule, flat-parameter re-homing and
the data-parallel gradient buckets over gloo (world_size 2)."""
import os
import re
import sys

import pytest
import torch

from conftest import ROOT, load_golden
from oracle import diffusion_ref as dref
from oracle import unet_ref as oref


def test_library_exports_every_declared_symbol():
    from diffusion_model_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "dm_hip.h")).read()
    declared = set(re.findall(r"\b(dm_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    h = _lib.lib()  # raises if the .so is missing or a bound symbol is not exported
    for name in declared:
        assert hasattr(h, name), f"{name} declared in include/dm_hip.h but not exported"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert h.dm_abi_version() == 4 and h.dm_target_arch() == b"gfx950"
    # pure host helpers may be called without a GPU
    assert h.dm_packed_weight_bytes(64, 3, 9) == 64 * 64 * 9 * 2
    assert h.dm_wgrad_workspace_bytes(64, 96, 96, 64, 64, 9, 0) > 0


def test_no_cpu_fallback():
    from diffusion_model_amd.unet import UNET
    m = UNET(input_channels=3, output_channels=3, context_dim=512)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 32, 32), torch.tensor([1]), torch.zeros(1, 512))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "diffusion_model_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{f} imports the oracle"


def test_schedule_tables_bit_exact():
    from diffusion_model_amd.utils import DDPM_beta_t_cosine_scheduler, DDPM_beta_t_linear_scheduler
    for T in (500, 1000):
        for bmin, nm in ((1e-4, "stl10"), (0.0, "mnist")):
            g = load_golden(f"schedule_cos_T{T}_{nm}")
            s = DDPM_beta_t_cosine_scheduler(T, beta_min=bmin)
            assert torch.equal(s.beta_t.cpu(), g["beta"])
            assert torch.equal(s.alphas.cpu(), g["alpha"])
            assert torch.equal(s.alpha_t.cpu(), g["alpha_bar"])
    g = load_golden("schedule_lin_T1000")
    s = DDPM_beta_t_linear_scheduler(1000)
    assert torch.equal(s.beta_t.cpu(), g["beta"]) and torch.equal(s.alpha_t.cpu(), g["alpha_bar"])
    a, b = s.call(200)
    assert round(float(a), 4) == 0.6563 and round(float(b), 4) == 0.0041  # notebook cell 3


def test_sinusoid_table_and_state_dict():
    from diffusion_model_amd.unet import UNET, SinusoidalEmbeddings
    g = load_golden("sinusoid")
    assert torch.equal(SinusoidalEmbeddings(1000, 256)(g["rows"]), g["emb"])
    assert torch.equal(SinusoidalEmbeddings(100, 64)(20), g["nb_cell7"])
    kw = dict(input_channels=3, output_channels=3, context_dim=512)
    want = oref.param_shapes(oref.UNetSpec(**kw))
    torch.manual_seed(0)
    m = UNET(**kw)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert list(got) == list(want) and got == want
    assert float(m.output_conv.weight.abs().max()) == 0.0  # zero-init (reference unet.py:324-326)
    ig = load_golden("unet_stl10_init")  # same seed => the reference's own initial weights
    for k, v in m.state_dict().items():
        if k != "output_conv.weight":
            assert float(torch.linalg.vector_norm(v)) == pytest.approx(float(ig["wn." + k]), rel=1e-6), k
    mn = UNET(Attentions=[False] * 6)
    assert list(mn.state_dict()) == list(oref.param_shapes(oref.UNetSpec(Attentions=[False] * 6)))


def test_ema_state_dict_keys_and_update():
    from diffusion_model_amd.ema import ModelEmaV2
    from diffusion_model_amd.unet import UNET
    torch.manual_seed(0)
    m = UNET(Attentions=[False] * 6)
    ema = ModelEmaV2(m, decay=0.9)
    assert list(ema.state_dict()) == ["module." + k for k in m.state_dict()]
    assert not ema.module.training
    before = {k: v.clone() for k, v in ema.module.state_dict().items()}
    with torch.no_grad():
        for p in m.parameters():
            p.add_(1.0)
    ema.update(m)
    ref = {k: v.clone() for k, v in before.items()}
    dref.ema_update(ref, {k: v for k, v in m.state_dict().items()}, 0.9)
    for k, v in ema.module.state_dict().items():
        assert torch.allclose(v, ref[k], atol=1e-7)


def test_sampler_coefficients_match_oracle_scalars():
    from diffusion_model_amd.sampler import step_coefficients
    beta, _, abar = dref.cosine_schedule(500)
    z = torch.randn(2, 3, 8, 8)
    vu, vc, n = torch.randn_like(z), torch.randn_like(z), torch.randn_like(z)
    for t, eta, g in ((499, 1.0, 6.0), (2, 1.0, 6.0), (1, 1.0, 6.0), (37, 0.0, 1.5), (0, 1.0, 6.0)):
        c = step_coefficients(beta, abar, t, g, eta, 2, final=(t == 0))
        sig, sa, ria, bs, ns, gg, cl = (c[0, i] for i in range(7))
        eps = sig * z + sa * (vu + gg * (vc - vu))
        got = ria * (z - bs * eps) + ns * n
        if cl:
            got = got.clamp(-1, 1)
        eps_ref = dref.cfg_eps_from_v(z, vu, vc, g, abar[torch.tensor([t, t])])
        ref = dref.ddpm_step(z, t, eps_ref, beta, abar, eta, n) if t > 0 else dref.final_step(z, eps_ref, beta, abar)
        assert torch.allclose(got, ref, rtol=1e-5, atol=1e-6), t
        assert (float(ns) != 0.0) == (t > 1 and eta > 0)


def test_lr_schedule_matches_oracle():
    from diffusion_model_amd.trainer import lr_lambda
    for s in (0, 1, 394, 395, 396, 4000, 7899):
        assert lr_lambda(s, 7900) == dref.lr_lambda(s, 7900)


def test_flat_params_rehoming_keeps_state_dict_and_values():
    from diffusion_model_amd.trainer import FlatParams, unet_stage_names
    from diffusion_model_amd.unet import UNET
    torch.manual_seed(0)
    m = UNET(input_channels=3, output_channels=3, context_dim=512)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    fp = FlatParams(m)
    assert list(m.state_dict()) == list(before)
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])
    assert fp.total >= sum(v.numel() for v in before.values()) and fp.total % 4 == 0
    # parameters alias the flat buffer, gradients alias the flat gradient
    p0 = next(m.parameters())
    fp.flat[0] = 123.0
    assert float(p0.reshape(-1)[0]) == 123.0
    # bucket ranges partition the buffer exactly
    total = 0
    for st, pref in unet_stage_names(m):
        total += sum(b - a for a, b in fp.ranges_for(pref))
    assert total == fp.total


def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        from diffusion_model_amd.trainer import FlatParams, GradBuckets, unet_stage_names
        from diffusion_model_amd.unet import UNET
        torch.manual_seed(0)  # identical init on every rank
        m = UNET(Attentions=[False] * 6)
        fp = FlatParams(m)
        bk = GradBuckets(fp, unet_stage_names(m))
        assert bk.world == world
        g = torch.Generator().manual_seed(100 + rank)  # rank-local "gradient"
        fp.grad.copy_(torch.randn(fp.total, generator=g))
        local = fp.grad.clone()
        for st, _ in unet_stage_names(m):  # reverse execution order, as the backward would call them
            bk.launch(st)
        bk.wait()
        # every rank must hold the SUM (the 1/world average rides in the optimizer's pre_scale)
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        want = sum(gathered)
        ok = bool(torch.allclose(fp.grad, want, atol=1e-5))
        err = float((fp.grad - want).abs().max())
        # the one-collective form (DM_DP_BUCKETS=1): nothing until the last stage, then the whole buffer
        os.environ["DM_DP_BUCKETS"] = "1"
        bk1 = GradBuckets(fp, unet_stage_names(m))
        fp.grad.copy_(local)
        names = [st for st, _ in unet_stage_names(m)]
        for st in names[:-1]:
            bk1.launch(st)
        ok = ok and not bk1.pending
        bk1.launch(names[-1])
        bk1.wait()
        ok = ok and bool(torch.allclose(fp.grad, want, atol=1e-5))
        q.put((rank, ok, max(err, float((fp.grad - want).abs().max()))))
    finally:
        dist.destroy_process_group()


def test_dp_gradient_buckets_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok, err in res:
        assert ok, f"rank {rank}: bucketed all-reduce differs from the gathered sum by {err}"


def _shard_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        from diffusion_model_amd.sampler import sample_sharded, shard_bounds
        g = torch.Generator().manual_seed(3)                      # same start noise / labels on every rank
        z, labels = torch.randn(5, 3, 4, 4, generator=g), torch.arange(5)
        seen = []

        def fn(zs, ls, scale=1.0):                                # stands in for CFGSampler.sample on the shard
            seen.append(zs.shape[0])
            return zs * scale + ls.view(-1, 1, 1, 1).float()

        out = sample_sharded(fn, z, labels, scale=2.0)
        want = z * 2.0 + labels.view(-1, 1, 1, 1).float()
        lo, hi = shard_bounds(5, rank, world)
        q.put((rank, bool(torch.equal(out, want)), seen == [hi - lo], (lo, hi)))
    finally:
        dist.destroy_process_group()


def test_sharded_sampling_gloo_world2():
    """SURVEY 8e, sampling: contiguous shards per rank, no collective in the loop, one final gather (ragged: 5 over 2)."""
    import torch.multiprocessing as mp
    from diffusion_model_amd.sampler import sample_sharded, shard_bounds
    assert [shard_bounds(704, r, 8) for r in (0, 7)] == [(0, 88), (616, 704)]
    assert [shard_bounds(5, r, 2) for r in range(2)] == [(0, 3), (3, 5)]
    assert [shard_bounds(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    z = torch.ones(3, 1, 2, 2)
    assert torch.equal(sample_sharded(lambda a, b: a + 1, z, torch.zeros(3)), z + 1)   # no process group: pass-through
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[3] for r in res] == [(0, 3), (3, 5)]
    for rank, same, one_call, _ in res:
        assert same and one_call, f"rank {rank}: gathered batch differs from the single-process result"


def _dp_init_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        from diffusion_model_amd import unet as U
        from diffusion_model_amd.ema import ModelEmaV2
        from diffusion_model_amd.trainer import Trainer
        torch.manual_seed(100 + rank)            # process-local RNG: the replicas START different
        m = U.UNET(Attentions=[False] * 6)
        with torch.no_grad():
            m.output_conv.weight.normal_(0, 0.05)
        before = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone()
        tr = Trainer(m, None, ema=ModelEmaV2(m, 0.995), distributed=True, start_step=3)
        assert tr.world == world and tr.rank == rank
        flats = [torch.empty_like(tr.fp.flat) for _ in range(world)]
        dist.all_gather(flats, tr.fp.flat)
        emas = [torch.empty_like(tr.ema_flat) for _ in range(world)]
        dist.all_gather(emas, tr.ema_flat)
        befores = [torch.empty_like(before) for _ in range(world)]
        dist.all_gather(befores, before)
        same = all(torch.equal(f, flats[0]) for f in flats) and all(torch.equal(e, emas[0]) for e in emas)
        # the module's parameters alias the flat buffer: they carry rank 0's values now, on every rank
        now = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
        aliased = torch.equal(now, befores[0])
        ema_follows = torch.equal(torch.cat([p.detach().reshape(-1) for p in tr.ema.module.parameters()]), befores[0])
        seeds = [None] * world
        dist.all_gather_object(seeds, [next(U._SEED) for _ in range(4)])
        q.put((rank, same, aliased and ema_follows, not torch.equal(befores[0], befores[1]), seeds))
    finally:
        dist.destroy_process_group()


def test_dp_trainer_init_identical_replicas_gloo_world2():
    """replicas built from different process-local RNG states continue from rank 0's parameters and EMA copy
    after Trainer init (one broadcast per flat buffer); dropout seed streams differ between the ranks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_init_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, same, aliased, started_different, seeds in res:
        assert started_different, "the test must start from different replicas"
        assert same, f"rank {rank}: flat parameter / EMA buffers differ between ranks after Trainer init"
        assert aliased, f"rank {rank}: module parameters do not carry rank 0's values"
        assert seeds[0] != seeds[1] and not (set(seeds[0]) & set(seeds[1])), "ranks must draw different dropout seeds"


def test_dropout_seed_stream_is_a_function_of_seed_rank_step():
    from diffusion_model_amd import unet as U
    a = U.DropoutSeeds(); a.reseed(7, 0, 0)
    b = U.DropoutSeeds(); b.reseed(7, 0, 0)
    c = U.DropoutSeeds(); c.reseed(7, 1, 0)
    d = U.DropoutSeeds(); d.reseed(7, 0, 5)     # resumed at optimizer step 5: not a replay of step 0
    sa, sb, sc, sd_ = ([next(s) for _ in range(8)] for s in (a, b, c, d))
    assert sa == sb and len(set(sa)) == 8
    assert not (set(sa) & set(sc)) and not (set(sa) & set(sd_))
    assert all(0 <= v < 2 ** 63 for v in sa)


def test_bench_launches_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus N` with no launcher environment starts N ranks itself (torch.distributed.run on
    127.0.0.1) before touching a GPU; with a launcher environment whose WORLD_SIZE differs it refuses to run."""
    import importlib
    import subprocess
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 0
        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit, match="WORLD_SIZE=2"):
        bench.main()
