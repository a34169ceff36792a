#!/usr/bin/env python3
"""ORACLE tooling (build container only; the reference never travels): pin the diffusion-math oracle
(oracle/diffusion_ref.py) with the reference's OWN code.

utils.py / visualization.py / main_stl10.py / main.py of the reference cannot be imported here (their module-level
imports need torchvision, timm, lpips, diffusers, matplotlib: ordinary ModuleNotFoundError).  The functions this path
needs depend on ``torch`` only, so this script parses those files with ``ast``, takes the UNMODIFIED source text of

  * utils.py ............ DDPM_beta_t_cosine_scheduler, DDPM_beta_t_linear_scheduler, _cfg_predict_eps_from_v
  * main.py ............. DDPM_beta_t_cosine_scheduler (the MNIST variant: clip at 0), DDPM_beta_t_linear_scheduler
  * visualization.py .... _ddpm_step, run_reverse_process
  * main_stl10.py ....... the nested lr_lambda of train(), and the statements of train()'s batch loop from the label
                          drop to the loss (the hot loop's q-sample / v-target / loss arithmetic)

executes it in a namespace that provides what the missing imports would (torch, nn, math, typing names, a CPU
``device``, a ``config`` with the STL10 constants, a context table), and writes inputs + outputs as small fixtures under
tests/golden/ref_*.npz.  Only DATA is written: no reference text is stored in the repository.

Re-run:  PYTHONDONTWRITEBYTECODE=1 python3 oracle/gen_golden_diffusion.py
"""
import ast
import math
import os
import sys
import textwrap
import types
from typing import Callable, Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True


def _tree(fname):
    src = open(os.path.join(REF, fname)).read()
    return src, ast.parse(src)


def top_level(fname, name):
    """Unmodified source of the top-level def/class ``name`` of a reference file."""
    src, tree = _tree(fname)
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name == name:
            return ast.get_source_segment(src, node)
    raise KeyError(f"{fname}: {name}")


def nested_def(fname, outer, name):
    src, tree = _tree(fname)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == outer:
            for sub in ast.walk(node):
                if isinstance(sub, ast.FunctionDef) and sub.name == name:
                    return textwrap.dedent("\n".join(src.splitlines()[sub.lineno - 1: sub.end_lineno]))
    raise KeyError(f"{fname}: {outer}.{name}")


def loop_statements(fname, outer, first_target, last_test_attr):
    """Source of the consecutive statements of the batch loop of ``outer`` from the assignment to ``first_target``
    through the ``if`` whose test mentions ``last_test_attr`` (the loss), dedented."""
    src, tree = _tree(fname)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == outer)
    for loop in ast.walk(fn):
        if not isinstance(loop, ast.For):
            continue
        body = loop.body
        start = end = None
        for i, st in enumerate(body):
            if (start is None and isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Name)
                    and st.targets[0].id == first_target):
                start = i
            if start is not None and isinstance(st, ast.If) and last_test_attr in ast.get_source_segment(src, st.test):
                end = i
                break
        if start is not None and end is not None:
            lines = src.splitlines()[body[start].lineno - 1: body[end].end_lineno]   # verbatim, comments included
            return textwrap.dedent("\n".join(lines))
    raise KeyError(f"{fname}: loop statements of {outer}")


def with_block_statements(fname, outer, first_target, last_target):
    """Source of the consecutive statements inside ``outer``'s ``with`` block from the assignment to ``first_target``
    through the assignment to ``last_target``, dedented (verbatim lines, comments included)."""
    src, tree = _tree(fname)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == outer)
    for w in ast.walk(fn):
        if not isinstance(w, ast.With):
            continue
        start = end = None
        for i, st in enumerate(w.body):
            if isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Name):
                if start is None and st.targets[0].id == first_target:
                    start = i
                if start is not None and st.targets[0].id == last_target:
                    end = i
                    break
        if start is not None and end is not None:
            lines = src.splitlines()[w.body[start].lineno - 1: w.body[end].end_lineno]
            return textwrap.dedent("\n".join(lines))
    raise KeyError(f"{fname}: with-block statements of {outer}")


def base_ns():
    cfg = types.SimpleNamespace(UNCOND_ID=10, USING_LOW_T_REWARDING=True, LOW_T_REWARDING_WARMUP_EPOCHS=20,
                                LOW_T_REWARDING_WARMUP_WEIGHT=0.01)
    return dict(torch=torch, nn=nn, math=math, device=torch.device("cpu"), config=cfg, Callable=Callable, Dict=Dict,
                Optional=Optional, Sequence=Sequence, Tuple=Tuple)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KB)")


sys.path.insert(0, ROOT)
from oracle.stub_model import StubNet, stub_model  # noqa: E402  (a deterministic stand-in for the UNet, shared with the tests)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)

    # ---------------------------------------------------------------- schedules (utils.py and main.py variants)
    ns = base_ns()
    exec(top_level("utils.py", "DDPM_beta_t_cosine_scheduler"), ns)
    exec(top_level("utils.py", "DDPM_beta_t_linear_scheduler"), ns)
    for T in (500, 1000):
        s = ns["DDPM_beta_t_cosine_scheduler"](T)
        save(f"ref_schedule_cos_T{T}_utils", beta=s.beta_t, alpha=s.alphas, alpha_bar=s.alpha_t)
    lin = ns["DDPM_beta_t_linear_scheduler"](1000)
    a200, b200 = lin.call(200)
    assert round(float(a200), 4) == 0.6563 and round(float(b200), 4) == 0.0041   # testing.ipynb cell 3
    save("ref_schedule_lin_T1000_utils", beta=lin.beta_t, alpha_bar=lin.alpha_t)
    ns_m = base_ns()
    exec(top_level("main.py", "DDPM_beta_t_cosine_scheduler"), ns_m)
    for T in (500, 1000):
        s = ns_m["DDPM_beta_t_cosine_scheduler"](T)
        save(f"ref_schedule_cos_T{T}_main", beta=s.beta_t, alpha=s.alphas, alpha_bar=s.alpha_t)
    sched = ns["DDPM_beta_t_cosine_scheduler"](500)

    # ---------------------------------------------------------------- CFG combine + v -> eps (utils.py:19-41)
    tab = torch.from_numpy(np.load(os.path.join(REF, "artifacts/clip/clip_text_emb_stl10_vitb32.npz"),
                                   allow_pickle=False)["embeddings_norm"].astype(np.float32))
    ns["_labels_to_context"] = lambda labels: tab[labels.long()]
    exec(top_level("utils.py", "_cfg_predict_eps_from_v"), ns)
    g = torch.Generator().manual_seed(11)
    z = torch.randn(3, 3, 8, 8, generator=g)
    labels = torch.tensor([0, 7, 10])
    cfg = {}
    for t, gs in ((499, 6.0), (250, 8.0), (2, 6.0), (0, 0.0)):
        tt = torch.tensor([t])
        eps = ns["_cfg_predict_eps_from_v"](StubNet(), sched, z, tt, labels, gs)
        v_u = stub_model(z, tt, tab[torch.full_like(labels, 10)])
        v_c = stub_model(z, tt, tab[labels])
        cfg.update({f"eps_t{t}": eps, f"vu_t{t}": v_u, f"vc_t{t}": v_c, f"g_t{t}": torch.tensor(gs)})
    # ... followed by the reference's own _ddpm_step on that eps (eta = 1): what the fused sampler kernel does in one go
    nv = base_ns()
    exec(top_level("visualization.py", "_ddpm_step"), nv)
    exec(top_level("visualization.py", "run_reverse_process"), nv)
    for t in (499, 250, 2):
        torch.manual_seed(2000 + t)
        cfg[f"step_t{t}"] = nv["_ddpm_step"](z.clone(), t, cfg[f"eps_t{t}"], sched, 1.0)
        torch.manual_seed(2000 + t)
        cfg[f"noise_t{t}"] = torch.randn_like(z)
    save("ref_cfg_eps", z=z, labels=labels, **cfg)

    # ---------------------------------------------------------------- one DDPM / eta step (visualization.py:92-109)
    steps = {}
    eps_in = torch.randn(3, 3, 8, 8, generator=g)
    for t, eta in ((499, 1.0), (300, 0.5), (2, 1.0), (1, 1.0), (37, 0.0)):
        torch.manual_seed(1000 + t)
        out = nv["_ddpm_step"](z.clone(), t, eps_in, sched, eta)
        torch.manual_seed(1000 + t)
        noise = torch.randn_like(z)          # the draw _ddpm_step made (its only RNG use), for t > 1 and eta > 0
        steps.update({f"out_t{t}": out, f"noise_t{t}": noise, f"eta_t{t}": torch.tensor(eta)})
    save("ref_ddpm_step", z=z, eps=eps_in, **steps)

    # ---------------------------------------------------------------- the reverse loop (visualization.py:112-152)
    T = 500
    z0 = torch.randn(2, 3, 8, 8, generator=g)
    lab2 = torch.tensor([3, 10])
    predict = lambda model, scheduler, zz, t_tensor, lbls, gsc: ns["_cfg_predict_eps_from_v"](model, scheduler, zz, t_tensor, lbls, gsc)  # noqa: E731
    # a short walk over the LAST steps of the real schedule: run_reverse_process starts at num_time_steps - 1
    for nsteps, eta, gs in ((12, 1.0, 6.0), (6, 0.0, 3.0)):
        torch.manual_seed(77)
        x0, snaps = nv["run_reverse_process"](StubNet(), sched, z0, lab2, gs, eta, nsteps, predict, capture_timesteps=[5, 2])
        torch.manual_seed(77)
        noises = torch.stack([torch.randn_like(z0) for t in range(nsteps - 1, 1, -1)]) if eta > 0 else torch.zeros(0)
        save(f"ref_reverse_T{nsteps}", z0=z0, labels=lab2, x0=x0, noises=noises, eta=torch.tensor(eta),
             guidance=torch.tensor(gs), **{f"snap_t{k}": v for k, v in snaps.items()})
    del T

    # ---------------------------------------------------------------- monitoring images (visualization.py:40-69, 183-330)
    # Image CONTENT of the monitoring functions: tensor_to_image, the collage's snapshots (run_reverse_process +
    # tensor_to_image, :205-222) and the reconstruction panels (:287-314).  The matplotlib figure around them is not
    # reproduced (matplotlib is absent); sample_noise() draws on a CUDA device by default, so the start noise is supplied.
    nm = base_ns()
    nm.update(LDM=False, IMAGE_NET=False, np=np, UNCOND_ID=10)
    exec(top_level("visualization.py", "tensor_to_image"), nm)
    exec(top_level("visualization.py", "_ddpm_step"), nm)
    exec(top_level("visualization.py", "run_reverse_process"), nm)
    mon = {}
    gm = torch.Generator().manual_seed(21)
    probe = torch.randn(2, 3, 6, 5, generator=gm) * 0.9
    mon["t2i_in"] = probe
    mon["t2i_out"] = nm["tensor_to_image"](probe)             # first of the batch, clamp((x+1)/2), HWC
    mon["t2i_chw_out"] = nm["tensor_to_image"](probe[1])
    zc = torch.randn(1, 3, 16, 16, generator=gm)
    labc = torch.tensor([6])
    caps = [9, 5, 2, 0]
    _, snaps = nm["run_reverse_process"](StubNet(), sched, zc, labc, 4.0, 0.0, 12, predict, capture_timesteps=caps)
    mon.update(collage_z=zc, collage_label=labc, collage_caps=np.array(caps), collage_guidance=np.array(4.0),
               collage_steps=np.array(12))
    for t in caps:
        mon[f"collage_img_t{t}"] = nm["tensor_to_image"](snaps[t])
    body = with_block_statements("visualization.py", "reconstruct_from_dataset_sample", "t0", "x0_hat")
    xr = torch.clamp(torch.randn(1, 3, 16, 16, generator=gm) * 0.5, -1, 1)
    for gs, tag in ((3.0, "cfg"), (0.0, "uncond")):
        lab = torch.tensor([2])
        nr = dict(nm)
        nr.update(num_time_steps=500, t_start=9, scheduler=sched, device=torch.device("cpu"), x=xr.clone(),
                  labels=(lab if gs != 0.0 else torch.full_like(lab, 10)),       # :284-285
                  predict_eps_fn=predict, model=StubNet(), guidance_scale=gs, sampling_eta=0.0)
        torch.manual_seed(515)
        exec(body, nr)
        mon.update({f"recon_{tag}_x": xr, f"recon_{tag}_xt": nr["x_t"], f"recon_{tag}_x0hat": nr["x0_hat"],
                    f"recon_{tag}_img_orig": nm["tensor_to_image"](torch.clamp(xr, -1, 1)),
                    f"recon_{tag}_img_xt": nm["tensor_to_image"](torch.clamp(nr["x_t"], -1, 1)),
                    f"recon_{tag}_img_recon": nm["tensor_to_image"](torch.clamp(nr["x0_hat"], -1, 1)),
                    f"recon_{tag}_t0": np.array(nr["t0"]), f"recon_{tag}_label": lab})
    save("ref_monitoring", **mon)

    # ---------------------------------------------------------------- LR lambda (main_stl10.py:405-412)
    for total in (7900, 200):
        nl = dict(math=math, total_steps_new=total, warmup_steps=max(1, int(0.05 * total)))   # main_stl10.py:402-403
        exec(nested_def("main_stl10.py", "train", "lr_lambda"), nl)
        steps_ = list(range(0, total, max(1, total // 97))) + [total - 1]
        save(f"ref_lr_lambda_{total}", steps=np.array(steps_), lam=np.array([nl["lr_lambda"](s) for s in steps_], dtype=np.float64))

    # ---------------------------------------------------------------- train-step arithmetic (main_stl10.py:453-482)
    body = loop_statements("main_stl10.py", "train", "drop_mask", "USING_LOW_T_REWARDING")
    utils_ns = types.SimpleNamespace(_labels_to_context=lambda labels: tab[labels.long()])
    for epoch, tag in ((0, "plain"), (25, "lowt")):
        nt = base_ns()
        B = 6
        gx = torch.Generator().manual_seed(5)
        x = torch.clamp(torch.randn(B, 3, 16, 16, generator=gx) * 0.5, -1, 1)
        labels = torch.randint(0, 10, (B,), generator=gx)
        nt.update(x=x, labels=labels, scheduler=sched, CFG_P_UNCOND=0.2, UNCOND_ID=10, NUM_TIME_STEPS=500,
                  CONTEXT_TABLE=tab, utils=utils_ns, model=StubNet(), epoch=epoch,
                  criterion=nn.MSELoss(reduction="none"))     # main_stl10.py:397-398 (USING_LOW_T_REWARDING)
        torch.manual_seed(4242)
        exec(body, nt)
        save(f"ref_train_step_{tag}", x=x, labels=labels, labels_dropped=nt["labels_dropped"], t=nt["t"], e=nt["e"],
             x_t=nt["x_t"], v_target=nt["v_target"], context=nt["context"], output=nt["output"], loss=nt["loss"],
             epoch=np.array(epoch))


if __name__ == "__main__":
    main()
