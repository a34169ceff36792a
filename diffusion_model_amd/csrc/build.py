#!/usr/bin/env python3
"""Build libdm_hip.so (gfx950) in-tree with hipcc.  No cmake/ninja; one object per .hip file,
compiled in parallel, linked into diffusion_model_amd/libdm_hip.so."""
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "libdm_hip.so")
OBJ = os.path.join(HERE, "build")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffast-math", "-fno-gpu-rdc",
         "-Wno-unused-result", "-DNDEBUG",
         # no SLP vectorizer: it pairs scalar fp32 adds/muls into v_pk_*_f32, which issue slower than the two
         # scalar ops they replace next to MFMA/exp streams (measured on the attention kernels)
         "-fno-slp-vectorize"]


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _compile(src):
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    deps = [src] + [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")] + [
        os.path.join(os.path.dirname(PKG), "include", "dm_hip.h")]
    if os.path.exists(obj) and all(os.path.getmtime(obj) > os.path.getmtime(d) for d in deps):
        return obj, ""
    r = subprocess.run([hipcc(), *FLAGS, "-c", src, "-o", obj], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    return obj, r.stderr


def build(verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".hip"))
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(_compile, srcs))
    objs = [o for o, _ in res]
    if verbose:
        for _, err in res:
            if err.strip():
                print(err)
    if os.path.exists(OUT) and all(os.path.getmtime(OUT) > os.path.getmtime(o) for o in objs):
        return OUT
    r = subprocess.run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return OUT


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
