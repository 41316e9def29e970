"""Build libsmh.so (hand-written HIP for gfx950 + the C ABI of include/smh.h) in-tree with hipcc.

    python -m sm_hpss_mtl_amd.build [--force]

hipcc cross-compiles without a GPU; the built library travels to the GPU box with the snapshot
(*.so is git-ignored, not gpurun-ignored).  No torch, no pybind: the boundary is a plain C ABI.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libsmh.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# per-file additions.  smh_median_split: pure selection on finite magnitudes; without the flag every fminf/fmaxf of
# a value loaded from LDS is preceded by a canonicalising v_max_f32 (IEEE mode), +30 % VALU in that kernel.
# smh_train: the weight-gradient accumulation uses float atomicAdd on device memory; without the flag hipcc emits a
# compare-and-swap loop per atomic instead of global_atomic_add_f32 (the backward kernel was 8.6 ms because of it).
# smh_tcn: relu / channel maximum of accumulator values; same canonicalisation issue (16 extra v_max per 16-frame tile).
EXTRA_FLAGS = {"smh_median_split.hip": ["-fno-honor-nans"], "smh_train.hip": ["-munsafe-fp-atomics"],
               "smh_tcn.hip": ["-fno-honor-nans"], "smh_tcn_bf16.hip": ["-fno-honor-nans"],
               "smh_train_bf16.hip": ["-munsafe-fp-atomics", "-fno-honor-nans"]}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "smh.h"))
    return max(os.path.getmtime(h) for h in hs)


LAB = False  # --lab: -DSMH_LAB, the measured-and-rejected implementation variants compiled in (smh_common.h: lab_env)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ, src[:-4] + ".o")
    srcp = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(srcp), _headers_mtime()):
        return obj
    cmd = ["hipcc", *FLAGS, *(["-DSMH_LAB"] if LAB else []), *EXTRA_FLAGS.get(src, []), "-c", srcp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = ["hipcc", "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", LIB, "from", len(srcs), "HIP sources")
    return LIB


if __name__ == "__main__":
    LAB = "--lab" in sys.argv
    build(force="--force" in sys.argv or LAB)
