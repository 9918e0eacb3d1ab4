"""Builds libomgsr_hip.so (the C-ABI kernel library, include/omgsr_hip.h) for gfx950 with hipcc.

The .so is written in-tree (omgsr_amd/lib/) so that it travels to the GPU box with the repo
snapshot; it is git-ignored. hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBNAME = "libomgsr_hip.so"
SOURCES = ["igemm.hip", "igemm_dma.hip", "igemm_p8.hip", "igemm_halo.hip", "igemm_halo_f16.hip", "igemm_halo_multi.hip", "igemm_halo_mx.hip", "attention.hip", "norm.hip", "elementwise.hip", "colorfix.hip", "preprocess.hip", "mfma_peak.hip"]
ARCH = "gfx950"


def lib_path() -> str:
    return os.path.join(LIBDIR, LIBNAME)


def _source_digest() -> str:
    h = hashlib.sha256()
    names = sorted(os.listdir(CSRC)) + [os.path.join("..", "..", "include", "omgsr_hip.h")]
    for n in names:
        p = os.path.join(CSRC, n)
        if os.path.isfile(p):
            h.update(n.encode())
            with open(p, "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source into one shared library; returns its path."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(LIBDIR, exist_ok=True)
    out = lib_path()
    stamp = out + ".sha256"
    # OMGSR_BUILD_ABLATIONS=1 adds the experiment-only kernel instantiations (parts of a kernel compiled out for timing runs, the
    # schedules kept for A/B: profiles/r02_dma_ablation.md); the shipped library leaves them out
    defs = ["-DOMGSR_BUILD_ABLATIONS"] if os.environ.get("OMGSR_BUILD_ABLATIONS", "0") == "1" else []
    defs += os.environ.get("OMGSR_EXTRA_DEFS", "").split()          # A/B builds of compile-time variants (e.g. -DOMGSR_NT_STORE)
    digest = _source_digest() + "".join(defs)
    if not force and os.path.isfile(out) and os.path.isfile(stamp) and open(stamp).read().strip() == digest:
        return out
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libomgsr_hip.so")
    from concurrent.futures import ThreadPoolExecutor

    hdr = hashlib.sha256()              # every header any source may include: a header edit recompiles everything, a .hip edit only itself
    for n in sorted(os.listdir(CSRC)) + [os.path.join("..", "..", "include", "omgsr_hip.h")]:
        if n.endswith(".h"):
            with open(os.path.join(CSRC, n), "rb") as f:
                hdr.update(n.encode() + f.read())

    def compile_one(src: str) -> str:
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        with open(os.path.join(CSRC, src), "rb") as f:
            want = hashlib.sha256(hdr.digest() + f.read() + "".join(defs).encode()).hexdigest()
        if not force and os.path.isfile(obj) and os.path.isfile(obj + ".sha256") and open(obj + ".sha256").read().strip() == want:
            return obj
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", *defs,
               "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        with open(obj + ".sha256", "w") as f:
            f.write(want)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:      # one hipcc per translation unit
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as f:
        f.write(digest)
    return out


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
