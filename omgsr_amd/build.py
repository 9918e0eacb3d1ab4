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
SOURCES = ["igemm.hip", "igemm_dma.hip", "igemm_p8.hip", "igemm_gmx.hip", "igemm_halo.hip", "igemm_halo_f16.hip", "igemm_halo_multi.hip", "igemm_halo_mx.hip", "igemm_halo_mx6.hip", "igemm_halo_mx6_flat.hip", "igemm_halo_out6.hip", "igemm_halo_flat.hip", "igemm_halo_gn.hip", "attention.hip", "norm.hip", "elementwise.hip", "colorfix.hip", "preprocess.hip", "mfma_peak.hip"]
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
        # -Rpass-analysis=kernel-resource-usage: the register / spill / scratch / LDS figures of every kernel, kept next to the object
        # (<src>.resources.txt) - tests/test_kernel_resources_cpu.py pins the hand-scheduled kernels' figures against a committed table
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", *defs,
               "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        import re
        remarks = [ln for ln in r.stderr.splitlines() if "remark:" in ln]
        # (each remark is followed by a source excerpt and a caret line; "In file included from" lines precede remarks in headers)
        rest, lines = [], r.stderr.splitlines()
        in_remark = False           # the source excerpt / caret lines that FOLLOW a remark are dropped with it; those of warnings and errors stay
        for i, ln in enumerate(lines):
            if "remark:" in ln:
                in_remark = True
                continue
            is_excerpt = bool(re.match(r"^\s*\d*\s*\|", ln)) or not ln.strip()
            if in_remark and is_excerpt:
                continue
            in_remark = False
            if ln.startswith("In file included from") and i + 1 < len(lines) and ("remark:" in lines[i + 1] or lines[i + 1].startswith("In file included")):
                continue
            if re.match(r"^\d+ (warning|remark)s? generated", ln):
                continue
            rest.append(ln)
        if rest:
            print("\n".join(rest), file=sys.stderr)
        if r.returncode != 0:
            raise subprocess.CalledProcessError(r.returncode, cmd)
        with open(os.path.join(LIBDIR, src + ".resources.txt"), "w") as f:
            f.write("\n".join(remarks) + "\n")
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


def _short_name(mangled: str) -> str:
    """`_ZN12_GLOBAL__N_117igemm_halo_kernelIDF16_Li0ELb0ELb0ELi9ELb1EEEv...` -> `igemm_halo_kernel<fp16,0,0,0,9,1>` (the image's c++filt
    predates the _Float16 / __bf16 manglings DF16_ / DF16b, so the few forms this library uses are decoded here)."""
    import re
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", mangled)
    if m:
        n = int(m.group(1))
        rest = mangled[m.end():]
        name, rest = rest[:n], rest[n:]
    else:
        m = re.match(r"_Z(\d+)", mangled)
        if not m:
            return mangled
        n = int(m.group(1))
        name, rest = mangled[m.end():m.end() + n], mangled[m.end() + n:]
    args = []
    if rest.startswith("I"):
        body = rest[1:]
        while body and not body.startswith("E"):
            for rx, fn in ((r"DF16_", lambda g: "fp16"), (r"DF16b", lambda g: "bf16"), (r"Lin(\d+)E", lambda g: "-" + g.group(1)),
                           (r"Li(\d+)E", lambda g: g.group(1)), (r"Lb([01])E", lambda g: g.group(1)), (r"f", lambda g: "float")):
                g = re.match(rx, body)
                if g:
                    args.append(fn(g))
                    body = body[g.end():]
                    break
            else:
                args.append("?")
                break
    return name + ("<" + ",".join(args) + ">" if args else "")


def kernel_resources() -> dict:
    """{demangled kernel name: {vgpr, agpr, sgpr, spill_vgpr, spill_sgpr, scratch, occupancy, lds}} of the library as built
    (parsed from the compiler's kernel-resource-usage remarks the build keeps next to each object)."""
    import re
    build_library()
    keys = {"VGPRs": "vgpr", "AGPRs": "agpr", "SGPRs": "sgpr", "VGPRs Spill": "spill_vgpr", "SGPRs Spill": "spill_sgpr",
            "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy", "LDS Size [bytes/block]": "lds"}
    out, mangled = {}, []
    for src in SOURCES:
        path = os.path.join(LIBDIR, src + ".resources.txt")
        if not os.path.isfile(path):        # object predates this build step: recompile it
            build_library(force=True)
        cur = None
        for line in open(path):
            m = re.search(r"remark: (?:[^:]+:\d+:\d+: +)?([A-Za-z \[\]/]+): +(\S+)", line)
            if not m:
                continue
            k, v = m.group(1).strip(), m.group(2)
            if k == "Function Name":
                cur = {"source": src}
                mangled.append((v, cur))
            elif cur is not None and k in keys:
                cur[keys[k]] = int(v)
    for m, rec in mangled:
        k = _short_name(m)
        if "?" in k or k in out:            # an undecoded template argument / two instantiations under one short name: key on the mangled name
            k = m
        out[k] = rec
    return out


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
