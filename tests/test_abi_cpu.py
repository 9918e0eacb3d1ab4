"""CPU checks of the C-ABI library: it builds for gfx950, loads, exports every symbol the header
declares (no compute calls without a GPU), and the product path refuses to run without a device."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib_path():
    from omgsr_amd.build import build_library
    return build_library()


def test_library_exports_every_declared_symbol():
    from omgsr_amd import _lib
    path = _lib_path()
    lib = ctypes.CDLL(path)
    header = open(os.path.join(ROOT, "include", "omgsr_hip.h")).read()
    declared = set(re.findall(r"\b(omgsr_[a-z0-9_]+)\s*\(", header))
    declared -= {"omgsr_igemm_args", "omgsr_attn_args", "omgsr_timing_entry"}
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported"
    lib.omgsr_abi_version.restype = ctypes.c_int
    assert lib.omgsr_abi_version() == _lib.ABI_VERSION
    lib.omgsr_error_string.restype = ctypes.c_char_p
    assert b"gfx950" in lib.omgsr_error_string(-3)
    lib.omgsr_groupnorm_nchunk.argtypes = [ctypes.c_int64]
    # 256-pixel chunks on big maps; smaller ones on the maps of a one-image call (round 5: a function of HW alone)
    assert lib.omgsr_groupnorm_nchunk(1 << 20) == 4096 and lib.omgsr_groupnorm_nchunk(4096) == 128 and lib.omgsr_groupnorm_nchunk(1) == 1


def test_struct_layout_matches_header(tmp_path):
    """ctypes mirrors of the argument structs: size and EVERY field offset agree with what a C compiler makes of
    include/omgsr_hip.h (gcc compiles a probe that prints sizeof / offsetof)."""
    import shutil
    import subprocess
    from omgsr_amd._lib import AttnArgs, GnMergeArgs, IgemmArgs, TimingEntry
    structs = {"omgsr_igemm_args": IgemmArgs, "omgsr_attn_args": AttnArgs, "omgsr_gn_merge_args": GnMergeArgs,
               "omgsr_timing_entry": TimingEntry}
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "omgsr_hip.h")}"', "int main(void) {"]
    for cname, cls in structs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname.rstrip("_")}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run([gcc, "-std=c11", "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    seen = 0
    for ln in out:
        if not ln.strip():
            continue
        cname, field, val = ln.split()
        cls = structs[cname]
        if field == "size":
            assert ctypes.sizeof(cls) == int(val), f"sizeof({cname}): C {val}, ctypes {ctypes.sizeof(cls)}"
        else:
            assert getattr(cls, field).offset == int(val), f"offsetof({cname}, {field}): C {val}, ctypes {getattr(cls, field).offset}"
        seen += 1
    assert seen == sum(len(c._fields_) + 1 for c in structs.values())


def test_bad_arguments_are_rejected_without_a_gpu():
    from omgsr_amd import _lib
    lib = _lib.load()
    assert lib.omgsr_igemm(None, None) == -1
    assert lib.omgsr_attention(None, None) == -1
    assert lib.omgsr_layernorm(None, None, None, None, 4, 320, 1e-5, 0, 0, None) == -1
    a = _lib.IgemmArgs()
    a.in_, a.weight, a.out = 8, 8, 8
    a.N = a.H = a.W = a.Ho = a.Wo = a.R = a.S = a.stride = a.batch = 1
    a.Cin, a.Cout, a.Cout_pad, a.K_pad = 12, 8, 128, 32          # Cin % 8 != 0
    assert lib.omgsr_igemm(ctypes.byref(a), None) == -2


def test_product_path_fails_loudly_on_cpu():
    from omgsr_amd import ops
    from omgsr_amd._lib import OmgsrError
    with pytest.raises(OmgsrError):
        ops.layer_norm(torch.zeros(2, 320, dtype=torch.bfloat16), None, None, 1e-5)
    from omgsr_amd.diffusers_api import AutoencoderKL
    vae = AutoencoderKL(block_out_channels=[32, 32, 32, 32], layers_per_block=1).to(torch.bfloat16)
    with pytest.raises(OmgsrError):
        vae.decode(torch.zeros(1, 4, 8, 8, dtype=torch.bfloat16))


def test_no_oracle_imports_in_product():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "omgsr_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{f} imports the oracle"
