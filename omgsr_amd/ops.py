"""Tensor-level wrappers over the C ABI (include/omgsr_hip.h).

torch is used for device memory and streams only: every function here validates shapes, allocates
the output with torch.empty and launches a hand-written gfx950 kernel on torch's current stream.
Activations are bf16, channels-last ([N, H, W, C]; token matrices are [B, L, C]).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import AttnArgs, IgemmArgs, check

ACT_NONE, ACT_SILU, ACT_GELU_TANH, ACT_GEGLU = 0, 1, 2, 3
OUT_BF16, OUT_F32 = 0, 1
LAYOUT_NHWC, LAYOUT_T = 0, 1


_ACT = torch.bfloat16


def act_dtype() -> torch.dtype:
    """The 16-bit element type of every activation / packed weight (bf16 default, fp16 optional)."""
    return _ACT


def set_compute_dtype(dtype: torch.dtype) -> None:
    """Process-wide switch (omgsr_set_compute_dtype): packed-weight caches are keyed on it and rebuild lazily."""
    global _ACT
    if dtype not in (torch.bfloat16, torch.float16):
        raise TypeError(f"compute dtype must be act_dtype() or torch.float16, got {dtype}")
    check(_lib.load().omgsr_set_compute_dtype(0 if dtype == torch.bfloat16 else 1), "set_compute_dtype")
    _ACT = dtype


def io_dtype(x: torch.Tensor) -> torch.dtype:
    """Boundary dtype the kernels can write directly for a caller holding `x`: f32 stays f32, else the compute dtype."""
    return torch.float32 if x.dtype == torch.float32 else _ACT


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _req(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_cuda:
        raise _lib.OmgsrError(f"{name}: tensor is on {t.device}; the OMGSR HIP path runs on an MI355X only")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return t


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def _igemm(a: "IgemmArgs", device, what: str) -> None:
    """Launch omgsr_igemm; when the library wants to split K (small-M / long-K problems) hand it an fp32 scratch."""
    lib = _lib.load()
    need = lib.omgsr_igemm_workspace_bytes(C.byref(a))
    ws = None
    if need > 0:
        ws = torch.empty(need // 4, device=device, dtype=torch.float32)
        a.workspace = ws.data_ptr()
    check(lib.omgsr_igemm(C.byref(a), _stream()), what)


# --------------------------------------------------------------------------------------------
# weight packing (load time)

@dataclass
class PackedWeight:
    """KRSC bf16 weight [Cout_pad, K_pad] for omgsr_igemm (+ optional f32 bias)."""
    w: torch.Tensor
    bias: Optional[torch.Tensor]
    cout: int          # logical output channels (GEGLU: halved)
    cin: int           # padded input channels (multiple of 8)
    R: int
    S: int
    geglu: bool = False
    w_cm: Optional[torch.Tensor] = None   # chunk-major second packing (3x3, Cin % 32 == 0): halo-tile kernel

    @property
    def cout_pad(self) -> int:
        return self.w.shape[0]

    @property
    def k_pad(self) -> int:
        return self.w.shape[1]


def pack_conv_weight(weight: torch.Tensor, bias: Optional[torch.Tensor], device=None, cout_multiple: int = 1) -> PackedWeight:
    """[Cout, Cin, R, S] (torch conv layout) -> [Cout_pad, roundup(R*S*Cin8, 32)] bf16, k = (r*S+s)*Cin8 + c.
    cout_multiple=8 widens the LOGICAL output to a multiple of 8 channels (zero weights, zero bias) so a
    3/4-channel conv writes 16-byte rows that the next kernel can consume directly."""
    cout, cin, R, S = weight.shape
    dev = device or weight.device
    if cout % cout_multiple:
        extra = _round_up(cout, cout_multiple) - cout
        weight = torch.cat([weight.detach().to(dev), torch.zeros((extra, cin, R, S), device=dev, dtype=weight.dtype)], dim=0)
        if bias is not None:
            bias = torch.cat([bias.detach().to(dev), torch.zeros(extra, device=dev, dtype=bias.dtype)], dim=0)
        cout += extra
    cin8 = _round_up(cin, 8)
    w = weight.detach().to(device=dev, dtype=torch.float32).permute(0, 2, 3, 1)  # [Cout,R,S,Cin]
    if cin8 != cin:
        w = torch.nn.functional.pad(w, (0, cin8 - cin))
    w = w.reshape(cout, R * S * cin8)
    k_pad = _round_up(w.shape[1], 32)
    cout_pad = _round_up(cout, 256 if cout >= 256 else 128)   # 256-row padding lets the 256x256 GEMM tile run
    out = torch.zeros(cout_pad, k_pad, device=dev, dtype=act_dtype())
    out[:cout, : w.shape[1]] = w.to(act_dtype())
    b = None if bias is None else bias.detach().to(device=dev, dtype=torch.float32).contiguous()
    w_cm = None
    if R == 3 and S == 3 and cin8 % 32 == 0:
        # slice-major: [Cin/32][9 taps][Cout_pad][32] - the 128 x 32 weight slice of one (chunk, tap) K-step is one
        # contiguous 8 KB run, so every LDS-DMA wave instruction reads 8 full 128-B lines
        w_cm = out.view(cout_pad, 9, cin8 // 32, 32).permute(2, 1, 0, 3).contiguous()
    return PackedWeight(out, b, cout, cin8, R, S, w_cm=w_cm)


def pack_linear_weight(weight: torch.Tensor, bias: Optional[torch.Tensor], device=None) -> PackedWeight:
    """[out, in] -> 1x1 'conv' weight."""
    return pack_conv_weight(weight[:, :, None, None], bias, device)


def pack_geglu_weight(weight: torch.Tensor, bias: Optional[torch.Tensor], device=None) -> PackedWeight:
    """GEGLU projection [2*inner, in] (rows [a | gate], diffusers `chunk(2, -1)`) -> rows interleaved in
    blocks of 32: [a_0..31, g_0..31, a_32..63, g_32..63, ...] so one 64-wide wave tile holds both halves."""
    two_inner, cin = weight.shape
    inner = two_inner // 2
    if inner % 32:
        raise ValueError("GEGLU inner dim must be a multiple of 32")
    a, g = weight[:inner], weight[inner:]
    w = torch.stack([a.reshape(inner // 32, 32, cin), g.reshape(inner // 32, 32, cin)], dim=1).reshape(two_inner, cin)
    pw = pack_linear_weight(w, None, device)
    if bias is not None:
        ba, bg = bias[:inner], bias[inner:]
        b = torch.stack([ba.reshape(-1, 32), bg.reshape(-1, 32)], dim=1).reshape(two_inner)
        pw.bias = b.detach().to(device=pw.w.device, dtype=torch.float32).contiguous()
    pw.cout = inner
    pw.geglu = True
    return pw


# --------------------------------------------------------------------------------------------
# K1-K3, K5, K6: implicit-GEMM conv / linear / bmm

def conv2d(x: torch.Tensor, pw: PackedWeight, *, stride: int = 1, pad: tuple[int, int, int, int] | int = 1,
           upsample: bool = False, act: int = ACT_NONE, residual: Optional[torch.Tensor] = None,
           gate: Optional[torch.Tensor] = None, out_dtype: int = OUT_BF16, alpha: float = 1.0,
           out: Optional[torch.Tensor] = None, gn_groups: int = 0) -> torch.Tensor:
    """x [N,H,W,Cin] bf16 -> [N,Ho,Wo,Cout]. pad = (top, bottom, left, right) on the (virtual) input.
    gn_groups > 0: the caller will GroupNorm the result with that many groups; when the kernel can, it emits the
    (sum, sum of squares) partials from its epilogue and group_norm_stats() skips its read pass over the tensor."""
    _req(x, act_dtype(), "x")
    N, H, W, Cin = x.shape
    if Cin != pw.cin:
        raise ValueError(f"conv2d: input has {Cin} channels, packed weight expects {pw.cin}")
    if isinstance(pad, int):
        pad = (pad, pad, pad, pad)
    pt, pb, pl, pr = pad
    Hv, Wv = (H * 2, W * 2) if upsample else (H, W)
    Ho = (Hv + pt + pb - pw.R) // stride + 1
    Wo = (Wv + pl + pr - pw.S) // stride + 1
    if pw.geglu:
        act = ACT_GEGLU
    if out is None:
        out = torch.empty((N, Ho, Wo, pw.cout), device=x.device,
                          dtype=act_dtype() if out_dtype == OUT_BF16 else torch.float32)
    if residual is not None:
        _req(residual, act_dtype(), "residual")
        if residual.shape != out.shape:
            raise ValueError(f"residual shape {tuple(residual.shape)} != output {tuple(out.shape)}")
    a = IgemmArgs()
    a.in_, a.weight, a.bias, a.gate = x.data_ptr(), pw.w.data_ptr(), _ptr(pw.bias), _ptr(gate)
    a.residual, a.out = _ptr(residual), out.data_ptr()
    a.weight_cm = _ptr(pw.w_cm)
    a.N, a.H, a.W, a.Cin = N, H, W, Cin
    a.Cout, a.Cout_pad, a.K_pad = pw.cout, pw.cout_pad, pw.k_pad
    a.R, a.S, a.stride, a.pad_top, a.pad_left, a.upsample = pw.R, pw.S, stride, pt, pl, int(upsample)
    a.Ho, a.Wo = Ho, Wo
    a.act, a.out_dtype, a.out_layout = act, out_dtype, LAYOUT_NHWC
    a.t_rows, a.t_ld = 0, 0
    a.batch, a.in_bstride, a.w_bstride, a.out_bstride = 1, 0, 0, 0
    a.alpha = alpha
    partial = None
    if gn_groups > 0:
        a.gn_groups = gn_groups
        lib = _lib.load()
        # a problem that _igemm will split over K (it hands over the workspace) finishes in the reduce pass: no statistics there
        nslot = 0 if lib.omgsr_igemm_workspace_bytes(C.byref(a)) > 0 else lib.omgsr_igemm_gn_slots(C.byref(a))
        if nslot > 0:
            a.gn_entries = lib.omgsr_igemm_gn_entries(C.byref(a))       # per group, or per channel for odd group sizes
            partial = torch.empty((N, nslot, a.gn_entries, 2), device=x.device, dtype=torch.float32)
            a.gn_partial = partial.data_ptr()
    _igemm(a, x.device, "omgsr_igemm(conv2d)")
    if partial is not None:
        out._omgsr_gn = (partial, gn_groups, out.data_ptr(), out._version)      # consumed by group_norm_stats(out)
    return out


def linear(x: torch.Tensor, pw: PackedWeight, *, act: int = ACT_NONE, residual: Optional[torch.Tensor] = None,
           gate: Optional[torch.Tensor] = None, out_dtype: int = OUT_BF16, alpha: float = 1.0, gn_groups: int = 0) -> torch.Tensor:
    """x [..., K] bf16 -> [..., Cout]. gn_groups > 0 (x [B, ..., K]): the result feeds a GroupNorm over each x[b]; the GEMM
    then runs as B images of prod(...) rows so its epilogue can leave the per-image statistics (see conv2d)."""
    lead = x.shape[:-1]
    M = 1
    for d in lead:
        M *= d
    B = lead[0] if (gn_groups > 0 and len(lead) >= 2) else 1
    x2 = x.reshape(B, 1, M // B, x.shape[-1])
    r2 = None if residual is None else residual.reshape(B, 1, M // B, pw.cout)
    y = conv2d(x2, pw, stride=1, pad=0, act=act, residual=r2, gate=gate, out_dtype=out_dtype, alpha=alpha, gn_groups=gn_groups)
    out = y.reshape(*lead, pw.cout)
    carry_gn(y, out)
    return out


def carry_gn(src: torch.Tensor, view: torch.Tensor) -> torch.Tensor:
    """Views are new Python objects: hand the fused-statistics handle of `src` on to a view of the same storage."""
    h = getattr(src, "_omgsr_gn", None)
    if h is not None and view.data_ptr() == src.data_ptr() and view.shape[0] == src.shape[0]:
        view._omgsr_gn = h
    return view


def linear_into(x: torch.Tensor, pw: PackedWeight, out: torch.Tensor, row0: int, col0: int, *, act: int = ACT_NONE,
                residual: Optional[torch.Tensor] = None, gate: Optional[torch.Tensor] = None) -> None:
    """out[row0:row0+M, col0:col0+Cout] = epilogue(x @ W^T): writes a projection straight into a slice of a
    larger 2-D token buffer `out` [rows, ld] (joint text+image sequences, [attn | mlp] concat)."""
    _req(x, act_dtype(), "x"); _req(out, act_dtype(), "out")
    M, K = x.numel() // x.shape[-1], x.shape[-1]
    ld = out.shape[-1]
    if K != pw.cin or out.dim() != 2 or row0 + M > out.shape[0] or col0 + pw.cout > ld or (col0 & 7):
        raise ValueError("linear_into: slice does not fit")
    if residual is not None:
        _req(residual, act_dtype(), "residual")
        if residual.numel() != M * pw.cout:
            raise ValueError("linear_into: residual must be a dense [M, Cout]")
    a = IgemmArgs()
    a.in_, a.weight, a.bias, a.gate = x.data_ptr(), pw.w.data_ptr(), _ptr(pw.bias), _ptr(gate)
    a.residual, a.out = _ptr(residual), out.data_ptr() + 2 * (row0 * ld + col0)
    a.N, a.H, a.W, a.Cin = 1, 1, M, K
    a.Cout, a.Cout_pad, a.K_pad = pw.cout, pw.cout_pad, pw.k_pad
    a.R, a.S, a.stride, a.pad_top, a.pad_left, a.upsample = 1, 1, 1, 0, 0, 0
    a.Ho, a.Wo = 1, M
    a.act, a.out_dtype, a.out_layout = act, OUT_BF16, LAYOUT_NHWC
    a.out_ld = ld
    a.batch, a.alpha = 1, 1.0
    _igemm(a, x.device, "omgsr_igemm(linear_into)")


def linear_t_into(x: torch.Tensor, pw: PackedWeight, out_t: torch.Tensor, key0: int) -> None:
    """out_t[n, key0 + l] = (x W^T + b)[l, n] for x [L, K]: transposed projection into a slice of a joint
    V^T buffer [Cout, ld]."""
    _req(x, act_dtype(), "x"); _req(out_t, act_dtype(), "out_t")
    L, K = x.numel() // x.shape[-1], x.shape[-1]
    if out_t.dim() != 2 or out_t.shape[0] != pw.cout or key0 + L > out_t.shape[1]:
        raise ValueError("linear_t_into: slice does not fit")
    a = IgemmArgs()
    a.in_, a.weight, a.bias, a.out = x.data_ptr(), pw.w.data_ptr(), _ptr(pw.bias), out_t.data_ptr() + 2 * key0
    a.N, a.H, a.W, a.Cin = 1, 1, L, K
    a.Cout, a.Cout_pad, a.K_pad = pw.cout, pw.cout_pad, pw.k_pad
    a.R, a.S, a.stride, a.pad_top, a.pad_left, a.upsample = 1, 1, 1, 0, 0, 0
    a.Ho, a.Wo = 1, L
    a.act, a.out_dtype, a.out_layout = ACT_NONE, OUT_BF16, LAYOUT_T
    a.t_rows, a.t_ld = L, out_t.shape[1]
    a.batch, a.alpha = 1, 1.0
    check(_lib.load().omgsr_igemm(C.byref(a), _stream()), "omgsr_igemm(linear_t_into)")


def linear_t(x: torch.Tensor, pw: PackedWeight, rows_per_batch: int, ld: Optional[int] = None) -> torch.Tensor:
    """Transposed-output projection: x [B, L, K] -> out [B, Cout, ld] with out[b, n, l] = (x W^T + bias)[b, l, n].
    This is how V reaches omgsr_attention (key index contiguous). Columns l >= L are zero."""
    _req(x, act_dtype(), "x")
    B, L, K = x.shape
    if L != rows_per_batch:
        raise ValueError("rows_per_batch mismatch")
    ld = ld or _round_up(L, 8)
    out = torch.zeros((B, pw.cout, ld), device=x.device, dtype=act_dtype()) if ld != L else \
        torch.empty((B, pw.cout, ld), device=x.device, dtype=act_dtype())
    a = IgemmArgs()
    a.in_, a.weight, a.bias, a.gate, a.residual, a.out = x.data_ptr(), pw.w.data_ptr(), _ptr(pw.bias), None, None, out.data_ptr()
    a.N, a.H, a.W, a.Cin = 1, 1, B * L, K
    a.Cout, a.Cout_pad, a.K_pad = pw.cout, pw.cout_pad, pw.k_pad
    a.R, a.S, a.stride, a.pad_top, a.pad_left, a.upsample = 1, 1, 1, 0, 0, 0
    a.Ho, a.Wo = 1, B * L
    a.act, a.out_dtype, a.out_layout = ACT_NONE, OUT_BF16, LAYOUT_T
    a.t_rows, a.t_ld = L, ld
    a.batch, a.in_bstride, a.w_bstride, a.out_bstride = 1, 0, 0, 0
    a.alpha = 1.0
    check(_lib.load().omgsr_igemm(C.byref(a), _stream()), "omgsr_igemm(linear_t)")
    return out


def bmm_nt(a_mat: torch.Tensor, b_mat: torch.Tensor, *, alpha: float = 1.0, out_dtype: int = OUT_BF16) -> torch.Tensor:
    """out[b] = alpha * a[b] @ b[b]^T ; a [B, M, K], b [B, Npad, K] bf16 with Npad % 128 == 0, K % 32 == 0.
    Returns [B, M, Npad]. (d=512 VAE attention scores / PV product.)"""
    _req(a_mat, act_dtype(), "a")
    _req(b_mat, act_dtype(), "b")
    B, M, K = a_mat.shape
    Bb, Np, Kb = b_mat.shape
    if Bb != B or Kb != K or K % 32:
        raise ValueError(f"bmm_nt: incompatible shapes {tuple(a_mat.shape)} x {tuple(b_mat.shape)}")
    if Np % 128:    # reduced test configs only (the real VAE has 512 channels / 128-padded key counts)
        bp = torch.zeros((B, _round_up(Np, 128), K), device=b_mat.device, dtype=act_dtype())
        bp[:, :Np] = b_mat
        return bmm_nt(a_mat, bp, alpha=alpha, out_dtype=out_dtype)[:, :, :Np].contiguous()
    out = torch.empty((B, M, Np), device=a_mat.device, dtype=act_dtype() if out_dtype == OUT_BF16 else torch.float32)
    a = IgemmArgs()
    a.in_, a.weight, a.bias, a.gate, a.residual, a.out = a_mat.data_ptr(), b_mat.data_ptr(), None, None, None, out.data_ptr()
    a.N, a.H, a.W, a.Cin = 1, 1, M, K
    a.Cout, a.Cout_pad, a.K_pad = Np, Np, K
    a.R, a.S, a.stride, a.pad_top, a.pad_left, a.upsample = 1, 1, 1, 0, 0, 0
    a.Ho, a.Wo = 1, M
    a.act, a.out_dtype, a.out_layout = ACT_NONE, out_dtype, LAYOUT_NHWC
    a.t_rows, a.t_ld = 0, 0
    a.batch, a.in_bstride, a.w_bstride, a.out_bstride = B, M * K, Np * K, M * Np
    a.alpha = alpha
    check(_lib.load().omgsr_igemm(C.byref(a), _stream()), "omgsr_igemm(bmm_nt)")
    return out


# --------------------------------------------------------------------------------------------
# K4: GroupNorm

def group_norm_stats(x: torch.Tensor, groups: int, eps: float):
    """x [N, ..., C] bf16 -> (mean [N,G], rstd [N,G], var [N,G]) f32 (biased variance)."""
    _req(x, act_dtype(), "x")
    N, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (N * Cc)
    lib = _lib.load()
    mean = torch.empty((N, groups), device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    var = torch.empty_like(mean)
    fused = getattr(x, "_omgsr_gn", None)
    if fused is not None and fused[1] == groups and fused[2] == x.data_ptr() and fused[3] == x._version \
            and fused[0].shape[0] == N:
        # the producing conv already reduced this tensor (omgsr_igemm gn_partial): fold its partials only
        check(lib.omgsr_groupnorm_finalize(fused[0].data_ptr(), mean.data_ptr(), rstd.data_ptr(), var.data_ptr(), N,
                                           fused[0].shape[1], groups, fused[0].shape[2], float(HW) * (Cc // groups), eps, _stream()),
              "omgsr_groupnorm_finalize")
        return mean, rstd, var
    nchunk = lib.omgsr_groupnorm_nchunk(HW)
    partial = torch.empty((N, nchunk, groups, 2), device=x.device, dtype=torch.float32)
    check(lib.omgsr_groupnorm_stats(x.data_ptr(), partial.data_ptr(), mean.data_ptr(), rstd.data_ptr(), var.data_ptr(),
                                    N, HW, Cc, groups, eps, _stream()), "omgsr_groupnorm_stats")
    return mean, rstd, var


def group_norm_partial(x: torch.Tensor, groups: int) -> torch.Tensor:
    """(sum, sum of squares) partials [N, nslot, G, 2] of x: the producing conv's fused ones when it left them
    (conv2d gn_groups), else one read pass over x."""
    _req(x, act_dtype(), "x")
    N, Cc = x.shape[0], x.shape[-1]
    fused = getattr(x, "_omgsr_gn", None)
    if fused is not None and fused[1] == groups and fused[2] == x.data_ptr() and fused[3] == x._version and fused[0].shape[0] == N:
        return fused[0]
    HW = x.numel() // (N * Cc)
    lib = _lib.load()
    partial = torch.empty((N, lib.omgsr_groupnorm_nchunk(HW), groups, 2), device=x.device, dtype=torch.float32)
    check(lib.omgsr_groupnorm_partial(x.data_ptr(), partial.data_ptr(), N, HW, Cc, groups, _stream()), "omgsr_groupnorm_partial")
    return partial


def group_norm_stats_merged(tensors, tiles, N: int, groups: int, eps: float):
    """Tiled-VAE GroupNorm statistics in ONE launch: tensors[k] is [tiles[k]*N, h_k, w_k, C] (tile-major); returns
    per-image (mean, rstd, var) [N, G] merged with pixel-count weights (the reference's GroupNormParam.summary)."""
    if len(tensors) > _lib.GN_MAX_GROUPS:
        raise ValueError(f"at most {_lib.GN_MAX_GROUPS} tile shape groups")
    a = _lib.GnMergeArgs()
    keep = []
    tot = float(sum(t.shape[1] * t.shape[2] * tiles[k] for k, t in enumerate(tensors)))
    for k, t in enumerate(tensors):
        part = group_norm_partial(t, groups)
        keep.append(part)
        a.partial[k] = part.data_ptr()
        a.nslot[k] = part.shape[1]
        a.entries[k] = part.shape[2]
        a.tiles[k] = tiles[k]
        a.count[k] = float(t.shape[1] * t.shape[2] * (t.shape[3] // groups))
        a.weight[k] = t.shape[1] * t.shape[2] / tot
    a.ngroups = len(tensors)
    dev = tensors[0].device
    mean = torch.empty((N, groups), device=dev, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    var = torch.empty_like(mean)
    check(_lib.load().omgsr_groupnorm_finalize_merged(C.byref(a), mean.data_ptr(), rstd.data_ptr(), var.data_ptr(), N, groups,
                                                      eps, _stream()), "omgsr_groupnorm_finalize_merged")
    return mean, rstd, var


def group_norm_apply_shared(x: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor, gamma, beta, groups: int,
                            act: int = ACT_NONE) -> torch.Tensor:
    """x [T*N, ..., C] tile-major; mean / rstd [N, G]: row r is normalised with the statistics of image r % N."""
    _req(x, act_dtype(), "x")
    rows, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (rows * Cc)
    y = torch.empty_like(x)
    check(_lib.load().omgsr_groupnorm_apply_shared(x.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _ptr(gamma),
                                                   _ptr(beta), rows, HW, Cc, groups, act, mean.shape[0], _stream()),
          "omgsr_groupnorm_apply_shared")
    return y


def group_norm_apply(x: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor, gamma: Optional[torch.Tensor],
                     beta: Optional[torch.Tensor], groups: int, act: int = ACT_NONE, inplace: bool = False) -> torch.Tensor:
    _req(x, act_dtype(), "x")
    N, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (N * Cc)
    y = x if inplace else torch.empty_like(x)
    check(_lib.load().omgsr_groupnorm_apply(x.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _ptr(gamma),
                                            _ptr(beta), N, HW, Cc, groups, act, _stream()), "omgsr_groupnorm_apply")
    return y


def group_norm(x: torch.Tensor, gamma, beta, groups: int, eps: float, act: int = ACT_NONE) -> torch.Tensor:
    mean, rstd, _ = group_norm_stats(x, groups, eps)
    return group_norm_apply(x, mean, rstd, gamma, beta, groups, act)


# --------------------------------------------------------------------------------------------
# K9/K10: LayerNorm (affine or AdaLN-modulated)

def layer_norm(x: torch.Tensor, a: Optional[torch.Tensor], b: Optional[torch.Tensor], eps: float) -> torch.Tensor:
    _req(x, act_dtype(), "x")
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    y = torch.empty_like(x)
    check(_lib.load().omgsr_layernorm(x.data_ptr(), y.data_ptr(), _ptr(a), _ptr(b), rows, Cc, eps, _stream()), "omgsr_layernorm")
    return y


# --------------------------------------------------------------------------------------------
# K7/K8: attention

def attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, heads: int, head_dim: int, scale: float,
              *, q_col: int = 0, k_col: int = 0, Lk: Optional[int] = None, out: Optional[torch.Tensor] = None,
              o_col: int = 0) -> torch.Tensor:
    """q [B, Lq, *] (heads at columns q_col + h*D), k [Bk, Lk, *], vt [Bk, heads*D, ld] -> o [B, Lq, heads*D].
    Bk == 1 broadcasts one K/V over the batch (constant cross-attention context)."""
    _req(q, act_dtype(), "q"); _req(k, act_dtype(), "k"); _req(vt, act_dtype(), "vt")
    B, Lq = q.shape[0], q.shape[1]
    Bk = k.shape[0]
    Lk = Lk if Lk is not None else k.shape[1]
    inner = heads * head_dim
    if out is None:
        out = torch.empty((B, Lq, inner), device=q.device, dtype=act_dtype())
    a = AttnArgs()
    esz = 2
    a.q = q.data_ptr() + q_col * esz
    a.k = k.data_ptr() + k_col * esz
    a.vt = vt.data_ptr()
    a.o = out.data_ptr() + o_col * esz
    a.B, a.H, a.D, a.Lq, a.Lk = B, heads, head_dim, Lq, Lk
    a.q_ld, a.k_ld, a.vt_ld, a.o_ld = q.shape[-1], k.shape[-1], vt.shape[-1], out.shape[-1]
    a.q_bstride = Lq * q.shape[-1]
    a.k_bstride = 0 if Bk == 1 and B > 1 else k.shape[1] * k.shape[-1]
    a.vt_bstride = 0 if Bk == 1 and B > 1 else vt.shape[1] * vt.shape[-1]
    a.o_bstride = Lq * out.shape[-1]
    a.scale = scale
    check(_lib.load().omgsr_attention(C.byref(a), _stream()), "omgsr_attention")
    return out


def softmax_rows(s: torch.Tensor, valid: Optional[int] = None) -> torch.Tensor:
    """softmax over the first `valid` columns of each row (the rest come out 0)."""
    _req(s, torch.float32, "s")
    L = s.shape[-1]
    rows = s.numel() // L
    p = torch.empty(s.shape, device=s.device, dtype=act_dtype())
    check(_lib.load().omgsr_softmax_rows(s.data_ptr(), p.data_ptr(), rows, L, valid or L, _stream()), "omgsr_softmax_rows")
    return p


def rmsnorm_rope_(x: torch.Tensor, w: torch.Tensor, cos: Optional[torch.Tensor], sin: Optional[torch.Tensor],
                  heads: int, head_dim: int, col0: int = 0, pos0: int = 0, eps: float = 1e-6) -> torch.Tensor:
    """In place on x [B, L, ld]: per head RMSNorm(head_dim) * w[h] then RoPE with cos/sin [>= pos0+L, D] (f32).
    w is [heads, D]: one call can cover the q heads and the k heads of a fused [q|k] buffer."""
    _req(x, act_dtype(), "x"); _req(w, torch.float32, "w")
    B, L, ld = x.shape
    if tuple(w.shape) != (heads, head_dim):
        raise ValueError(f"rmsnorm_rope_: w must be [{heads}, {head_dim}], got {tuple(w.shape)}")
    check(_lib.load().omgsr_rmsnorm_rope(x.data_ptr(), w.data_ptr(), _ptr(cos), _ptr(sin), B, L, heads, head_dim, ld,
                                         col0, pos0, eps, _stream()), "omgsr_rmsnorm_rope")
    return x


# --------------------------------------------------------------------------------------------
# K14: layout + latent algebra

def nchw_to_nhwc(x: torch.Tensor, cpad: Optional[int] = None) -> torch.Tensor:
    if x.dtype not in (torch.float32, act_dtype()):
        x = x.to(act_dtype())          # the other 16-bit type / f64: one conversion at the boundary
    _req(x, x.dtype, "x")
    N, Cc, H, W = x.shape
    cpad = cpad or _round_up(Cc, 8)
    y = torch.empty((N, H, W, cpad), device=x.device, dtype=act_dtype())
    check(_lib.load().omgsr_nchw_to_nhwc(x.data_ptr(), y.data_ptr(), N, Cc, H, W, cpad, int(x.dtype == torch.float32), _stream()),
          "omgsr_nchw_to_nhwc")
    return y


def nhwc_to_nchw(x: torch.Tensor, channels: Optional[int] = None, dtype=None,
                 clamp: Optional[tuple[float, float]] = None) -> torch.Tensor:
    _req(x, act_dtype(), "x")
    dtype = dtype or act_dtype()
    if dtype not in (torch.float32, act_dtype()):
        raise TypeError(f"nhwc_to_nchw: output dtype must be float32 or {act_dtype()}, got {dtype}")
    N, H, W, ld = x.shape
    Cc = channels or ld
    y = torch.empty((N, Cc, H, W), device=x.device, dtype=dtype)
    lo, hi = clamp if clamp else (0.0, 0.0)
    check(_lib.load().omgsr_nhwc_to_nchw(x.data_ptr(), y.data_ptr(), N, Cc, H, W, ld, int(dtype == torch.float32),
                                         int(clamp is not None), lo, hi, _stream()), "omgsr_nhwc_to_nchw")
    return y


def concat_channels(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """torch.cat([a, b], dim=channel) for NHWC bf16."""
    _req(a, act_dtype(), "a"); _req(b, act_dtype(), "b")
    Ca, Cb = a.shape[-1], b.shape[-1]
    rows = a.numel() // Ca
    out = torch.empty((*a.shape[:-1], Ca + Cb), device=a.device, dtype=act_dtype())
    lib = _lib.load()
    check(lib.omgsr_copy_channels(a.data_ptr(), out.data_ptr(), rows, Ca, Ca, Ca + Cb, 0, _stream()), "omgsr_copy_channels")
    check(lib.omgsr_copy_channels(b.data_ptr(), out.data_ptr(), rows, Cb, Cb, Ca + Cb, Ca, _stream()), "omgsr_copy_channels")
    return out


def vae_sample(moments: torch.Tensor, eps: torch.Tensor, latent_channels: int, shift: float, scale: float,
               ld_out: Optional[int] = None) -> torch.Tensor:
    """moments [N,h,w,2C] bf16, eps [N,h,w,C] f32 -> z [N,h,w,ld_out] bf16 (zero padded channels)."""
    _req(moments, act_dtype(), "moments"); _req(eps, torch.float32, "eps")
    N, h, w, _ = moments.shape
    ld_out = ld_out or _round_up(latent_channels, 8)
    z = torch.empty((N, h, w, ld_out), device=moments.device, dtype=act_dtype())
    check(_lib.load().omgsr_vae_sample(moments.data_ptr(), eps.data_ptr(), z.data_ptr(), N * h * w, latent_channels, ld_out,
                                       shift, scale, _stream()), "omgsr_vae_sample")
    return z


def axpby(x: torch.Tensor, y: Optional[torch.Tensor], a: float, b: float, c: float = 0.0, d: float = 1.0,
          bf16_steps: bool = False) -> torch.Tensor:
    _req(x, act_dtype(), "x")
    out = torch.empty_like(x)
    check(_lib.load().omgsr_axpby(x.data_ptr(), _ptr(y), out.data_ptr(), x.numel(), a, b, c, d, int(bf16_steps), _stream()),
          "omgsr_axpby")
    return out


def tile_accumulate(tile: Optional[torch.Tensor], w: torch.Tensor, acc: torch.Tensor, y0: int, x0: int,
                    channels: Optional[int] = None) -> None:
    """acc [N,H,W,C] f32 += tile[..., :C] * w[th,tw]; tile None accumulates the weights alone (acc [1,H,W,1])."""
    _req(w, torch.float32, "w"); _req(acc, torch.float32, "acc")
    N, H, W, Cc = acc.shape
    th, tw = w.shape
    tile_ld = 0
    if tile is not None:
        _req(tile, act_dtype(), "tile")
        tile_ld = tile.shape[-1]
    check(_lib.load().omgsr_tile_accumulate(_ptr(tile), w.data_ptr(), acc.data_ptr(), N, Cc, th, tw, tile_ld, H, W, y0, x0,
                                            _stream()), "omgsr_tile_accumulate")


def tile_normalise(acc: torch.Tensor, wsum: torch.Tensor, ld: Optional[int] = None) -> torch.Tensor:
    N, H, W, Cc = acc.shape
    ld = ld or _round_up(Cc, 8)
    out = torch.empty((N, H, W, ld), device=acc.device, dtype=act_dtype())
    check(_lib.load().omgsr_tile_normalise(acc.data_ptr(), wsum.data_ptr(), out.data_ptr(), N, H * W, Cc, ld, _stream()),
          "omgsr_tile_normalise")
    return out


def crop_nhwc(x: torch.Tensor, y0: int, x0: int, th: int, tw: int) -> torch.Tensor:
    _req(x, act_dtype(), "x")
    N, H, W, Cc = x.shape
    out = torch.empty((N, th, tw, Cc), device=x.device, dtype=act_dtype())
    check(_lib.load().omgsr_crop_nhwc(x.data_ptr(), out.data_ptr(), N, H, W, Cc, y0, x0, th, tw, _stream()), "omgsr_crop_nhwc")
    return out


def paste_nhwc(src: torch.Tensor, dst: torch.Tensor, sy0: int, sx0: int, dy0: int, dx0: int, th: int, tw: int) -> None:
    """dst[:, dy0:dy0+th, dx0:dx0+tw, :] = src[:, sy0:sy0+th, sx0:sx0+tw, :] (bf16 NHWC, same N and C)."""
    _req(src, act_dtype(), "src"); _req(dst, act_dtype(), "dst")
    N, sH, sW, Cc = src.shape
    if dst.shape[0] != N or dst.shape[3] != Cc:
        raise ValueError("paste_nhwc: batch/channel mismatch")
    check(_lib.load().omgsr_paste_nhwc(src.data_ptr(), dst.data_ptr(), N, Cc, sH, sW, sy0, sx0, dst.shape[1], dst.shape[2],
                                       dy0, dx0, th, tw, _stream()), "omgsr_paste_nhwc")


def flux_pack(x: torch.Tensor, channels: int) -> torch.Tensor:
    """NHWC [N,H,W,ld] (first `channels`) -> tokens [N, (H/2)(W/2), 4*channels]."""
    _req(x, act_dtype(), "x")
    N, H, W, ld = x.shape
    out = torch.empty((N, (H // 2) * (W // 2), 4 * channels), device=x.device, dtype=act_dtype())
    check(_lib.load().omgsr_flux_pack(x.data_ptr(), out.data_ptr(), N, H, W, channels, ld, 0, _stream()), "omgsr_flux_pack")
    return out


def flux_unpack(tok: torch.Tensor, H: int, W: int, ld: Optional[int] = None) -> torch.Tensor:
    _req(tok, act_dtype(), "tok")
    N, _, c4 = tok.shape
    Cc = c4 // 4
    ld = ld or _round_up(Cc, 8)
    out = torch.zeros((N, H, W, ld), device=tok.device, dtype=act_dtype()) if ld != Cc else \
        torch.empty((N, H, W, ld), device=tok.device, dtype=act_dtype())
    check(_lib.load().omgsr_flux_pack(tok.data_ptr(), out.data_ptr(), N, H, W, Cc, ld, 1, _stream()), "omgsr_flux_pack")
    return out
