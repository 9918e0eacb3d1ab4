"""Tensor-level wrappers over the C ABI (include/omgsr_hip.h).

torch is used for device memory and streams only: every function here validates shapes, allocates
the output with torch.empty and launches a hand-written gfx950 kernel on torch's current stream.
Activations are channels-last ([N, H, W, C]; token matrices are [B, L, C]).

Two tensor classes (include/omgsr_hip.h, OMGSR_EL_*):
  operand tensors  feed an MFMA: the 16-bit compute type `act_dtype()`; optionally the two-term split
                   x = hi + lo stored as [hi (C) | lo (C)] per row (`split=2`), consumed by a weight packed with
                   duplicated input channels, so the fp32 accumulator receives hi*W + lo*W
  stream tensors   everything between two GEMMs: `stream_dtype()` = the compute type in the fast tiers
                   (--weight_dtype bf16 | fp16), fp32 in the accurate tier (--weight_dtype fp32)
Every function dispatches on the dtype of the tensors it is handed.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import AttnArgs, IgemmArgs, check

ACT_NONE, ACT_SILU, ACT_GELU_TANH, ACT_GEGLU = 0, 1, 2, 3
OUT_STREAM, OUT_BF16, OUT_F32 = -1, 0, 1      # conv / linear outputs: stream tensor (default) | 16-bit operand | fp32
LAYOUT_NHWC, LAYOUT_T = 0, 1
EL_16, EL_F32, EL_SPLIT, EL_MX, EL_MX6 = 0, 1, 2, 3, 4
MX_LO_SHIFT = 11          # OMGSR_MX_LO_SHIFT (csrc/common.hip.h): a_lo' = (a - a_hi) * 2^11 as fp8


def _el_of_split(split: int) -> int:
    """Operand form of a `split` code: 1 plain, 2 two-term split [hi | lo] (both fp16), 3 the mixed-precision form
    [hi fp16 | lo' fp8 | hi' fp8] (OMGSR_EL_MX: same row width as 2, its correction segments run as block-scaled fp8 MFMAs)."""
    return {1: EL_16, 2: EL_SPLIT, 3: EL_MX, 4: EL_MX6}[split]


_ACT = torch.bfloat16
_PRECISE = False


def act_dtype() -> torch.dtype:
    """The 16-bit element type of every MFMA operand / packed weight (bf16 default, fp16 optional)."""
    return _ACT


def stream_dtype() -> torch.dtype:
    """Element type of the tensors between GEMMs: the compute type, or fp32 in the accurate tier."""
    return torch.float32 if _PRECISE else _ACT


def precise() -> bool:
    return _PRECISE


def attn_split() -> bool:
    """The UNet's attention takes q, k (from their projections' epilogues) and P (in registers) as two-term splits: in the range-fallback tier
    (fp32 stream, bf16 operands: 8-bit mantissas), where single q / k / P cost the north-star tolerance on OMGSR-S 256->1024 (1.15e-3).
    OMGSR_ATTN_SPLIT=0 | 1 forces it off / on in the accurate tiers (A/B runs)."""
    env = os.environ.get("OMGSR_ATTN_SPLIT")
    if env is not None and _PRECISE:
        return env != "0"
    return bool(_PRECISE and _ACT == torch.bfloat16)


def mode_key() -> tuple:
    """Cache key of anything derived from weights under the current tier."""
    return (_ACT, _PRECISE)


def set_compute_dtype(dtype: torch.dtype, operand_dtype: Optional[torch.dtype] = None) -> None:
    """Process-wide tier switch; packed-weight caches are keyed on it and rebuild lazily. Mirrors the reference's
    --weight_dtype (infer/infer_omgsr_s.py:134-149):
      bf16 / fp16  fast tiers: operands AND stream tensors in that 16-bit type (omgsr_set_compute_dtype)
      fp32         accurate tier: fp32 stream tensors, fp16 MFMA operands with fp32 accumulation, two-term split
                   operands / weights on the layers a precision policy names (omgsr_amd.precision)
    operand_dtype (accurate tier only): torch.bfloat16 runs the same tier with bf16 operands - fp32's exponent range, 8-bit
    mantissas (16 with the two-term split): the range-safe fallback the fp16 range guard (overflow_seen) switches to."""
    global _ACT, _PRECISE
    if dtype not in (torch.bfloat16, torch.float16, torch.float32):
        raise TypeError(f"compute dtype must be bfloat16, float16 or float32, got {dtype}")
    if operand_dtype is not None and (dtype != torch.float32 or operand_dtype not in (torch.float16, torch.bfloat16)):
        raise TypeError("operand_dtype applies to the accurate tier (dtype float32) and must be float16 or bfloat16")
    act = torch.bfloat16 if dtype == torch.bfloat16 else (operand_dtype or torch.float16)
    check(_lib.load().omgsr_set_compute_dtype(0 if act == torch.bfloat16 else 1), "set_compute_dtype")
    # fast tiers: deferred softmax maximum (+5 % attention); accurate tier: exact running maximum
    # (OMGSR_ATTN_DEFER_ACCURATE=<log2 threshold>: A/B runs of a deferred maximum in the accurate tier)
    acc_defer = float(os.environ.get("OMGSR_ATTN_DEFER_ACCURATE", "0"))
    check(_lib.load().omgsr_set_attention_defer_max(acc_defer if dtype == torch.float32 else 8.0), "set_attention_defer_max")
    _ACT, _PRECISE = act, dtype == torch.float32


# ---- fp16 range guard (accurate tier) -----------------------------------------------------------------------------------------
# fp16 operands saturate at +-65504 (pack2 clips instead of producing inf). Every kernel that writes a 16-bit operand from values
# that are not normalised - GEMM / conv epilogues (q, k, v, FF hidden, operand-only outputs), stream -> operand casts - ORs 1 into a
# device word when it clips one. The pipelines read the word once per call, at the synchronisation the reference's forward() already
# has, and fall back to bf16 operands for that call instead of returning a silently clipped image.
_GUARD = True
_ovf_words: dict = {}


def set_range_guard(on: bool) -> None:
    global _GUARD
    _GUARD = bool(on)


def _ovf(device) -> Optional[int]:
    """Device pointer of the overflow word when the guard applies (accurate tier with fp16 operands), else None."""
    if not (_GUARD and _PRECISE and _ACT == torch.float16):
        return None
    key = str(device)
    t = _ovf_words.get(key)
    if t is None:
        t = _ovf_words[key] = torch.zeros(1, device=device, dtype=torch.int32)
    return t.data_ptr()


_weight_overflow = False        # a WEIGHT beyond the fp16 range was packed for fp16 operands (pack_conv_weight saturates it at +-65504)


def _note_weight_range(w: torch.Tensor) -> torch.Tensor:
    """Accurate tier, fp16 operands: a weight beyond +-65504 cannot be carried by an fp16 operand (the reason the reference defaults
    FLUX to bf16). Saturate it like the kernels saturate activations and raise the same guard the activations raise, so the pipelines
    fall back to bf16 operands instead of multiplying by inf."""
    global _weight_overflow
    if _PRECISE and _ACT == torch.float16 and w.numel() and float(w.abs().max()) > 65504.0:
        if _GUARD:
            _weight_overflow = True
        return w.clamp(-65504.0, 65504.0)
    return w


def overflow_seen(reset: bool = True) -> bool:
    """True when a kernel clipped an fp16 operand (or a weight beyond the fp16 range was packed) since the last reset (synchronises:
    one 4-byte read per device in use)."""
    global _weight_overflow
    seen = _weight_overflow
    if reset:
        _weight_overflow = False
    global _mx_saturated
    for t in _ovf_words.values():
        v = int(t.item())
        if v & 2:
            _mx_saturated = True
        if v & 1:
            seen = True
        if v and reset:
            t.zero_()
    return seen


_mx_saturated = False


def mx_saturation_seen(reset: bool = True) -> bool:
    """Diagnostic: True when, since the last reset, a kernel wrote a value beyond +-448 into a mixed-precision (OMGSR_EL_MX)
    operand. Its fp8 correction fields use fixed scales and clamp there, so that element carried single-rounding fp16 accuracy (nothing
    becomes NaN, the fp16 main term is exact). The word is read with the range guard's (overflow_seen, at the pipelines' own sync)."""
    global _mx_saturated
    for t in _ovf_words.values():
        if int(t.item()) & 2:
            _mx_saturated = True
            if reset:
                t.bitwise_and_(1)
    seen = _mx_saturated
    if reset:
        _mx_saturated = False
    return seen


_BATCH_INVARIANT = False


def set_batch_invariant(on: bool) -> None:
    """Process-wide: make kernel-family / split-K choices depend on one sample's size instead of the batch's (omgsr_set_batch_invariant):
    a batch of B then equals B batch-1 calls bit for bit, at some cost in speed for small batches."""
    global _BATCH_INVARIANT
    check(_lib.load().omgsr_set_batch_invariant(int(bool(on))), "omgsr_set_batch_invariant")
    _BATCH_INVARIANT = bool(on)


STAGE_NONE, STAGE_ENCODE, STAGE_DENOISE, STAGE_DECODE = 0, 1, 2, 3


def timing_stage(stage: int) -> None:
    """Tag the launches that follow for the per-launch timing leg (omgsr_timing_stage): the pipelines mark VAE encode / denoiser /
    VAE decode so bench.py can report per-stage time and the denoiser-only MFMA fraction. One host-side store; free when timing is off."""
    _lib.load().omgsr_timing_stage(stage)


def compute_dtype_name() -> str:
    return "fp32" if _PRECISE else ("bf16" if _ACT == torch.bfloat16 else "fp16")


def io_dtype(x: torch.Tensor) -> torch.dtype:
    """Boundary dtype the kernels can write directly for a caller holding `x`: f32 stays f32, else the compute dtype."""
    return torch.float32 if x.dtype == torch.float32 else _ACT


def _el(t: torch.Tensor, name: str) -> int:
    """Element kind of a stream-or-operand tensor argument."""
    if not t.is_cuda:
        raise _lib.OmgsrError(f"{name}: tensor is on {t.device}; the OMGSR HIP path runs on an MI355X only")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if t.dtype == torch.float32:
        return EL_F32
    if t.dtype != _ACT:
        raise TypeError(f"{name}: expected {_ACT} or float32, got {t.dtype}")
    return EL_16


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


def _igemm(a: "IgemmArgs", device, what: str, z_batched_api: bool = False) -> None:
    """Launch omgsr_igemm; when the library wants to split K (small-M / long-K problems) hand it an fp32 scratch.
    z_batched_api: the caller is one of the entry points that put the images of a call on grid.z (linear_into / linear_rows / bmm_nt).
    The library cannot split K there when there is more than one image, so in batch-invariant mode it must not split the one-image
    call either (round 4: OMGSR-F batch 1 == batch N bit for bit under ops.set_batch_invariant)."""
    lib = _lib.load()
    need = 0 if (z_batched_api and _BATCH_INVARIANT) else lib.omgsr_igemm_workspace_bytes(C.byref(a))
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
    split: int = 1     # 2: the input is a two-term split operand [hi | lo]: every input channel packed twice ([w | w]);
                       # 3: the mixed-precision form (OMGSR_EL_MX operand; fp16 [w_hi] + fp8 [w_hi' | w_lo'] per tap, `mx`)
    mx: Optional[tuple] = None      # split 3: (fp16 chunks per tap, E8M0 scale of w_hi', of a_lo', of w_lo', of a_hi')
    mx_fmt: int = 0                 # 6: split 4, the fp6 form (OMGSR_EL_MX6: e2m3 codes + per-block scale bytes in the data; `mx` scales unused)
    w_split: int = 1   # 2: the weight itself is carried as w_hi + w_lo: one more K segment [w_lo] that re-reads the operand's
                       # first (hi) half - the contraction WRAPS (omgsr_igemm_args.in_ld); `cin` counts every segment
    in_ld: int = 0     # physical channels of the operand row this weight expects (split * padded Cin); 0 = cin
    w_ph: Optional[torch.Tensor] = None   # 3x3 convs applied to a nearest-2x upsampled map: the four phase-summed 2 x 2 kernels,
                                          # [4][Kc/32][4 taps][Cout_pad][32] (omgsr_igemm_args.weight_ph): 4 / 9 of the MFMA work

    @property
    def cout_pad(self) -> int:
        return self.w.shape[0]

    @property
    def k_pad(self) -> int:
        return self.w.shape[1]

    @property
    def row_channels(self) -> int:
        """Channels of one operand row ([a] or [a_hi | a_lo]); < cin when the weight carries its own low half (w_split 2)."""
        return self.in_ld or self.cin


def _segments(w: torch.Tensor, split: int, w_split: int) -> torch.Tensor:
    """[..., Cin8] fp32 -> [..., Kc]: the K-concatenation of pack_conv_weight ([w_hi] * split, then [w_lo] when w_split 2), every value
    already rounded to the compute type."""
    w_hi = w.to(act_dtype()).float()
    segs = [w_hi] * split           # [hi (cin8) | lo (cin8)] operand rows: both halves meet the same (rounded) weights
    if w_split == 2:
        segs.append((w - w_hi).to(act_dtype()).float())      # ... and the operand's hi half meets the weights' low halves
    return torch.cat(segs, dim=-1) if len(segs) > 1 else w_hi


def _mx_rows(w: torch.Tensor, s1: int, s2: int) -> torch.Tensor:
    """[..., C] fp32 -> [..., 4C] uint8: [w_hi fp16 | fp8(w_hi 2^s1) | fp8((w - w_hi) 2^s2)]."""
    w_hi = w.to(torch.float16)
    hi8 = (w_hi.float() * 2.0 ** s1).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
    lo8 = ((w - w_hi.float()) * 2.0 ** s2).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
    return torch.cat([w_hi.contiguous().view(torch.uint8).reshape(*w.shape[:-1], 2 * w.shape[-1]), hi8, lo8], dim=-1)


def _e2m3_blocks(v: torch.Tensor) -> torch.Tensor:
    """[..., C] fp32 (C % 64 == 0) -> [..., C] uint8: one correction third of an OMGSR_EL_MX6 row (include/omgsr_hip.h). Every 64-byte group covers
    64 channels = two 32-channel blocks; block h owns bytes [16h, 16h + 16) and [32 + 16h, 40 + 16h) (its 192-bit string of e2m3 codes, channel i
    at bits [6i, 6i + 6)), byte 40 + 16h (E8M0 scale = max(0, biased exponent of the block's largest magnitude - 2)) and 7 zero bytes. Codes round
    to nearest even and saturate at +-7.5. The SAME integer arithmetic as the device producers (csrc/common.hip.h store8_mx6): tests compare bytes."""
    lead, Cc = v.shape[:-1], v.shape[-1]
    b = v.reshape(-1, Cc // 32, 32).float()
    m = b.abs().amax(-1)
    sb = ((m.view(torch.int32) >> 23) - 2).clamp(min=0)                                    # [R, nb] scale bytes
    inv = ((254 - sb) << 23).to(torch.int32).view(torch.float32).unsqueeze(-1)
    s_ = b * inv
    a = s_.abs().clamp(max=7.5)
    eb = (a.view(torch.int32) >> 23).clamp(min=127)
    q = torch.round(a * ((257 - eb) << 23).to(torch.int32).view(torch.float32)).to(torch.int64)      # torch.round: half to even, like v_rndne
    code = (q + ((eb.to(torch.int64) - 127) << 3)) | (((s_.view(torch.int32).to(torch.int64) >> 26) & 32))
    # four codes = 24 bits = three bytes: code i sits at bits [6i, 6i + 6) of the block's little-endian 192-bit string
    c4 = code.reshape(*code.shape[:-1], 8, 4)
    v24 = c4[..., 0] | (c4[..., 1] << 6) | (c4[..., 2] << 12) | (c4[..., 3] << 18)
    string = torch.stack([v24 & 0xFF, (v24 >> 8) & 0xFF, (v24 >> 16) & 0xFF], dim=-1).to(torch.uint8).reshape(*code.shape[:-1], 24)
    R, nb = string.shape[0], string.shape[1]
    out = torch.zeros((R, nb // 2, 64), dtype=torch.uint8, device=v.device)
    st = string.reshape(R, nb // 2, 2, 24)
    sbv = sb.reshape(R, nb // 2, 2).to(torch.uint8)
    for h in (0, 1):
        out[..., 16 * h:16 * h + 16] = st[..., h, :16]
        out[..., 32 + 16 * h:40 + 16 * h] = st[..., h, 16:]
        out[..., 40 + 16 * h] = sbv[..., h]
    return out.reshape(*lead, Cc)


def _mx6_rows(w: torch.Tensor) -> torch.Tensor:
    """[..., C] fp32 -> [..., 4C] uint8: [w_hi fp16 | e2m3 blocks of w_hi | e2m3 blocks of w - w_hi] (meets an OMGSR_EL_MX6 operand's [a_hi | a_lo' | a_hi'])."""
    w_hi = w.to(torch.float16)
    return torch.cat([w_hi.contiguous().view(torch.uint8).reshape(*w.shape[:-1], 2 * w.shape[-1]), _e2m3_blocks(w_hi.float()), _e2m3_blocks(w - w_hi.float())], dim=-1)


def _mx_shift(t: torch.Tensor) -> int:
    """s with max |t| 2^s in [128, 256): e4m3 tops out at 448, its 17 binades below cover values 5 orders of magnitude under the largest."""
    import math
    m = float(t.abs().max())
    return max(-100, min(100, 7 - math.floor(math.log2(m)))) if m > 0 else 0


def _pack_mx(w: torch.Tensor, bias, cout: int, cin: int, R: int, S: int, dev, upsample_phases: bool = False, fmt: int = 8) -> "PackedWeight":
    """[Cout, R, S, C] fp32 -> the mixed-precision weight of an OMGSR_EL_MX operand: per tap 4C bytes = 2C 16-bit slots,
    [w_hi fp16 (2C B) | w_hi' fp8 (C B) | w_lo' fp8 (C B)] with w_hi = fp16(w), w_hi' = fp8(w_hi 2^s1), w_lo' = fp8((w - w_hi) 2^s2): the
    first C / 32 chunks meet a_hi in fp16 MFMAs, then C / 64 fp8 chunks meet a_lo' (w_hi') and C / 64 meet a_hi' (w_lo') in block-scaled
    fp8 MFMAs whose E8M0 scale operands undo s1 / s2 and the operand's 2^11.
    fmt 6 (split 4, OMGSR_EL_MX6; 3x3 convs of the halo-tile kernel only): the two correction thirds hold fp6 (e2m3) codes with a scale byte per
    32-channel block inside the data (_e2m3_blocks) - the MFMA runs them in half the passes of fp8."""
    if fmt == 6 and (R, S) != (3, 3):
        raise ValueError("the fp6 form (split 4) serves the 3x3 stride-1 convolutions of the halo-tile kernel")
    rows = (lambda t: _mx6_rows(t)) if fmt == 6 else (lambda t: _mx_rows(t, s1, s2))
    if (R, S) not in ((3, 3), (1, 1)) or cin % 64 or act_dtype() != torch.float16:
        raise ValueError("the mixed-precision (MX) form serves 3x3 and 1x1 convolutions (linears) with Cin % 64 == 0 in the fp16 compute type")
    ph = _phase_kernels(w) if upsample_phases else None       # [4, Cout, 2, 2, C]: one set of scales for the taps AND the phase sums
    both = w if ph is None else torch.cat([w.reshape(-1), ph.reshape(-1)])
    both_hi = both.to(torch.float16).float()
    s1, s2 = _mx_shift(both_hi), _mx_shift(both - both_hi)
    kslots = 2 * cin                                           # 16-bit slots per tap
    cout_pad = _round_up(cout, 256 if cout >= 256 else 128)
    out = torch.zeros((cout_pad, R * S * kslots), device=dev, dtype=torch.float16)
    out[:cout] = rows(w).reshape(cout, R * S * 4 * cin).contiguous().view(torch.float16)
    # 3x3: slice-major copy for the halo-tile kernel; 1x1 (a Linear): `out` itself is what igemm_gmx_kernel streams, row by row
    w_cm = out.view(cout_pad, 9, kslots // 32, 32).permute(2, 1, 0, 3).contiguous() if R == 3 else None
    w_ph = None
    if ph is not None:
        full = torch.zeros((4, cout_pad, 4 * kslots), device=dev, dtype=torch.float16)
        full[:, :cout] = rows(ph).reshape(4, cout, 4 * 4 * cin).contiguous().view(torch.float16)
        w_ph = full.view(4, cout_pad, 4, kslots // 32, 32).permute(0, 3, 2, 1, 4).contiguous()         # [phase][chunk][tap][Cout_pad][32]
    b = None if bias is None else bias.detach().to(device=dev, dtype=torch.float32).contiguous()
    if fmt == 6:
        return PackedWeight(out, b, cout, kslots, R, S, w_cm=w_cm, split=4, w_split=1, in_ld=kslots, w_ph=w_ph, mx=(cin // 32, 127, 127, 127, 127), mx_fmt=6)
    return PackedWeight(out, b, cout, kslots, R, S, w_cm=w_cm, split=3, w_split=1, in_ld=kslots, w_ph=w_ph,
                        mx=(cin // 32, 127 - s1, 127 - MX_LO_SHIFT, 127 - s2, 127))


def _phase_kernels(w: torch.Tensor) -> torch.Tensor:
    """[Cout, 3, 3, C] -> [4, Cout, 2, 2, C]: conv3x3(nearest_up2(x)) at output pixel (2y + a, 2x + b) only sees input rows
    y - 1 + a, y + a (columns likewise): the taps that land on one input pixel are summed (in fp64), phase index 2a + b."""
    rows = {0: ([0], [1, 2]), 1: ([0, 1], [2])}
    w64 = w.double()
    out = []
    for a in (0, 1):
        for b in (0, 1):
            k = torch.zeros((w.shape[0], 2, 2, w.shape[3]), dtype=torch.float64, device=w.device)
            for dy, rs in enumerate(rows[a]):
                for dx, ss in enumerate(rows[b]):
                    for r in rs:
                        for s_ in ss:
                            k[:, dy, dx] += w64[:, r, s_]
            out.append(k.float())
    return torch.stack(out, 0)


def pack_conv_weight(weight: torch.Tensor, bias: Optional[torch.Tensor], device=None, cout_multiple: int = 1,
                     split: int = 1, w_split: int = 1, upsample_phases: bool = False) -> PackedWeight:
    """[Cout, Cin, R, S] (torch conv layout) -> [Cout_pad, roundup(R*S*Kc, 32)] in the compute type, k = (r*S+s)*Kc + c.
    cout_multiple=8 widens the LOGICAL output to a multiple of 8 channels (zero weights, zero bias) so a
    3/4-channel conv writes 16-byte rows that the next kernel can consume directly.
    Per tap the contraction is a K-concatenation of segments of Cin8 channels, Kc = Cin8 * (split + w_split - 1):
      split 1, w_split 1   [w_hi]                 x [a]            one rounding of a, one of w
      split 2, w_split 1   [w_hi | w_hi]          x [a_hi | a_lo]  a to 2^-22
      split 1, w_split 2   [w_hi | w_lo]          x [a] (wraps)    w to 2^-22
      split 2, w_split 2   [w_hi | w_hi | w_lo]   x [a_hi | a_lo] (the third segment wraps to a_hi): both to 2^-22
    with w_hi = round(w), w_lo = round(w - w_hi) in the compute type; everything accumulates in ONE fp32 accumulator."""
    cout, cin, R, S = weight.shape
    dev = device or weight.device
    if cout % cout_multiple:
        extra = _round_up(cout, cout_multiple) - cout
        weight = torch.cat([weight.detach().to(dev), torch.zeros((extra, cin, R, S), device=dev, dtype=weight.dtype)], dim=0)
        if bias is not None:
            bias = torch.cat([bias.detach().to(dev), torch.zeros(extra, device=dev, dtype=bias.dtype)], dim=0)
        cout += extra
    cin8 = _round_up(cin, 8)
    w = _note_weight_range(weight.detach().to(device=dev, dtype=torch.float32)).permute(0, 2, 3, 1)  # [Cout,R,S,Cin]
    if cin8 != cin:
        w = torch.nn.functional.pad(w, (0, cin8 - cin))
    if split in (3, 4):
        if cin8 != cin:
            raise ValueError("the mixed-precision (MX) form needs Cin % 64 == 0")
        return _pack_mx(w, bias if bias is None else bias.detach().to(dev), cout, cin, R, S, dev, upsample_phases, fmt=6 if split == 4 else 8)
    if split not in (1, 2) or w_split not in (1, 2):
        raise ValueError("split / w_split must be 1 or 2")
    if w_split == 2 and bool(torch.equal(w.to(act_dtype()).float(), w)):
        # every value is exact in the compute type (an fp16 / bf16 checkpoint's layers that no LoRA merge touched): w_lo == 0, and a
        # segment of zeros adds exactly 0 to the accumulator - drop it (same bits, one K segment less). The phase form of an up-sampling conv
        # carries SUMS of 2 - 4 taps (_phase_kernels), which are not representable just because the taps are (round 6: with bf16-representable
        # weights the range-fallback tier rounded those sums to 8 bits and measured 3.4e-3 where full-mantissa weights measure 1.1e-3): the
        # segment goes only if the summed kernels are exact too - one w_split serves both packings, their K layouts must agree
        ph_exact = True
        if upsample_phases and R == 3 and S == 3:
            ph0 = _phase_kernels(w)
            ph_exact = bool(torch.equal(ph0.to(act_dtype()).float(), ph0))
        if ph_exact:
            w_split = 1
    in_ld = cin8 * split
    w_raw = w
    w = _segments(w, split, w_split)
    cin8 = w.shape[-1]
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
    w_ph = None
    if upsample_phases and w_cm is not None and in_ld % 32 == 0:
        ph = _segments(_phase_kernels(w_raw), split, w_split).reshape(4, cout, 4 * cin8)          # k = (2 dy + dx) * Kc + c
        full = torch.zeros((4, cout_pad, 4 * cin8), device=dev, dtype=act_dtype())
        full[:, :cout] = ph.to(act_dtype())
        w_ph = full.view(4, cout_pad, 4, cin8 // 32, 32).permute(0, 3, 2, 1, 4).contiguous()         # [phase][chunk][tap][Cout_pad][32]
    return PackedWeight(out, b, cout, cin8, R, S, w_cm=w_cm, split=split, w_split=w_split, in_ld=in_ld, w_ph=w_ph)


def pack_linear_weight(weight: torch.Tensor, bias: Optional[torch.Tensor], device=None, split: int = 1, w_split: int = 1) -> PackedWeight:
    """[out, in] -> 1x1 'conv' weight."""
    return pack_conv_weight(weight[:, :, None, None], bias, device, split=split, w_split=w_split)


def pack_geglu_weight(weight: torch.Tensor, bias: Optional[torch.Tensor], device=None, split: int = 1, w_split: int = 1) -> PackedWeight:
    """GEGLU projection [2*inner, in] (rows [a | gate], diffusers `chunk(2, -1)`) -> rows interleaved in
    blocks of 32: [a_0..31, g_0..31, a_32..63, g_32..63, ...] so one 64-wide wave tile holds both halves."""
    two_inner, cin = weight.shape
    inner = two_inner // 2
    if inner % 32:
        raise ValueError("GEGLU inner dim must be a multiple of 32")
    a, g = weight[:inner], weight[inner:]
    w = torch.stack([a.reshape(inner // 32, 32, cin), g.reshape(inner // 32, 32, cin)], dim=1).reshape(two_inner, cin)
    pw = pack_linear_weight(w, None, device, split=split, w_split=w_split)
    if bias is not None:
        ba, bg = bias[:inner], bias[inner:]
        b = torch.stack([ba.reshape(-1, 32), bg.reshape(-1, 32)], dim=1).reshape(two_inner)
        pw.bias = b.detach().to(device=pw.w.device, dtype=torch.float32).contiguous()
    pw.cout = inner
    pw.geglu = True
    return pw


# --------------------------------------------------------------------------------------------
# K1-K3, K5, K6: implicit-GEMM conv / linear / bmm

def to_operand(x: torch.Tensor, split: int = 1) -> torch.Tensor:
    """Stream tensor -> MFMA operand [..., split*C]: a no-op for a tensor already in the compute type (fast tiers),
    one rounding (split 1) or the two-term split [hi | lo] (split 2) of an fp32 stream tensor."""
    if x.dtype == _ACT:
        if split != 1:
            raise ValueError("to_operand: a 16-bit tensor cannot be split (split operands exist in the accurate tier only)")
        return x
    _req(x, torch.float32, "x")
    Cc = x.shape[-1]
    y = torch.empty((*x.shape[:-1], min(split, 2) * Cc), device=x.device, dtype=_ACT)
    check(_lib.load().omgsr_to_operand(x.data_ptr(), y.data_ptr(), x.numel() // Cc, Cc, _el_of_split(split), _ovf(x.device),
                                       _stream()), "omgsr_to_operand")
    return y


def _out_tensor(shape, cout: int, out_dtype: int, out_split: int, device) -> torch.Tensor:
    if out_dtype == OUT_STREAM:
        out_dtype = OUT_F32 if _PRECISE else OUT_BF16
        if out_split != 1:
            raise ValueError("out_split applies to operand outputs (out_dtype=OUT_BF16)")
    if out_dtype == OUT_F32:
        return torch.empty((*shape, cout), device=device, dtype=torch.float32)
    return torch.empty((*shape, cout * min(out_split, 2)), device=device, dtype=_ACT)       # split 3 (MX): 4 bytes per channel, like split 2


def _fill_k(a: IgemmArgs, pw: PackedWeight) -> None:
    """Contraction geometry of a packed weight: Cin = every K segment of a tap, in_ld = the operand row it wraps over."""
    a.Cin = pw.cin
    a.overflow_flag = _ovf(pw.w.device)
    a.in_ld = pw.row_channels if pw.row_channels != pw.cin else 0
    a.in_split = int(pw.split >= 2)            # split 3 (MX): 2C slots per tap for C logical channels, like the two-term split
    a.w_split = int(pw.w_split == 2)
    if pw.mx is not None:
        a.mx_chunks16, a.mx_scale_w1, a.mx_scale_a1, a.mx_scale_w2, a.mx_scale_a2 = pw.mx
        a.mx_fmt = pw.mx_fmt


def _fill_out(a: IgemmArgs, out: torch.Tensor, out_split: int, residual: Optional[torch.Tensor], cout: int) -> None:
    a.out = out.data_ptr()
    a.out_dtype = OUT_F32 if out.dtype == torch.float32 else OUT_BF16
    a.out_lo_off = cout if out_split == 2 else 0
    a.out_mx = {3: 1, 4: 6}.get(out_split, 0)
    if residual is not None:
        a.residual = residual.data_ptr()
        a.res_el = _el(residual, "residual")


def _conv_args(a: IgemmArgs, x: torch.Tensor, pw: PackedWeight, stride, pad, upsample, act, residual, gate, out_dtype, alpha, out, out_split,
               sample_rows):
    """Fill `a` for one conv problem; returns (x as operand, out). Keep both alive until the launch."""
    if x.dtype == torch.float32:
        x = to_operand(x, pw.split)
    _req(x, act_dtype(), "x")
    N, H, W, Cin = x.shape
    if Cin != pw.row_channels:
        raise ValueError(f"conv2d: input has {Cin} channels, packed weight expects {pw.row_channels}")
    if isinstance(pad, int):
        pad = (pad, pad, pad, pad)
    pt, pb, pl, pr = pad
    Hv, Wv = (H * 2, W * 2) if upsample else (H, W)
    Ho = (Hv + pt + pb - pw.R) // stride + 1
    Wo = (Wv + pl + pr - pw.S) // stride + 1
    if pw.geglu:
        act = ACT_GEGLU
    if out is None:
        out = _out_tensor((N, Ho, Wo), pw.cout, out_dtype, out_split, x.device)
    if residual is not None and tuple(residual.shape) != (N, Ho, Wo, pw.cout):
        raise ValueError(f"residual shape {tuple(residual.shape)} != output {(N, Ho, Wo, pw.cout)}")
    a.in_, a.weight, a.bias, a.gate = x.data_ptr(), pw.w.data_ptr(), _ptr(pw.bias), _ptr(gate)
    _fill_out(a, out, out_split, residual, pw.cout)
    a.weight_cm = _ptr(pw.w_cm)
    a.weight_ph = _ptr(pw.w_ph) if upsample else None
    a.N, a.H, a.W = N, H, W
    _fill_k(a, pw)
    a.Cout, a.Cout_pad, a.K_pad = pw.cout, pw.cout_pad, pw.k_pad
    a.R, a.S, a.stride, a.pad_top, a.pad_left, a.upsample = pw.R, pw.S, stride, pt, pl, int(upsample)
    a.Ho, a.Wo = Ho, Wo
    a.act, a.out_layout = act, LAYOUT_NHWC
    a.t_rows, a.t_ld = 0, 0
    a.out_ld = out.shape[-1] if out_split == 2 else 0
    a.batch, a.in_bstride, a.w_bstride, a.out_bstride = 1, 0, 0, 0
    a.alpha = alpha
    a.sample_rows = sample_rows or Ho * Wo
    return x, out


def _conv_gn(a: IgemmArgs, gn_groups: int, out_split: int, device):
    """Ask the library whether this problem's epilogue can leave the GroupNorm statistics; allocates the partials if so."""
    if gn_groups <= 0 or out_split != 1:
        return None
    a.gn_groups = gn_groups
    lib = _lib.load()
    # a problem that will be split over K (it is handed the workspace) finishes in the reduce pass: no statistics there
    nslot = 0 if lib.omgsr_igemm_workspace_bytes(C.byref(a)) > 0 else lib.omgsr_igemm_gn_slots(C.byref(a))
    if nslot <= 0:
        return None
    a.gn_entries = lib.omgsr_igemm_gn_entries(C.byref(a))       # per group, or per channel for odd group sizes
    partial = torch.empty((a.N, nslot, a.gn_entries, 2), device=device, dtype=torch.float32)
    a.gn_partial = partial.data_ptr()
    return partial


class GnSpec:
    """A GroupNorm waiting for its consumer: statistics (mean / rstd [nimg, G]; rows n of the tensor use image n % nimg), affine, activation.
    conv2d(..., gn=spec) runs it as the conv's patch producer when the library says the problem qualifies (omgsr_igemm_gn_fusable: the
    halo-tile kernel's spatial 3x3 form over a 16-bit stream tensor, plain operand and weight) - the apply pass and the conv's read of its
    output disappear - and as the separate apply pass in front of the conv otherwise. Same values either way up to the 16-bit rounding the
    apply pass performs too (the fused form evaluates x * scale + shift from the same fp32 (scale, shift) table)."""

    def __init__(self, mean: torch.Tensor, rstd: torch.Tensor, gamma, beta, groups: int, act: int = ACT_NONE):
        self.mean, self.rstd, self.gamma, self.beta, self.groups, self.act = mean, rstd, gamma, beta, groups, act
        self._table = None

    def table(self, channels: int) -> torch.Tensor:
        """f32 [nimg, C, 2] = (rstd gamma, beta - mean rstd gamma): one tiny launch per GroupNorm, shared by every tile-shape group."""
        if self._table is None:
            nimg = self.mean.shape[0]
            t = torch.empty((nimg, channels, 2), device=self.mean.device, dtype=torch.float32)
            check(_lib.load().omgsr_groupnorm_scale_shift(self.mean.data_ptr(), self.rstd.data_ptr(), _ptr(self.gamma), _ptr(self.beta),
                                                          t.data_ptr(), nimg, channels, self.groups, _stream()), "omgsr_groupnorm_scale_shift")
            self._table = t
        return self._table

    def apply(self, x: torch.Tensor, split: int = 1) -> torch.Tensor:
        """The separate pass: stream tensor -> normalised operand in the form the consumer reads."""
        if self.mean.shape[0] != x.shape[0]:
            return group_norm_apply_shared(x, self.mean, self.rstd, self.gamma, self.beta, self.groups, self.act, split=split)
        return group_norm_apply(x, self.mean, self.rstd, self.gamma, self.beta, self.groups, self.act, split=split)


def _gn_candidate(x: torch.Tensor, pw: PackedWeight) -> bool:
    """Host-side part of the fusable test (the library decides the rest from the geometry): a 16-bit stream tensor, plain operand / weight."""
    return x.dtype == _ACT and pw.split == 1 and pw.w_split == 1 and pw.mx is None and pw.R == 3 and pw.S == 3


def _gn_fusable(a: IgemmArgs, gn: "GnSpec") -> bool:
    return gn.act == ACT_SILU and bool(_lib.load().omgsr_igemm_gn_fusable(C.byref(a)))


def _attach_gn(a: IgemmArgs, gn: "GnSpec", channels: int) -> torch.Tensor:
    t = gn.table(channels)
    a.gn_scale_shift, a.gn_nimg, a.gn_act, a.in_el = t.data_ptr(), t.shape[0], gn.act, EL_16
    return t


def conv2d(x: torch.Tensor, pw: PackedWeight, *, stride: int = 1, pad: tuple[int, int, int, int] | int = 1,
           upsample: bool = False, act: int = ACT_NONE, residual: Optional[torch.Tensor] = None,
           gate: Optional[torch.Tensor] = None, out_dtype: int = OUT_STREAM, alpha: float = 1.0,
           out: Optional[torch.Tensor] = None, gn_groups: int = 0, out_split: int = 1, sample_rows: int = 0,
           gn: Optional["GnSpec"] = None) -> torch.Tensor:
    """x [N,H,W,Cin] operand (or a stream tensor: cast / split here) -> [N,Ho,Wo,Cout]. pad = (top, bottom, left, right) on
    the (virtual) input. out_dtype: OUT_STREAM (a stream tensor: default), OUT_BF16 (a 16-bit operand for the next GEMM,
    `out_split` 2 = written as the two-term split) or OUT_F32. residual: a stream tensor of the output's shape.
    gn_groups > 0: the caller will GroupNorm the result with that many groups; when the kernel can, it emits the
    (sum, sum of squares) partials from its epilogue and group_norm_stats() skips its read pass over the tensor.
    gn: x is the stream tensor a GroupNorm (+ SiLU) reads and the conv consumes the normalised tensor (GnSpec): fused into the conv's
    patch producer where the library can, the apply pass in front of the conv otherwise."""
    a = IgemmArgs()
    if gn is not None and not _gn_candidate(x, pw):
        x, gn = gn.apply(x, pw.split), None
    x, out = _conv_args(a, x, pw, stride, pad, upsample, act, residual, gate, out_dtype, alpha, out, out_split, sample_rows)
    if out_split == 4 and not _lib.load().omgsr_igemm_out_mx6_ok(C.byref(a)):
        # the fp6 operand form comes out of the halo-tile kernel's dedicated instantiations only: anything else writes a stream tensor and the
        # cast kernel makes the operand (one more pass over the tensor; small maps and GEMM-shaped problems)
        return to_operand(conv2d(x, pw, stride=stride, pad=pad, upsample=upsample, act=act, residual=residual, gate=gate, out_dtype=OUT_STREAM,
                                 alpha=alpha, sample_rows=sample_rows, gn=gn), 4)
    keep = None
    if gn is not None:
        a.in_el = EL_16
        if _gn_fusable(a, gn):
            keep = _attach_gn(a, gn, x.shape[-1])
        else:
            x = gn.apply(x, pw.split)
            a.in_ = x.data_ptr()
    partial = _conv_gn(a, gn_groups, out_split, x.device)
    _igemm(a, x.device, "omgsr_igemm(conv2d)")
    del keep
    if partial is not None:
        out._omgsr_gn = (partial, gn_groups, out.data_ptr(), out._version)      # consumed by group_norm_stats(out)
    return out


def conv2d_multi(xs, pw: PackedWeight, *, stride: int = 1, pad: tuple[int, int, int, int] | int = 1, upsample: bool = False,
                 act: int = ACT_NONE, residuals=None, out_dtype: int = OUT_STREAM, gn_groups: int = 0, out_split: int = 1,
                 gn: Optional["GnSpec"] = None):
    """conv2d of several inputs with ONE weight (the tile-shape groups of a tiled-VAE layer: each x is its own dense [T*N, h, w, C]
    tensor) through omgsr_igemm_multi: the problems that take the halo-tile kernel run as one launch, with the kernel choice and the
    fused GroupNorm statistics planned for the group. Same results as a conv2d call per input. Returns the list of outputs."""
    n = len(xs)
    if n == 1:
        return [conv2d(xs[0], pw, stride=stride, pad=pad, upsample=upsample, act=act, residual=None if residuals is None else residuals[0],
                       out_dtype=out_dtype, gn_groups=gn_groups, out_split=out_split, gn=gn)]
    lib = _lib.load()
    if gn is not None and not all(_gn_candidate(x, pw) for x in xs):
        xs, gn = [gn.apply(x, pw.split) for x in xs], None
    arr = (IgemmArgs * n)()
    keep, outs = [], []
    for i in range(n):
        x, out = _conv_args(arr[i], xs[i], pw, stride, pad, upsample, act, None if residuals is None else residuals[i], None, out_dtype, 1.0,
                            None, out_split, 0)
        keep.append(x)
        outs.append(out)
        if gn is not None:
            arr[i].in_el = EL_16
    check(lib.omgsr_igemm_multi_plan(arr, n), "omgsr_igemm_multi_plan")
    if out_split == 4 and not all(lib.omgsr_igemm_out_mx6_ok(C.byref(arr[i])) for i in range(n)):
        ys = conv2d_multi(keep, pw, stride=stride, pad=pad, upsample=upsample, act=act, residuals=residuals, out_dtype=OUT_STREAM, gn=gn)
        return [to_operand(y, 4) for y in ys]
    if gn is not None:
        # one form per layer: every tile-shape group runs the normalising patch producer, or the apply pass runs for all of them
        if all(_gn_fusable(arr[i], gn) for i in range(n)):
            for i in range(n):
                keep.append(_attach_gn(arr[i], gn, keep[i].shape[-1]))
        else:
            for i in range(n):
                y = gn.apply(keep[i], pw.split)
                keep.append(y)
                arr[i].in_ = y.data_ptr()
    partials = []
    for i in range(n):
        partials.append(_conv_gn(arr[i], gn_groups, out_split, outs[i].device))
        need = lib.omgsr_igemm_workspace_bytes(C.byref(arr[i]))
        if need > 0:
            ws = torch.empty(need // 4, device=outs[i].device, dtype=torch.float32)
            keep.append(ws)
            arr[i].workspace = ws.data_ptr()
    check(lib.omgsr_igemm_multi(arr, n, _stream()), "omgsr_igemm_multi(conv2d_multi)")
    for out, partial in zip(outs, partials):
        if partial is not None:
            out._omgsr_gn = (partial, gn_groups, out.data_ptr(), out._version)
    return outs


def linear(x: torch.Tensor, pw: PackedWeight, *, act: int = ACT_NONE, residual: Optional[torch.Tensor] = None,
           gate: Optional[torch.Tensor] = None, out_dtype: int = OUT_STREAM, alpha: float = 1.0, gn_groups: int = 0,
           out_split: int = 1) -> torch.Tensor:
    """x [..., K] -> [..., Cout]. gn_groups > 0 (x [B, ..., K]): the result feeds a GroupNorm over each x[b]; the GEMM
    then runs as B images of prod(...) rows so its epilogue can leave the per-image statistics (see conv2d)."""
    lead = x.shape[:-1]
    M = 1
    for d in lead:
        M *= d
    B = lead[0] if (gn_groups > 0 and len(lead) >= 2) else 1
    x2 = x.reshape(B, 1, M // B, x.shape[-1])
    r2 = None if residual is None else residual.reshape(B, 1, M // B, pw.cout)
    y = conv2d(x2, pw, stride=1, pad=0, act=act, residual=r2, gate=gate, out_dtype=out_dtype, alpha=alpha, gn_groups=gn_groups,
               out_split=out_split, sample_rows=(M // lead[0]) if len(lead) >= 2 else M)
    out = y.reshape(*lead, y.shape[-1])
    carry_gn(y, out)
    return out


def carry_gn(src: torch.Tensor, view: torch.Tensor) -> torch.Tensor:
    """Views are new Python objects: hand the fused-statistics handle of `src` on to a view of the same storage."""
    h = getattr(src, "_omgsr_gn", None)
    if h is not None and view.data_ptr() == src.data_ptr() and view.shape[0] == src.shape[0]:
        view._omgsr_gn = h
    return view


def linear_into(x: torch.Tensor, pw: PackedWeight, out: torch.Tensor, row0: int, col0: int, *, act: int = ACT_NONE,
                residual: Optional[torch.Tensor] = None, gate: Optional[torch.Tensor] = None, out_split: int = 1,
                lo_col0: Optional[int] = None, sample_rows: int = 0) -> None:
    """out[row0:row0+M, col0:col0+Cout] = epilogue(x @ W^T): writes a projection straight into a slice of a
    larger 2-D operand buffer `out` [rows, ld] (joint text+image sequences, [attn | mlp] concat). out_split 2: the low
    halves of the two-term split go to columns lo_col0 ... lo_col0+Cout of the same rows (default col0 + Cout)."""
    if x.dtype == torch.float32:
        x = to_operand(x, pw.split)
    _req(x, act_dtype(), "x"); _req(out, act_dtype(), "out")
    # out [rows, ld], or [B, rows, ld] with x [B, M, K]: image b's M rows land at rows row0 .. row0+M of out[b] (grid.z = B)
    Bz = out.shape[0] if out.dim() == 3 else 1
    M, K = x.numel() // x.shape[-1] // Bz, x.shape[-1]
    ld, nrows = out.shape[-1], out.shape[-2]
    if K != pw.row_channels or out.dim() not in (2, 3) or row0 + M > nrows or col0 + pw.cout > ld or (col0 & 7):
        raise ValueError("linear_into: slice does not fit")
    if residual is not None and (Bz != 1 or residual.numel() != M * pw.cout):
        raise ValueError("linear_into: residual must be a dense [M, Cout] (unbatched call)")
    a = IgemmArgs()
    a.in_, a.weight, a.bias, a.gate = x.data_ptr(), pw.w.data_ptr(), _ptr(pw.bias), _ptr(gate)
    a.out = out.data_ptr() + 2 * (row0 * ld + col0)
    a.out_dtype = OUT_BF16
    if out_split == 2:
        a.out_lo_off = (lo_col0 - col0) if lo_col0 is not None else pw.cout
        if a.out_lo_off < pw.cout or col0 + a.out_lo_off + pw.cout > ld:
            raise ValueError("linear_into: the low halves do not fit the row")
    if residual is not None:
        a.residual, a.res_el = residual.data_ptr(), _el(residual, "residual")
    a.N, a.H, a.W = 1, 1, M
    _fill_k(a, pw)
    a.Cout, a.Cout_pad, a.K_pad = pw.cout, pw.cout_pad, pw.k_pad
    a.R, a.S, a.stride, a.pad_top, a.pad_left, a.upsample = 1, 1, 1, 0, 0, 0
    a.Ho, a.Wo = 1, M
    a.act, a.out_layout = act, LAYOUT_NHWC
    a.out_ld = ld
    a.batch, a.alpha = Bz, 1.0
    a.in_bstride, a.w_bstride, a.out_bstride = M * K, 0, nrows * ld
    a.sample_rows = sample_rows or M              # rows of ONE image when the caller flattened a batch into M (batch-invariant dispatch)
    _igemm(a, x.device, "omgsr_igemm(linear_into)", z_batched_api=True)


def linear_rows(x_buf: torch.Tensor, row0: int, rows: int, pw: PackedWeight, *, act: int = ACT_NONE, residual: Optional[torch.Tensor] = None,
                gate: Optional[torch.Tensor] = None, out_dtype: int = OUT_STREAM) -> torch.Tensor:
    """out[b] = epilogue(x_buf[b, row0:row0+rows] @ W^T) for an operand buffer x_buf [B, L, K]: a projection of a row range of
    every image's joint sequence (to_out / to_add_out after joint attention) without gathering the rows first. Returns a dense
    [B, rows, Cout] stream tensor; residual: dense [B, rows, Cout]."""
    _req(x_buf, act_dtype(), "x_buf")
    B, L, K = x_buf.shape
    if K != pw.row_channels or row0 < 0 or row0 + rows > L:
        raise ValueError("linear_rows: row range / channels do not fit")
    out = _out_tensor((B, rows), pw.cout, out_dtype, 1, x_buf.device)
    if residual is not None and tuple(residual.shape) != (B, rows, pw.cout):
        raise ValueError("linear_rows: residual must be [B, rows, Cout]")
    a = IgemmArgs()
    a.in_, a.weight, a.bias, a.gate = x_buf.data_ptr() + 2 * row0 * K, pw.w.data_ptr(), _ptr(pw.bias), _ptr(gate)
    _fill_out(a, out, 1, residual, pw.cout)
    a.N, a.H, a.W = 1, 1, rows
    _fill_k(a, pw)
    a.Cout, a.Cout_pad, a.K_pad = pw.cout, pw.cout_pad, pw.k_pad
    a.R, a.S, a.stride, a.pad_top, a.pad_left, a.upsample = 1, 1, 1, 0, 0, 0
    a.Ho, a.Wo = 1, rows
    a.act, a.out_layout = act, LAYOUT_NHWC
    a.batch, a.alpha = B, 1.0
    a.in_bstride, a.w_bstride, a.out_bstride = L * K, 0, rows * pw.cout
    a.sample_rows = rows
    check(_lib.load().omgsr_igemm(C.byref(a), _stream()), "omgsr_igemm(linear_rows)")
    return out


def linear_t_into(x: torch.Tensor, pw: PackedWeight, out_t: torch.Tensor, key0: int) -> None:
    """out_t[n, key0 + l] = (x W^T + b)[l, n] for x [L, K]: transposed projection into a slice of a joint
    V^T buffer [Cout, ld]."""
    if x.dtype == torch.float32:
        x = to_operand(x, pw.split)
    _req(x, act_dtype(), "x"); _req(out_t, act_dtype(), "out_t")
    # out_t [Cout, ld], or [B, Cout, ld] with x [B, L, K] (grid.z = B)
    Bz = out_t.shape[0] if out_t.dim() == 3 else 1
    L, K = x.numel() // x.shape[-1] // Bz, x.shape[-1]
    if out_t.dim() not in (2, 3) or out_t.shape[-2] != pw.cout or key0 + L > out_t.shape[-1] or K != pw.row_channels:
        raise ValueError("linear_t_into: slice does not fit")
    a = IgemmArgs()
    a.in_, a.weight, a.bias, a.out = x.data_ptr(), pw.w.data_ptr(), _ptr(pw.bias), out_t.data_ptr() + 2 * key0
    a.N, a.H, a.W = 1, 1, L
    _fill_k(a, pw)
    a.Cout, a.Cout_pad, a.K_pad = pw.cout, pw.cout_pad, pw.k_pad
    a.R, a.S, a.stride, a.pad_top, a.pad_left, a.upsample = 1, 1, 1, 0, 0, 0
    a.Ho, a.Wo = 1, L
    a.act, a.out_dtype, a.out_layout = ACT_NONE, OUT_BF16, LAYOUT_T
    a.t_rows, a.t_ld = L, out_t.shape[-1]
    a.batch, a.alpha = Bz, 1.0
    a.in_bstride, a.w_bstride, a.out_bstride = L * K, 0, pw.cout * out_t.shape[-1]
    a.sample_rows = L
    check(_lib.load().omgsr_igemm(C.byref(a), _stream()), "omgsr_igemm(linear_t_into)")


def linear_t(x: torch.Tensor, pw: PackedWeight, rows_per_batch: int, ld: Optional[int] = None) -> torch.Tensor:
    """Transposed-output projection: x [B, L, K] -> out [B, Cout, ld] with out[b, n, l] = (x W^T + bias)[b, l, n].
    This is how V reaches omgsr_attention (key index contiguous). Columns l >= L are zero."""
    if x.dtype == torch.float32:
        x = to_operand(x, pw.split)
    _req(x, act_dtype(), "x")
    B, L, K = x.shape
    if L != rows_per_batch or K != pw.row_channels:
        raise ValueError("linear_t: shape mismatch")
    ld = ld or _round_up(L, 8)
    out = torch.zeros((B, pw.cout, ld), device=x.device, dtype=act_dtype()) if ld != L else \
        torch.empty((B, pw.cout, ld), device=x.device, dtype=act_dtype())
    a = IgemmArgs()
    a.in_, a.weight, a.bias, a.gate, a.residual, a.out = x.data_ptr(), pw.w.data_ptr(), _ptr(pw.bias), None, None, out.data_ptr()
    a.N, a.H, a.W = 1, 1, B * L
    _fill_k(a, pw)
    a.Cout, a.Cout_pad, a.K_pad = pw.cout, pw.cout_pad, pw.k_pad
    a.R, a.S, a.stride, a.pad_top, a.pad_left, a.upsample = 1, 1, 1, 0, 0, 0
    a.Ho, a.Wo = 1, B * L
    a.act, a.out_dtype, a.out_layout = ACT_NONE, OUT_BF16, LAYOUT_T
    a.t_rows, a.t_ld = L, ld
    a.batch, a.in_bstride, a.w_bstride, a.out_bstride = 1, 0, 0, 0
    a.alpha = 1.0
    a.sample_rows = L
    check(_lib.load().omgsr_igemm(C.byref(a), _stream()), "omgsr_igemm(linear_t)")
    return out


def transpose_split(x: torch.Tensor, ld: Optional[int] = None) -> torch.Tensor:
    """x fp32 [B, L, C] (a projection's stream output) -> [B, 2C, ld] in the operand type: rows [0, C) = round(x)^T, rows [C, 2C) = the low halves.
    The attention's V^T operand as a two-term split (range-fallback tier, ops.attn_split). Columns >= L are zero."""
    _req(x, torch.float32, "x")
    B, L, Cc = x.shape
    ld = ld or _round_up(L, 8)
    out = (torch.zeros if ld != L else torch.empty)((B, 2 * Cc, ld), device=x.device, dtype=act_dtype())
    check(_lib.load().omgsr_transpose_split(x.data_ptr(), out.data_ptr(), B, L, Cc, ld, _stream()), "omgsr_transpose_split")
    return out


def bmm_nt(a_mat: torch.Tensor, b_mat: torch.Tensor, *, alpha: float = 1.0, out_dtype: int = OUT_BF16, out_split: int = 1,
           both_split: bool = False) -> torch.Tensor:
    """out[b] = alpha * a[b] @ b[b]^T ; a [B, M, K], b [B, Npad, K] operands with Npad % 128 == 0, K % 32 == 0.
    Returns [B, M, Npad] (OUT_F32: fp32; OUT_BF16: a 16-bit operand, [B, M, 2*Npad] as a two-term split when out_split 2).
    (d=512 VAE attention scores / PV product.)
    both_split: BOTH factors carry two-term splits - a = [a_hi | a_lo] ([B, M, 2C], what a GEMM epilogue writes with out_split 2) and
    b = [b_hi | b_hi | b_lo] ([B, Npad, 3C], split_rows_hhl): the contraction runs the three segments a_hi b_hi + a_lo b_hi + a_hi b_lo,
    the third one wrapping back to a_hi (omgsr_igemm_args.in_ld) exactly like a Linear with operand and weight splits."""
    _req(a_mat, act_dtype(), "a")
    _req(b_mat, act_dtype(), "b")
    B, M, K = a_mat.shape
    Bb, Np, Kb = b_mat.shape
    if both_split:
        if Bb != B or K % 64 or Kb * 2 != K * 3:
            raise ValueError(f"bmm_nt(both_split): incompatible shapes {tuple(a_mat.shape)} x {tuple(b_mat.shape)}")
    elif Bb != B or Kb != K or K % 32:
        raise ValueError(f"bmm_nt: incompatible shapes {tuple(a_mat.shape)} x {tuple(b_mat.shape)}")
    if Np % 128:    # reduced test configs only (the real VAE has 512 channels / 128-padded key counts)
        bp = torch.zeros((B, _round_up(Np, 128), Kb), device=b_mat.device, dtype=act_dtype())
        bp[:, :Np] = b_mat
        full = bmm_nt(a_mat, bp, alpha=alpha, out_dtype=out_dtype, out_split=out_split, both_split=both_split)
        Npp = bp.shape[1]
        if out_split == 2 and out_dtype == OUT_BF16:
            return torch.cat([full[:, :, :Np], full[:, :, Npp:Npp + Np]], dim=-1).contiguous()
        return full[:, :, :Np].contiguous()
    split = out_split if out_dtype == OUT_BF16 else 1
    out = torch.empty((B, M, Np * split), device=a_mat.device, dtype=act_dtype() if out_dtype == OUT_BF16 else torch.float32)
    a = IgemmArgs()
    a.in_, a.weight, a.bias, a.gate, a.residual, a.out = a_mat.data_ptr(), b_mat.data_ptr(), None, None, None, out.data_ptr()
    a.N, a.H, a.W, a.Cin = 1, 1, M, Kb
    a.Cout, a.Cout_pad, a.K_pad = Np, Np, Kb
    if both_split:
        a.in_ld, a.in_split, a.w_split = K, 1, 1
        a.overflow_flag = _ovf(a_mat.device)
    a.R, a.S, a.stride, a.pad_top, a.pad_left, a.upsample = 1, 1, 1, 0, 0, 0
    a.Ho, a.Wo = 1, M
    a.act, a.out_dtype, a.out_layout = ACT_NONE, out_dtype, LAYOUT_NHWC
    a.t_rows, a.t_ld = 0, 0
    a.out_lo_off = Np if split == 2 else 0
    a.out_ld = Np * split if split == 2 else 0
    a.batch, a.in_bstride, a.w_bstride, a.out_bstride = B, M * K, Np * Kb, M * Np * split
    a.alpha = alpha
    check(_lib.load().omgsr_igemm(C.byref(a), _stream()), "omgsr_igemm(bmm_nt)")
    return out


def split_rows_hhl(x: torch.Tensor, pad_rows: int = 0) -> torch.Tensor:
    """[B, L, 2C] two-term split rows [hi | lo] -> [B, max(L, pad_rows), 3C] rows [hi | hi | lo] (zero rows past L): the second factor of
    bmm_nt(both_split=True). Plain device copies (a few MB: the keys of the VAE's one-head attention)."""
    B, L, C2 = x.shape
    Cc = C2 // 2
    out = (torch.zeros if pad_rows > L else torch.empty)((B, max(L, pad_rows), 3 * Cc), device=x.device, dtype=x.dtype)
    out[:, :L, :Cc] = x[:, :, :Cc]
    out[:, :L, Cc:2 * Cc] = x[:, :, :Cc]
    out[:, :L, 2 * Cc:] = x[:, :, Cc:]
    return out


# --------------------------------------------------------------------------------------------
# K4: GroupNorm

def _fused_gn(x: torch.Tensor, groups: int, N: int):
    fused = getattr(x, "_omgsr_gn", None)
    if fused is not None and fused[1] == groups and fused[2] == x.data_ptr() and fused[3] == x._version and fused[0].shape[0] == N:
        return fused[0]
    return None


def group_norm_stats(x: torch.Tensor, groups: int, eps: float):
    """x [N, ..., C] stream tensor -> (mean [N,G], rstd [N,G], var [N,G]) f32 (biased variance)."""
    xel = _el(x, "x")
    N, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (N * Cc)
    lib = _lib.load()
    mean = torch.empty((N, groups), device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    var = torch.empty_like(mean)
    fused = _fused_gn(x, groups, N)
    if fused is not None:
        # the producing conv already reduced this tensor (omgsr_igemm gn_partial): fold its partials only
        check(lib.omgsr_groupnorm_finalize(fused.data_ptr(), mean.data_ptr(), rstd.data_ptr(), var.data_ptr(), N,
                                           fused.shape[1], groups, fused.shape[2], float(HW) * (Cc // groups), eps, _stream()),
              "omgsr_groupnorm_finalize")
        return mean, rstd, var
    nchunk = lib.omgsr_groupnorm_nchunk(HW)
    partial = torch.empty((N, nchunk, groups, 2), device=x.device, dtype=torch.float32)
    check(lib.omgsr_groupnorm_stats(x.data_ptr(), partial.data_ptr(), mean.data_ptr(), rstd.data_ptr(), var.data_ptr(),
                                    N, HW, Cc, groups, eps, xel, _stream()), "omgsr_groupnorm_stats")
    return mean, rstd, var


def group_norm_partial(x: torch.Tensor, groups: int) -> torch.Tensor:
    """(sum, sum of squares) partials [N, nslot, G, 2] of x: the producing conv's fused ones when it left them
    (conv2d gn_groups), else one read pass over x."""
    xel = _el(x, "x")
    N, Cc = x.shape[0], x.shape[-1]
    fused = _fused_gn(x, groups, N)
    if fused is not None:
        return fused
    HW = x.numel() // (N * Cc)
    lib = _lib.load()
    partial = torch.empty((N, lib.omgsr_groupnorm_nchunk(HW), groups, 2), device=x.device, dtype=torch.float32)
    check(lib.omgsr_groupnorm_partial(x.data_ptr(), partial.data_ptr(), N, HW, Cc, groups, xel, _stream()), "omgsr_groupnorm_partial")
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


def _operand_like(x: torch.Tensor, split: int) -> torch.Tensor:
    if split not in (1, 2, 3, 4):
        raise ValueError("split must be 1, 2, 3 (MX) or 4 (MX6)")
    return torch.empty((*x.shape[:-1], x.shape[-1] * min(split, 2)), device=x.device, dtype=_ACT)


def _cast_twin(x: torch.Tensor, xel: int, also_cast: int):
    """The optional second output of the GroupNorm apply kernels: x itself as an operand (a fp32 stream tensor only; a 16-bit
    stream tensor IS its own operand)."""
    if not also_cast:
        return None
    if xel != EL_F32:
        if also_cast != 1:
            raise ValueError("a 16-bit tensor cannot be split")
        return x
    return _operand_like(x, also_cast)


def group_norm_apply_shared(x: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor, gamma, beta, groups: int,
                            act: int = ACT_NONE, split: int = 1, also_cast: int = 0):
    """x [T*N, ..., C] tile-major; mean / rstd [N, G]: row r is normalised with the statistics of image r % N.
    also_cast 1 | 2 | 3: returns (y, x as a plain / split / mixed-precision operand) - see group_norm_apply."""
    xel = _el(x, "x")
    rows, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (rows * Cc)
    y = _operand_like(x, split)
    y2 = _cast_twin(x, xel, also_cast)
    fused = y2 is not None and y2 is not x
    check(_lib.load().omgsr_groupnorm_apply_shared(x.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _ptr(gamma),
                                                   _ptr(beta), rows, HW, Cc, groups, act, mean.shape[0], xel,
                                                   _el_of_split(split), y2.data_ptr() if fused else None,
                                                   _el_of_split(also_cast) if also_cast else EL_16, _ovf(x.device) if fused else None, _stream()),
          "omgsr_groupnorm_apply_shared")
    return (y, y2) if also_cast else y


def group_norm_apply(x: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor, gamma: Optional[torch.Tensor],
                     beta: Optional[torch.Tensor], groups: int, act: int = ACT_NONE, inplace: bool = False, split: int = 1,
                     also_cast: int = 0):
    """Stream tensor -> normalised (+SiLU) MFMA operand [..., split*C]. also_cast 1 | 2 | 3: additionally returns x itself as a
    plain / two-term split / mixed-precision operand (the input of a ResnetBlock's 1x1 shortcut conv), written by the same pass."""
    xel = _el(x, "x")
    N, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (N * Cc)
    y = x if (inplace and xel == EL_16 and split == 1 and not also_cast) else _operand_like(x, split)
    y2 = _cast_twin(x, xel, also_cast)
    fused = y2 is not None and y2 is not x
    check(_lib.load().omgsr_groupnorm_apply(x.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _ptr(gamma),
                                            _ptr(beta), N, HW, Cc, groups, act, xel, _el_of_split(split),
                                            y2.data_ptr() if fused else None, _el_of_split(also_cast) if also_cast else EL_16,
                                            _ovf(x.device) if fused else None, _stream()),
          "omgsr_groupnorm_apply")
    return (y, y2) if also_cast else y


def group_norm(x: torch.Tensor, gamma, beta, groups: int, eps: float, act: int = ACT_NONE, split: int = 1, also_cast: int = 0):
    mean, rstd, _ = group_norm_stats(x, groups, eps)
    return group_norm_apply(x, mean, rstd, gamma, beta, groups, act, split=split, also_cast=also_cast)


# --------------------------------------------------------------------------------------------
# K9/K10: LayerNorm (affine or AdaLN-modulated)

def layer_norm(x: torch.Tensor, a: Optional[torch.Tensor], b: Optional[torch.Tensor], eps: float, split: int = 1) -> torch.Tensor:
    """Stream tensor rows -> normalised MFMA operand [..., split*C]."""
    xel = _el(x, "x")
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    y = _operand_like(x, split)
    check(_lib.load().omgsr_layernorm(x.data_ptr(), y.data_ptr(), _ptr(a), _ptr(b), rows, Cc, eps, xel, _el_of_split(split), _stream()),
          "omgsr_layernorm")
    return y


# --------------------------------------------------------------------------------------------
# K7/K8: attention

def attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, heads: int, head_dim: int, scale: float,
              *, q_col: int = 0, k_col: int = 0, Lk: Optional[int] = None, out: Optional[torch.Tensor] = None,
              o_col: int = 0, out_split: int = 1, o_lo_col: Optional[int] = None, q_lo_col: Optional[int] = None,
              k_lo_col: Optional[int] = None, p_split: bool = True) -> torch.Tensor:
    """q [B, Lq, *] (heads at columns q_col + h*D), k [Bk, Lk, *], vt [Bk, heads*D, ld] -> o [B, Lq, heads*D]
    q_lo_col / k_lo_col (both or neither, head_dim 64): q and k are two-term splits whose low halves start at those columns of the same rows
    (a projection written with out_split = 2); the scores then run three MFMA passes and, with p_split, the probabilities two (attn_split()).
    (an operand for the output projection; out_split 2: [B, Lq, 2*heads*D] as the two-term split; 3: the same bytes per row in the
    mixed-precision form OMGSR_EL_MX, for an output projection that runs as an MX GEMM).
    Bk == 1 broadcasts one K/V over the batch (constant cross-attention context)."""
    _req(q, act_dtype(), "q"); _req(k, act_dtype(), "k"); _req(vt, act_dtype(), "vt")
    B, Lq = q.shape[0], q.shape[1]
    Bk = k.shape[0]
    Lk = Lk if Lk is not None else k.shape[1]
    inner = heads * head_dim
    if out is None:
        out = torch.empty((B, Lq, inner * min(out_split, 2)), device=q.device, dtype=act_dtype())      # split 3 (MX): 4 bytes per channel
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
    a.o_lo_off = 0 if out_split != 2 else ((o_lo_col - o_col) if o_lo_col is not None else inner)
    a.o_mx = int(out_split == 3)
    if (q_lo_col is None) != (k_lo_col is None):
        raise ValueError("attention: q_lo_col and k_lo_col come together")
    a.q_lo_off = 0 if q_lo_col is None else q_lo_col - q_col
    a.k_lo_off = 0 if k_lo_col is None else k_lo_col - k_col
    a.p_split = int(q_lo_col is not None and p_split)
    if vt.shape[1] == 2 * inner:                      # [Bk, 2 * inner, ld]: V^T as the transposed two-term split (transpose_split)
        if q_lo_col is None:
            raise ValueError("attention: a split V^T comes with split q / k")
        a.vt_lo_off = inner * vt.shape[-1]
    elif vt.shape[1] != inner:
        raise ValueError(f"attention: vt has {vt.shape[1]} rows, expected {inner} (or {2 * inner} as a two-term split)")
    if out_split == 3 and o_col != 0:
        raise ValueError("attention: an MX output owns its whole row (o_col must be 0)")
    check(_lib.load().omgsr_attention(C.byref(a), _stream()), "omgsr_attention")
    return out


def softmax_rows(s: torch.Tensor, valid: Optional[int] = None, split: bool = False) -> torch.Tensor:
    """softmax over the first `valid` columns of each row (the rest come out 0). split: [..., 2 L] rows [p_hi | p_lo] (the first factor of
    bmm_nt(both_split=True): the range-fallback tier's PV product)."""
    _req(s, torch.float32, "s")
    L = s.shape[-1]
    rows = s.numel() // L
    p = torch.empty(s.shape[:-1] + ((2 * L) if split else L,), device=s.device, dtype=act_dtype())
    fn = _lib.load().omgsr_softmax_rows_split if split else _lib.load().omgsr_softmax_rows
    check(fn(s.data_ptr(), p.data_ptr(), rows, L, valid or L, _stream()), "omgsr_softmax_rows")
    return p


def rmsnorm_rope_(x: torch.Tensor, w: torch.Tensor, cos: Optional[torch.Tensor], sin: Optional[torch.Tensor],
                  heads: int, head_dim: int, col0: int = 0, pos0: int = 0, eps: float = 1e-6,
                  w_after: Optional[torch.Tensor] = None, split_at: int = 0) -> torch.Tensor:
    """In place on x [B, L, ld]: per head RMSNorm(head_dim) * w[h] then RoPE with cos/sin [>= pos0+L, D] (f32).
    w is [heads, D]: one call can cover the q heads and the k heads of a fused [q|k] buffer. w_after / split_at: rows at
    sequence positions >= split_at use w_after instead (joint [text ; image] sequences)."""
    _req(x, act_dtype(), "x"); _req(w, torch.float32, "w")
    B, L, ld = x.shape
    if tuple(w.shape) != (heads, head_dim) or (w_after is not None and tuple(w_after.shape) != (heads, head_dim)):
        raise ValueError(f"rmsnorm_rope_: w must be [{heads}, {head_dim}], got {tuple(w.shape)}")
    if cos is not None and (cos.shape[0] < pos0 + L or sin.shape[0] < pos0 + L or cos.shape[-1] != head_dim):
        raise ValueError(f"rmsnorm_rope_: rope tables {tuple(cos.shape)} do not cover positions {pos0}..{pos0 + L - 1} x {head_dim}")
    check(_lib.load().omgsr_rmsnorm_rope(x.data_ptr(), w.data_ptr(), _ptr(w_after), split_at, _ptr(cos), _ptr(sin), B, L, heads, head_dim, ld,
                                         col0, pos0, eps, _stream()), "omgsr_rmsnorm_rope")
    return x


# --------------------------------------------------------------------------------------------
# K14: layout + latent algebra

def nchw_to_nhwc(x: torch.Tensor, cpad: Optional[int] = None, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """NCHW (f32 or the compute type) -> NHWC stream tensor (channels zero-padded to cpad); dtype overrides stream_dtype()."""
    if x.dtype not in (torch.float32, act_dtype()):
        x = x.to(act_dtype() if not _PRECISE else torch.float32)          # the other 16-bit type / f64: one conversion at the boundary
    _req(x, x.dtype, "x")
    N, Cc, H, W = x.shape
    cpad = cpad or _round_up(Cc, 8)
    dtype = dtype or stream_dtype()
    y = torch.empty((N, H, W, cpad), device=x.device, dtype=dtype)
    check(_lib.load().omgsr_nchw_to_nhwc(x.data_ptr(), y.data_ptr(), N, Cc, H, W, cpad, int(x.dtype == torch.float32),
                                         EL_F32 if dtype == torch.float32 else EL_16, _stream()), "omgsr_nchw_to_nhwc")
    return y


def nhwc_to_nchw(x: torch.Tensor, channels: Optional[int] = None, dtype=None,
                 clamp: Optional[tuple[float, float]] = None) -> torch.Tensor:
    xel = _el(x, "x")
    dtype = dtype or (torch.float32 if xel == EL_F32 else act_dtype())
    if dtype not in (torch.float32, act_dtype()):
        raise TypeError(f"nhwc_to_nchw: output dtype must be float32 or {act_dtype()}, got {dtype}")
    N, H, W, ld = x.shape
    Cc = channels or ld
    y = torch.empty((N, Cc, H, W), device=x.device, dtype=dtype)
    lo, hi = clamp if clamp else (0.0, 0.0)
    check(_lib.load().omgsr_nhwc_to_nchw(x.data_ptr(), y.data_ptr(), N, Cc, H, W, ld, int(dtype == torch.float32),
                                         int(clamp is not None), lo, hi, xel, _stream()), "omgsr_nhwc_to_nchw")
    return y


def concat_channels(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """torch.cat([a, b], dim=channel) for two NHWC stream tensors of one element kind."""
    el = _el(a, "a")
    if _el(b, "b") != el:
        raise TypeError("concat_channels: element kinds differ")
    Ca, Cb = a.shape[-1], b.shape[-1]
    rows = a.numel() // Ca
    out = torch.empty((*a.shape[:-1], Ca + Cb), device=a.device, dtype=a.dtype)
    lib = _lib.load()
    check(lib.omgsr_copy_channels(a.data_ptr(), out.data_ptr(), rows, Ca, Ca, Ca + Cb, 0, el, _stream()), "omgsr_copy_channels")
    check(lib.omgsr_copy_channels(b.data_ptr(), out.data_ptr(), rows, Cb, Cb, Ca + Cb, Ca, el, _stream()), "omgsr_copy_channels")
    return out


def vae_sample(moments: torch.Tensor, eps: torch.Tensor, latent_channels: int, shift: float, scale: float,
               ld_out: Optional[int] = None) -> torch.Tensor:
    """moments [N,h,w,2C] stream, eps [N,h,w,C] f32 -> z [N,h,w,ld_out] stream of the same kind (zero padded channels)."""
    el = _el(moments, "moments"); _req(eps, torch.float32, "eps")
    N, h, w, _ = moments.shape
    ld_out = ld_out or _round_up(latent_channels, 8)
    z = torch.empty((N, h, w, ld_out), device=moments.device, dtype=moments.dtype)
    check(_lib.load().omgsr_vae_sample(moments.data_ptr(), eps.data_ptr(), z.data_ptr(), N * h * w, latent_channels, ld_out,
                                       shift, scale, el, _stream()), "omgsr_vae_sample")
    return z


def axpby(x: torch.Tensor, y: Optional[torch.Tensor], a: float, b: float, c: float = 0.0, d: float = 1.0,
          bf16_steps: bool = False) -> torch.Tensor:
    el = _el(x, "x")
    if y is not None and _el(y, "y") != el:
        raise TypeError("axpby: element kinds differ")
    out = torch.empty_like(x)
    check(_lib.load().omgsr_axpby(x.data_ptr(), _ptr(y), out.data_ptr(), x.numel(), a, b, c, d, int(bf16_steps), el, _stream()),
          "omgsr_axpby")
    return out


def linear_f32(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, silu_in: bool = False) -> torch.Tensor:
    """fp32 constant folding: y = act(x) @ weight.T + bias with x [..., K], weight [N, K] (any float dtype; widened to fp32 first), in the
    library's own fixed-order kernel (omgsr_linear_f32) - the time / guidance / pooled-text embeddings and the modulation vectors a model
    derives once per (t*, guidance, prompt), so no vendor BLAS runs in the product process. Device tensors only: weights still on the host
    (CPU-side tooling and the `not gpu` tests, where no kernel of this library can run either) take torch's fp32 math instead."""
    w32 = weight.detach().float().contiguous()
    b32 = None if bias is None else bias.detach().float().contiguous()
    x32 = x.detach().float().contiguous()
    if not w32.is_cuda:
        xin = torch.nn.functional.silu(x32) if silu_in else x32
        return torch.nn.functional.linear(xin, w32, b32)
    N, K = w32.shape
    if x32.shape[-1] != K:
        raise ValueError(f"linear_f32: x[..., {x32.shape[-1]}] against weight [{N}, {K}]")
    rows = x32.numel() // K
    y = torch.empty(x32.shape[:-1] + (N,), dtype=torch.float32, device=w32.device)
    check(_lib.load().omgsr_linear_f32(x32.data_ptr(), w32.data_ptr(), _ptr(b32), y.data_ptr(), rows, K, N, int(silu_in), _stream()),
          "omgsr_linear_f32")
    return y


def tile_accumulate(tile: Optional[torch.Tensor], w: torch.Tensor, acc: torch.Tensor, y0: int, x0: int,
                    channels: Optional[int] = None) -> None:
    """acc [N,H,W,C] f32 += tile[..., :C] * w[th,tw]; tile None accumulates the weights alone (acc [1,H,W,1])."""
    _req(w, torch.float32, "w"); _req(acc, torch.float32, "acc")
    N, H, W, Cc = acc.shape
    th, tw = w.shape
    tile_ld, el = 0, EL_16
    if tile is not None:
        el = _el(tile, "tile")
        tile_ld = tile.shape[-1]
    check(_lib.load().omgsr_tile_accumulate(_ptr(tile), w.data_ptr(), acc.data_ptr(), N, Cc, th, tw, tile_ld, H, W, y0, x0,
                                            el, _stream()), "omgsr_tile_accumulate")


def tile_normalise(acc: torch.Tensor, wsum: torch.Tensor, ld: Optional[int] = None, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    N, H, W, Cc = acc.shape
    ld = ld or _round_up(Cc, 8)
    dtype = dtype or stream_dtype()
    out = torch.empty((N, H, W, ld), device=acc.device, dtype=dtype)
    check(_lib.load().omgsr_tile_normalise(acc.data_ptr(), wsum.data_ptr(), out.data_ptr(), N, H * W, Cc, ld,
                                           EL_F32 if dtype == torch.float32 else EL_16, _stream()), "omgsr_tile_normalise")
    return out


def crop_nhwc(x: torch.Tensor, y0: int, x0: int, th: int, tw: int) -> torch.Tensor:
    el = _el(x, "x")
    N, H, W, Cc = x.shape
    out = torch.empty((N, th, tw, Cc), device=x.device, dtype=x.dtype)
    check(_lib.load().omgsr_crop_nhwc(x.data_ptr(), out.data_ptr(), N, H, W, Cc, y0, x0, th, tw, el, _stream()), "omgsr_crop_nhwc")
    return out


def resize_nearest_exact(x: torch.Tensor, scale_factor: float) -> torch.Tensor:
    """F.interpolate(x_nchw, scale_factor=s, mode="nearest-exact") on an NHWC tensor [N, H, W, C] (any element kind): output
    [N, floor(H s), floor(W s), C], source index min(floorf((dst + .5) * (float)(1 / s)), in - 1) - ATen's formula (omgsr_resize_nearest_exact_nhwc)."""
    import math
    el = _el(x, "x")
    N, H, W, Cc = x.shape
    Ho, Wo = int(math.floor(float(H) * scale_factor)), int(math.floor(float(W) * scale_factor))
    out = torch.empty((N, Ho, Wo, Cc), device=x.device, dtype=x.dtype)
    inv = float(torch.tensor(1.0 / scale_factor, dtype=torch.float32))            # (float)(1 / s), as compute_scales_value<float> makes it
    check(_lib.load().omgsr_resize_nearest_exact_nhwc(x.data_ptr(), out.data_ptr(), N, H, W, Cc, Ho, Wo, inv, inv, el, _stream()),
          "omgsr_resize_nearest_exact_nhwc")
    return out


def paste_nhwc(src: torch.Tensor, dst: torch.Tensor, sy0: int, sx0: int, dy0: int, dx0: int, th: int, tw: int) -> None:
    """dst[:, dy0:dy0+th, dx0:dx0+tw, :] = src[:, sy0:sy0+th, sx0:sx0+tw, :] (NHWC, same N, C and element kind)."""
    el = _el(src, "src")
    if _el(dst, "dst") != el:
        raise TypeError("paste_nhwc: element kinds differ")
    N, sH, sW, Cc = src.shape
    if dst.shape[0] != N or dst.shape[3] != Cc:
        raise ValueError("paste_nhwc: batch/channel mismatch")
    check(_lib.load().omgsr_paste_nhwc(src.data_ptr(), dst.data_ptr(), N, Cc, sH, sW, sy0, sx0, dst.shape[1], dst.shape[2],
                                       dy0, dx0, th, tw, el, _stream()), "omgsr_paste_nhwc")


def flux_pack(x: torch.Tensor, channels: int) -> torch.Tensor:
    """NHWC [N,H,W,ld] (first `channels`) -> tokens [N, (H/2)(W/2), 4*channels]."""
    el = _el(x, "x")
    N, H, W, ld = x.shape
    out = torch.empty((N, (H // 2) * (W // 2), 4 * channels), device=x.device, dtype=x.dtype)
    check(_lib.load().omgsr_flux_pack(x.data_ptr(), out.data_ptr(), N, H, W, channels, ld, 0, el, _stream()), "omgsr_flux_pack")
    return out


def flux_unpack(tok: torch.Tensor, H: int, W: int, ld: Optional[int] = None) -> torch.Tensor:
    el = _el(tok, "tok")
    N, _, c4 = tok.shape
    Cc = c4 // 4
    ld = ld or _round_up(Cc, 8)
    out = torch.zeros((N, H, W, ld), device=tok.device, dtype=tok.dtype) if ld != Cc else \
        torch.empty((N, H, W, ld), device=tok.device, dtype=tok.dtype)
    check(_lib.load().omgsr_flux_pack(tok.data_ptr(), out.data_ptr(), N, H, W, Cc, ld, 1, el, _stream()), "omgsr_flux_pack")
    return out
