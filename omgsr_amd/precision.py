"""Precision policy of the accurate tier (`--weight_dtype fp32`, ops.set_compute_dtype(torch.float32)).

The reference's `--weight_dtype fp32` (infer/infer_omgsr_s.py:134-149) runs every layer in fp32. gfx950 has no fast
fp32 / TF32 matrix path (fp32 MFMA runs at 1/16 of the 16-bit rate), so the accurate tier keeps every tensor BETWEEN two
GEMMs in fp32 and feeds the MFMAs fp16 operands with fp32 accumulation. What is left is one fp16 rounding of each GEMM
input (2^-11 relative); tests/emulate_numerics.py measures what those roundings cost at the full SD2.1 shapes against the
fp32 oracle (rel-L2 1.5e-3 in total) and how the total splits over the layers. The layers named here receive their input as
the two-term split x = hi + lo (both fp16, the weights see both halves: twice the MFMA work of that layer, 2^-22
relative error), which brings the pipeline under the north-star tolerance (rel-L2 <= 1e-3, PSNR >= 60 dB vs the fp32 oracle).

A policy is a list of regular expressions over module names; every Conv2d / Linear (omgsr_amd.nn) whose qualified name
matches gets `op_split = 2`. Producers (norm / cast kernels, GEMM and attention epilogues) ask their consumer which form to
write, so a policy needs no other change.
"""
from __future__ import annotations

import contextlib
import re
from typing import Iterable, Optional

import torch
import torch.nn as nn

from .nn import Conv2d, Linear

# Where the error comes from (tests/emulate_numerics.py --budget f16_levels / f16_subs on OMGSR-S 128->512, squared rel-L2 in 1e-8
# units; everything unsplit: 224 = (1.5e-3)^2). An operand rounding hurts in proportion to how much of the SIGNAL passes through
# it: inside a residual branch (ResnetBlock conv1 / conv2, attention and feed-forward linears) it perturbs only the branch, while
# the input of a 1x1 shortcut, of an up / down-sampling conv, of proj_in / proj_out, of conv_in / conv_out carries the whole
# tensor. Shares: latent-sized tensors 35; UNet shortcuts 27, proj_in / proj_out 24, up / down-samplers 9, conv_out 5, the
# 64 x 64 level's resnet convs 13 and linears 10 (32 x 32 level: 1.3 + 3, 16 x 16: less); VAE decoder shortcuts 11, conv_out 4,
# upsampling convs 17.5 (28 % of the decoder's FLOPs: left unsplit), resnet convs 6; encoder downsampling convs 11 (5 % of its
# FLOPs), shortcuts 5, conv_out 3, resnet convs 24 (unsplit). The lists below split the full-signal operands (a few % of the
# FLOPs) and the UNet's 64 x 64 level: emulated 7.6e-4, and ~40 % of the extra MFMA work of splitting whole UNet levels
# (the first policy of this round: every UNet level but 8 x 8 plus the encoder's 512-px level, 8.2e-4 emulated / 8.0e-4 measured).
# Inside the 64 x 64 level the q / k / v projections stay unsplit: the LayerNorm'd operand they read contributes 0.4 of the 224.
_L64 = r"^(down_blocks\.0|up_blocks\.3)\."
UNET_DEFAULT = [r"^conv_in$", r"^conv_out$", r"\.conv_shortcut$", r"\.attentions\.\d+\.proj_(in|out)$", r"samplers\.0\.conv$",
                _L64 + r"resnets\.", _L64 + r"attentions\.\d+\.transformer_blocks\.\d+\.(ff\.|attn[12]\.to_out\.)"]
VAE_DEFAULT = [r"^encoder\.conv_in$", r"^encoder\.conv_out$", r"^quant_conv$", r"^post_quant_conv$", r"^decoder\.conv_in$",
               r"^decoder\.conv_out$", r"\.conv_shortcut$", r"^encoder\..*downsamplers\.0\.conv$"]
FLUX_DEFAULT = [r"^x_embedder$", r"^proj_out$"]
# "Inner" tensors: a ResnetBlock's conv1 output is read by norm2 and nothing else. Keeping it in fp16 instead of fp32 halves its
# write and its read; measured with the emulator on OMGSR-S 128->512 (squared rel-L2, 1e-8 units, on top of the 57.9 of the
# policy above): decoder levels at >= 1/2 of the output resolution +1.5 (83 % of the decoder's inner-tensor bytes), the whole
# decoder +~6, encoder levels at >= 1/2 resolution +11.6, the whole VAE +18. Only the first is taken.
VAE_INNER16 = [r"^decoder\.up_blocks\.[23]\.resnets\.\d+\.conv1$"]
# ---- the shipped lists (round 3) -------------------------------------------------------------------------------------------------
# Round 2's lists (`*_DEFAULT` above, activation side only) were sized on ONE weight draw whose values were bf16-representable. With
# full-mantissa fp32 weights (every real checkpoint after the reference's fp32 LoRA merge) the WEIGHT rounding is a second error
# source of the same size: tests/emulate_numerics.py --fp32-weights --named r2 gives 1.72e-3 (295 units of 1e-8 squared rel-L2: 58
# from the unsplit operands, 233 from the weights), and the measured spread over weight / input draws is a factor 1.6 in rel-L2
# (tests/test_fullsize_parity_gpu.py: 7.0e-4 ... 1.1e-3 for one and the same policy). The budget of every layer group, operand
# side (A) and weight side (W) separately, on OMGSR-S 128->512 (everything else exact; floor 1.7 = attention internals):
#   encoder resnet convs   512 px 14.7 / 13.9   256 px 4.6 / 3.9   128 px 2.6 / 2.6   64 px + mid 3.2 / 3.9   mid attention 0.5 / 0.4
#   decoder resnet convs   mid + 64 px 2.6 / 2.4   128 px 1.3 / 1.0   256 px 1.2 / 1.0   512 px 1.1 / 1.1   mid attention 0.2 / 0.3
#   decoder upsampling convs (3)   5.8 / 5.0 each
#   UNet beyond round 2's list   64 x 64 q / k / v 0.4 / 6.4   32 x 32 resnets 1.4 / 1.0, linears 3.0 / 5.0   16 x 16 resnets 0.3 / 0.2,
#                                linears 0.9 / 1.2   8 x 8 + mid 0.1 / 0.0
# The shipped policy splits BOTH sides of everything whose share per FLOP is not negligible and leaves out the decoder's resnet convs
# above 64 px (6.7 units for 52 % of the decoder's FLOPs), the VAE attention operands (until round 4's 40-draw sweep: see VAE_ACT below), the UNet's 16 x 16 resnets and 8 x 8 level on
# the operand side and the 64 x 64 q / k / v operand (its GEMMs are HBM-bound: a split operand doubles their bytes for 0.4 units):
# 11.1 units emulated = 3.3e-4 on the reference draw; measured over 3 weight x 2 input draws 3.4e-4 ... 4.3e-4
# (tests/test_fullsize_parity_gpu.py).
# A budget from ONE draw does not transfer: also leaving out the decoder's mid / 64 px resnets and the encoder's 128 px level
# (+10 units on the reference draw = 4.6e-4 emulated, 17 ms of the 246 ms S-1024 step saved) measured 4.7e-4 ... 5.9e-4 on five of
# those six draws and 9.2e-4 on the sixth - so they stay split. precision_policy="all" is the 2^-22 floor.
# A weight split costs MFMA time only (no producer changes, no extra activation bytes); an operand split also doubles the bytes of
# the operand it splits.
# Round 4: the 9.2e-4 outlier of round 3's trimmed policy was attributed with the emulator on THAT draw (tests/emulate_numerics.py
# --test-draw 1,0 reproduces the GPU test's weights / input / noise; shipped policy 4.39e-4 emulated, 4.51e-4 measured), squared rel-L2
# in 1e-8 units:
#   shipped 19.3 | + decoder mid / 64 px resnets unsplit 25.5 (+6.2) | + encoder 128 px level unsplit 56.0 (+36.7) | both 64.9 (8.1e-4)
# The encoder's 128 px level costs 5 units on the reference draw and 37 on this one (same weights, other input: 5.9e-4 measured): it
# stays split. The decoder's mid block / first level was then tried alone, and the UNet's 16 x 16 transformer blocks, on four draws
# (weight, input) = (0,0) (1,0) (11,0) (4,1):
#   shipped                                   21.0   19.3   27.5   13.6
#   decoder mid + up_blocks.0 unsplit         24.3   25.5   42.9   11.2     (+3 ... +15: draw-dependent, 6.5e-4 on the worst; -4 ms)
#   UNet 16 x 16 transformer blocks unsplit   22.6   23.4   30.8   11.1     (+1.6 ... +4.1; -6 ms)
#   both                                      25.9   28.0   46.6   13.6     (6.8e-4 on the worst of four draws)
# Taken: the UNet's 16 x 16 transformer blocks on BOTH sides (thirty-odd long-K GEMMs at three K segments, 8.7 ms of the S-1024 step,
# for <= 4 units on every draw tried). NOT taken: the decoder trim - its cost moves by a factor 5 between draws, and two draw-sensitive
# groups together would eat most of the head-room under 8e-4 for 4 ms. OMGSR_POLICY_TRIM=0 restores round 3's lists (A/B runs),
# OMGSR_POLICY_TRIM=2 adds the decoder trim (A/B runs).
import os as _os
_TRIM = int(_os.environ.get("OMGSR_POLICY_TRIM", "1"))
_VAE_SINGLE = (r"decoder\.(mid_block|up_blocks\.[0123])\.resnets\.\d+\.conv[12]$" if _TRIM >= 2 else
               r"decoder\.up_blocks\.[123]\.resnets\.\d+\.conv[12]$")
# Round 4, 40 weight x 2 input draws (profiles/r04_robustness_40x2.log): 78 of 80 at 2.9e-4 ... 6.0e-4, one at 8.1e-4, one at 1.26e-3 - under this
# policy AND under round 3's. The emulator on the 1.26e-3 draw (--test-draw 16,0: 1.51e-3 emulated, 228 units): all of it is the single-term
# OPERAND of the VAE's mid-block attention projections (228 -> 44 with that operand split and nothing else changed; decoder resnets: no change).
# On the reference draw the same operand costs 0.2-0.5 units: a 4096-key softmax over one 512-wide head amplifies the rounding of q and k by
# the magnitude of that draw's logits. Eight tiny GEMMs (+0.5 ms per S-1024 step at three K segments): split. OMGSR_POLICY_VAE_ATTN=0 = before.
# What is left of that draw (44 units, 7.2e-4 measured) is the block's INTERNAL rounding: q and k of the decoder's attention 13.5, of the
# encoder's 2.7, P and v of both 7.4 (--attn-exact dec:qk | enc:qk,dec:qk | enc,dec). The decoder's q / k therefore leave their projections as
# two-term splits and its score GEMM runs three segments (VAE_QK_SPLIT; OMGSR_POLICY_VAE_QK=0 none, 2 = the encoder's too: A/B runs).
_VAE_ATTN_SINGLE = _os.environ.get("OMGSR_POLICY_VAE_ATTN", "1") == "0"
_VAE_QK = int(_os.environ.get("OMGSR_POLICY_VAE_QK", "1"))
VAE_QK_SPLIT = [] if _VAE_QK <= 0 else ([r"^decoder\.mid_block\.attentions\.\d+$"] if _VAE_QK == 1 else [r"mid_block\.attentions\.\d+$"])
VAE_ACT = [r"^(?!" + _VAE_SINGLE + (r"|.*attentions\." if _VAE_ATTN_SINGLE else "") + r")"]
VAE_W = [r"^(?!" + _VAE_SINGLE + r")"]
_L32 = r"^(down_blocks\.1|up_blocks\.2)\."
_L16 = r"^(down_blocks\.2|up_blocks\.1)\."
UNET_ACT = UNET_DEFAULT + [_L32 + r"resnets\.\d+\.conv[12]$", _L32 + r"attentions\.\d+\.transformer_blocks\."] + \
    ([] if _TRIM >= 1 else [_L16 + r"attentions\.\d+\.transformer_blocks\."])
# weight side: everything but the 16 x 16 level's resnet convs (0.2 units) and transformer blocks (round 4) and the 8 x 8 level + mid
# block (0.0): long-K convs / GEMMs on tiny maps (round 3, resnet convs only: 215.8 -> 210.6 ms on one box, six draws 3.4e-4 ... 4.5e-4
# -> 3.6e-4 ... 4.5e-4)
UNET_W = [r"^(?!(down_blocks\.2|up_blocks\.1)\.(resnets\.\d+\.conv[12]$" + (r"|attentions\.\d+\.transformer_blocks\." if _TRIM >= 1 else "") +
          r")|(down_blocks\.3|up_blocks\.0|mid_block)\.(resnets\.\d+\.conv[12]$|attentions\.\d+\.transformer_blocks\.))"]
FLUX_ACT, FLUX_W = FLUX_DEFAULT, FLUX_DEFAULT
# Layers whose both-sides split runs in the MIXED-PRECISION form (op_split 3, OMGSR_EL_MX): the product a_hi w_hi in fp16 MFMAs and
# the two correction terms a_lo w_hi, a_hi w_lo - which only need a few bits of their own - as block-scaled fp8 MFMAs at twice the
# fp16 rate (v_mfma_scale_f32_32x32x64_f8f6f4): 2 instead of 3 segments of MFMA time and operand traffic. 3x3 stride-1 convs with
# Cin % 64 == 0 whose operand a GroupNorm pass writes (ResnetBlock conv1 / conv2); they always take the halo-tile kernel.
# ... and the nearest-2x upsampling convs (phase-decomposed form), whose operand the previous block's conv / linear epilogue writes.
_DEC_UPS = r"^decoder\..*upsamplers\.0\.conv$"
# ... and (round 4) the ResnetBlocks' 1x1 shortcut convs: GEMM-shaped, they run on igemm_gmx_kernel; their operand is the second output of
# norm1's apply pass (also_cast 3)
VAE_MX = [r"^encoder\..*resnets\.\d+\.conv[12]$", r"^decoder\.(mid_block|up_blocks\.0)\.resnets\.\d+\.conv[12]$", _DEC_UPS, r"\.conv_shortcut$"]
UNET_MX = [_L64 + r"resnets\.\d+\.conv[12]$", _L32 + r"resnets\.\d+\.conv[12]$", r"upsamplers\.0\.conv$", r"\.conv_shortcut$"]


# Bumped by every setter below: captured hipGraphs (pipelines/graphed.py) key on it, so a policy edit after a capture is a new graph
_POLICY_EPOCH = 0


def policy_epoch() -> int:
    return _POLICY_EPOCH


def _touch() -> None:
    global _POLICY_EPOCH
    _POLICY_EPOCH += 1


def set_operand_split(model: nn.Module, patterns: Iterable[str], split: int = 2) -> int:
    """Mark the Conv2d / Linear layers whose qualified name matches any pattern; returns how many were marked."""
    if split not in (1, 2):
        raise ValueError("split must be 1 or 2")
    _touch()
    regs = [re.compile(p) for p in patterns]
    n = 0
    for name, m in model.named_modules():
        if isinstance(m, (Conv2d, Linear)):
            hit = any(r.search(name) for r in regs)
            m.op_split = split if hit else 1
            n += int(hit)
    check_policy(model)
    return n


def set_weight_split(model: nn.Module, patterns: Iterable[str], split: int = 2) -> int:
    """Mark the Conv2d / Linear layers whose WEIGHT is carried as the two-term split w_hi + w_lo (one more K segment that wraps
    over the operand's hi half: +1x the layer's MFMA work, no change to any producer); returns how many were marked."""
    if split not in (1, 2):
        raise ValueError("split must be 1 or 2")
    _touch()
    regs = [re.compile(p) for p in patterns]
    n = 0
    for name, m in model.named_modules():
        if isinstance(m, (Conv2d, Linear)):
            hit = any(r.search(name) for r in regs)
            m.w_split = split if hit else 1
            n += int(hit)
    check_policy(model)
    return n


# Linears whose both-sides split runs in the mixed-precision form (round 4, igemm_gmx_kernel): every Linear of the UNet's 64 x 64 and
# 32 x 32 transformer blocks that the lists above split on BOTH sides - q / k / v of the 32 x 32 level (one LayerNorm'd operand), the
# attention output projections, the cross-attention query, the GEGLU projection and the feed-forward output, proj_in / proj_out. The
# producers write the three-part operand (LayerNorm / GroupNorm apply / attention epilogue / GEGLU and plain GEMM epilogues); the
# cross-attention K / V projections read the PROMPT (1024 channels, computed once per prompt tensor) and keep the two-term split.
UNET_MX_LIN = [r"\.attentions\.\d+\.(proj_in|proj_out)$", r"\.transformer_blocks\.\d+\.(attn1\.to_(q|k|v|out\.0)|attn2\.to_(q|out\.0)|ff\.net\.(0\.proj|2))$"]


def set_mx_linear(model: nn.Module, patterns: Iterable[str]) -> int:
    """Move the both-sides split of the matching Linear layers (in_features % 64 == 0) to the mixed-precision form (op_split 3): the
    layer then runs on igemm_gmx_kernel with 2 instead of 3 K segments of MFMA time and operand traffic. Layers that read ONE operand
    (q / k / v of a self-attention) move together or not at all. OMGSR_MX_LINEAR=0 switches it off (A/B runs). Returns how many moved."""
    import os
    if os.environ.get("OMGSR_MX", "1") == "0" or os.environ.get("OMGSR_MX_LINEAR", "1") == "0":
        return 0
    _touch()
    regs = [re.compile(p) for p in patterns]
    ok = lambda m: isinstance(m, Linear) and m.in_features % 64 == 0 and m.op_split == 2 and m.w_split == 2      # noqa: E731
    n = 0
    grouped = set()
    for name, m in model.named_modules():
        qkv = [getattr(m, a, None) for a in ("to_q", "to_k", "to_v")]
        if all(isinstance(g, Linear) for g in qkv) and not getattr(m, "is_cross", False):
            grouped.update(id(g) for g in qkv)
            hit = all(any(r.search(f"{name}.{a}") for r in regs) for a in ("to_q", "to_k", "to_v"))
            if hit and all(ok(g) for g in qkv) and getattr(m, "_shares_input_with", None) is None:
                for g in qkv:
                    g.op_split = 3
                n += 3
    for name, m in model.named_modules():
        if id(m) in grouped or not ok(m) or not any(r.search(name) for r in regs):
            continue
        m.op_split = 3
        n += 1
    check_policy(model)
    return n


def set_mx(model: nn.Module, patterns: Iterable[str]) -> int:
    """Move the both-sides split of the matching 3x3 stride-1 (halo-tile kernel) and 1x1 (MX GEMM kernel) convolutions (Cin % 64 == 0,
    >= 96 output channels) to the mixed-precision form (op_split 3); layers that do not qualify keep what they had. OMGSR_MX=0
    switches the form off (A/B runs).
    Returns how many layers were moved."""
    import os
    mx_env = os.environ.get("OMGSR_MX", "1")
    if mx_env == "0":
        return 0
    _touch()
    regs = [re.compile(p) for p in patterns]
    n = 0
    is_vae = hasattr(model, "decoder") and hasattr(model, "encoder")
    for name, m in model.named_modules():
        if isinstance(m, Conv2d) and any(r.search(name) for r in regs):
            conv3 = m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1)          # halo-tile kernel
            conv1 = m.kernel_size == (1, 1) and m.stride == (1, 1) and m.padding == (0, 0)          # GEMM-shaped: igemm_gmx_kernel (round 4)
            if (conv3 or conv1) and m.in_channels % 64 == 0 and m.out_channels >= 96 \
                    and m.out_channels % 8 == 0 and m.op_split == 2 and m.w_split == 2:
                # fp6 correction segments (OMGSR_EL_MX6, op_split 4; round 5): the 3x3 convs whose operand a GroupNorm apply writes - a
                # ResnetBlock2D's conv1 / conv2 - and the VAE's up-sampling convs, whose operand the previous ResnetBlock's conv2 writes from the
                # halo-tile kernel's OUT6 instantiations (ops.conv2d falls back to a stream tensor + the cast kernel where those cannot run; the
                # UNet's up-samplers sit behind GEMM-shaped producers and small maps and stay fp8). Half the matrix-pipe passes of fp8 on the
                # correction chunks (1.5x instead of 2x per layer), the same 3 mantissa bits with a scale per 32-channel block. The GEMM-shaped
                # MX kernels are not matrix-pipe bound and stay fp8. OMGSR_MX=8 keeps every layer at fp8 (A/B runs).
                is_up = bool(getattr(m, "phase_upsample", False))
                fp6 = conv3 and mx_env != "8" and ((is_up and is_vae) or (not is_up and name.rsplit(".", 1)[-1] in ("conv1", "conv2")))
                m.op_split = 4 if fp6 else 3
                n += 1
    return n


def set_inner16(model: nn.Module, patterns: Iterable[str]) -> int:
    """Mark the Conv2d layers whose output stays in the 16-bit compute type in the accurate tier; returns how many."""
    _touch()
    regs = [re.compile(p) for p in patterns]
    n = 0
    for name, m in model.named_modules():
        if isinstance(m, Conv2d):
            m.out_inner16 = any(r.search(name) for r in regs)
            n += int(m.out_inner16)
    return n


def set_qk_split(model: nn.Module, patterns: Iterable[str]) -> int:
    """Mark the VAE attention blocks (modules with group_norm + to_q: one head over the whole map) whose q and k leave their projections
    as two-term splits, so the score GEMM runs q_hi k_hi + q_lo k_hi + q_hi k_lo (autoencoder_kl.VaeAttention.attend); returns how many."""
    _touch()
    regs = [re.compile(p) for p in patterns]
    n = 0
    for name, m in model.named_modules():
        if hasattr(m, "to_q") and hasattr(m, "group_norm"):
            m.qk_split = any(r.search(name) for r in regs)
            n += int(m.qk_split)
    return n


def policy_fingerprint(model: nn.Module) -> str:
    """Hash of the per-layer (op_split, w_split, out_inner16) assignment: constants folded under one policy (cross-attention
    K / V^T, omgsr_amd.constants) are refused under another."""
    import hashlib
    h = hashlib.sha256()
    for name, m in model.named_modules():
        if isinstance(m, (Conv2d, Linear)):
            h.update(f"{name}:{m.op_split}:{m.w_split}:{int(m.out_inner16)};".encode())
        elif getattr(m, "qk_split", False):
            h.update(f"{name}:qk;".encode())
    return h.hexdigest()[:16]


def _same(mods, attr: str, what: str) -> None:
    vals = {getattr(g, attr) for g in mods}
    if len(vals) > 1:
        raise ValueError(f"precision policy: {what} must agree on {attr}, got {sorted(vals)}")


def check_policy(model: nn.Module) -> None:
    """Layers that read ONE shared operand must agree on its form: q / k / v (and Flux's proj_mlp) of a self-attention, Flux's
    add_q / add_k / add_v projections, and to_out / to_add_out (one attention-output buffer). Layers packed into ONE weight
    (the fused q | k projections) must also agree on the weight split."""
    for name, m in model.named_modules():
        add = [getattr(m, a, None) for a in ("add_q_proj", "add_k_proj", "add_v_proj")]
        if all(isinstance(g, Linear) for g in add):
            _same(add, "op_split", f"{name}.add_q_proj / add_k_proj / add_v_proj (one LayerNorm'd context operand)")
            _same(add[:2], "w_split", f"{name}.add_q_proj / add_k_proj (one fused weight)")
        outs = [getattr(m, "to_out", None), getattr(m, "to_add_out", None)]
        if isinstance(outs[1], Linear) and isinstance(outs[0], nn.ModuleList) and isinstance(outs[0][0], Linear):
            _same([outs[0][0], outs[1]], "op_split", f"{name}.to_out.0 / to_add_out (one attention-output operand)")
        qk = [getattr(m, a, None) for a in ("to_q", "to_k")]
        if all(isinstance(g, Linear) for g in qk) and not getattr(m, "is_cross", False):
            _same(qk, "w_split", f"{name}.to_q / to_k (one fused weight)")
        group = [getattr(m, a, None) for a in ("to_q", "to_k", "to_v")]
        if all(isinstance(g, Linear) for g in group):
            cross = getattr(m, "is_cross", False)
            shared = group[:1] if cross else group
            extra = getattr(m, "_shares_input_with", None)
            if extra is not None:
                shared = shared + list(extra)
            splits = {g.op_split for g in shared}
            if len(splits) > 1:
                raise ValueError(f"precision policy: {name}.to_q / to_k / to_v read one operand but disagree on its split {splits}")
            if cross and group[1].op_split != group[2].op_split:
                raise ValueError(f"precision policy: {name}.to_k / to_v read one operand (the prompt) but disagree on its split")


def apply_default_policy(vae: Optional[nn.Module] = None, unet: Optional[nn.Module] = None, flux: Optional[nn.Module] = None) -> None:
    if vae is not None:
        set_operand_split(vae, VAE_ACT)
        set_weight_split(vae, VAE_W)
        set_mx(vae, VAE_MX)
        set_inner16(vae, VAE_INNER16)
        set_qk_split(vae, VAE_QK_SPLIT)
    if unet is not None:
        import os
        set_operand_split(unet, UNET_ACT)
        set_weight_split(unet, [r"."] if os.environ.get("OMGSR_UNET_W_ALL") == "1" else UNET_W)      # (A/B switch: every UNet weight split)
        set_mx(unet, UNET_MX)
        set_mx_linear(unet, UNET_MX_LIN)
    if flux is not None:
        set_operand_split(flux, FLUX_ACT)
        set_weight_split(flux, FLUX_W)


def resolve(policy, **models) -> None:
    """The pipelines' `precision_policy=` argument (accurate tier only): None / "default" = the shipped lists above; "all" = every
    layer's operand AND weight as two-term splits (3x the MFMA work, the 2^-22 floor: the fallback when a checkpoint's
    statistics are far from what the shipped lists were sized on); or {"vae" | "unet" | "flux": dict(act=[...], weight=[...],
    inner16=[...])} with regular expressions over module names. OMGSR_PRECISION_POLICY=all|default overrides None."""
    import os
    if policy is None:
        policy = os.environ.get("OMGSR_PRECISION_POLICY") or "default"
    if policy == "default":
        return apply_default_policy(**models)
    for key, m in models.items():
        if m is None:
            continue
        if policy == "all":
            apply_policy(m, [r"."], [r"."], qk=[r"."])
        elif isinstance(policy, dict):
            if key in policy:
                apply_policy(m, **policy[key])
            else:
                apply_default_policy(**{key: m})
        else:
            raise ValueError(f"precision_policy must be None, 'default', 'all' or a dict, got {policy!r}")


@contextlib.contextmanager
def bf16_operand_fallback(*models):
    """The fp16 range guard's fallback (ops.overflow_seen): inside the block the accurate tier runs with bf16 operands (fp32's
    exponent range) and every layer's operand and weight as two-term splits (16-bit mantissas: the best bf16 operands can do;
    attention probabilities stay single bf16 values), then tier and policy are put back. Costs a re-pack of the weights - it is a
    fallback for a call that would otherwise return a clipped image, not a mode to run in."""
    from . import ops
    models = [m for m in models if m is not None]
    saved = [[(m, m.op_split, m.w_split, m.out_inner16) for m in model.modules() if isinstance(m, (Conv2d, Linear))] for model in models]
    ops.set_compute_dtype(torch.float32, operand_dtype=torch.bfloat16)
    try:
        for model in models:
            apply_policy(model, [r"."], [r"."])
        yield
    finally:
        for rows in saved:
            for m, a, w, i16 in rows:
                m.op_split, m.w_split, m.out_inner16 = a, w, i16
        _touch()
        ops.set_compute_dtype(torch.float32)


class RangeFallback:
    """Per-pipeline state of the fp16 range guard. The first call whose fp16 operands clip is recomputed with bf16 operands (every
    operand and weight as a two-term split) - and the pipeline then STAYS in that mode (`sticky`): with a checkpoint whose
    activations leave the fp16 range (FLUX: the reason the reference defaults to bf16) every image would otherwise pay an fp16 pass,
    a re-pack of every weight, a bf16 pass and a re-pack back. `count` = calls that overflowed; `sticky` = running range-safe;
    `reset()` returns to fp16 operands and the policy that was active when the pipeline was built."""

    def __init__(self, *models, weight_dtype=None):
        self.models = [m for m in models if m is not None]
        # the tier this pipeline was built for (its --weight_dtype): the operand dtype is PROCESS-wide state (ops.set_compute_dtype), so
        # every forward() re-asserts what its own pipeline needs - after pipeline A went sticky (fp32 stream, bf16 operands) a second
        # accurate pipeline B would otherwise run its fp16-only policy (MX layers, q / k splits) under bf16 operands and fail every call
        self.weight_dtype = weight_dtype
        self.mx_saturation_count = 0    # calls in which a mixed-precision operand carried a value beyond +-448 (its fp8 correction fields saturated)
        self.mx_demoted = False         # round 6: after such a call the fixed-scale fp8 layers run fp16 correction segments (demote_mx), sticky
        self._mx_saved = None
        self.count = 0
        self.sticky = False
        self._saved = None
        self.on_mode_change = None       # optional callback (the pipelines drop their captured hipGraphs: other kernels, other packed weights)

    def enter(self) -> None:
        from . import ops
        if self._saved is None:
            self._saved = [[(m, m.op_split, m.w_split, m.out_inner16) for m in model.modules() if isinstance(m, (Conv2d, Linear))]
                           for model in self.models]
            for model in self.models:
                apply_policy(model, [r"."], [r"."])
        ops.set_compute_dtype(torch.float32, operand_dtype=torch.bfloat16)
        self.sticky = True
        if self.on_mode_change:
            self.on_mode_change()

    def demote_mx(self) -> int:
        """the fp8 correction fields of the mixed-precision form (op_split 3: UNET_MX_LIN, the GEMM-shaped MX convs, the UNet's
        up-samplers) use FIXED scales and clamp at +-448, and the layers that carry them read un-normalised activations (GEGLU hidden states,
        attention outputs, raw residual streams) that a trained checkpoint drives into the hundreds or thousands. When the saturation bit fires,
        those layers fall back to fp16 correction segments (op_split 2 with the weight split kept: three K segments, no range limit below fp16's
        own, which the range guard covers) - per layer FORM, for the whole pipeline, sticky like the range fallback. The fp6 layers (op_split 4)
        carry a scale per 32-channel block and never saturate. Returns how many layers moved."""
        moved = []
        for model in self.models:
            for m in model.modules():
                if isinstance(m, (Conv2d, Linear)) and m.op_split == 3:
                    m.op_split = 2
                    moved.append(m)
        if moved:
            self._mx_saved = (self._mx_saved or []) + moved
            self.mx_demoted = True
            _touch()
            if self.on_mode_change:
                self.on_mode_change()
        return len(moved)

    def reassert(self) -> None:
        """Called at the top of every forward(): another pipeline may have switched the process-wide tier in between."""
        from . import ops
        if self.sticky:
            if not ops.precise() or ops.act_dtype() != torch.bfloat16:
                ops.set_compute_dtype(torch.float32, operand_dtype=torch.bfloat16)
        elif self.weight_dtype == torch.float32:
            if not ops.precise() or ops.act_dtype() != torch.float16:
                ops.set_compute_dtype(torch.float32)
        elif self.weight_dtype in (torch.bfloat16, torch.float16):
            if ops.precise() or ops.act_dtype() != self.weight_dtype:
                ops.set_compute_dtype(self.weight_dtype)

    def reset(self) -> None:
        from . import ops
        if self._saved is not None:
            for rows in self._saved:
                for m, a, w, i16 in rows:
                    m.op_split, m.w_split, m.out_inner16 = a, w, i16
            self._saved = None
            _touch()
        if self._mx_saved is not None:           # (after the rows above: a range fallback entered AFTER a demotion saved the demoted marks)
            for m in self._mx_saved:
                m.op_split = 3
            _touch()
        self._mx_saved, self.mx_demoted = None, False
        if self.sticky:
            ops.set_compute_dtype(torch.float32)
        self.sticky = False
        if self.on_mode_change:
            self.on_mode_change()

    def run(self, run, what: str):
        """run() -> result (synchronised by the caller's contract: this method synchronises before reading the guard word)."""
        import warnings
        from . import ops
        self.reassert()
        out = run()
        torch.cuda.synchronize()
        overflowed = (not self.sticky) and ops.precise() and ops.overflow_seen()          # (reads the guard word: also latches the MX diagnostic bit)
        saturated = ops.precise() and ops.mx_saturation_seen()
        if saturated:
            # not an error - the fp16 main term of that element is exact, its two correction terms were clamped, so it carried
            # single-rounding accuracy. Seeded weights never get here; a checkpoint whose feed-forward / attention outputs reach the hundreds
            # does, and then the accuracy evidence of this tier (all from O(1) activations) does not cover it: say so once - and (round 6)
            # stop using the fixed-scale form: demote_mx() + one recompute, sticky (OMGSR_MX_SAT_FALLBACK=0: warn only, round 5's behaviour)
            self.mx_saturation_count += 1
            fallback = _os.environ.get("OMGSR_MX_SAT_FALLBACK", "1") != "0" and not overflowed and not self.sticky
            if self.mx_saturation_count == 1:
                warnings.warn(f"{what} accurate tier: a mixed-precision (fp16 + fp8 correction) operand held values beyond +-448; their correction "
                              f"terms saturated (single-rounding accuracy for those elements). " +
                              ("This call is recomputed with fp16 correction segments on those layers and the pipeline stays in that form "
                               "(pipe.range_fallback.reset() returns to the fp8 form)" if fallback else
                               "OMGSR_MX_LINEAR=0 / OMGSR_MX=0 run those layers with fp16 correction segments instead"))
            if fallback and self.demote_mx():
                out = run()
                torch.cuda.synchronize()
                overflowed = ops.overflow_seen()
                ops.mx_saturation_seen()                 # (the fp6 layers never set it; the demoted ones cannot)
        if overflowed:
            self.count += 1
            warnings.warn(f"{what} accurate tier: an fp16 MFMA operand exceeded 65504; this call is recomputed with bf16 operands and "
                          f"the pipeline stays range-safe (pipe.range_fallback.reset() returns to fp16 operands)")
            self.enter()
            out = run()
            torch.cuda.synchronize()
        return out


def apply_policy(model: nn.Module, act: Iterable[str], weight: Iterable[str] = (), inner16: Iterable[str] = (), qk=None) -> None:
    """An explicit policy for one model (pipelines: `precision_policy=` / the CLI's --precision_policy all): lists of regular
    expressions over module names; [r"."] splits every layer (activation and weight to 2^-22: 3x the MFMA work). qk: the VAE attention
    blocks whose q / k carry splits (None leaves the marks as they are: the range-guard fallback keeps them)."""
    set_operand_split(model, list(act))
    set_weight_split(model, list(weight))
    set_inner16(model, list(inner16))
    if qk is not None:
        set_qk_split(model, list(qk))
