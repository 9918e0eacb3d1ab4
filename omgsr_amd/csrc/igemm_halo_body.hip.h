// 3x3 stride-1 pad-1 convolution, halo-tile variant of the LDS-DMA implicit GEMM (gfx950).
//
// profiles/r01_pmc_igemm.md: the LDS-DMA GEMM kernel spends ~70 % of its time just streaming operands
// L2 -> LDS (24 KB per 2.1 MFLOP K-step); 9 of every 10 A bytes are the SAME input pixels fetched again
// for another tap. This kernel cuts the bytes instead of chasing the stream:
//
//  * an output tile is SPATIAL: 8 rows x 32 pixels (256 outputs) x 128 channels; for one 32-channel chunk
//    the (8+2) x (32+2) input patch is DMA'd into LDS ONCE (21.3 KB) and all 9 taps read their MFMA A
//    fragments from it at shifted rows (fragment = 32 consecutive pixels of one tile row, so the swizzled
//    linear image stays conflict-free at any shift)      -> A traffic 144 KB -> 24 KB per chunk
//  * K order is chunk-major: k = (cc*9 + tap)*32 + c, and the weights are packed SLICE-major (`weight_cm` =
//    [Cin/32][9][Cout_pad][32]): a K-step = (chunk, tap) streams its 8 KB weight slice, one contiguous run of
//    full 128-B lines, through a 3-deep ring (row-major weights made every DMA instruction touch 16 half lines)
//  * per 18.9 MFLOP chunk: 24 KB (A, double-buffered, prefetched one chunk ahead) + 72 KB (B) = 5.1 KB/MFLOP
//    vs 12 KB/MFLOP for the GEMM-shaped kernel
//  * 4 waves, 128 x 64 per wave (4 tile rows x 64 couts), one raw s_barrier per K-step, counted vmcnt
//  * image borders / ragged W,H: patch pixels outside the image come from the zero page, output columns
//    past W and rows past H are dropped in the epilogue (tiled-VAE tiles are 86, 172, 320 ... wide)
//  * TAPS = 4: nearest-2x upsampling + 3x3 conv WITHOUT the redundant taps. Output pixel (2y + a, 2x + b) of conv3x3(up2(in)) only
//    ever sees the 2 x 2 input pixels rows {y - 1 + a, y + a} x cols {x - 1 + b, x + b}: taps that land on the same input pixel
//    have their weights SUMMED at pack time (phase a = 0: row y - 1 <- w[0], row y <- w[1] + w[2]; a = 1: row y <- w[0] + w[1],
//    row y + 1 <- w[2]; same along x), so each of the four output phases is a 2 x 2 convolution of the LOW-resolution map: 4 / 9 of
//    the MFMA work of the gather form. One launch, blockIdx.y = phase; the tile is 8 x 32 low-res pixels whose outputs land at
//    stride 2 in the high-res map; patch (8+1) x (32+1), weight ring of 4 stages (stage = tap).
#pragma once
#include "common.hip.h"
#include "../../include/omgsr_hip.h"
#include "igemm_epilogue.hip.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int TH = 8, TW = 32;
// Geometry of the two tap sets. 3 x 3: patch (8+2) x (32+2) = 340 pixels = 22 1-KiB DMA pieces (16 patch rows each), every wave
// issues 6 so the vmcnt arithmetic is uniform (pieces 22, 23 copy the zero page to a dummy KiB), weight ring 3 deep (9 % 3 == 0:
// stage = tap % 3). 2 x 2 (phase-decomposed upsampling): patch 9 x 33 = 297 pixels = 19 pieces, 5 per wave, ring 4 deep (stage = tap).
// BIG (FLAT form only): a 27-piece patch (432 rows) for maps up to 80 pixels wide - 2 x 27 KB + 2 KB + 24 KB = exactly the 80 KB that still let two
// workgroups share a CU
template <int TAPS, bool BIG = false> struct HaloGeo {
    static constexpr int KS = TAPS == 9 ? 3 : 2;
    static constexpr int PW = TW + KS - 1, PH = TH + KS - 1, PROWS = PH * PW;
    static constexpr int APIECES = BIG ? 27 : (PROWS + 15) / 16, APW = (APIECES + 3) / 4;
    static constexpr int A_BYTES = APIECES * 1024;
    static constexpr int NB = TAPS == 9 ? 3 : 4;
    static constexpr int DUMMY_OFF = 2 * A_BYTES, B_OFF = DUMMY_OFF + 2048;
    static constexpr int LDS_BYTES = B_OFF + NB * 128 * 64;
};
constexpr int BN = 128, B_BYTES = BN * 64, BPW = 2;       // BN / B_BYTES: the wide shape; the narrow one uses 2 KB of each stage
constexpr int LDS_BYTES = HaloGeo<9>::LDS_BYTES > HaloGeo<4>::LDS_BYTES ? HaloGeo<9>::LDS_BYTES : HaloGeo<4>::LDS_BYTES;   // <= 72 KB: two workgroups per CU
static_assert(HaloGeo<9>::APIECES == 22 && HaloGeo<9>::APW == 6 && HaloGeo<4>::APIECES == 19 && HaloGeo<4>::APW == 5, "piece counts");
constexpr int LDS_BYTES_BIG = HaloGeo<9, true>::LDS_BYTES;
// GNF instantiations: + the (scale, shift) table of one image, Cin <= 1024 channels x 2 floats (still two workgroups per CU)
constexpr int GN_MAX_CIN = 1024, LDS_BYTES_GN = HaloGeo<9>::LDS_BYTES + GN_MAX_CIN * 8;
static_assert(LDS_BYTES_GN <= 81920, "GNF: two workgroups per CU");
static_assert(HaloGeo<9, true>::APW == 7 && LDS_BYTES_BIG == 81920, "big FLAT patch: two workgroups per CU");

__device__ __attribute__((aligned(16))) unsigned int g_zero_page_h[4] = {0u, 0u, 0u, 0u};

OMGSR_DEVINL void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

// PRIO: raise the wave's priority around its MFMA cluster. Measured +6..10 % on K-heavy layers (Cin >= 512) and
// -3..5 % on the epilogue-heavy 128/256-channel layers with row-major weights; with the slice-major packing it no longer
// pays anywhere (S-1024 step 136.3 with it on Cin >= 384, 135.7 without): off by default, OMGSR_HALO_PRIO_CIN=<n> for A/B.
// NARROW: Cout <= 32 (the VAE's conv_out, 128 -> 3): the four waves split the 8 tile rows (2 each) over ONE 32-column
// fragment instead of 2 x 2 waves over 128 columns. The im2col kernels gather every input pixel nine times out of L2
// (3.2 GB for a 1.4 MPixel x 128-channel map: 350 us, L2-bound at 70 TFLOP/s); here the patch is read once.
// GNF (round 5): GroupNorm apply (+ SiLU) as the patch PRODUCER (omgsr_igemm_args.gn_scale_shift). 0 = `in` is a ready 16-bit operand.
// 1 = `in` is the 16-bit STREAM tensor the GroupNorm reads: the patch of the next chunk is LDS-DMA'd exactly as before (raw values), and
// between its arrival (the counted wait of tap 2 covers a wave's OWN pieces) and its first use one chunk later every wave normalises the
// pieces it fetched IN PLACE: at the end of taps 2..7, behind the step's MFMAs when the fragment registers are dead, a lane reads back the 16
// bytes it DMA'd (8 channels of one patch pixel), applies x * scale[c] + shift[c] (the (scale, shift) table of the tile's image sits in LDS
// behind the weight ring), SiLU, rounds once and writes them back; pixels outside the image stay the exact zeros the zero page supplied
// (the conv pads the NORMALISED map). No other wave touches those bytes before the next chunk's first barrier. The separate apply pass
// (read stream, write operand) and the conv's read of that operand are gone: 4 of 6 bytes per element. Spatial nine-tap form, plain
// 16-bit operand and weight only (no split / MX / wrap). An fp32 stream (the accurate tier's) would need its patch register-staged (a raw
// fp32 chunk is 42 KB: no LDS room): loads held in VGPRs across K-steps next to inline-asm DMA traffic the compiler cannot count - not built.
// SPLITK (round 5): the workgroup runs the chunk range IgemmGeo.cc0 .. cc1 of the contraction (split-K launch groups). A template flag, not a run-time
// test: with the two extra loop bounds live the FLAT instantiations - already at 256 registers - went from 3 to 38 spilled VGPRs.
template <typename T, int ABL, bool PRIO, bool NARROW = false, int TAPS = 9, int MX = 0, int FLAT = 0, int GNF = 0, bool SPLITK = false, bool OUT6 = false>      // OUT6: the epilogue writes OMGSR_EL_MX6 and nothing else (igemm_halo_out6.hip); MX: 0 plain, 1 fp8 correction chunks, 6 fp6 ones; FLAT: 0 spatial tiles, 1 FLAT form, 2 FLAT form with the 27-piece patch
OMGSR_DEVINL void halo_body(const omgsr_igemm_args& p, const IgemmGeo& g, const int tile, const int bidy) {      // tile: logical (XCD-remapped) tile index
    static_assert(GNF == 0 || (TAPS == 9 && !MX && FLAT == 0 && ABL == 0 && !PRIO), "GNF: spatial nine-tap form only");
    constexpr int WTN = NARROW ? 32 : 64, FM = NARROW ? 2 : 4, FN = NARROW ? 1 : 2, BNK = NARROW ? 32 : 128;
    using HG = HaloGeo<TAPS, FLAT == 2>;
    constexpr int KS = HG::KS, PW = HG::PW, PROWS = HG::PROWS, APIECES = HG::APIECES, APW = HG::APW, A_BYTES = HG::A_BYTES;
    constexpr int NB = HG::NB, DUMMY_OFF = HG::DUMMY_OFF, B_OFF = HG::B_OFF;
    constexpr bool PHASE = TAPS == 4;
    static_assert(!(PHASE && NARROW), "the phase-decomposed form has no narrow shape");
    const int ph_a = PHASE ? (bidy >> 1) : 0, ph_b = PHASE ? (bidy & 1) : 0;    // output phase (row, column parity)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = NARROW ? wave : (wave >> 1), wn = NARROW ? 0 : (wave & 1);

    const int tn = tile % g.ntn, tm = tile / g.ntn;
    const int per_img = g.tiles_x * g.tiles_y;
    const int img = tm / per_img;
    const int trem = tm - img * per_img;
    const int ty = trem / g.tiles_x, tx = trem - ty * g.tiles_x;
    const int y0 = ty * TH, x0 = tx * TW, n0 = tn * BNK;
    // FLAT form (g.flat = P = W + 2, nine taps, no upsampling): the tile is 256 consecutive positions f0 .. f0 + 255 of the image's flattened
    // padded map (f = y P + x', x' = x + 1; x' = 0 and P - 1 are zero border columns), the patch the positions f0 - P - 1 .. f0 + 256 + P:
    // a tap (ky, kx) is the uniform shift ky P + kx inside the patch, whatever row of the image a position belongs to. A narrow map
    // (the tiled VAE's 1/8-resolution tile images are 36-40 pixels wide) then wastes 2 of P columns instead of up to half of a 32-pixel tile.
    // FLAT is a template parameter so that the spatial instantiations keep their compile-time fragment geometry (and their register count)
    const int FP = (FLAT && TAPS == 9 && !NARROW) ? g.flat : 0;
    const int f0 = trem * (TH * TW);

    const T* __restrict__ in = (const T*)p.in;
    // phase form: weight_ph holds the four phase-summed 2 x 2 kernels back to back, [4][Cin/32][4 taps][Cout_pad][32]
    const T* __restrict__ wt = PHASE ? (const T*)p.weight_ph + (int64_t)bidy * (p.Cin / 32) * 4 * p.Cout_pad * 32 : (const T*)p.weight_cm;
    typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
    const unsigned lds_base = (unsigned)(size_t)(lds_byte_t*)lds;

    const int lrow = lane >> 2;
    const int kc = (lane & 3) ^ ((lane >> 4) & 3);           // source chunk for LDS position (lane & 3)

    // A patch pieces: piece j = wave*6 + i covers patch rows [16j, 16j+16); patch row -> (py, px)
    const unsigned char* a_ptr[APW];
    int a_inc[APW];
    unsigned gn_ok = 0;
    const int ild = p.in_ld > 0 ? p.in_ld : p.Cin;      // physical channels per pixel row
    const int wrap_at = ild / 32;                       // chunk index at which the patch pointer returns to channel 0 (w_lo segment)
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        const int pr = 16 * (wave * APW + i) + lrow;
        int py = pr / PW, px = pr - py * PW;
        bool in_patch = pr < PROWS;
        if (FP) {               // patch position pr is flattened position f0 - P - 1 + pr = row py, padded column px of the image
            const int fq = f0 - 1 + FP + pr;                  // + 2 P: non-negative
            py = fq / FP - 1; px = fq - (py + 1) * FP;           // row y = py - 1, padded column x' = px: (vy, vx) below = (y, x' - 1)
            in_patch = pr < TH * TW + 2 * FP + 2;
        }
        // (vy, vx): coordinates in the virtual (optionally nearest-2x upsampled) input = output coordinates; phase form: the tile
        // lives on the LOW-res grid and phase (a, b) reads input rows y - 1 + a, y + a (columns likewise)
        const int vy = (FP ? 0 : y0) - 1 + (PHASE ? ph_a : 0) + py, vx = (FP ? 0 : x0) - 1 + (PHASE ? ph_b : 0) + px;
        const bool ok = (wave * APW + i) < APIECES && in_patch && (unsigned)vy < (unsigned)(PHASE ? p.H : p.Ho) && (unsigned)vx < (unsigned)(PHASE ? p.W : p.Wo);
        const int iy = PHASE ? vy : (vy >> p.upsample), ix = PHASE ? vx : (vx >> p.upsample);
        const int64_t pix = ((int64_t)img * p.H + iy) * p.W + ix;
        a_ptr[i] = ok ? reinterpret_cast<const unsigned char*>(in + pix * ild + kc * 8)
                      : reinterpret_cast<const unsigned char*>(g_zero_page_h);
        a_inc[i] = ok ? 64 : 0;
        if constexpr (GNF != 0) gn_ok |= (ok ? 1u : 0u) << i;      // (a zero-page pixel must stay zero: silu(shift) is not)
    }
    const unsigned char* b_ptr[BPW];
#pragma unroll
    for (int i = 0; i < BPW; ++i)       // narrow shape: only pieces 0, 1 are real, the rest copy the zero page to the dummy KiB (uniform vmcnt arithmetic)
        b_ptr[i] = (wave * BPW + i) * 16 < BNK
                       ? reinterpret_cast<const unsigned char*>(wt + (int64_t)(n0 + 16 * (wave * BPW + i) + lrow) * 32 + kc * 8)
                       : reinterpret_cast<const unsigned char*>(g_zero_page_h);
    const int64_t b_step = (int64_t)p.Cout_pad * 64;       // bytes between consecutive (chunk, tap) slices
    // split-K (IgemmGeo.cc0 / cc1): this workgroup's share of the contraction is the chunk range [cbeg, ncc); patch and weight streams start there
    const int cbeg = (SPLITK && g.cc1 > 0) ? g.cc0 : 0;
    if (cbeg > 0) {
#pragma unroll
        for (int i = 0; i < APW; ++i) a_ptr[i] += (int64_t)a_inc[i] * cbeg;
#pragma unroll
        for (int i = 0; i < BPW; ++i)
            if ((wave * BPW + i) * 16 < BNK) b_ptr[i] += b_step * TAPS * cbeg;
    }

    auto issue_a = [&](int buf, const int chunk) {
        if (chunk == wrap_at) {
#pragma unroll
            for (int i = 0; i < APW; ++i) a_ptr[i] -= (int64_t)a_inc[i] * wrap_at;
        }
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            const int piece = wave * APW + i;
            const unsigned dst = piece < APIECES ? lds_base + buf * A_BYTES + piece * 1024
                                                 : lds_base + DUMMY_OFF + (piece - APIECES) * 1024;
            glds16(a_ptr[i], __builtin_amdgcn_readfirstlane(dst));
            a_ptr[i] += a_inc[i];
        }
    };
    auto issue_b = [&](int stage) {
        const unsigned dst = lds_base + B_OFF + stage * B_BYTES + (wave * BPW) * 1024;
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            const bool real = (wave * BPW + i) * 16 < BNK;
            glds16(b_ptr[i], __builtin_amdgcn_readfirstlane(real ? dst + i * 1024 : lds_base + DUMMY_OFF));
            if (real) b_ptr[i] += b_step;
        }
    };

    // ---- GNF: normalise this wave's piece i of patch buffer `buf` (chunk `chunk`) in place -------------------------------------
    // read (the lane's 16 raw bytes + its (scale, shift) octet) / math (~70 VALU instructions, SiLU is the only activation: omgsr_igemm
    // refuses anything else) / write-back; the dummy pieces of the last wave normalise the dummy KiB (wave-uniform address select)
    constexpr int GN_TAB_OFF = HG::LDS_BYTES;                   // (scale, shift) pairs of the tile's image, Cin x 2 floats, behind the weight ring
    const int gn_lane16 = lane * 16, gn_kc64 = kc * 64;
    auto gn_slot = [&](const int i, const int buf) -> u32x4_t* {
        const int piece = wave * APW + i;                        // (wave-uniform)
        const int off = piece < APIECES ? buf * A_BYTES + piece * 1024 : DUMMY_OFF + (piece - APIECES) * 1024;
        return reinterpret_cast<u32x4_t*>(lds + off + gn_lane16);
    };
    auto gn_read = [&](const int i, const int buf, const int chunk, u32x4_t& raw, f32x4_t (&tb)[4]) {
        raw = *gn_slot(i, buf);
        const f32x4_t* tp = reinterpret_cast<const f32x4_t*>(lds + GN_TAB_OFF + chunk * 256 + gn_kc64);
        tb[0] = tp[0]; tb[1] = tp[1]; tb[2] = tp[2]; tb[3] = tp[3];
    };
    auto gn_math = [&](const int i, const u32x4_t raw, const f32x4_t (&tb)[4]) -> u32x4_t {
        float f[8];
        unpack8<T>(raw, f);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f[2 * q] = silu_f(f[2 * q] * tb[q][0] + tb[q][1]);
            f[2 * q + 1] = silu_f(f[2 * q + 1] * tb[q][2] + tb[q][3]);
        }
        u32x4_t o = pack8<T>(f);
        if (!((gn_ok >> i) & 1u)) o = (u32x4_t){0u, 0u, 0u, 0u};      // outside the image: the conv pads the NORMALISED map with zeros
        return o;
    };

    f32x16_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int ncc = (SPLITK && g.cc1 > 0) ? g.cc1 : g.nk;      // (one past) the last 32-channel chunk this workgroup runs
    const int nsteps = ncc * TAPS;         // K-step index s = chunk * TAPS + tap is absolute: only differences and comparisons with nsteps are used

    // The first versions of this loop were instruction-issue bound (profiles/r01_pmc_igemm.md §6: 5.3 VALU +
    // 4.4 SALU per MFMA, mostly LDS address arithmetic and tap bookkeeping). Everything is now static:
    //   * the 9 taps are unrolled; the weight-ring stage of step (cc, tap) is tap % 3 because 9 % 3 == 0
    //   * chunks are unrolled by 2 so the patch buffer parity is an immediate offset
    //   * the swizzled LDS offset of each lane's A fragment is precomputed per (tap, tile row): 36 VGPRs;
    //     the second half-K fragment is the same address ^ 32
    const int frow = lane & 31;
    const int half = lane >> 5;
    const int bsw = (frow >> 2) & 3;
    const int boff0 = (wn * WTN) * 64 + frow * 64 + ((half) ^ bsw) * 16;
    const int boff1 = (wn * WTN) * 64 + frow * 64 + ((2 + half) ^ bsw) * 16;
    // FLAT: consecutive row blocks of a wave are 32 patch rows apart - the swizzle term ((row >> 2) & 3) is the same for all of them and the
    // 2 KiB between them is an immediate of the ds_read: 9 address registers instead of 36
    constexpr int AFM = FLAT ? 1 : FM, AIMM = FLAT ? TW * 64 : 0;
    int aoff[TAPS][AFM];
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int i = 0; i < AFM; ++i) {
            const int row = FLAT ? (FM * wm + i) * TW + frow + (tp / KS) * FP + (tp % KS) : (FM * wm + i) * PW + frow + (tp / KS) * PW + (tp % KS);
            aoff[tp][i] = row * 64 + ((half ^ ((row >> 2) & 3)) << 4);
        }

    if constexpr (ABL != 2) {
        issue_a(0, cbeg);
        issue_b(0);
        issue_b(1);
    }
    if constexpr (GNF != 0) {
        // the (scale, shift) table of the tile's image -> LDS (rows of a tile-major tensor share the statistics of image n % gn_nimg), then
        // the first chunk's pieces, which nothing hides: ~0.7 us per tile, once (every later chunk is normalised behind the previous one's MFMAs)
        const f32x4_t* tsrc = reinterpret_cast<const f32x4_t*>(p.gn_scale_shift + (int64_t)(img % p.gn_nimg) * ild * 2);
        f32x4_t* tdst = reinterpret_cast<f32x4_t*>(lds + GN_TAB_OFF);
        for (int q = t; q < ild / 2; q += 256) tdst[q] = tsrc[q];
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");        // this wave's patch pieces (older than the two weight slices) have landed
        __syncthreads();                                         // the table is complete
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            u32x4_t raw; f32x4_t tb[4];
            gn_read(i, 0, cbeg, raw, tb);
            *gn_slot(i, 0) = gn_math(i, raw, tb);
        }
    }

    // one K-step with compile-time tap and patch parity
    // F8: the chunk holds 64 fp8 channels per pixel / weight row (the correction segments of an OMGSR_EL_MX operand): the same 64-byte
    // rows, the same fragment addresses - a lane's two 16-byte reads are simply the 32 consecutive k of ITS k-block for
    // v_mfma_scale_f32_32x32x64_f8f6f4 (A and B are read at the same byte positions, so the products pair up whatever the byte order
    // inside a row means) - one instruction of 16 passes where the fp16 path issues two of 8: same matrix-pipe time, twice the channels.
    const int n16 = (MX && p.mx_chunks16 > 0) ? p.mx_chunks16 : g.nk;
    const int seg2_at = n16 + (g.nk - n16) / 2;          // first chunk of the a_hi' x w_lo' segment
    auto step = [&](auto tap_c, auto par_c, auto f8_c, const int cc, const int s) {
        constexpr int tap = decltype(tap_c)::value, par = decltype(par_c)::value;
        constexpr int FMT = (int)decltype(f8_c)::value;          // 0: fp16 / bf16 chunk; 1: fp8 (e4m3) correction chunk; 2: fp6 (e2m3) correction chunk
        constexpr bool F8 = FMT != 0;
        // wait for slice s: only what the PREVIOUS step issued may still be in flight. lgkmcnt(0): every fragment read of the
        // previous step has RETURNED before this wave passes the barrier that lets the others overwrite that stage - hipcc
        // is free to sink MFMAs (and the lgkmcnt wait in front of them) below the barrier, and a ds_read still queued
        // there raced the next LDS-DMA write about once per 10^5 tiles (one wave, a few weight rows of one K-step: found by
        // the bit-repeatability test; the s_setprio variant pins the MFMAs and never showed it)
        if constexpr (ABL == 2) {
        } else if constexpr (tap == 1) {
            // the previous step (tap 0) issued the next chunk's patch (APW pieces) and one weight slice (2)
            if (cc + 1 < ncc) {
                if constexpr (APW == 7) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");
                else if constexpr (APW == 6) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
            }
            else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        } else if constexpr (tap == TAPS - 1) {
            if (s + 1 < nsteps) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // LATE: this step's LDS-DMA pieces are issued after the first half of its MFMAs instead of in front of them, so their
        // issue slots (60-185 cycles per piece) overlap matrix-pipe time: +1.5...3 % on every VAE shape (profiles/r02_halo_late.md;
        // ABL 5 = the early schedule, OMGSR_HALO_VARIANT=0, for A/B). Safe: the stage they fill was last read in the previous
        // step, and every wave passed this step's barrier with those reads retired (lgkmcnt(0) above).
        constexpr bool LATE = (ABL == 0) && !PRIO && !NARROW;
        auto issue_dma = [&]() {
            if constexpr (ABL != 2) {
                if constexpr (tap == 0) { if (cc + 1 < ncc) issue_a(par ^ 1, cc + 1); }
                if (s + 2 < nsteps) issue_b((tap + 2) % NB);
            }
        };
        if constexpr (!LATE) issue_dma();
        if constexpr (ABL == 3) return;

        const unsigned char* As = lds + par * A_BYTES;
        const unsigned char* Bs = lds + B_OFF + (tap % NB) * B_BYTES;
        if constexpr (F8) {
            typedef int i32x4_t __attribute__((ext_vector_type(4)));
            // the scale operands are VGPRs written by VALU moves: set them up in front of the fragment reads (the asm MFMAs that read them
            // are invisible to the compiler's hazard padding; the LDS reads and their wait put >> the 2 required wait states in between)
            if constexpr (MX == 6) {
                // MX == 6 (OMGSR_EL_MX6, round 5): the same 64-byte rows and fragment addresses; of a lane's 32 bytes the first 24 are ITS 32-channel
                // block (e2m3 codes: 6 registers), byte 24 is the block's E8M0 scale. The 8-pass fp6 form of the instruction (cbsz / blgp = 2) takes
                // the six registers and, per lane, the scale bytes of both fragments. A template value, not a run-time branch: both forms in one
                // kernel cost 100+ spilled registers (the allocator loses track of the tied accumulators); 24 data bytes + 1 scale byte per read
                // instead of the whole 32: two registers less per fragment (24 spilled registers with 8-register fragments; ds_read_b128 +
                // ds_read_b96 - scale in the third dword, one LDS instruction less - spills 15: measured at build time, not shipped).
                typedef int i32x2_t __attribute__((ext_vector_type(2)));
                typedef int i32x6_t __attribute__((ext_vector_type(6)));
                i32x6_t a6[FM], b6[FN];
                int sa6[FM], sb6[FN];
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const unsigned char* q = As + i * AIMM + aoff[tap][FLAT ? 0 : i];
                    const unsigned char* q2 = As + i * AIMM + (aoff[tap][FLAT ? 0 : i] ^ 32);
                    const i32x4_t lo = *reinterpret_cast<const i32x4_t*>(q);
                    const i32x2_t hi = *reinterpret_cast<const i32x2_t*>(q2);
                    a6[i] = i32x6_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1]};
                    sa6[i] = q2[8];
                }
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const i32x4_t lo = *reinterpret_cast<const i32x4_t*>(Bs + j * 32 * 64 + boff0);
                    const i32x2_t hi = *reinterpret_cast<const i32x2_t*>(Bs + j * 32 * 64 + boff1);
                    b6[j] = i32x6_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1]};
                    sb6[j] = Bs[j * 32 * 64 + boff1 + 8];
                }
#pragma unroll
                for (int i = 0; i < FM; ++i) {
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        // s_nop 3: at this register pressure hipcc assembles some 6-register fragments with v_mov copies placed right in front of
                        // the MFMA that reads them, and it pads no hazards around inline asm - without the wait states the matrix pipe read stale
                        // registers now and then (found by the bit-repeatability assertions of test_conv_mx6 / the loader test; the FLAT forms
                        // showed it first). Four cycles in front of an instruction that occupies the pipe for 32: hidden behind its predecessor.
                        asm volatile("s_nop 3\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:2 blgp:2"
                                     : "+v"(acc[i][j]) : "v"(b6[j]), "v"(a6[i]), "v"(sb6[j]), "v"(sa6[i]));   // transposed tile
                    if constexpr (LATE) {
                        if (i == FM / 2 - 1) {
                            __builtin_amdgcn_sched_barrier(0);
                            issue_dma();
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                return;
            }
            i32x8_t a8[FM], b8[FN];
            int sw = cc < seg2_at ? p.mx_scale_w1 : p.mx_scale_w2, sa = cc < seg2_at ? p.mx_scale_a1 : p.mx_scale_a2;
            asm volatile("" : "+v"(sw), "+v"(sa));
#pragma unroll
            for (int i = 0; i < FM; ++i)
                a8[i] = __builtin_shufflevector(*reinterpret_cast<const i32x4_t*>(As + i * AIMM + aoff[tap][FLAT ? 0 : i]),
                                                *reinterpret_cast<const i32x4_t*>(As + i * AIMM + (aoff[tap][FLAT ? 0 : i] ^ 32)), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int j = 0; j < FN; ++j)
                b8[j] = __builtin_shufflevector(*reinterpret_cast<const i32x4_t*>(Bs + j * 32 * 64 + boff0), *reinterpret_cast<const i32x4_t*>(Bs + j * 32 * 64 + boff1),
                                                0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    // inline asm: with the builtin the register allocator moves whole accumulators between the fp16 and the fp8 loop
                    // (800+ spilled VGPRs at this kernel's pressure); tied in place it keeps them where they are. Dependent MFMAs on one
                    // accumulator are 7 instructions of 16 passes apart, far beyond the hazard window the compiler would otherwise pad.
                    // s_nop 3 (round 6): hipcc pads no hazards around inline asm, and a v_mov copy of a fragment register placed right in
                    // front of the MFMA raced the matrix pipe in the fp6 form (above); the fp8 sites carry the same padding instead of depending on
                    // the current register allocation
                    asm volatile("s_nop 3\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]"
                                 : "+v"(acc[i][j]) : "v"(b8[j]), "v"(a8[i]), "v"(sw), "v"(sa));   // transposed tile
                if constexpr (LATE) {
                    if (i == FM / 2 - 1) {
                        __builtin_amdgcn_sched_barrier(0);
                        issue_dma();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            return;
        }
        x8_t<T> af[2][FM], bf[2][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            af[0][i] = *reinterpret_cast<const x8_t<T>*>(As + i * AIMM + aoff[tap][FLAT ? 0 : i]);
            af[1][i] = *reinterpret_cast<const x8_t<T>*>(As + i * AIMM + (aoff[tap][FLAT ? 0 : i] ^ 32));
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            bf[0][j] = *reinterpret_cast<const x8_t<T>*>(Bs + j * 32 * 64 + boff0);
            bf[1][j] = *reinterpret_cast<const x8_t<T>*>(Bs + j * 32 * 64 + boff1);
        }
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = mfma32(bf[ks][j], af[ks][i], acc[i][j]);   // transposed tile
            if constexpr (LATE) {
                if (ks == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue_dma();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if constexpr (GNF != 0 && tap >= 2 && tap < 2 + APW) {
            // GNF: the next chunk's patch was issued in tap 0 and this wave's OWN pieces are covered by the counted wait of tap 2: piece
            // tap - 2 is normalised in place at the end of taps 2 .. 7, behind the step's MFMAs (the fragment registers are dead here).
            // Interleaving the ~70 VALU instructions with the second half of the MFMAs instead (sched_group_barrier, 256 registers)
            // measured the same: the step's cost is the VALU issue time itself, not its placement (profiles/r05_experiments.md)
            if (cc + 1 < ncc) {
                __builtin_amdgcn_sched_barrier(0);
                u32x4_t g_raw;
                f32x4_t g_tb[4];
                gn_read(tap - 2, par ^ 1, cc + 1, g_raw, g_tb);
                *gn_slot(tap - 2, par ^ 1) = gn_math(tap - 2, g_raw, g_tb);
            }
        }
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    };
    auto chunk = [&](auto par_c, auto f8_c, const int cc) {
        const int s0 = cc * TAPS;
        step(std::integral_constant<int, 0>{}, par_c, f8_c, cc, s0 + 0);
        step(std::integral_constant<int, 1>{}, par_c, f8_c, cc, s0 + 1);
        step(std::integral_constant<int, 2>{}, par_c, f8_c, cc, s0 + 2);
        step(std::integral_constant<int, 3>{}, par_c, f8_c, cc, s0 + 3);
        if constexpr (TAPS == 9) {
            step(std::integral_constant<int, 4>{}, par_c, f8_c, cc, s0 + 4);
            step(std::integral_constant<int, 5>{}, par_c, f8_c, cc, s0 + 5);
            step(std::integral_constant<int, 6>{}, par_c, f8_c, cc, s0 + 6);
            step(std::integral_constant<int, 7>{}, par_c, f8_c, cc, s0 + 7);
            step(std::integral_constant<int, 8>{}, par_c, f8_c, cc, s0 + 8);
        }
    };
    if (g.main_prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (g.main_prio == 2) __builtin_amdgcn_s_setprio(2);
    int cc = cbeg;
    const int e16 = (SPLITK && n16 > ncc) ? ncc : n16;      // (a split-K range lies on one side of the fp16 / fp8 boundary, so patch parity restarts at 0 on both)
    for (; cc < e16; cc += 2) {
        chunk(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, cc);
        if (cc + 1 < e16) chunk(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, cc + 1);
    }
    if constexpr (MX && std::is_same<T, f16_t>::value && !NARROW && ABL == 0 && !PRIO) {
        // the fp8 segments of an MX problem (mx_chunks16 even, so the patch parity starts over at 0; n16 == ncc otherwise)
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");        // (the asm MFMAs below are invisible to the compiler's hazard padding)
        for (; cc < ncc; cc += 2) {
            chunk(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, cc);
            if (cc + 1 < ncc) chunk(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, cc + 1);
        }
        // the last asm MFMAs (16 passes each) must have written the accumulators before the epilogue's VALU reads them
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    }

    if (g.main_prio) __builtin_amdgcn_s_setprio(0);
    int mb[FM], nv[FM];
    int colsv = (PHASE ? p.W : p.Wo) - x0; colsv = colsv > TW ? TW : colsv;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int y = y0 + FM * wm + i;
        // phase form: low-res pixel (y, x0 + j) of phase (a, b) is output pixel (2y + a, 2 (x0 + j) + b): stride 2 along the row
        mb[i] = PHASE ? (img * p.Ho + 2 * y + ph_a) * p.Wo + 2 * x0 + ph_b : (img * p.Ho + y) * p.Wo + x0;
        nv[i] = (y < (PHASE ? p.H : p.Ho)) ? colsv : 0;
        if (FP) {               // block i = positions f0 + 32 (FM wm + i) .. + 31; positions past the last row of the map are dropped
            mb[i] = f0 + TW * (FM * wm + i);
            const int left = p.Ho * FP - mb[i];
            nv[i] = left < 0 ? 0 : (left > TW ? TW : left);
        }
    }
    if constexpr (ABL == 1) { if (p.alpha != 12345.0f) return; }      // timing experiment: no epilogue (never true at run time)
    {
        float* epi = reinterpret_cast<float*>(lds) + wave * 32 * (WTN + 4);
        // fused GroupNorm statistics: slot = (spatial tile, upper / lower 4 tile rows), [N][2*tiles][G][2]
        // ... phase form: four launches' worth of slots per spatial tile, [N][tiles][2][4 phases][G][2]
        const int64_t slot = PHASE ? ((int64_t)(img * per_img + trem) * 2 + wm) * 4 + bidy : (int64_t)(img * per_img + trem) * 2 + wm;
        float* gn_dst = (p.gn_partial && !NARROW) ? p.gn_partial + slot * p.gn_entries * 2 : nullptr;
        igemm_epilogue<T, WTN, FM, FN, OUT6>(p, acc, epi, lane, mb, nv, n0 + wn * WTN, 0, gn_dst, 0, PHASE ? 2 : (FP ? -FP : 1), FP ? img * p.Ho * p.Wo : 0);
    }
}

// Block b of an x-only phase-form grid -> (logical tile, phase). T = tiles of the problem; n8 = its 8-aligned block range in a multi-problem
// launch (every XCD then owns n8 / 8 tiles, tiles >= T are fillers), 0 = single problem (XCD ranges as xcd_remap hands them out; the grid is
// 8 x 4 x ceil(T / 8) blocks). chunked: see igemm_halo_kernel. Returns false for a filler block.
OMGSR_DEVINL bool phase_block_map(const int b, const int T, const int n8, const bool chunked, int& tile, int& phase) {
    constexpr int CH = 64;                             // one round of workgroups per XCD: 32 CUs x 2
    const int xcd = b & 7, idx = b >> 3;
    int base, cnt;
    if (n8 > 0) { cnt = n8 >> 3; base = xcd * cnt; }
    else {
        const int q = T >> 3, r = T & 7;
        base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        cnt = xcd < r ? q + 1 : q;
    }
    if (idx >= 4 * cnt) return false;
    if (chunked) {
        const int chunk = idx / (4 * CH), r = idx - chunk * 4 * CH;
        const int left = cnt - chunk * CH, sz = left < CH ? left : CH;
        phase = r / sz;
        tile = base + chunk * CH + (r - phase * sz);
    } else {
        tile = base + (idx >> 2); phase = idx & 3;
    }
    return tile < T;
}

template <typename T, int ABL, bool PRIO, bool NARROW = false, int TAPS = 9, int MX = 0, int FLAT = 0, int GNF = 0, bool OUT6 = false>
__global__ __launch_bounds__(256, 2) void igemm_halo_kernel(const omgsr_igemm_args p, const IgemmGeo g) {
    int tile, phase;
    // Phase form (TAPS = 4). The four phases of a tile read the SAME low-res patch and each its own phase-summed weights. Three block orders:
    //  0  blockIdx.y = phase: phase 0 sweeps every tile, then phase 1 ...: one phase's weights stay hot in an XCD's L2, but the input comes
    //     back after a whole sweep (1.5 GB at the decoder's last upsampler: beyond L2 + MALL, i.e. from DRAM four times)
    //  1  phases of a tile adjacent (logical block 4 tile + phase): the patch is shared in L2, but the weights of all four phases stream
    //     through every L2 at once (measured: FETCH_SIZE +35 % on the decoder's upsamplers, x2 on the UNet's whose weights are 26-105 MB)
    //  2  (default) CHUNKED: per XCD, phase 0 of 64 tiles (one round of its workgroups), then phase 1 of the same 64 tiles, ...: one
    //     phase's weights hot at a time AND the input chunk (a few MB) re-read one tile-time later, out of L2 / MALL
    if (TAPS == 4 && g.interleave) {
        if (!phase_block_map((int)blockIdx.x, g.ntm * g.ntn, 0, g.interleave == 2, tile, phase)) return;
    } else {
        tile = xcd_remap((int)blockIdx.x, g.ntm * g.ntn); phase = (int)blockIdx.y;
    }
    halo_body<T, ABL, PRIO, NARROW, TAPS, MX, FLAT, GNF, false, OUT6>(p, g, tile, phase);
}

// Several problems that share weights and epilogue options in ONE launch (the tiled VAE runs every layer once per tile-shape group:
// corner / edge / interior tiles are separate dense tensors): the workgroup looks its problem up in a prefix table of tile counts
// (each problem's range starts on a multiple of 8 blocks so the XCD remap of its local index keeps its meaning; the filler blocks
// exit). Small groups no longer pay their own partial last round of workgroups, and a layer is one launch instead of 3-4.
constexpr int HALO_MULTI_MAX = 8;
struct HaloMulti {
    omgsr_igemm_args p[HALO_MULTI_MAX];
    IgemmGeo g[HALO_MULTI_MAX];
    int start[HALO_MULTI_MAX + 1];
    int count;
};
template <typename T, bool NARROW, int TAPS, int MX = 0, int FLAT = 0, int GNF = 0, bool SPLITK = false, bool OUT6 = false>
__global__ __launch_bounds__(256, 2) void igemm_halo_multi_kernel(const HaloMulti m) {
    int s = 0, tile, phase;
    if (TAPS == 4 && m.g[0].interleave) {      // x-only grid of 4 x the 8-aligned ranges; block order inside a problem's range: see igemm_halo_kernel
        while (s + 1 < m.count && (int)blockIdx.x >= 4 * m.start[s + 1]) ++s;      // wave-uniform
        if (!phase_block_map((int)blockIdx.x - 4 * m.start[s], m.g[s].ntm * m.g[s].ntn, m.start[s + 1] - m.start[s], m.g[0].interleave == 2, tile, phase)) return;
    } else {
        while (s + 1 < m.count && (int)blockIdx.x >= m.start[s + 1]) ++s;          // wave-uniform
        const int bid = (int)blockIdx.x - m.start[s];
        if (bid >= m.g[s].ntm * m.g[s].ntn) return;                                 // filler block of the 8-aligned range
        tile = xcd_remap(bid, m.g[s].ntm * m.g[s].ntn); phase = (int)blockIdx.y;
    }
    halo_body<T, 0, false, NARROW, TAPS, MX, FLAT, GNF, SPLITK, OUT6>(m.p[s], m.g[s], tile, phase);
}


// FLAT form of the nine-tap kernel (see halo_body): pitch P = W + 2 when the map is narrow enough for the patch (256 + 2 P + 2 <= 352 positions:
// the same 22 LDS-DMA pieces). halo_flat_eligible: what the form can run at all (the epilogue maps positions to pixels on its 16-byte-row path
// only). halo_flat_pitch: the decision - for a problem on its own, when walking the flattened padded map takes fewer workgroup tiles than the
// 8 x 32 spatial grid; for a problem of a launch group (omgsr_igemm_multi_plan), what the plan decided for the WHOLE group (group_tiles < 0 =
// every problem of the group in the FLAT form: the tile-shape groups of a tiled-VAE level are 40 x 40, 40 x 32, 32 x 40, 32 x 32 - the 32-wide
// ones gain nothing on their own, but a launch runs one form). OMGSR_HALO_FLAT=0 switches the form off (A/B runs).
static inline int halo_flat_eligible(const omgsr_igemm_args& a) {
    static const char* off = getenv("OMGSR_HALO_FLAT");
    if (off && off[0] == '0') return 0;
    if (a.gn_scale_shift) return 0;                 // the normalising patch producer exists in the spatial form only (omgsr_igemm_gn_fusable said so)
    if (a.out_mx == 6) return 0;                    // ... and so do the instantiations whose epilogue writes the fp6 operand form (igemm_halo_out6.hip)
    static const char* mw = getenv("OMGSR_HALO_FLAT_MAXW");        // A/B runs: 45 = only the 22-piece patch
    static const int maxw = mw ? atoi(mw) : 80;                   // 256 + 2 (W + 2) + 2 <= 432 patch rows (27 pieces)
    const int logical_cols = (a.act == OMGSR_ACT_GEGLU) ? 2 * a.Cout : a.Cout;
    const int64_t ldo = a.out_ld > 0 ? a.out_ld : a.Cout;
    if (a.R != 3 || a.S != 3 || a.stride != 1 || a.upsample || a.Wo != a.W || a.Ho != a.H || a.Wo > maxw || a.Wo < 4 || logical_cols < 96 || (a.Cout & 7) || (ldo & 7) ||
        a.out_layout != OMGSR_LAYOUT_NHWC || a.act == OMGSR_ACT_GEGLU) return 0;
    return a.Wo + 2;
}
static inline int halo_flat_tiles_per_image(const omgsr_igemm_args& a) { return (a.Ho * (a.Wo + 2) + TH * TW - 1) / (TH * TW); }
static inline int halo_grid_tiles_per_image(const omgsr_igemm_args& a) { return ((a.Wo + TW - 1) / TW) * ((a.Ho + TH - 1) / TH); }
static inline int halo_flat_pitch(const omgsr_igemm_args& a) {
    const int P = halo_flat_eligible(a);
    if (!P) return 0;
    if (a.group_tiles < 0) return P;               // the plan put the whole launch group on the FLAT form
    if (a.group_tiles > 0) return 0;               // ... or on the spatial form
    return halo_flat_tiles_per_image(a) < halo_grid_tiles_per_image(a) ? P : 0;
}

// tile grid of one problem (the host side of both translation units: igemm_halo.hip launches one problem per grid, igemm_halo_multi.hip
// several problems / the mixed-precision form)
static inline bool halo_geo(const omgsr_igemm_args& a, IgemmGeo& g, const bool phase) {
    const int logical_cols = (a.act == OMGSR_ACT_GEGLU) ? 2 * a.Cout : a.Cout;
    static const char* mp = getenv("OMGSR_HALO_MAINPRIO");         // A/B runs: "1" | "2" = waves raise their priority for the K loop
    g.main_prio = mp ? atoi(mp) : 0;
    static const char* il = getenv("OMGSR_PHASE_INTERLEAVE");      // A/B runs: "0" = blockIdx.y = phase, "1" = phases adjacent, "2" = chunked (default)
    g.interleave = phase ? (il ? atoi(il) : 2) : 0;
    g.nk = a.Cin / 32;
    g.tiles_x = ((phase ? a.W : a.Wo) + TW - 1) / TW;          // phase form: tiles of the LOW-res map, four phases each
    g.tiles_y = ((phase ? a.H : a.Ho) + TH - 1) / TH;
    g.flat = phase ? 0 : halo_flat_pitch(a);
    if (g.flat) { g.tiles_x = (a.Ho * g.flat + TH * TW - 1) / (TH * TW); g.tiles_y = 1; }
    g.ntm = a.N * g.tiles_x * g.tiles_y;
    const bool narrow = logical_cols <= 32 && !phase;
    g.ntn = narrow ? 1 : (logical_cols + BN - 1) / BN;
    return narrow;
}
static inline int halo_set_lds_attr(const void* const* fns, const int n, const int bytes = LDS_BYTES) {
    hipError_t e = hipSuccess;
    for (int i = 0; i < n; ++i)
        if (e == hipSuccess) e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return (int)e;
}

}  // namespace
