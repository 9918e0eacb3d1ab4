// Probe (build: hipcc --offload-arch=gfx950 -O2 mx_fp8_probe.hip -o mx_fp8_probe): semantics of the gfx950 block-scaled MFMA
// v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 e4m3 operands and of v_cvt_pk_fp8_f32, as the accurate tier's correction segments
// would use them: lane l holds row (l & 31), 32 consecutive k of block (l >> 5); the E8M0 scale in byte 0 of a per-lane word.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <stdint.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ unsigned pk4(float a, float b, float c, float d) {
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (unsigned)v;
}

__global__ void probe(const float* A, const float* B, float* C, int sa, int sb, unsigned* cvt_out, const float* cvt_in) {
    const int l = threadIdx.x, row = l & 31, kb = l >> 5;
    i32x8 a, b;
    for (int w = 0; w < 8; ++w) {
        const int k = kb * 32 + 4 * w;
        a[w] = (int)pk4(A[row * 64 + k], A[row * 64 + k + 1], A[row * 64 + k + 2], A[row * 64 + k + 3]);
        b[w] = (int)pk4(B[(k) * 32 + row], B[(k + 1) * 32 + row], B[(k + 2) * 32 + row], B[(k + 3) * 32 + row]);   // B[k][col = row]
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    int scale_a = sa, scale_b = sb;
    asm volatile("" : "+v"(scale_a), "+v"(scale_b));          // keep them runtime VGPR values
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
    for (int r = 0; r < 16; ++r) {
        const int crow = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), ccol = l & 31;
        C[crow * 32 + ccol] = c[r];
    }
    if (l < 16) cvt_out[l] = pk4(cvt_in[4 * l], cvt_in[4 * l + 1], cvt_in[4 * l + 2], cvt_in[4 * l + 3]);
}

static float e4m3(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf((float)m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
    if (e == 15 && m == 7) x = NAN;
    return s ? -x : x;
}
static float q8(float x) {      // round to nearest e4m3 by brute force
    float best = 0; float bd = 1e30f;
    for (int v = 0; v < 256; ++v) { float y = e4m3((uint8_t)v); if (isnan(y)) continue; float d = fabsf(y - x); if (d < bd) { bd = d; best = y; } }
    return best;
}
int main() {
    float hA[32 * 64], hB[64 * 32], hC[32 * 32], cin[64];
    srand(1);
    for (int i = 0; i < 32 * 64; ++i) hA[i] = (rand() / (float)RAND_MAX - 0.5f) * 8.f;
    for (int i = 0; i < 64 * 32; ++i) hB[i] = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    const float probes[64] = {0.f, 1.f, -1.f, 1.0625f, 1.1875f, 447.f, 448.f, 449.f, 480.f, 500.f, 1000.f, 1e6f, -1e6f, 0.001953125f, 0.0009765625f, 0.0014f,
                              0.015625f, 0.0175f, 3.3f, 7.7f, 100.f, 240.f, 256.f, 300.f, 0.3f, 0.05f, -0.05f, 17.f, 18.f, 19.f, 20.f, 21.f};
    for (int i = 0; i < 64; ++i) cin[i] = i < 32 ? probes[i] : 0.f;
    float *dA, *dB, *dC, *dcin; unsigned* dcv;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC)); hipMalloc(&dcin, sizeof(cin)); hipMalloc(&dcv, 64);
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice); hipMemcpy(dcin, cin, sizeof(cin), hipMemcpyHostToDevice);
    for (int trial = 0; trial < 3; ++trial) {
        const int sa = trial == 0 ? 127 : trial == 1 ? 127 - 12 : 127 + 3, sb = trial == 2 ? 127 - 6 : 127;
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, sa, sb, dcv, dcin);
        hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
        double maxerr = 0, maxref = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double ref = 0;
            for (int k = 0; k < 64; ++k) ref += (double)q8(hA[i * 64 + k]) * (double)q8(hB[k * 32 + j]);
            ref *= ldexp(1.0, sa - 127) * ldexp(1.0, sb - 127);
            maxerr = fmax(maxerr, fabs(ref - hC[i * 32 + j])); maxref = fmax(maxref, fabs(ref));
        }
        printf("scale_a 2^%d scale_b 2^%d: max |C - ref| = %.3e (max |ref| %.3e) C[0][0] %.6g C[3][5] %.6g\n", sa - 127, sb - 127, maxerr, maxref, hC[0], hC[3 * 32 + 5]);
    }
    unsigned cv[16];
    hipMemcpy(cv, dcv, 64, hipMemcpyDeviceToHost);
    for (int i = 0; i < 32; ++i) {
        uint8_t byte = (cv[i / 4] >> (8 * (i % 4))) & 255;
        printf("cvt %g -> 0x%02x = %g (nearest e4m3 %g)\n", probes[i], byte, e4m3(byte), q8(fminf(fmaxf(probes[i], -448.f), 448.f)));
    }
    return 0;
}
