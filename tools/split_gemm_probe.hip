// Can the fp32 GEMMs of the MLP kernels leave the fp32 MFMA?  On gfx950 v_mfma_f32_32x32x2_f32 runs at the packed-fp32 VALU rate
// (and shares the SIMD's time with the VALU: coexec_pad_probe.hip); v_mfma_f32_32x32x16_bf16 is 16 times faster per FLOP.
// An fp32 value is EXACTLY the sum of three bf16 values (its 24-bit significand cut into three 8-bit pieces by truncation:
// x = a0 + a1 + a2), so  a.b = a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0) + [a1b2 + a2b1 + a2b2: <= 3 x 2^-24 |ab|, dropped]:
// six bf16 MFMA products, each exact in the fp32 accumulator (8 x 8-bit significands).
// This probe: one wave computes C[32 x 32] = A[32 x K] . B[K x 32], K = 384, (1) on the fp32 MFMA, (2) as six bf16 products with
// the operands split on the fly, (3) the same with operands split beforehand (registers), and reports the clocks of each and the
// largest error of (1) and (2) against an fp64 host reference.
//   hipcc --offload-arch=gfx950 -O2 tools/split_gemm_probe.hip -o build/split_gemm_probe && ./build/split_gemm_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define K 384
struct Split { bf16x8 p0, p1, p2; };
// eight fp32 -> three bf16x8 planes by truncation (exact: x = p0 + p1 + p2)
__device__ __forceinline__ Split split8(const float (&x)[8]) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const unsigned u = __float_as_uint(x[i]);
        h[i] = u & 0xffff0000u;
        const float r1 = x[i] - __uint_as_float(h[i]);
        m[i] = __float_as_uint(r1) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(m[i]);
        l[i] = __float_as_uint(r2) & 0xffff0000u;
    }
    u32x4 a, b, c;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        a[i] = (h[2 * i] >> 16) | h[2 * i + 1];
        b[i] = (m[2 * i] >> 16) | m[2 * i + 1];
        c[i] = (l[2 * i] >> 16) | l[2 * i + 1];
    }
    Split s;
    s.p0 = __builtin_bit_cast(bf16x8, a); s.p1 = __builtin_bit_cast(bf16x8, b); s.p2 = __builtin_bit_cast(bf16x8, c);
    return s;
}
// A row-major [32][K], B as [col 32][K] (both k-contiguous per lane)
__global__ __launch_bounds__(64) void k_probe(const float *__restrict__ A, const float *__restrict__ B, float *C32, float *Csp,
                                              unsigned long long *t, int reps) {
    const int lane = threadIdx.x, rc = lane & 31, g = lane >> 5;
    // (1) fp32 MFMA: lane holds A[rc][2s + g], B[2s + g][rc]
    f32x16 acc = {0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
        f32x16 c = {0};
        for (int s = 0; s < K / 2; s++) c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[rc * K + 2 * s + g], B[rc * K + 2 * s + g], c, 0, 0, 0);
        acc = c;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 16; i++) C32[((i & 3) + 8 * (i >> 2) + 4 * g) * 32 + rc] = acc[i];
    // (2) six bf16 products, split on the fly: lane holds k = 16 b + 8 g .. + 7 of its row / column
    f32x16 hi = {0}, lo = {0};
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
        f32x16 c0 = {0}, c1 = {0};
        for (int b = 0; b < K / 16; b++) {
            float xa[8], xb[8];
#pragma unroll
            for (int i = 0; i < 8; i++) { xa[i] = A[rc * K + 16 * b + 8 * g + i]; xb[i] = B[rc * K + 16 * b + 8 * g + i]; }
            const Split a = split8(xa), w = split8(xb);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p0, w.p2, c1, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p1, w.p1, c1, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p2, w.p0, c1, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p0, w.p1, c1, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p1, w.p0, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p0, w.p0, c0, 0, 0, 0);
        }
        hi = c0; lo = c1;
    }
    unsigned long long t3 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 16; i++) Csp[((i & 3) + 8 * (i >> 2) + 4 * g) * 32 + rc] = hi[i] + lo[i];
    // (3) the same six products with the operands split beforehand (the first 16-k block's planes reused: timing only)
    float xa[8], xb[8];
    for (int i = 0; i < 8; i++) { xa[i] = A[rc * K + 8 * g + i]; xb[i] = B[rc * K + 8 * g + i]; }
    const Split a = split8(xa), w = split8(xb);
    f32x16 d0 = {0}, d1 = {0};
    unsigned long long t4 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++)
        for (int b = 0; b < K / 16; b++) {
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p0, w.p2, d1, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p1, w.p1, d1, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p2, w.p0, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p0, w.p1, d0, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p1, w.p0, d0, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p0, w.p0, d0, 0, 0, 0);
        }
    unsigned long long t5 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { t[0] = t1 - t0; t[1] = t3 - t2; t[2] = t5 - t4; }
    if (d0[0] + d1[1] == 12345.678f) Csp[0] = d0[0];
}
int main() {
    std::vector<float> hA(32 * K), hB(32 * K);
    srand(7);
    auto rnd = [] { return (float) rand() / RAND_MAX * 2.f - 1.f; };
    for (auto &v : hA) v = rnd() * expf(rnd() * 3.f);   // mixed magnitudes
    for (auto &v : hB) v = rnd() * expf(rnd() * 3.f);
    float *dA, *dB, *dC, *dS; unsigned long long *dt;
    if (hipMalloc(&dA, hA.size() * 4) || hipMalloc(&dB, hB.size() * 4) || hipMalloc(&dC, 4096) || hipMalloc(&dS, 4096) || hipMalloc(&dt, 64)) return 1;
    (void) hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    (void) hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    const int reps = 200;
    for (int trial = 0; trial < 2; trial++) hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dS, dt, reps);
    std::vector<float> c32(1024), csp(1024);
    unsigned long long ht[3];
    if (hipMemcpy(c32.data(), dC, 4096, hipMemcpyDeviceToHost) || hipMemcpy(csp.data(), dS, 4096, hipMemcpyDeviceToHost) || hipMemcpy(ht, dt, 24, hipMemcpyDeviceToHost)) return 1;
    double e32 = 0, esp = 0, scale = 0;
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) {
            double ref = 0, mag = 0;
            for (int k = 0; k < K; k++) { ref += (double) hA[i * K + k] * hB[j * K + k]; mag += fabs((double) hA[i * K + k] * hB[j * K + k]); }
            e32 = fmax(e32, fabs(c32[i * 32 + j] - ref) / mag); esp = fmax(esp, fabs(csp[i * 32 + j] - ref) / mag);
            scale = fmax(scale, mag);
        }
    printf("C[32x32] = A[32x%d] B[%dx32], one wavefront, %d repetitions, clocks per repetition:\n", K, K, reps);
    printf("  fp32 MFMA (v_mfma_f32_32x32x2_f32 x %d)                         %8.0f\n", K / 2, (double) ht[0] / reps);
    printf("  six bf16 products per 16-k block, operands split on the fly     %8.0f\n", (double) ht[1] / reps);
    printf("  six bf16 products per 16-k block, operands split beforehand     %8.0f\n", (double) ht[2] / reps);
    printf("largest |C - C_fp64| / sum_k |a_k b_k| over the tile:  fp32 MFMA %.3g,  six bf16 products %.3g   (2^-24 = %.3g)\n", e32, esp, ldexp(1.0, -24));
    return 0;
}
