// Follow-up of coexec_probe.hip: a wave that issues MFMAs back to back starves the VALU work of the wave it shares the SIMD with.
// Does it still, when the MFMA wave steps aside between its MFMAs (s_nop padding that covers the matrix pipe's busy time), so
// that the arbiter is offered no VALU-class instruction by it?  One 512-thread workgroup per CU: waves 0-3 MFMAs (+ PAD x
// `s_nop 15` = 16 idle cycles each after every MFMA), waves 4-7 independent v_fma_f32.
// Finding (profiles/r4s_coexec_pad.txt): yes.  `s_nop n` holds the wave for 4 (n + 1) clocks; with the padding just under the
// MFMA's pipe time the MFMA wave keeps its rate and the VALU wave of the same SIMD runs beside it.
//   hipcc --offload-arch=gfx950 -O2 tools/coexec_pad_probe.hip -o build/coexec_pad_probe && ./build/coexec_pad_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int PAD> __device__ __forceinline__ void pad() {  // PAD = the s_nop operand + 1 (0: none); one unit = 4 clocks
    if (PAD > 0) asm volatile("s_nop %0" ::"n"(PAD > 0 ? PAD - 1 : 0));
}
template <int KIND, int PAD>  // KIND 0: 32x32x2 f32 (64 cycles), 1: 16x16x4 f32 (32 cycles)
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *t, int iters, int run_mfma, int run_valu) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool mf = wave < 4;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float res = 0.f;
    if (mf && run_mfma) {
        const float x = lane * 0.001f, y = 1.0f;
        if (KIND == 0) {
            f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
            for (int i = 0; i < iters; i++) {
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y)); pad<PAD>();
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y)); pad<PAD>();
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(x), "v"(y)); pad<PAD>();
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a3) : "v"(x), "v"(y)); pad<PAD>();
            }
            res = a0[0] + a1[1] + a2[2] + a3[3];
        } else {
            f32x4 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
            for (int i = 0; i < iters; i++) {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y)); pad<PAD>();
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y)); pad<PAD>();
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(x), "v"(y)); pad<PAD>();
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a3) : "v"(x), "v"(y)); pad<PAD>();
            }
            res = a0[0] + a1[1] + a2[2] + a3[3];
        }
    }
    if (!mf && run_valu) {
        float v0 = lane, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
        const float m = 1.0001f, c = 0.5f;
        for (int i = 0; i < iters * 4; i++) {
            v0 = __builtin_fmaf(v0, m, c); v1 = __builtin_fmaf(v1, m, c); v2 = __builtin_fmaf(v2, m, c); v3 = __builtin_fmaf(v3, m, c);
            v4 = __builtin_fmaf(v4, m, c); v5 = __builtin_fmaf(v5, m, c); v6 = __builtin_fmaf(v6, m, c); v7 = __builtin_fmaf(v7, m, c);
        }
        res = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) t[blockIdx.x * 8 + wave] = t1 - t0;
    if (res == 12345.678f) out[0] = res;
}
// the same wave: NV independent v_fma_f32 behind every MFMA (do they issue in the MFMA's shadow?)
template <int KIND, int NV>
__global__ __launch_bounds__(256) void ks(float *out, unsigned long long *t, int iters) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float x = lane * 0.001f, y = 1.0f;
    float v0 = lane, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
    const float m = 1.0001f, c = 0.5f;
    f32x16 a0 = {0}, a1 = {0};
    f32x4 b0 = {0}, b1 = {0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define VFMA(v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(m), "v"(c));
#define SHADOW()                                                             \
    if (NV >= 1) VFMA(v0) if (NV >= 2) VFMA(v1) if (NV >= 3) VFMA(v2) if (NV >= 4) VFMA(v3) \
    if (NV >= 5) VFMA(v4) if (NV >= 6) VFMA(v5) if (NV >= 7) VFMA(v6) if (NV >= 8) VFMA(v7) \
    if (NV >= 9) VFMA(v0) if (NV >= 10) VFMA(v1) if (NV >= 11) VFMA(v2) if (NV >= 12) VFMA(v3) \
    if (NV >= 13) VFMA(v4) if (NV >= 14) VFMA(v5) if (NV >= 15) VFMA(v6) if (NV >= 16) VFMA(v7)
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y)); SHADOW()
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y)); SHADOW()
        } else {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(b0) : "v"(x), "v"(y)); SHADOW()
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(b1) : "v"(x), "v"(y)); SHADOW()
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) t[blockIdx.x * 4 + wave] = t1 - t0;
    const float res = a0[0] + a1[1] + b0[0] + b1[1] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    if (res == 12345.678f) out[0] = res;
}
template <int KIND, int NV> static void run_shadow(float *d, unsigned long long *t, std::vector<unsigned long long> &h) {
    const int nwg = 256, iters = 4000;
    for (int trial = 0; trial < 2; trial++) {
        hipLaunchKernelGGL((ks<KIND, NV>), dim3(nwg), dim3(256), 0, 0, d, t, iters);
        if (hipMemcpy(h.data(), t, nwg * 4 * 8, hipMemcpyDeviceToHost) != hipSuccess) exit(1);
    }
    double sm = 0;
    for (int i = 0; i < nwg * 4; i++) sm += (double) h[i];
    printf("%s same wave, %2d v_fma_f32 behind each MFMA: %.1f clocks per MFMA\n", KIND == 0 ? "32x32x2" : "16x16x4", NV, sm / (nwg * 4) / (2.0 * iters));
}
template <int KIND, int PAD> static void run(float *d, unsigned long long *t, std::vector<unsigned long long> &h) {
    const int nwg = 256, iters = 2000;
    double r[3][2];
    for (int mode = 0; mode < 3; mode++) {
        const int rm = mode != 1, rv = mode != 0;
        for (int trial = 0; trial < 2; trial++) {
            hipLaunchKernelGGL((k<KIND, PAD>), dim3(nwg), dim3(512), 0, 0, d, t, iters, rm, rv);
            if (hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) exit(1);
        }
        double sm = 0, sv = 0;
        for (int b = 0; b < nwg; b++)
            for (int w = 0; w < 8; w++) (w < 4 ? sm : sv) += (double) h[b * 8 + w];
        r[mode][0] = sm / (nwg * 4); r[mode][1] = sv / (nwg * 4);
    }
    printf("%s s_nop x %2d: MFMA alone %7.0f (%.1f per MFMA) | VALU alone %7.0f | together: MFMA %7.0f VALU %7.0f (serial would be %.0f)\n",
           KIND == 0 ? "32x32x2" : "16x16x4", PAD, r[0][0], r[0][0] / 8000, r[1][1], r[2][0], r[2][1], r[0][0] + r[1][1]);
}
int main() {
    float *d; unsigned long long *t;
    if (hipMalloc(&d, 4096) != hipSuccess || hipMalloc(&t, 256 * 8 * 8) != hipSuccess) return 1;
    std::vector<unsigned long long> h(256 * 8);
    run<0, 0>(d, t, h); run<0, 8>(d, t, h); run<0, 11>(d, t, h); run<0, 12>(d, t, h); run<0, 13>(d, t, h); run<0, 14>(d, t, h);
    run<0, 15>(d, t, h); run<0, 16>(d, t, h);
    run<1, 0>(d, t, h); run<1, 3>(d, t, h); run<1, 4>(d, t, h); run<1, 5>(d, t, h); run<1, 6>(d, t, h); run<1, 7>(d, t, h); run<1, 8>(d, t, h);
    run_shadow<0, 0>(d, t, h); run_shadow<0, 4>(d, t, h); run_shadow<0, 8>(d, t, h); run_shadow<0, 12>(d, t, h); run_shadow<0, 16>(d, t, h);
    run_shadow<1, 0>(d, t, h); run_shadow<1, 2>(d, t, h); run_shadow<1, 4>(d, t, h); run_shadow<1, 6>(d, t, h); run_shadow<1, 8>(d, t, h);
    return 0;
}
