// Do fp32 MFMA (v_mfma_f32_32x32x2_f32) and fp32 VALU work of ANOTHER wave on the same SIMD overlap?
// One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run a stream of MFMAs, waves 4-7 (their SIMD partners) a stream of
// independent v_fma_f32.  Times: MFMA waves alone, VALU waves alone, both.  Also the bf16 MFMA for comparison.
//   hipcc --offload-arch=gfx950 -O2 tools/coexec_probe.hip -o tools/coexec_probe && ./tools/coexec_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int KIND>  // 0: f32 MFMA, 1: bf16 MFMA
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *t, int iters, int run_mfma, int run_valu, int prio) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool mf = wave < 4;
    if (prio == 1 && !mf) __builtin_amdgcn_s_setprio(3);  // VALU waves above the MFMA waves
    if (prio == 2 && mf) __builtin_amdgcn_s_setprio(3);   // the reverse
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float res = 0.f;
    if (mf && run_mfma) {
        f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        const float x = lane * 0.001f, y = 1.0f;
        bf16x8 xb, yb;
        for (int i = 0; i < 8; i++) { xb[i] = (__bf16) x; yb[i] = (__bf16) y; }
        for (int i = 0; i < iters; i++) {
            if (KIND == 0) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
            } else {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a3, 0, 0, 0);
            }
        }
        res = a0[0] + a1[1] + a2[2] + a3[3];
    }
    if (!mf && run_valu) {
        float v0 = lane, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
        const float m = 1.0001f, c = 0.5f;
        for (int i = 0; i < iters * 8; i++) {  // 8 independent fma per trip
            v0 = __builtin_fmaf(v0, m, c); v1 = __builtin_fmaf(v1, m, c); v2 = __builtin_fmaf(v2, m, c); v3 = __builtin_fmaf(v3, m, c);
            v4 = __builtin_fmaf(v4, m, c); v5 = __builtin_fmaf(v5, m, c); v6 = __builtin_fmaf(v6, m, c); v7 = __builtin_fmaf(v7, m, c);
        }
        res = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        t[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
        t[(blockIdx.x * 8 + wave) * 2 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
    if (res == 12345.678f) out[0] = res;
}
int main() {
    const int nwg = 256, iters = 2000;
    float *d; unsigned long long *t;
    if (hipMalloc(&d, 4096) != hipSuccess || hipMalloc(&t, nwg * 8 * 2 * 8) != hipSuccess) return 1;
    std::vector<unsigned long long> h(nwg * 16);
    for (int kind = 0; kind < 2; kind++) {
        double r[5][2];
        for (int mode = 0; mode < 5; mode++) {
            const int rm = mode != 1, rv = mode != 0, prio = mode >= 3 ? mode - 2 : 0;
            for (int trial = 0; trial < 2; trial++) {
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(nwg), dim3(512), 0, 0, d, t, iters, rm, rv, prio);
                else hipLaunchKernelGGL(k<1>, dim3(nwg), dim3(512), 0, 0, d, t, iters, rm, rv, prio);
                if (hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
            }
            double sm = 0, sv = 0;
            for (int b = 0; b < nwg; b++)
                for (int w = 0; w < 8; w++) (w < 4 ? sm : sv) += (double) h[(b * 8 + w) * 2];
            r[mode][0] = sm / (nwg * 4); r[mode][1] = sv / (nwg * 4);
        }
        int same = 0;
        for (int w = 0; w < 4; w++) same += ((h[w * 2 + 1] >> 4) & 3) == ((h[(w + 4) * 2 + 1] >> 4) & 3);
        printf("%s: waves w and w + 4 on the same SIMD: %d of 4\n", kind == 0 ? "v_mfma_f32_32x32x2_f32" : "v_mfma_f32_32x32x16_bf16", same);
        printf("  MFMA waves alone  %8.0f cycles (%d MFMAs: %.1f per MFMA)\n", r[0][0], iters * 4, r[0][0] / (iters * 4));
        printf("  VALU waves alone  %8.0f cycles (%d v_fma_f32: %.2f per instruction)\n", r[1][1], iters * 64, r[1][1] / (iters * 64));
        printf("  together          MFMA waves %8.0f, VALU waves %8.0f cycles  (sum of the two alone: %.0f)\n", r[2][0], r[2][1], r[0][0] + r[1][1]);
        printf("  together, s_setprio 3 on the VALU waves: MFMA waves %8.0f, VALU waves %8.0f cycles\n", r[3][0], r[3][1]);
        printf("  together, s_setprio 3 on the MFMA waves: MFMA waves %8.0f, VALU waves %8.0f cycles\n", r[4][0], r[4][1]);
    }
    return 0;
}
