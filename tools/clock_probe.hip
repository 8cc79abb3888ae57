// Diagnostic: shader clock under an MFMA-only load and under a VALU-only load (s_memtime ticks / HIP-event time).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k_mfma(int iters, unsigned long long *ticks, float *sink) {
    f32x16 a0 = {0}, a1 = {0};
    float x = threadIdx.x * 1e-3f, y = 1.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[3];
}
__global__ void k_valu(int iters, unsigned long long *ticks, float *sink) {
    float a = threadIdx.x, b = 1.0001f, c = 0.5f, d = 0.25f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) { a = fmaf(a, b, c); d = fmaf(d, b, a); c = fmaf(c, b, d); }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a + c + d;
}
int main() {
    unsigned long long *ticks; float *sink;
    hipMalloc(&ticks, 8); hipMalloc(&sink, 1024 * 256 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kind = 0; kind < 2; kind++)
        for (int blocks : {64, 256, 1024}) {
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, 20000, ticks, sink);
                else hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, 0, 400000, ticks, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                unsigned long long h; hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
                if (rep == 2) printf("%s blocks=%4d  %.3f ms  %llu ticks  -> %.0f MHz (ticks/time of block 0)\n", kind == 0 ? "mfma" : "valu", blocks, ms, h, h / (ms * 1e3));
            }
        }
    return 0;
}
