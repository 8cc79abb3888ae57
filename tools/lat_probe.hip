// Dependent-chain latency of scalar vs packed fp32 FMA on one wave per SIMD (the regime of the lane-group sweeps kernel).
//   hipcc --offload-arch=gfx950 -O3 tools/lat_probe.hip -o build/lat_probe && build/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float P2 __attribute__((ext_vector_type(2)));
#define N 4096
__global__ void k_probe(float *out, unsigned long long *cyc, float a, float b) {
    float x = a + threadIdx.x * 1e-9f, y = b;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 64
    for (int i = 0; i < N; i++) x = __builtin_fmaf(x, y, a);              // dependent scalar fma chain
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    P2 p = {x, b + 1e-9f}, q = {y, y}, r = {a, a};
#pragma unroll 64
    for (int i = 0; i < N; i++) p = __builtin_elementwise_fma(p, q, r);    // dependent packed fma chain
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    // two independent packed chains interleaved (ILP 2)
    P2 p1 = p, p2 = {p.y, p.x};
#pragma unroll 32
    for (int i = 0; i < N; i++) { p1 = __builtin_elementwise_fma(p1, q, r); p2 = __builtin_elementwise_fma(p2, q, r); }
    unsigned long long t3 = __builtin_amdgcn_s_memtime();
    float s1 = x, s2 = y + x;
#pragma unroll 32
    for (int i = 0; i < N; i++) { s1 = __builtin_fmaf(s1, y, a); s2 = __builtin_fmaf(s2, y, a); }
    unsigned long long t4 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x + p.x + p.y + p1.x + p2.y + s1 + s2;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; }
}
int main() {
    float *out; unsigned long long *cyc, h[4];
    hipMalloc(&out, 64 * 4); hipMalloc(&cyc, 32);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, out, cyc, 0.5f, 0.999f);
    hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    printf("dependent v_fma_f32      : %.2f cycles per instruction\n", (double) h[0] / N);
    printf("dependent v_pk_fma_f32   : %.2f cycles per instruction\n", (double) h[1] / N);
    printf("2 interleaved pk chains  : %.2f cycles per instruction\n", (double) h[2] / (2.0 * N));
    printf("2 interleaved fma chains : %.2f cycles per instruction\n", (double) h[3] / (2.0 * N));
    return 0;
}
