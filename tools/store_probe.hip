// What does a wave pay to issue global stores?  Each wave writes the same bytes either as dword stores in the 32x32 MFMA
// accumulator layout (2 rows x 128 B per instruction) or as dwordx4 stores (1 KiB contiguous per instruction); 2 workgroups
// of 4 waves per CU, every CU storing.  Prints cycles per store instruction and per KiB, seen by the issuing wave.
//   hipcc --offload-arch=gfx950 -O2 tools/store_probe.hip -o tools/store_probe && ./tools/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *t, int reps) {
    extern __shared__ float sm[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *base = out + (size_t) blockIdx.x * reps * 32 * 256;  // reps tiles of 32 rows x 256 floats
    sm[threadIdx.x] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < reps; it++) {
        float *tile = base + (size_t) it * 32 * 256;
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), c = wave * 64 + j * 32 + (lane & 31);
                    tile[row * 256 + c] = (float) (it + r);
                }
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int q = (i * 4 + wave) * 64 + lane;  // float4 index inside the tile
                reinterpret_cast<f32x4 *>(tile)[q] = f32x4{(float) it, 1.f, 2.f, (float) i};
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { t[(blockIdx.x * 4 + wave) * 2] = t1 - t0; t[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0; }
}
int main() {
    const int nwg = 512, reps = 16;
    float *d; unsigned long long *t;
    if (hipMalloc(&d, (size_t) nwg * reps * 32 * 256 * 4) != hipSuccess || hipMalloc(&t, nwg * 8 * 8) != hipSuccess) return 1;
    std::vector<unsigned long long> h(nwg * 8);
    for (int mode = 0; mode < 2; mode++)
        for (int trial = 0; trial < 2; trial++) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nwg), dim3(256), 50000, 0, d, t, reps);
            else hipLaunchKernelGGL(k<1>, dim3(nwg), dim3(256), 50000, 0, d, t, reps);
            if (hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
            double issue = 0, done = 0;
            for (int i = 0; i < nwg * 4; i++) { issue += h[2 * i]; done += h[2 * i + 1]; }
            issue /= nwg * 4; done /= nwg * 4;
            const int n_instr = reps * (mode == 0 ? 32 : 8);
            printf("%s: issue %.0f cycles (%.1f per instruction, %.1f per KiB), drained after %.0f cycles; %d KiB per wave\n",
                   mode == 0 ? "dword, accumulator layout" : "dwordx4, contiguous     ", issue, issue / n_instr, issue / (reps * 8.0), done, reps * 8);
        }
    return 0;
}
