// Where do the four waves of a 256-thread workgroup land?  Prints, for a launch of 2 workgroups per CU (50 KB LDS each), how many
// workgroups have their waves on four different SIMDs and how the co-resident pairs share SIMDs.
//   hipcc --offload-arch=gfx950 -O2 tools/hwid_probe.hip -o tools/hwid_probe && ./tools/hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256) void k(unsigned *out, int spin) {
    extern __shared__ float sm[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
    float v = threadIdx.x;
    for (int i = 0; i < spin; i++) v = v * 1.0001f + 0.5f;
    sm[threadIdx.x] = v;
    if ((threadIdx.x & 63) == 0) {
        unsigned *o = out + ((size_t) blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
        o[0] = hw; o[1] = xcc + (v == 1234.5f);
    }
}
int main(int argc, char **argv) {
    const int nwg = 512, lds = argc > 1 ? atoi(argv[1]) : 50000;
    unsigned *d; hipMalloc(&d, nwg * 4 * 2 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, d, 200000);
    std::vector<unsigned> h(nwg * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int distinct4 = 0; std::map<int, int> hist;
    for (int b = 0; b < nwg; b++) {
        int mask = 0;
        for (int w = 0; w < 4; w++) mask |= 1 << ((h[(b * 4 + w) * 2] >> 4) & 3);
        hist[__builtin_popcount(mask)]++;
        if (__builtin_popcount(mask) == 4) distinct4++;
    }
    for (auto &kv : hist) printf("workgroups whose 4 waves sit on %d different SIMDs: %d\n", kv.first, kv.second);
    for (int b = 0; b < 6; b++) {
        printf("wg %d:", b);
        for (int w = 0; w < 4; w++) {
            const unsigned hw = h[(b * 4 + w) * 2], xc = h[(b * 4 + w) * 2 + 1];
            printf("  [xcc %u se %u cu %u simd %u slot %u]", xc & 15, (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15);
        }
        printf("\n");
    }
    return 0;
}
