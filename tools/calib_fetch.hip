// FETCH_SIZE calibration (MI355X_MICROARCH.md, HBM): the counter is calibrated only for 16 B/lane streaming reads
// (reports 1/2 of the bytes).  This reads a 1 GiB buffer once with each access width the dynamics kernel uses, so
// that profiles/ can state the factor for dword reads too.  Run under `rocprofv3 --pmc FETCH_SIZE`.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read_dword(const float *p, size_t n, float *out) {
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) s += p[i];
    if (s == 123.456f) out[0] = s;
}
__global__ void read_dwordx4(const float4 *p, size_t n, float *out) {
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        const float4 v = p[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) out[0] = s;
}
int main() {
    const size_t bytes = (size_t) 1 << 30;
    float *buf, *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    read_dword<<<4096, 256>>>(buf, bytes / 4, out);
    read_dwordx4<<<4096, 256>>>((const float4 *) buf, bytes / 16, out);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("read %zu bytes per kernel\n", bytes);
    return 0;
}
