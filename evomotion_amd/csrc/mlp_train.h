// Device pieces shared by the training kernels (ppo_kernels.hip: actor / critic of PPO; q_kernels.hip: the twin Q
// networks of SAC): tile moves between HBM and the k-split LDS layout, the forward epilogue that keeps activations,
// the LayerNorm / Mish backward of a row, block reductions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mlp_tile.h"

namespace evm {

#define PRT 1  // 32-row MFMA tiles per wave in the training kernels (TM = 32 rows per workgroup)

// ---------------------------------------------------------------------------------------------------------
// tile helpers (k-split activation tile [TM][ALD2] <-> row-major [rows][256] in HBM)
// ---------------------------------------------------------------------------------------------------------
// Thread t moves the column pair (2k, 2k + 1), k = t & 127, of the rows r = (t >> 7), + 2, ...: 8 contiguous bytes per lane
// in HBM (512 B per wave instruction) and the two k-split halves [k], [128 + k] in LDS (conflict free).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int TM_>
__device__ __forceinline__ void tile_load(float *T, const float *__restrict__ src, int row0, int n) {
    const int k = threadIdx.x & 127, r0 = threadIdx.x >> 7;
    f32x2 v[TM_ / 2];
#pragma unroll
    for (int j = 0; j < TM_ / 2; j++) {
        const int gr = row0 + r0 + 2 * j;
        v[j] = *reinterpret_cast<const f32x2 *>(src + (size_t) (gr < n ? gr : n - 1) * 256 + 2 * k);
        if (gr >= n) v[j] = f32x2{0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < TM_ / 2; j++) {
        float *d = T + (r0 + 2 * j) * ALD2 + k;
        d[0] = v[j][0];
        d[128] = v[j][1];
    }
}
template <int TM_>
__device__ __forceinline__ void tile_store(const float *T, float *__restrict__ dst, int row0, int n) {
    const int k = threadIdx.x & 127, r0 = threadIdx.x >> 7;
#pragma unroll 8
    for (int j = 0; j < TM_ / 2; j++) {
        const int r = r0 + 2 * j;
        const float *sp = T + r * ALD2 + k;
        const f32x2 v = {sp[0], sp[128]};
        if (row0 + r < n) *reinterpret_cast<f32x2 *>(dst + (size_t) (row0 + r) * 256 + 2 * k) = v;
    }
}
// thread t sums the stored position t of every row: column QCOL(t)
template <int TM_>
__device__ __forceinline__ float tile_colsum(const float *T) {
    const int q = threadIdx.x;
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < TM_; r++) s += T[r * ALD2 + q];
    return s;
}
// MFMA accumulators (C layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) -> tile, plus a per-column bias
template <int RT>
__device__ __forceinline__ void acc_to_tile(const f32x16 (&acc)[RT][2], const float *__restrict__ bias, float *T, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < RT; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int c = wave * 64 + j * 32 + (lane & 31);
            const float b = bias ? bias[c] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                T[HIDX(row, c)] = acc[i][j][r] + b;
            }
        }
}

// ---------------------------------------------------------------------------------------------------------
// forward with the activations kept
// ---------------------------------------------------------------------------------------------------------
// TM x K1 tile of the padded observations [n][K1] (16-byte aligned rows) -> k-split LDS tile (row stride ALD1)
template <int TM_>
__device__ __forceinline__ void stage_padded_ksplit(float *xs, const float *__restrict__ xp, int row0, int n) {
    constexpr int NIT = TM_ * K1 / 4 / PT;
    static_assert(TM_ * K1 / 4 % PT == 0, "tile must divide among the threads");
    f32x4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int q = it * PT + (int) threadIdx.x, r = q / (K1 / 4), c4 = (q - r * (K1 / 4)) * 4;
        const int gr = row0 + r;
        v[it] = *reinterpret_cast<const f32x4 *>(xp + (size_t) (gr < n ? gr : n - 1) * K1 + c4);
        if (gr >= n) v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int q = it * PT + (int) threadIdx.x, r = q / (K1 / 4), c4 = (q - r * (K1 / 4)) * 4;
        // k = c4 .. c4 + 3: even k -> first half, odd k -> second half of the k-split row
        float *dst = xs + r * ALD1 + (c4 >> 1);
        dst[0] = v[it][0]; dst[K1 / 2] = v[it][1]; dst[1] = v[it][2]; dst[K1 / 2 + 1] = v[it][3];
    }
}

// forward epilogue of a layer with the activations kept: mish_ln_epilogue (mlp_tile.h) with z = acc + bias, the LayerNorm
// output and (mean, rstd) -> st[row * st_stride + st_off ..] written to HBM; zg / ag / st may be NULL (a forward that keeps
// nothing).  red: EVM_RED_FLOATS of LDS outside the tiles.
template <int RT>
__device__ __forceinline__ void train_epilogue(f32x16 (&acc)[RT][2], const float *__restrict__ bias,
                                               const float *__restrict__ gamma, const float *__restrict__ beta, float *hb, float *red,
                                               int wave, int lane, int row0, int n, float *zg, float *ag, float *st, int st_off, int st_stride) {
    static_assert(RT == 1, "one 32-row MFMA tile per wave");
    mish_ln_epilogue(acc[0], bias, gamma, beta, hb, red, wave, lane, row0, n, zg, ag, st, st_off, st_stride);
}


__device__ __forceinline__ double block_sum_double(double v, double *sh) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < nw; i++) s += sh[i];
    return s;
}

// one float per thread -> one double atomicAdd per workgroup
__device__ __forceinline__ void block_accumulate(float v, double *dst, float *sh) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < nw; i++) s += (double) sh[i];
        if (s != 0.0) atomicAdd(dst, s);
    }
}


// one float per thread -> this workgroup's partial sum (double), summed later in a fixed order (block_partial_sum).  Thousands of
// workgroups adding doubles to ONE address serialise in L2: the PPO actor loss kernel spent 60 of its 82 us there.
__device__ __forceinline__ void block_partial(float v, double *part, float *sh) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < nw; i++) s += (double) sh[i];
        part[blockIdx.x] = s;
    }
}

// LayerNorm + Mish backward of a 32-row tile in the MFMA accumulator layout (the counterpart of mish_ln_epilogue, mlp_tile.h):
//   g[j][r]: in d loss / d a (LayerNorm output), out d loss / d z, for column wave * 64 + 32 j + (lane & 31) and row
//   (r & 3) + 8 (r >> 2) + 4 (lane >> 5); in: the layer's pre-activations and LayerNorm statistics in the same layout (ln_bwd_load:
//   half a wave reads 32 consecutive columns of a row).
//   cs[k][j]: this lane's column sums over its 16 rows of  k = 0: g * xhat (dgamma), 1: g (dbeta), 2: dz (dbias);
//   the caller adds the two halves of the wave.
// The two row means of the LayerNorm backward are reduced like the forward statistics: DPP inside the 32-lane halves, the four
// waves' partial sums through red (EVM_RED_FLOATS of LDS), one LDS-only barrier.
// operands of ln_mish_backward_c, requested early (the caller puts a GEMM between this and their use): the layer's
// pre-activations and LayerNorm statistics of the lane's 2 x 16 elements / 16 rows, in the accumulator layout
struct LnBwdIn {
    f32x16 z[2];
    float mean[16], rstd[16];
};
template <bool FULL>  // FULL: the tile has all 32 rows (row0 + 32 <= n): no per-row guards
__device__ __forceinline__ void ln_bwd_load(LnBwdIn &in, const float *__restrict__ zsrc, const float *__restrict__ st, int st_off, int st_stride,
                                            int wave, int lane, int row0, int n) {
    const int cl = lane & 31, hf = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hf;
        const bool ok = FULL || row0 + row < n;
        const unsigned gr = (unsigned) (ok ? row0 + row : n - 1);  // byte offsets fit 32 bits: rows * 1 KiB < 4 GiB
        const unsigned so = (gr * (unsigned) st_stride + (unsigned) st_off) * 4u, zo = (gr * 256u + (unsigned) (wave * 64 + cl)) * 4u;
        const float m = ldg_off(st, so), rs = ldg_off(st, so + 4u), z0 = ldg_off(zsrc, zo), z1 = ldg_off(zsrc, zo + 128u);
        in.mean[r] = ok ? m : 0.f;
        in.rstd[r] = ok ? rs : 0.f;
        in.z[0][r] = ok ? z0 : 0.f;
        in.z[1][r] = ok ? z1 : 0.f;
    }
}
__device__ __forceinline__ void ln_mish_backward_c(f32x16 (&g)[2], const LnBwdIn &in, const float *__restrict__ gamma, float *red, int wave,
                                                   int lane, float (&cs)[3][2]) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    const int cl = lane & 31, hf = lane >> 5;
    const int c0 = wave * 64 + cl, c1 = c0 + 32;
    const float gam[2] = {gamma[c0], gamma[c1]};
    f32x16 xh[2], mp[2];
#pragma unroll
    for (int k = 0; k < 3; k++) cs[k][0] = cs[k][1] = 0.f;
    // Registers: g, xhat and Mish' (96) live across the barrier; the row sums go to LDS as soon as they are reduced.
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hf;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const float zz = in.z[j][r];
            const float nn = __expf(fminf(zz, 20.f));
            const float mm = nn * (nn + 2.f);
            const float tt = mm * __builtin_amdgcn_rcpf(mm + 2.f);  // tanh(softplus(z))
            float mpv = tt + zz * (1.f - tt * tt) * (nn * __builtin_amdgcn_rcpf(1.f + nn));  // Mish'(z) = tanh(sp) + z sigmoid(z) (1 - tanh(sp)^2)
            float xhv = (zz * tt - in.mean[r]) * in.rstd[r];
            // pin both here: their only other use is behind the barrier, and the compiler otherwise sinks the whole exp / rcp
            // chain down there, keeping z, e^z and the statistics of all 32 elements alive instead (440 registers)
            asm volatile("" : "+v"(mpv), "+v"(xhv));
            mp[j][r] = mpv;
            xh[j][r] = xhv;
            cs[0][j] += g[j][r] * xh[j][r];
            cs[1][j] += g[j][r];
            const float gy = g[j][r] * gam[j];
            g[j][r] = gy;
            s1 += gy;
            s2 += gy * xh[j][r];
        }
        s1 = half_wave_sum(s1);
        s2 = half_wave_sum(s2);
        if (cl == 0) *reinterpret_cast<f32x2_ *>(red + (row * 4 + wave) * 2) = f32x2_{s1, s2};
    }
    lds_barrier();
    float *mine = red + 256 + wave * 64;  // this wave's [32 rows][mean of gy, mean of gy * xhat]
    {
        const f32x4 p0 = *reinterpret_cast<const f32x4 *>(red + cl * 8), p1 = *reinterpret_cast<const f32x4 *>(red + cl * 8 + 4);
        const float m1 = ((p0[0] + p0[2]) + (p1[0] + p1[2])) * (1.f / 256.f), m2 = ((p0[1] + p0[3]) + (p1[1] + p1[3])) * (1.f / 256.f);
        if (hf == 0) *reinterpret_cast<f32x2_ *>(mine + cl * 2) = f32x2_{m1, m2};
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const f32x4 ma = *reinterpret_cast<const f32x4 *>(mine + (8 * q + 4 * hf) * 2), mb = *reinterpret_cast<const f32x4 *>(mine + (8 * q + 4 * hf) * 2 + 4);
        const float m1[4] = {ma[0], ma[2], mb[0], mb[2]}, m2[4] = {ma[1], ma[3], mb[1], mb[3]};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int r = 4 * q + u;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const float dz = in.rstd[r] * (g[j][r] - m1[u] - xh[j][r] * m2[u]) * mp[j][r];
                g[j][r] = dz;
                cs[2][j] += dz;
            }
        }
    }
}

// thread (row, part): gradient w.r.t. the layer's pre-activation z from the gradient w.r.t. its LayerNorm output.
//   g[i]: in d loss / d a (LayerNorm output) of the thread's run, out d loss / d z; Tz: the z tile, left holding
//   da * xhat (the terms of dgamma).
template <int RUN, int PARTS>
__device__ __forceinline__ void ln_mish_backward(float (&g)[RUN], float *Tz, int row, int part, float mean, float rstd,
                                                 const float *__restrict__ gamma) {
    float xh[RUN], mp[RUN];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < RUN; i++) {
        const int q = 4 * CHUNK(i >> 2, part, PARTS) + (i & 3);
        const int c = QCOL(q);
        const float z = Tz[row * ALD2 + q];
        const float nn = __expf(fminf(z, 20.f));
        const float mm = nn * (nn + 2.f);
        const float tt = __fdividef(mm, mm + 2.f);  // tanh(softplus(z))
        mp[i] = tt + z * (1.f - tt * tt) * __fdividef(nn, 1.f + nn);  // Mish'(z) = tanh(sp) + z sigmoid(z) (1 - tanh(sp)^2)
        xh[i] = (z * tt - mean) * rstd;
        Tz[row * ALD2 + q] = g[i] * xh[i];
        const float gy = g[i] * gamma[c];
        c1 += gy;
        c2 += gy * xh[i];
        g[i] = gy;
    }
#pragma unroll
    for (int m = 1; m < PARTS; m <<= 1) { c1 += __shfl_xor(c1, m); c2 += __shfl_xor(c2, m); }
    c1 *= (1.f / 256.f);
    c2 *= (1.f / 256.f);
#pragma unroll
    for (int i = 0; i < RUN; i++) g[i] = rstd * (g[i] - c1 - xh[i] * c2) * mp[i];
}


}  // namespace evm
