// PPO / GAE update on the device (gfx950, fp32 MFMA): replaces the PyTorch-ROCm autograd pass of
//   PpoGaeAgent::train                    evo_motion_networks/src/agents/ppo_gae.cpp:117-190
//   ActorModule / CriticModule            evo_motion_networks/src/networks/actor.cpp:9-48, critic.cpp:8-35
//   truncated_normal_log_pdf / _entropy   evo_motion_networks/src/functions.cpp:94-128
//   clip_grad_norm_ + torch::optim::Adam  ppo_gae.cpp:170-186
//
// One epoch = k_ppo_forward (both networks, activations kept) -> k_ppo_loss_* (loss + gradient at the head
// pre-activations) -> k_ppo_backward (per 32-row tile: heads, LayerNorm, Mish, the Linear(256,256) dgrad on MFMA)
// -> k_ppo_wgrad (weight gradients: split-K MFMA GEMMs over the rows) + reductions -> gradient norm, clip, Adam and
// the repack of the new weights into the operand layout of the forward kernels.  Every reduction runs in a fixed
// order: the update is deterministic.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mlp_tile.h"
#include "mlp_train.h"
#include "ppo_dev.h"

namespace evm {


#ifdef EVM_FSTAMPS  // diagnostic build (tools/fstamps.py): s_memtime at the phase boundaries of every wave of k_ppo_forward
__device__ unsigned long long g_fstamps[8192 * 4 * 8];
#define FSTAMP { __builtin_amdgcn_sched_barrier(0); fs_t[fs_n++] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define FSTAMP
#endif
#ifndef EVM_PPO_FWD_SPLIT
#define EVM_PPO_FWD_SPLIT 0   // 1: the training forward's hidden layers as six bf16 MFMA products per fp32 product (dense_layer_split)
#endif
template <int RT>
__global__ __launch_bounds__(PT) void k_ppo_forward(PolicyDev p, PpoDev d, int n, const float *__restrict__ states) {
    constexpr int TM = 32 * RT, PARTS = PT / TM, RUN = 256 / PARTS;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *xs = sm, *hb = sm;
    const int net = blockIdx.y;
    const int row0 = blockIdx.x * TM;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const NetDev &N = net == 0 ? p.actor : p.critic;
    const PpoNet &B = net == 0 ? d.actor : d.critic;
#ifdef EVM_FSTAMPS
    unsigned long long fs_t[8];
    int fs_n = 0;
#endif
    FSTAMP
    stage_padded_ksplit<TM>(xs, states, row0, n);  // the padded copy: [n][K1]
    lds_barrier();
    FSTAMP
    f32x16 acc[RT][2];
    float *red = sm + TM * ALD1;  // behind both tiles
    if constexpr (EVM_PPO_FWD_SPLIT && RT == 1) dense_layer_split<K1>(xs, ALD1, N.w1s, wave, lane, acc);
    else dense_layer<K1, RT>(xs, ALD1, N.w1t, wave, lane, acc);
    FSTAMP
    train_epilogue<RT>(acc, N.b1, N.g1, N.be1, hb, red, wave, lane, row0, n, B.z1, B.a1, B.st, 0, 4);
    FSTAMP
    if constexpr (EVM_PPO_FWD_SPLIT && RT == 1) dense_layer_split<256>(hb, ALD2, N.w2s, wave, lane, acc);
    else dense_layer<256, RT>(hb, ALD2, N.w2t, wave, lane, acc);
    FSTAMP
    train_epilogue<RT>(acc, N.b2, N.g2, N.be2, hb, red, wave, lane, row0, n, B.z2, B.a2, B.st, 2, 4);
    FSTAMP

    // heads, as in k_policy_forward
    const int t = threadIdx.x, row = t / PARTS, part = t % PARTS;
    const int gr = row0 + row;
    const int A = p.A;
    const int nout = net == 1 ? 1 : 2 * A;
    // head GEMM on the matrix pipe (mlp_tile.h): the four waves' partial tiles land behind the activation tile (inside the
    // dead observation tile), their sum over the activation tile itself once every wave has read its operands
    static_assert(TM == 32, "the head GEMM is one 32-row MFMA tile");
    float *hs4 = sm + TM * ALD2;  // [4 waves][32 rows][32 cols]
    float *hs = sm;               // [TM][32] pre-activations
    head_gemm(hb, N.whp, hs4, wave, lane);
    lds_barrier();
    for (int o = part; o < nout; o += PARTS)
        hs[row * 32 + o] = ((hs4[row * 32 + o] + hs4[(32 + row) * 32 + o]) + (hs4[(64 + row) * 32 + o] + hs4[(96 + row) * 32 + o])) + N.bh[o];
    lds_barrier();
    FSTAMP
#ifdef EVM_FSTAMPS
    if (lane == 0 && blockIdx.x < 4096) {
        unsigned long long *o = g_fstamps + ((size_t) (blockIdx.y * 4096 + blockIdx.x) * 4 + wave) * 8;
        for (int i = 0; i < 7; i++) o[i] = fs_t[i];
        o[7] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long) __builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);  // HW_ID, XCC_ID
    }
#endif
    if (net == 1) {
        if (part == 0 && gr < n) B.head[gr] = hs[row * 32];
        return;
    }
    for (int a = part; a < A; a += PARTS) {
        if (gr >= n) continue;
        B.head[(size_t) gr * 2 * A + a] = tanhf(hs[row * 32 + a]);
        B.head[(size_t) gr * 2 * A + A + a] = softplus_f(hs[row * 32 + A + a]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// losses and their gradients at the head pre-activations
// ---------------------------------------------------------------------------------------------------------
// clipped surrogate + entropy bonus (ppo_gae.cpp:155-166); one thread per (row, action dimension).
// inv_count = 1 / (masked rows of ALL ranks x A): the loss is the mean over the selected elements.
// dh columns >= 2A are zero from the allocation and never written.
__global__ __launch_bounds__(256) void k_ppo_loss_actor(PpoDev d, int n, const float *__restrict__ actions,
                                                        const float *__restrict__ logp_old, const float *__restrict__ adv,
                                                        const uint8_t *__restrict__ mask, float inv_count, float eps, float ef) {
    // inv_count < 0: the global count lives on the device (evm_ppo_gae, merged over the ranks); nothing selected -> zero gradients
    if (inv_count < 0.f) inv_count = d.gae[0] >= 1.0 ? (float) ((1.0 / d.gae[0]) / (double) d.A) : 0.f;
    __shared__ float sh[4];
    const int A = d.A;
    const size_t e = (size_t) blockIdx.x * 256 + threadIdx.x;
    float lsum = 0.f;
    if (e < (size_t) n * A) {
        const size_t row = e / A;
        const int o = (int) (e - row * A);
        float *dh = d.actor.dh + row * 32;
        {
            float g_mu = 0.f, g_sg = 0.f;
            if (mask[row]) {
                const float mu = d.actor.head[row * 2 * A + o], sg = d.actor.head[row * 2 * A + A + o];
                const float x = actions[e], lpo = logp_old[e], Adv = adv[row];
                const float INV_SQRT_2PI = 0.39894228040143267794f;
                const float s = fminf(fmaxf(sg, 1e-6f), 1e6f);
                const float cs = (sg >= 1e-6f && sg <= 1e6f) ? 1.f : 0.f;
                const float is = 1.0f / s;
                const float ar = (-1.f - mu) / s, br = (1.f - mu) / s;
                const float al = fminf(fmaxf(ar, -5.f), 5.f), be = fminf(fmaxf(br, -5.f), 5.f);
                const float ca = (ar >= -5.f && ar <= 5.f) ? 1.f : 0.f, cb = (br >= -5.f && br <= 5.f) ? 1.f : 0.f;
                const float ta = theta_f(al), tb = theta_f(be);
                const float Z = tb - ta, iZ = 1.0f / Z;
                const float pa = expf(-0.5f * al * al) * INV_SQRT_2PI, pb = expf(-0.5f * be * be) * INV_SQRT_2PI;
                const float q = (x - mu) / s;
                const float lp = -0.91893853320467274178f - logf(s) - 0.5f * (q * q) - logf(Z);
                const float Nn = al * pa - be * pb;
                const float ent = logf(4.13273135412249293846f * s * Z) + 0.5f * Nn / Z;  // sqrt(2 pi e)
                const float ratio = expf(lp - lpo);
                const float s1 = ratio * Adv;
                const float s2 = fminf(fmaxf(ratio, 1.f - eps), 1.f + eps) * Adv;
                lsum = -(fminf(s1, s2) + ef * ent) * inv_count;
                // torch.min(s1, s2): the gradient goes to the smaller one, half each on a tie; clamp passes it inside [lo, hi]
                const float w1 = s1 < s2 ? 1.f : (s1 == s2 ? 0.5f : 0.f);
                const float w2 = s2 < s1 ? 1.f : (s1 == s2 ? 0.5f : 0.f);
                const float inside = (ratio >= 1.f - eps && ratio <= 1.f + eps) ? 1.f : 0.f;
                const float dlp = -inv_count * Adv * (w1 + w2 * inside) * ratio;
                const float dent = -inv_count * ef;
                const float lp_mu = q * is + iZ * is * (pb * cb - pa * ca);
                const float lp_s = (q * q - 1.f) * is + iZ * is * (pb * be * cb - pa * al * ca);
                const float e_al = -pa * iZ + 0.5f * (pa * (1.f - al * al) * iZ + Nn * pa * iZ * iZ);
                const float e_be = pb * iZ + 0.5f * (-pb * (1.f - be * be) * iZ - Nn * pb * iZ * iZ);
                const float ent_mu = -(e_al * ca + e_be * cb) * is;
                const float ent_s = is - (e_al * ca * al + e_be * cb * be) * is;
                g_mu = (dlp * lp_mu + dent * ent_mu) * (1.f - mu * mu);            // tanh'
                g_sg = (dlp * lp_s + dent * ent_s) * cs * (-expm1f(-sg));          // softplus' = sigmoid(pre) = 1 - e^-sigma
            }
            dh[o] = g_mu;
            dh[A + o] = g_sg;
        }
    }
    block_partial(lsum, d.loss_part, sh);
}

// SAC's actor step: gradients w.r.t. (mu, sigma) supplied by the caller -> gradients at the head pre-activations
__global__ __launch_bounds__(256) void k_actor_head_grad(PpoDev d, int n, const float *__restrict__ dmu, const float *__restrict__ dsigma) {
    const int A = d.A;
    const size_t e = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t) n * A) return;
    const size_t row = e / A;
    const int o = (int) (e - row * A);
    const float mu = d.actor.head[row * 2 * A + o], sg = d.actor.head[row * 2 * A + A + o];
    float *dh = d.actor.dh + row * 32;
    dh[o] = dmu[e] * (1.f - mu * mu);        // tanh'
    dh[A + o] = dsigma[e] * (-expm1f(-sg));  // softplus' = sigmoid(pre) = 1 - e^-sigma
}
__global__ __launch_bounds__(256) void k_actor_head_out(PpoDev d, int n, float *__restrict__ mu, float *__restrict__ sigma) {
    const int A = d.A;
    const size_t e = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t) n * A) return;
    const size_t row = e / A;
    const int o = (int) (e - row * A);
    mu[e] = d.actor.head[row * 2 * A + o];
    sigma[e] = d.actor.head[row * 2 * A + A + o];
}
hipError_t launch_actor_head_grad(const PpoDev &d, size_t rows, const float *dmu, const float *dsigma, hipStream_t s) {
    hipLaunchKernelGGL(k_actor_head_grad, dim3((unsigned) ((rows * d.A + 255) / 256)), dim3(256), 0, s, d, (int) rows, dmu, dsigma);
    return hipGetLastError();
}
hipError_t launch_actor_head_out(const PpoDev &d, size_t rows, float *mu, float *sigma, hipStream_t s) {
    hipLaunchKernelGGL(k_actor_head_out, dim3((unsigned) ((rows * d.A + 255) / 256)), dim3(256), 0, s, d, (int) rows, mu, sigma);
    return hipGetLastError();
}

// critic_loss_factor * mean((value - returns)^2) over the selected rows (ppo_gae.cpp:176-179)
__global__ __launch_bounds__(256) void k_ppo_loss_critic(PpoDev d, int n, const float *__restrict__ returns,
                                                         const uint8_t *__restrict__ mask, float inv_rows, float cf, int part_off) {
    if (inv_rows < 0.f) inv_rows = d.gae[0] >= 1.0 ? (float) (1.0 / d.gae[0]) : 0.f;
    __shared__ float sh[4];
    const size_t row = (size_t) blockIdx.x * 256 + threadIdx.x;
    float lsum = 0.f;
    if (row < (size_t) n) {
        float dv = 0.f;
        if (mask[row]) {
            const float diff = d.critic.head[row] - returns[row];
            lsum = cf * diff * diff * inv_rows;
            dv = cf * 2.f * diff * inv_rows;
        }
        d.critic.dh[row * 32] = dv;
    }
    block_partial(lsum, d.loss_part + part_off, sh);
}

// ---------------------------------------------------------------------------------------------------------
// backward through heads, LayerNorm, Mish and the second Linear for one 32-row tile
// ---------------------------------------------------------------------------------------------------------
// Everything stays in the MFMA accumulator layout (wave: 32 rows x 64 columns, half a wave = 32 consecutive columns of a row):
//   d a2 = dh * W_heads           K = 32 GEMM on the matrix pipe (A: the dh tile, B: whd)
//   d z2 = LayerNorm' Mish' (d a2) ln_mish_backward_c, z2 read from HBM in that layout; -> HBM and the k-split tile Td
//   d a1 = d z2 * W2              K = 256 GEMM (B: w2d)
//   d z1 = LayerNorm' Mish' (d a1) -> HBM
// and the per-tile column sums (LayerNorm / bias gradients) are sums over a lane's 16 registers plus the other half of the
// wave.  Three LDS-only barriers.  (The row-wise form this replaces moved every tile through LDS three times behind twelve
// full barriers: 0.64 ms per launch at 131 072 rows, 36 % of what the matrix pipe alone needs.)
#define BWD_DH_LD 36  // k-split dh tile: 2 x 16 + 4
template <bool FULL>
__device__ __forceinline__ void ppo_backward_tile(const PolicyDev &p, const PpoDev &d, int n, float *sm) {
    constexpr int TM = 32, RT = 1;
    float *Td = sm;                    // d z2, k-split: the A operand of the dgrad GEMM
    float *dhs = sm + TM * ALD2;       // dh tile, k-split over the 32 head outputs
    float *red = dhs + TM * BWD_DH_LD; // row-mean exchange of ln_mish_backward_c
    const int net = blockIdx.y;
    const int row0 = blockIdx.x * TM;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cl = lane & 31, hf = lane >> 5;
    const int t = threadIdx.x;
    const NetDev &N = net == 0 ? p.actor : p.critic;
    const PpoNet &B = net == 0 ? d.actor : d.critic;
    const int nout = net == 1 ? 1 : 2 * p.A;
    float *cp = B.colpart + (size_t) blockIdx.x * PPO_COLSLOTS * 256;

    for (int e = threadIdx.x; e < TM * 32; e += PT) {
        const int r = e >> 5, o = e & 31;
        dhs[r * BWD_DH_LD + (o & 1) * 16 + (o >> 1)] = ((FULL || row0 + r < n) && o < nout) ? B.dh[(size_t) (row0 + r) * 32 + o] : 0.f;
    }
    LnBwdIn in;
    ln_bwd_load<FULL>(in, B.z2, B.st, 2, 4, wave, lane, row0, n);  // in flight across the head GEMM
    lds_barrier();
    if (t < 32) {  // head bias gradient: column sums of the dh tile
        float s = 0.f;
        for (int r = 0; r < TM; r++) s += dhs[r * BWD_DH_LD + (t & 1) * 16 + (t >> 1)];
        cp[6 * 256 + t] = s;
    }
    f32x16 acc[RT][2];
    dense_layer<32, RT>(dhs, BWD_DH_LD, B.whd, wave, lane, acc);  // d a2
    float cs[3][2];
    auto put_colsums = [&](int slot0) {
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const float v = cs[k][j] + __shfl_xor(cs[k][j], 32);
                if (hf == 0) cp[(slot0 + k) * 256 + wave * 64 + 32 * j + cl] = v;
            }
    };
    // stores of a gradient tile in the accumulator layout, 32-bit offsets from the uniform base
    auto put_rows = [&](float *dst, int r, float v0, float v1) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hf;
        if (FULL || row0 + row < n) {
            const unsigned o = ((unsigned) (row0 + row) * 256u + (unsigned) (wave * 64 + cl)) * 4u;
            stg_off(dst, o, v0);
            stg_off(dst, o + 128u, v1);
        }
    };
    ln_mish_backward_c(acc[0], in, N.g2, red, wave, lane, cs);  // acc <- d z2
    put_colsums(0);  // dgamma2, dbeta2, dbias2
    ln_bwd_load<FULL>(in, B.z1, B.st, 0, 4, wave, lane, row0, n);  // in flight across the dgrad GEMM
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hf;
        Td[HIDX(row, wave * 64 + cl)] = acc[0][0][r];
        Td[HIDX(row, wave * 64 + 32 + cl)] = acc[0][1][r];
        put_rows(B.dz2, r, acc[0][0][r], acc[0][1][r]);
    }
    lds_barrier();
    dense_layer<256, RT>(Td, ALD2, B.w2d, wave, lane, acc);  // d a1 = d z2 * W2
    ln_mish_backward_c(acc[0], in, N.g1, red, wave, lane, cs);  // acc <- d z1
    put_colsums(3);  // dgamma1, dbeta1, dbias1
#pragma unroll
    for (int r = 0; r < 16; r++) put_rows(B.dz1, r, acc[0][0][r], acc[0][1][r]);
}
template <int RT>
__global__ __launch_bounds__(PT) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_ppo_backward(PolicyDev p, PpoDev d, int n) {
    static_assert(RT == 1, "one 32-row MFMA tile per wave");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    if ((int) blockIdx.x * 32 + 32 <= n) ppo_backward_tile<true>(p, d, n, sm);
    else ppo_backward_tile<false>(p, d, n, sm);  // the ragged last tile
}

// ---------------------------------------------------------------------------------------------------------
// weight gradients: C[i][j] = sum_m P[m][i] Q[m][j], split over row chunks (blockIdx.y), 128 x 128 tile per workgroup
// ---------------------------------------------------------------------------------------------------------
template <bool ALIGNED>
__device__ __forceinline__ void wg_fetch(f32x4 (&r)[4], const float *__restrict__ X, int ld, int ncols, int c0, int m0, int m_end) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int e = u * PT + (int) threadIdx.x, rr = e >> 5, c4 = (e & 31) * 4;
        const int m = m0 + rr, c = c0 + c4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < m_end) {
            const float *src = X + (size_t) m * ld + c;
            if (ALIGNED) {
                if (c < ncols) v = *reinterpret_cast<const f32x4 *>(src);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) if (c + k < ncols) v[k] = src[k];
            }
        }
        r[u] = v;
    }
}
__device__ __forceinline__ void wg_stash(const f32x4 (&r)[4], float *S) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int e = u * PT + (int) threadIdx.x;
        reinterpret_cast<f32x4 *>(S)[e] = r[u];  // [32 rows][128 cols], row major
    }
}
template <bool PA, bool QA>
__global__ __launch_bounds__(PT) void k_ppo_wgrad(const float *__restrict__ P, int ldp, int np, const float *__restrict__ Q, int ldq,
                                                  int nq, int M, int rows_per_chunk, int i_tiles, float *__restrict__ part, int I, int ldo) {
    __shared__ __attribute__((aligned(16))) float Ps[2][32 * 128];
    __shared__ __attribute__((aligned(16))) float Qs[2][32 * 128];
    const int it = blockIdx.x % i_tiles, jt = blockIdx.x / i_tiles;
    const int i0 = it * 128, j0 = jt * 128;
    const int m_begin = blockIdx.y * rows_per_chunk;
    const int m_end = min(M, m_begin + rows_per_chunk);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wi = wave >> 1, wj = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
    f32x4 pr[4], qr[4];
    wg_fetch<PA>(pr, P, ldp, np, i0, m_begin, m_end);
    wg_fetch<QA>(qr, Q, ldq, nq, j0, m_begin, m_end);
    wg_stash(pr, Ps[0]);
    wg_stash(qr, Qs[0]);
    __syncthreads();
    int buf = 0;
    for (int m0 = m_begin; m0 < m_end; m0 += 32) {
        const bool more = m0 + 32 < m_end;
        if (more) {
            wg_fetch<PA>(pr, P, ldp, np, i0, m0 + 32, m_end);
            wg_fetch<QA>(qr, Q, ldq, nq, j0, m0 + 32, m_end);
        }
        const float *ps = Ps[buf] + wi * 64 + li, *qs = Qs[buf] + wj * 64 + li;
#pragma unroll
        for (int s = 0; s < 16; s++) {
            const int k = 2 * s + lh;
            const float a0 = ps[k * 128], a1 = ps[k * 128 + 32];
            const float b0 = qs[k * 128], b1 = qs[k * 128 + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) {
            wg_stash(pr, Ps[buf ^ 1]);
            wg_stash(qr, Qs[buf ^ 1]);
        }
        __syncthreads();
        buf ^= 1;
    }
    float *out = part + (size_t) blockIdx.y * I * ldo;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int j = j0 + wj * 64 + b * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int i = i0 + wi * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (i < I && j < ldo) out[(size_t) i * ldo + j] = acc[a][b][r];
            }
        }
}

// the heads' weight gradient: C[o][j] = sum_m dh[m][o] a2[m][j], 32 x 256 per workgroup (one 32-row MFMA tile, each wave
// 64 columns), split over row chunks like k_ppo_wgrad
__global__ __launch_bounds__(PT) void k_ppo_wgrad_heads(const float *__restrict__ P, const float *__restrict__ Q, int M, int rows_per_chunk,
                                                        float *__restrict__ part, int I) {
    __shared__ __attribute__((aligned(16))) float Ps[2][32 * 32];
    __shared__ __attribute__((aligned(16))) float Qs[2][32 * 256];
    const int m_begin = blockIdx.y * rows_per_chunk;
    const int m_end = min(M, m_begin + rows_per_chunk);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int t = threadIdx.x;
    f32x16 acc[2];
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[b][r] = 0.f;
    f32x4 pr, qr[8];
    auto fetch = [&](int m0) {
        {   // dh rows: 32 x 32 floats = 256 float4
            const int rr = t >> 3, c4 = (t & 7) * 4, m = m0 + rr;
            pr = m < m_end ? *reinterpret_cast<const f32x4 *>(P + (size_t) m * 32 + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {  // a2 rows: 32 x 256 floats = 2048 float4
            const int e = u * PT + t, rr = e >> 6, c4 = (e & 63) * 4, m = m0 + rr;
            qr[u] = m < m_end ? *reinterpret_cast<const f32x4 *>(Q + (size_t) m * 256 + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stash = [&](int buf) {
        reinterpret_cast<f32x4 *>(Ps[buf])[t] = pr;
#pragma unroll
        for (int u = 0; u < 8; u++) reinterpret_cast<f32x4 *>(Qs[buf])[u * PT + t] = qr[u];
    };
    fetch(m_begin);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int m0 = m_begin; m0 < m_end; m0 += 32) {
        const bool more = m0 + 32 < m_end;
        if (more) fetch(m0 + 32);
        const float *ps = Ps[buf] + li, *qs = Qs[buf] + wave * 64 + li;
#pragma unroll
        for (int s = 0; s < 16; s++) {
            const int k = 2 * s + lh;
            const float a0 = ps[k * 32];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, qs[k * 256], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, qs[k * 256 + 32], acc[1], 0, 0, 0);
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    float *out = part + (size_t) blockIdx.y * I * 256;
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const int j = wave * 64 + b * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (i < I) out[(size_t) i * 256 + j] = acc[b][r];
        }
    }
}

// sum of the split-K partials -> the flat gradient (rows >= split_row land `extra` floats further: the actor's sigma head)
__global__ __launch_bounds__(256) void k_ppo_wreduce(const float *__restrict__ part, int sk, int I, int ldo, int J, float *__restrict__ dst,
                                                     int dst_ld, int split_row, int extra) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= I * J) return;
    const int i = e / J, j = e - i * J;
    float s = 0.f;
    const float *src = part + (size_t) i * ldo + j;
    const size_t stride = (size_t) I * ldo;
    int k = 0;
    for (; k + 8 <= sk; k += 8) {  // eight loads in flight; the additions stay in chunk order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = src[(size_t) (k + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; u++) s += v[u];
    }
    for (; k < sk; k++) s += src[(size_t) k * stride];
    dst[(size_t) i * dst_ld + j + (i >= split_row ? extra : 0)] = s;
}

// column-sum partials: [tiles][W] -> [groups][W] (group g sums tiles g, g + groups, ...)
__global__ __launch_bounds__(256) void k_ppo_colreduce(const float *__restrict__ src, int tiles, int W, float *__restrict__ dst) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= W) return;
    float s = 0.f;
    int k = blockIdx.y;
    const int g = gridDim.y;
    for (; k + 7 * g < tiles; k += 8 * g) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = src[(size_t) (k + u * g) * W + e];
#pragma unroll
        for (int u = 0; u < 8; u++) s += v[u];
    }
    for (; k < tiles; k += g) s += src[(size_t) k * W + e];
    dst[(size_t) blockIdx.y * W + e] = s;
}
// [groups][7][256] -> the flat gradient slots
struct ColSlots { int off[PPO_COLSLOTS]; int A; int actor; int sg_extra; };
__global__ __launch_bounds__(256) void k_ppo_colfinish(const float *__restrict__ src, int groups, ColSlots cs, float *__restrict__ grad) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= PPO_COLSLOTS * 256) return;
    const int slot = e >> 8, c = e & 255;
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= groups; k += 8) {  // eight loads in flight, summed in index order (a handful of workgroups: pure latency otherwise)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = src[(size_t) (k + u) * PPO_COLSLOTS * 256 + e];
#pragma unroll
        for (int u = 0; u < 8; u++) s += v[u];
    }
    for (; k < groups; k++) s += src[(size_t) k * PPO_COLSLOTS * 256 + e];
    if (slot < 6) { grad[cs.off[slot] + c] = s; return; }
    if (cs.actor) {
        if (c < cs.A) grad[cs.off[6] + c] = s;                                  // mu.0.bias
        else if (c < 2 * cs.A) grad[cs.off[6] + cs.sg_extra + (c - cs.A)] = s;  // sigma.0.bias
    } else if (c == 0) {
        grad[cs.off[6]] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------
// clip_grad_norm_ + Adam (torch::optim::Adam defaults: betas 0.9 / 0.999, eps 1e-8, no weight decay)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ppo_sqnorm(PpoDev d) {  // grid (PPO_NORM_PARTS, 2): fixed partition, fixed order
    __shared__ double sh[4];
    const PpoNet &B = blockIdx.y == 0 ? d.actor : d.critic;
    double s = 0.0;
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < B.n_params; i += (size_t) PPO_NORM_PARTS * 256) {
        const double g = B.grad[i];
        s += g * g;
    }
    const double tot = block_sum_double(s, sh);
    if (threadIdx.x == 0) B.normp[blockIdx.x] = tot;
}
// One flat parameter -> the trainer's own dgrad operands: Linear(256,256) as B[k = out j][col = in i] (w2d) and the head weights as
// B[k = head output o < 32][col = in i] (whd), both k-split as in mlp_tile.h.  (k_ppo_pack_w2d / _whd do the same from a whole vector.)
__device__ __forceinline__ void ppo_dgrad_pack_write(const PpoNet &B, int S, int A, int actor, size_t i, float v) {
    const size_t o_w2 = (size_t) 256 * S + 3 * 256, o_h = o_w2 + 65536 + 3 * 256;
    if (i >= o_w2 && i < o_w2 + 65536) {
        const int e = (int) (i - o_w2), k = e >> 8, col = e & 255;
        const int st = k >> 1, h = k & 1, s4 = st >> 2, tt = st & 3;
        B.w2d[(((size_t) s4 * 256 + col) * 2 + h) * 4 + tt] = v;
        return;
    }
    if (i < o_h) return;
    size_t e = i - o_h;
    int k;
    if (actor) {
        const size_t hw = (size_t) A * 256;
        if (e < hw) k = (int) (e >> 8);                                  // mu.0.weight row
        else if (e >= hw + A && e < 2 * hw + A) { e -= hw + A; k = A + (int) (e >> 8); }  // sigma.0.weight row
        else return;                                                     // a bias
    } else {
        if (e >= 256) return;
        k = 0;
    }
    const int col = (int) (e & 255);
    const int st = k >> 1, h = k & 1, s4 = st >> 2, tt = st & 3;
    B.whd[(((size_t) s4 * 256 + col) * 2 + h) * 4 + tt] = v;
}

// clip + Adam + the repack of every new weight into the forward / dgrad operand layouts (policy_pack_write, ppo_dgrad_pack_write)
__global__ __launch_bounds__(256) void k_ppo_adam(PolicyDev p, PpoDev d, float max_norm, float lr, float bc1_a, float bc2s_a, float bc1_c, float bc2s_c) {
    const PpoNet &B = blockIdx.y == 0 ? d.actor : d.critic;
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    const float bc1 = blockIdx.y == 0 ? bc1_a : bc1_c, bc2s = blockIdx.y == 0 ? bc2s_a : bc2s_c;
    // nothing selected on any rank (the merged count lives on the device; the reference returns before it trains,
    // ppo_gae.cpp:63-66): no step — the moments do not decay, the weights do not drift on residual momentum
    if (d.dev_count && d.gae[0] < 1.0) return;
    double sq = 0.0;
#pragma unroll 8
    for (int k = 0; k < PPO_NORM_PARTS; k++) sq += B.normp[k];
    const float coef = fminf(max_norm / ((float) sqrt(sq) + 1e-6f), 1.0f);
    if (i >= B.n_params) return;
    const float g = B.grad[i] * coef;
    const float m = B.m[i] + (g - B.m[i]) * 0.1f;           // exp_avg.lerp_(grad, 1 - beta1)
    const float v = B.v[i] * 0.999f + (g * g) * 0.001f;    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    B.m[i] = m;
    B.v[i] = v;
    const float denom = sqrtf(v) / bc2s + 1e-8f;
    const float th = B.theta[i] - (lr / bc1) * (m / denom);
    B.theta[i] = th;
    const int actor = blockIdx.y == 0;
    policy_pack_write(actor ? p.actor : p.critic, p.S, p.A, actor, i, th);
    ppo_dgrad_pack_write(B, p.S, p.A, actor, i, th);
}

// the actor alone, no clipping, step count on the device (SAC's actor step inside a captured graph)
__global__ __launch_bounds__(256) void k_actor_adam_dev(PolicyDev p, PpoDev d, float lr) {
    const PpoNet &B = d.actor;
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= B.n_params) return;
    const int step = d.step_dev[0] + 1;
    const float bc1 = (float) (1.0 - pow(0.9, (double) step));
    const float bc2s = (float) sqrt(1.0 - pow(0.999, (double) step));
    const float g = B.grad[i];
    const float m = B.m[i] + (g - B.m[i]) * 0.1f;
    const float v = B.v[i] * 0.999f + (g * g) * 0.001f;
    B.m[i] = m;
    B.v[i] = v;
    const float th = B.theta[i] - (lr / bc1) * (m / (sqrtf(v) / bc2s + 1e-8f));
    B.theta[i] = th;
    policy_pack_write(p.actor, p.S, p.A, 1, i, th);
    ppo_dgrad_pack_write(B, p.S, p.A, 1, i, th);
}
__global__ void k_actor_step_inc(PpoDev d) {
    if (threadIdx.x == 0) d.step_dev[0] += 1;
}
hipError_t launch_actor_apply(const PolicyDev &p, const PpoDev &d, float lr, hipStream_t s) {
    hipLaunchKernelGGL(k_actor_adam_dev, dim3((unsigned) ((d.actor.n_params + 255) / 256)), dim3(256), 0, s, p, d, lr);
    hipLaunchKernelGGL(k_actor_step_inc, dim3(1), dim3(64), 0, s, d);
    return hipGetLastError();
}

// Linear(256,256) weight -> the B operand of the dgrad GEMM: B[k = j][col = i] = W2[j][i], k-split as in mlp_tile.h
__global__ __launch_bounds__(256) void k_ppo_pack_w2d(const float *__restrict__ w2, float *__restrict__ w2d) {
    const int e = blockIdx.x * 256 + threadIdx.x;  // e = j * 256 + i
    const int k = e >> 8, col = e & 255;
    const int st = k >> 1, h = k & 1, s4 = st >> 2, tt = st & 3;
    w2d[(((size_t) s4 * 256 + col) * 2 + h) * 4 + tt] = w2[e];
}

// ---------------------------------------------------------------------------------------------------------
// GAE (ppo_gae.cpp:127-150), time-major [T][N] rollouts
// ---------------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
// one thread per env: backward scan over the horizon; per-block (count, sum) of the selected advantages
__global__ __launch_bounds__(256) void k_ppo_gae_scan(int T, int N, const float *__restrict__ rewards, const uint8_t *__restrict__ done,
                                                      const float *__restrict__ cv, const float *__restrict__ nv,
                                                      const uint8_t *__restrict__ mask, float gamma, float gl, float *__restrict__ adv,
                                                      double *__restrict__ part) {
    __shared__ double sh[4];
    const int n = blockIdx.x * 256 + threadIdx.x;
    double cnt = 0.0, sum = 0.0;
    if (n < N) {
        float g = 0.f;
        for (int t = T - 1; t >= 0; t--) {
            const size_t i = (size_t) t * N + n;
            const float m = mask[i] ? 1.f : 0.f;
            const float nd = (mask[i] && !done[i]) ? 1.f : 0.f;  // 1 - done, with done forced to 1 outside the mask
            const float delta = rewards[i] + (nd * gamma) * nv[i] - cv[i];
            g = delta * m + (gl * nd) * g;
            g = g * m;
            adv[i] = g;
            if (mask[i]) { cnt += 1.0; sum += (double) g; }
        }
    }
    const double c_all = block_sum_double(cnt, sh);
    const double s_all = block_sum_double(sum, sh);
    if (threadIdx.x == 0) { part[3 * blockIdx.x] = c_all; part[3 * blockIdx.x + 1] = s_all; }
}
// partials -> count and mean (fixed order)
__global__ __launch_bounds__(64) void k_ppo_gae_mean(int nblocks, const double *__restrict__ part, double *__restrict__ stats) {
    if (threadIdx.x != 0) return;
    double c = 0.0, s = 0.0;
    for (int b = 0; b < nblocks; b++) { c += part[3 * b]; s += part[3 * b + 1]; }
    stats[0] = c;
    stats[1] = c > 0.0 ? s / c : 0.0;
}
__global__ __launch_bounds__(256) void k_ppo_gae_m2(int T, int N, const uint8_t *__restrict__ mask, const float *__restrict__ adv,
                                                    const double *__restrict__ stats, double *__restrict__ part) {
    __shared__ double sh[4];
    const int n = blockIdx.x * 256 + threadIdx.x;
    const double mean = stats[1];
    double m2 = 0.0;
    if (n < N)
        for (int t = 0; t < T; t++) {
            const size_t i = (size_t) t * N + n;
            if (mask[i]) { const double dd = (double) adv[i] - mean; m2 += dd * dd; }
        }
    const double all = block_sum_double(m2, sh);
    if (threadIdx.x == 0) part[3 * blockIdx.x + 2] = all;
}
__global__ __launch_bounds__(64) void k_ppo_gae_m2sum(int nblocks, const double *__restrict__ part, double *__restrict__ stats) {
    if (threadIdx.x != 0) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; b++) s += part[3 * b + 2];
    stats[2] = s;
}
__global__ __launch_bounds__(256) void k_ppo_gae_finish(size_t total, const double *__restrict__ stats, const float *__restrict__ cv,
                                                        float *__restrict__ adv, float *__restrict__ returns) {
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float mean = (float) stats[1];
    const float sd = (float) sqrt(stats[2] / fmax(stats[0] - 1.0, 1.0));
    const float a = (adv[i] - mean) / (sd + 1e-8f);
    adv[i] = a;
    returns[i] = a + cv[i];  // returns = NORMALISED advantages + V (ppo_gae.cpp:150)
}
#pragma clang fp contract(fast)

// ---------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------
static size_t fwd_lds_bytes() {
    constexpr int TM = 32 * PRT;
    const size_t a = (size_t) TM * ALD1 + EVM_RED_FLOATS, b = (size_t) TM * ALD2 + 4 * 32 * 32;
    return (a > b ? a : b) * sizeof(float);
}
static size_t bwd_lds_bytes() { return ((size_t) 32 * ALD2 + 32 * BWD_DH_LD + EVM_RED_FLOATS) * sizeof(float); }
size_t ppo_wpart_floats() { return (size_t) 128 * 256 * 256 > (size_t) 86 * 256 * 384 ? (size_t) 128 * 256 * 256 : (size_t) 86 * 256 * 384; }

// head weights -> the B operand of the heads' dgrad GEMM: B[k = head output o][col = i], K padded to 32 with zero rows.
// Actor: o < A is mu.0.weight row o, A <= o < 2A is sigma.0.weight row o - A (each followed by its bias in theta); critic: one row.
__global__ __launch_bounds__(256) void k_ppo_pack_whd(const float *__restrict__ heads, int A, int actor, float *__restrict__ whd) {
    const int e = blockIdx.x * 256 + threadIdx.x;  // e = o * 256 + i, o < 32
    const int k = e >> 8, col = e & 255;
    float v = 0.f;
    if (actor) {
        if (k < A) v = heads[(size_t) k * 256 + col];
        else if (k < 2 * A) v = heads[(size_t) A * 256 + A + (size_t) (k - A) * 256 + col];
    } else if (k == 0) v = heads[col];
    const int st = k >> 1, h = k & 1, s4 = st >> 2, tt = st & 3;
    whd[(((size_t) s4 * 256 + col) * 2 + h) * 4 + tt] = v;
}
hipError_t launch_ppo_pack_w2d(const PpoNet &n, int S, int A, bool actor, hipStream_t s) {
    const float *w2 = n.theta + (size_t) 256 * S + 3 * 256;
    hipLaunchKernelGGL(k_ppo_pack_w2d, dim3(256), dim3(256), 0, s, w2, n.w2d);
    hipLaunchKernelGGL(k_ppo_pack_whd, dim3(32), dim3(256), 0, s, w2 + 65536 + 3 * 256, A, actor ? 1 : 0, n.whd);
    return hipGetLastError();
}

// observations -> [rows][K1] with zero padding: 16-byte aligned rows for the forward staging and the weight-gradient GEMM
__global__ __launch_bounds__(256) void k_ppo_pad(const float *__restrict__ src, int S, size_t rows, float *__restrict__ dst) {
    const size_t e = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (e >= rows * K1) return;
    const size_t r = e / K1;
    const int c = (int) (e - r * K1);
    dst[e] = c < S ? src[r * S + c] : 0.f;
}
hipError_t launch_ppo_pad(const PpoDev &d, size_t rows, const float *states, hipStream_t s) {
    hipLaunchKernelGGL(k_ppo_pad, dim3((unsigned) ((rows * K1 + 255) / 256)), dim3(256), 0, s, states, d.S, rows, d.xpad);
    return hipGetLastError();
}

// hipFuncSetAttribute is per DEVICE: a process that creates policies / trainers on a second device must set it there too
static bool &evm_attr_done_for_current_device() {
    static bool done[64] = {};
    int dev = 0;
    (void) hipGetDevice(&dev);
    return done[dev >= 0 && dev < 64 ? dev : 0];
}

hipError_t launch_ppo_forward(const PolicyDev &p, const PpoDev &d, size_t rows, const float *states, hipStream_t s, int nets) {
    bool &attr = evm_attr_done_for_current_device();
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ppo_forward<PRT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int) fwd_lds_bytes());
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ppo_backward<PRT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int) bwd_lds_bytes());
        if (e != hipSuccess) return e;
        attr = true;
    }
    constexpr int TM = 32 * PRT;
    dim3 grid((unsigned) ((rows + TM - 1) / TM), nets);
    (void) states;
    hipLaunchKernelGGL(k_ppo_forward<PRT>, grid, dim3(PT), fwd_lds_bytes(), s, p, d, (int) rows, d.xpad);
    return hipGetLastError();
}

// the two loss values: the workgroups' partial sums in index order (one workgroup; a strided pass, then a fixed tree)
__global__ __launch_bounds__(256) void k_ppo_loss_sum(PpoDev d, int na, int nc) {
    __shared__ double sh[256];
    for (int which = 0; which < 2; which++) {
        const double *p = d.loss_part + (which ? na : 0);
        const int cnt = which ? nc : na;
        double s = 0.0;
        int i = threadIdx.x;
        for (; i + 7 * 256 < cnt; i += 8 * 256) {  // eight loads in flight, added in index order
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = p[i + u * 256];
#pragma unroll
            for (int u = 0; u < 8; u++) s += v[u];
        }
        for (; i < cnt; i += 256) s += p[i];
        sh[threadIdx.x] = s;
        __syncthreads();
        for (int m = 128; m >= 1; m >>= 1) {
            if ((int) threadIdx.x < m) sh[threadIdx.x] += sh[threadIdx.x + m];
            __syncthreads();
        }
        if (threadIdx.x == 0) d.loss[which] = (d.dev_count && d.gae[0] < 1.0) ? (double) __builtin_nanf("") : sh[0];   // an empty update says so
        __syncthreads();
    }
}
hipError_t launch_ppo_loss(const PpoDev &d, size_t rows, const float *actions, const float *logp_old, const float *adv,
                           const float *returns, const uint8_t *mask, double inv_rows, float epsilon, float entropy_factor,
                           float critic_loss_factor, hipStream_t s) {
    const float inv_count = inv_rows < 0.0 ? -1.f : (float) (inv_rows / (double) d.A);
    const unsigned na = (unsigned) ((rows * d.A + 255) / 256), nc = (unsigned) ((rows + 255) / 256);
    hipLaunchKernelGGL(k_ppo_loss_actor, dim3(na), dim3(256), 0, s, d, (int) rows, actions, logp_old, adv, mask, inv_count, epsilon,
                       entropy_factor);
    hipLaunchKernelGGL(k_ppo_loss_critic, dim3(nc), dim3(256), 0, s, d, (int) rows, returns, mask, (float) inv_rows, critic_loss_factor,
                       (int) na);
    hipLaunchKernelGGL(k_ppo_loss_sum, dim3(1), dim3(256), 0, s, d, (int) na, (int) nc);
    return hipGetLastError();
}

hipError_t launch_ppo_backward(const PolicyDev &p, const PpoDev &d, size_t rows, hipStream_t s, int nets) {
    constexpr int TM = 32 * PRT;
    dim3 grid((unsigned) ((rows + TM - 1) / TM), nets);
    hipLaunchKernelGGL(k_ppo_backward<PRT>, grid, dim3(PT), bwd_lds_bytes(), s, p, d, (int) rows);
    return hipGetLastError();
}

void wgrad_one(const float *P, int ldp, int np, bool pa, const float *Q, int ldq, int nq, bool qa, int M, float *part, int I, int J,
                      float *dst, int dst_ld, int split_row, int extra, hipStream_t s) {
    const int ldo = (J + 127) / 128 * 128;
    int chunks = (M + 31) / 32;
    const int want = 512 / (((I + 127) / 128) * (ldo / 128));  // two workgroups per CU (64 KB of LDS each)
    if (chunks > want) chunks = want;
    int rpc = ((M + chunks - 1) / chunks + 31) / 32 * 32;
    chunks = (M + rpc - 1) / rpc;
    const int i_tiles = (I + 127) / 128, j_tiles = ldo / 128;
    dim3 grid(i_tiles * j_tiles, chunks);
    if (pa && qa) hipLaunchKernelGGL((k_ppo_wgrad<true, true>), grid, dim3(PT), 0, s, P, ldp, np, Q, ldq, nq, M, rpc, i_tiles, part, I, ldo);
    else if (pa) hipLaunchKernelGGL((k_ppo_wgrad<true, false>), grid, dim3(PT), 0, s, P, ldp, np, Q, ldq, nq, M, rpc, i_tiles, part, I, ldo);
    else hipLaunchKernelGGL((k_ppo_wgrad<false, false>), grid, dim3(PT), 0, s, P, ldp, np, Q, ldq, nq, M, rpc, i_tiles, part, I, ldo);
    hipLaunchKernelGGL(k_ppo_wreduce, dim3((I * J + 255) / 256), dim3(256), 0, s, part, chunks, I, ldo, J, dst, dst_ld, split_row, extra);
}

void wgrad_heads(const float *dh, const float *a_last, float *wpart, int M, int I, float *dst, int split_row, int extra, hipStream_t s) {
    int chunks = (M + 31) / 32;
    if (chunks > 4 * PPO_SK) chunks = 4 * PPO_SK;  // a workgroup's tile is small: more row chunks to fill the chip
    int rpc = ((M + chunks - 1) / chunks + 31) / 32 * 32;
    chunks = (M + rpc - 1) / rpc;
    hipLaunchKernelGGL(k_ppo_wgrad_heads, dim3(1, chunks), dim3(PT), 0, s, dh, a_last, M, rpc, wpart, I);
    hipLaunchKernelGGL(k_ppo_wreduce, dim3((I * 256 + 255) / 256), dim3(256), 0, s, wpart, chunks, I, 256, 256, dst, 256, split_row, extra);
}

int launch_colreduce(const float *colpart, int tiles, int W, float *colpart2, hipStream_t s) {
    const int groups = tiles < 64 ? tiles : 64;
    hipLaunchKernelGGL(k_ppo_colreduce, dim3((W + 255) / 256, groups), dim3(256), 0, s, colpart, tiles, W, colpart2);
    return groups;
}
hipError_t launch_pad_rows(const float *src, int S, size_t rows, float *dst, hipStream_t s) {
    hipLaunchKernelGGL(k_ppo_pad, dim3((unsigned) ((rows * K1 + 255) / 256)), dim3(256), 0, s, src, S, rows, dst);
    return hipGetLastError();
}

hipError_t launch_ppo_wgrads(const PpoDev &d, size_t rows, const float *states, hipStream_t s, int nets) {
    constexpr int TM = 32 * PRT;
    const int S = d.S, A = d.A, M = (int) rows;
    const int tiles = (int) ((rows + TM - 1) / TM);
    (void) states;
    for (int net = 0; net < nets; net++) {
        const PpoNet &B = net == 0 ? d.actor : d.critic;
        const size_t o_w1 = 0, o_b1 = (size_t) 256 * S, o_g1 = o_b1 + 256, o_be1 = o_g1 + 256, o_w2 = o_be1 + 256;
        const size_t o_b2 = o_w2 + 65536, o_g2 = o_b2 + 256, o_be2 = o_g2 + 256, o_h = o_be2 + 256;
        // head.0.weight [256][S] = dz1^T states; head.3.weight [256][256] = dz2^T a1; heads [nout][256] = dh^T a2
        wgrad_one(B.dz1, 256, 256, true, d.xpad, K1, K1, true, M, B.wpart, 256, S, B.grad + o_w1, S, 1 << 30, 0, s);
        wgrad_one(B.dz2, 256, 256, true, B.a1, 256, 256, true, M, B.wpart, 256, 256, B.grad + o_w2, 256, 1 << 30, 0, s);
        ColSlots cs;
        cs.off[0] = (int) o_g2; cs.off[1] = (int) o_be2; cs.off[2] = (int) o_b2;
        cs.off[3] = (int) o_g1; cs.off[4] = (int) o_be1; cs.off[5] = (int) o_b1;
        cs.A = A; cs.actor = net == 0;
        if (net == 0) {
            // mu.0.weight [A][256], mu.0.bias [A], sigma.0.weight [A][256], sigma.0.bias [A]
            wgrad_heads(B.dh, B.a2, B.wpart, M, 2 * A, B.grad + o_h, A, A, s);
            cs.off[6] = (int) (o_h + (size_t) A * 256);
            cs.sg_extra = A * 256 + A;
        } else {
            wgrad_heads(B.dh, B.a2, B.wpart, M, 1, B.grad + o_h, 1 << 30, 0, s);
            cs.off[6] = (int) (o_h + 256);
            cs.sg_extra = 0;
        }
        const int W = PPO_COLSLOTS * 256;
        const int groups = tiles < 64 ? tiles : 64;
        hipLaunchKernelGGL(k_ppo_colreduce, dim3((W + 255) / 256, groups), dim3(256), 0, s, B.colpart, tiles, W, B.colpart2);
        hipLaunchKernelGGL(k_ppo_colfinish, dim3((W + 255) / 256), dim3(256), 0, s, B.colpart2, groups, cs, B.grad);
    }
    return hipGetLastError();
}

hipError_t launch_ppo_apply(const PolicyDev &p, PpoDev &d, float lr, float clip_grad_norm, hipStream_t s) {
    d.actor.step++;
    d.critic.step++;
    auto bc1 = [](int t) { return (float) (1.0 - pow(0.9, (double) t)); };
    auto bc2s = [](int t) { return (float) sqrt(1.0 - pow(0.999, (double) t)); };
    hipLaunchKernelGGL(k_ppo_sqnorm, dim3(PPO_NORM_PARTS, 2), dim3(256), 0, s, d);
    const size_t nmax = d.actor.n_params > d.critic.n_params ? d.actor.n_params : d.critic.n_params;
    // the step and the repack of the new weights into every operand layout in one launch (policy_pack_write, ppo_dgrad_pack_write)
    hipLaunchKernelGGL(k_ppo_adam, dim3((unsigned) ((nmax + 255) / 256), 2), dim3(256), 0, s, p, d, clip_grad_norm, lr, bc1(d.actor.step),
                       bc2s(d.actor.step), bc1(d.critic.step), bc2s(d.critic.step));
    return hipGetLastError();
}

// (count, mean, M2) triples of `world` ranks -> the trainer's own statistics, Chan's pairwise merge in rank order (the same
// order on every rank, so the replicas normalise with identical numbers); one thread
__global__ __launch_bounds__(64) void k_ppo_gae_merge(const double *__restrict__ all, int world, double *__restrict__ stats) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double n = all[0], mean = all[1], m2 = all[2];
    for (int r = 1; r < world; r++) {
        const double nb = all[3 * r], mb = all[3 * r + 1], m2b = all[3 * r + 2];
        const double tot = n + nb;
        if (tot == 0.0) continue;
        const double dlt = mb - mean;
        mean = mean + dlt * nb / tot;
        m2 = m2 + m2b + dlt * dlt * n * nb / tot;
        n = tot;
    }
    stats[0] = n; stats[1] = mean; stats[2] = m2;
}
hipError_t launch_ppo_gae_merge(const PpoDev &d, const double *all, int world, hipStream_t s) {
    hipLaunchKernelGGL(k_ppo_gae_merge, dim3(1), dim3(64), 0, s, all, world, d.gae);
    return hipGetLastError();
}
hipError_t launch_ppo_gae_scan(const PpoDev &d, int T, int N, const float *rewards, const uint8_t *done, const float *curr_values,
                               const float *next_values, const uint8_t *mask, float gamma, float lam, float *adv, hipStream_t s) {
    const float gl = (float) ((double) gamma * (double) lam);
    const int nblocks = (N + 255) / 256;
    hipLaunchKernelGGL(k_ppo_gae_scan, dim3(nblocks), dim3(256), 0, s, T, N, rewards, done, curr_values, next_values, mask, gamma, gl, adv, d.gae_part);
    hipLaunchKernelGGL(k_ppo_gae_mean, dim3(1), dim3(64), 0, s, nblocks, d.gae_part, d.gae);
    hipLaunchKernelGGL(k_ppo_gae_m2, dim3(nblocks), dim3(256), 0, s, T, N, mask, adv, d.gae, d.gae_part);
    hipLaunchKernelGGL(k_ppo_gae_m2sum, dim3(1), dim3(64), 0, s, nblocks, d.gae_part, d.gae);
    return hipGetLastError();
}
hipError_t launch_ppo_gae_finish(int T, int N, const double *stats, const float *curr_values, const uint8_t *mask, float *adv,
                                 float *returns, hipStream_t s) {
    (void) mask;
    const size_t total = (size_t) T * N;
    hipLaunchKernelGGL(k_ppo_gae_finish, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, s, total, stats, curr_values, adv, returns);
    return hipGetLastError();
}


// ---- selected rows only -------------------------------------------------------------------------------------------------
// After GAE the update is a sum over rows in which the rows outside the mask (reset()'s settle calls and emissions) weigh
// nothing; evm_ppo_select_rows gathers the selected ones, in their order, so that the epochs' GEMMs do not push the others
// through forward, backward and weight gradients.  One block: every thread counts a contiguous slice, the block scans the
// counts, every thread writes its slice's row numbers.
__global__ __launch_bounds__(1024) void k_ppo_select_scan(size_t rows, const uint8_t *__restrict__ mask, int *__restrict__ sel_idx,
                                                          int *__restrict__ sel_count) {
    __shared__ int part[1024];
    const size_t per = (rows + 1023) / 1024;
    const size_t lo = (size_t) threadIdx.x * per, hi = lo + per < rows ? lo + per : rows;
    int c = 0;
    for (size_t r = lo; r < hi; r++) c += mask[r] != 0;
    part[threadIdx.x] = c;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // inclusive Hillis-Steele scan
        const int v = (int) threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int at = part[threadIdx.x] - c;
    for (size_t r = lo; r < hi; r++)
        if (mask[r] != 0) sel_idx[at++] = (int) r;
    if (threadIdx.x == 1023) sel_count[0] = part[1023];
}
// one wavefront per selected row
__global__ __launch_bounds__(256) void k_ppo_select_gather(size_t n_sel, const int *__restrict__ sel_idx, int S, int A,
                                                           const float *__restrict__ states, const float *__restrict__ actions,
                                                           const float *__restrict__ logp, const float *__restrict__ adv,
                                                           const float *__restrict__ returns, float *o_states, float *o_actions,
                                                           float *o_logp, float *o_adv, float *o_returns) {
    const size_t i = (size_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_sel) return;
    const int lane = threadIdx.x & 63;
    const size_t r = (size_t) sel_idx[i];
    for (int k = lane; k < S; k += 64) o_states[i * S + k] = states[r * S + k];
    if (lane < A) { o_actions[i * A + lane] = actions[r * A + lane]; o_logp[i * A + lane] = logp[r * A + lane]; }
    if (lane == 0) { o_adv[i] = adv[r]; o_returns[i] = returns[r]; }
}
hipError_t launch_ppo_select_scan(size_t rows, const uint8_t *mask, int *sel_idx, int *sel_count, hipStream_t s) {
    hipLaunchKernelGGL(k_ppo_select_scan, dim3(1), dim3(1024), 0, s, rows, mask, sel_idx, sel_count);
    return hipGetLastError();
}
hipError_t launch_ppo_select_gather(size_t n_sel, const int *sel_idx, int S, int A, const float *states, const float *actions,
                                    const float *logp, const float *adv, const float *returns, float *o_states, float *o_actions,
                                    float *o_logp, float *o_adv, float *o_returns, hipStream_t s) {
    if (n_sel == 0) return hipSuccess;
    hipLaunchKernelGGL(k_ppo_select_gather, dim3((unsigned) ((n_sel + 3) / 4)), dim3(256), 0, s, n_sel, sel_idx, S, A, states, actions,
                       logp, adv, returns, o_states, o_actions, o_logp, o_adv, o_returns);
    return hipGetLastError();
}

}  // namespace evm

#ifdef EVM_FSTAMPS
extern "C" int evm_debug_fstamps(unsigned long long *out) {
    return (int) hipMemcpyFromSymbol(out, HIP_SYMBOL(evm::g_fstamps), sizeof(unsigned long long) * 8192 * 4 * 8);
}
#endif
