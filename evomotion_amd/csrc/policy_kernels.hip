// Fused actor-critic forward + truncated-normal sampling for the PPO rollout (gfx950, fp32 MFMA).
//
// Replaces, per environment (reference paths relative to its repo root):
//   PpoGaeAgent::act                     evo_motion_networks/src/agents/ppo_gae.cpp:29-45
//   ActorModule / CriticModule forward   evo_motion_networks/src/networks/actor.cpp:30-48, critic.cpp:23-35
//   truncated_normal_sample / _log_pdf   evo_motion_networks/src/functions.cpp:53-68,94-111
//
// One workgroup (4 waves) owns a 64-row tile of the batch and ONE of the two networks (blockIdx.y: 0 actor,
// 1 critic).  Both hidden layers run on v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain): each wave
// owns 64 rows x 64 output columns as 2x2 accumulator tiles.  A operands come from LDS (observations staged
// in 32-wide K chunks, activations kept in a [64][257] tile), B operands are read straight from the
// pre-transposed [K][256] weights (two 128-byte segments per wave-instruction, L2 resident).  Bias, Mish,
// LayerNorm, the tanh/softplus heads, inverse-CDF sampling and the log-pdf are fused behind the GEMMs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "policy_dev.h"

namespace evm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define PT 256       // threads per workgroup
#define TM 64        // rows per workgroup
#define HLD 257      // activation tile leading dimension (bank = (row + k) mod 32)
#define XLD 33       // observation chunk leading dimension

__device__ __forceinline__ float mish_f(float x) { return x * tanhf(log1pf(expf(x))); }
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float theta_f(float x) { return 0.5f * (1.0f + erff(x / 1.41421356237309504880f)); }

// one dense layer: acc[2][2] += A(64 x K, from LDS or staged from global) * Wt(K x 256)
template <bool FROM_GLOBAL>
__device__ __forceinline__ void dense_layer(const float *__restrict__ X, int n_rows, int row0, int S, int Kpad,
                                            const float *__restrict__ Wt, float *xs, const float *hb, int wave, int lane,
                                            f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const int ai = lane & 31, ak = lane >> 5;
    const int col = wave * 64 + (lane & 31);
    if (FROM_GLOBAL) {
        const int t = threadIdx.x;
        const int srow = t >> 2, sk = (t & 3) * 8;
        for (int kc = 0; kc < Kpad; kc += 32) {
            // stage X[row0 .. row0+64)[kc .. kc+32) -> xs[64][XLD]
            const int gr = row0 + srow;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int k = kc + sk + u;
                float v = 0.f;
                if (gr < n_rows && k < S) v = X[(size_t) gr * S + k];
                xs[srow * XLD + sk + u] = v;
            }
            __syncthreads();
#pragma unroll 4
            for (int kk = 0; kk < 32; kk += 2) {
                const float a0 = xs[ai * XLD + kk + ak], a1 = xs[(32 + ai) * XLD + kk + ak];
                const float *wr = Wt + (size_t) (kc + kk + ak) * 256 + col;
                const float b0 = wr[0], b1 = wr[32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            __syncthreads();
        }
    } else {
#pragma unroll 4
        for (int k = 0; k < Kpad; k += 2) {
            const float a0 = hb[ai * HLD + k + ak], a1 = hb[(32 + ai) * HLD + k + ak];
            const float *wr = Wt + (size_t) (k + ak) * 256 + col;
            const float b0 = wr[0], b1 = wr[32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
}

// bias + Mish into the activation tile, then LayerNorm(256) per row
__device__ __forceinline__ void epilogue_mish_ln(f32x16 (&acc)[2][2], const float *__restrict__ bias,
                                                 const float *__restrict__ gamma, const float *__restrict__ beta,
                                                 float *hb, int wave, int lane) {
    // C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int c = wave * 64 + j * 32 + (lane & 31);
            const float b = bias[c];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                hb[row * HLD + c] = mish_f(acc[i][j][r] + b);
            }
        }
    __syncthreads();
    const int t = threadIdx.x, row = t >> 2, part = t & 3;
    float s = 0.f;
    for (int c = part * 64; c < part * 64 + 64; c++) s += hb[row * HLD + c];
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
    const float mean = s / 256.f;
    float v = 0.f;
    for (int c = part * 64; c < part * 64 + 64; c++) { const float d = hb[row * HLD + c] - mean; v += d * d; }
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2);
    const float rstd = 1.0f / sqrtf(v / 256.f + 1e-5f);
    for (int c = part * 64; c < part * 64 + 64; c++)
        hb[row * HLD + c] = (hb[row * HLD + c] - mean) * rstd * gamma[c] + beta[c];
    __syncthreads();
}

__device__ __forceinline__ float rng_uniform(uint64_t seed, uint64_t counter, uint32_t row, uint32_t dim) {
    // counter-based (stateless) generator: splitmix64 finaliser over (seed, counter, row, dim) -> 24-bit uniform [0,1)
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (counter + 1) + ((uint64_t) row << 20) + dim;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float) (uint32_t) (z >> 40) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(PT) void k_policy_forward(PolicyDev p, int n, const float *__restrict__ obs,
                                                       const float *__restrict__ uniform, uint64_t seed, uint64_t counter,
                                                       float *action, float *logp, float *value, float *mu_out,
                                                       float *sigma_out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *hb = sm;               // [64][HLD]
    float *xs = sm + TM * HLD;    // [64][XLD]
    const int net = blockIdx.y;   // 0 actor, 1 critic
    const int row0 = blockIdx.x * TM;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const NetDev &N = net == 0 ? p.actor : p.critic;
    f32x16 acc[2][2];
    dense_layer<true>(obs, n, row0, p.S, p.K1pad, N.w1t, xs, hb, wave, lane, acc);
    epilogue_mish_ln(acc, N.b1, N.g1, N.be1, hb, wave, lane);
    dense_layer<false>(nullptr, n, row0, p.S, 256, N.w2t, xs, hb, wave, lane, acc);
    __syncthreads();  // every wave has finished reading the layer-1 activations
    epilogue_mish_ln(acc, N.b2, N.g2, N.be2, hb, wave, lane);

    const int t = threadIdx.x, row = t >> 2, part = t & 3;
    const int gr = row0 + row;
    if (net == 1) {  // critic head: Linear(256, 1)
        float s = 0.f;
        for (int c = part * 64; c < part * 64 + 64; c++) s += hb[row * HLD + c] * N.wh[c];
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
        if (part == 0 && gr < n) value[gr] = s + N.bh[0];
        return;
    }
    // actor heads: mu = tanh(Linear(256, A)), sigma = softplus(Linear(256, A)); 2A outputs split over 4 threads
    const int A = p.A;
    for (int o = part; o < 2 * A; o += 4) {
        const float *w = N.wh + (size_t) o * 256;
        float s = 0.f;
        for (int c = 0; c < 256; c++) s += hb[row * HLD + c] * w[c];
        s += N.bh[o];
        xs[row * XLD + o] = s;  // pre-activation heads parked in the idle staging tile; 2A <= 32 is checked on the host
    }
    __syncthreads();
    for (int a = part; a < A; a += 4) {
        if (gr >= n) continue;
        const float mu = tanhf(xs[row * XLD + a]);
        const float sigma = softplus_f(xs[row * XLD + A + a]);
        // truncated_normal_sample(mu, sigma, -1, 1)
        const float ss = fminf(fmaxf(sigma, 1e-6f), 1e6f);
        const float al = fminf(fmaxf((-1.f - mu) / ss, -5.f), 5.f);
        const float be = fminf(fmaxf((1.f - mu) / ss, -5.f), 5.f);
        const float ta = theta_f(al), tb = theta_f(be);
        const float u = uniform ? uniform[(size_t) gr * A + a] : rng_uniform(seed, counter, (uint32_t) gr, (uint32_t) a);
        const float cdf = fminf(fmaxf(ta + u * (tb - ta), 0.f), 1.f);
        const float inv = 1.41421356237309504880f * erfinvf(2.0f * cdf - 1.0f);
        const float act = fminf(fmaxf(inv * ss + mu, -1.f), 1.f);
        // truncated_normal_log_pdf(action, mu, sigma, -1, 1)
        const float z = tb - ta;
        const float q = (act - mu) / ss;
        const float lp = -0.91893853320467274178f - logf(ss) - 0.5f * (q * q) - logf(z);
        const size_t o = (size_t) gr * A + a;
        action[o] = act;
        logp[o] = lp;
        if (mu_out) mu_out[o] = mu;
        if (sigma_out) sigma_out[o] = sigma;
    }
}

size_t policy_lds_bytes() { return (size_t) (TM * HLD + TM * XLD) * sizeof(float); }

hipError_t launch_policy_forward(const PolicyDev &p, int n, const float *obs, const float *uniform, uint64_t seed,
                                 uint64_t counter, float *action, float *logp, float *value, float *mu, float *sigma,
                                 hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_policy_forward),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int) policy_lds_bytes());
        if (e != hipSuccess) return e;
        attr = true;
    }
    dim3 grid((n + TM - 1) / TM, 2);
    hipLaunchKernelGGL(k_policy_forward, grid, dim3(PT), policy_lds_bytes(), s, p, n, obs, uniform, seed, counter, action,
                       logp, value, mu, sigma);
    return hipGetLastError();
}

}  // namespace evm
