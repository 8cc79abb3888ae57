// Fused actor-critic forward + truncated-normal sampling for the PPO rollout (gfx950, fp32 MFMA).
//
// Replaces, per environment (reference paths relative to its repo root):
//   PpoGaeAgent::act                     evo_motion_networks/src/agents/ppo_gae.cpp:29-45
//   ActorModule / CriticModule forward   evo_motion_networks/src/networks/actor.cpp:30-48, critic.cpp:23-35
//   truncated_normal_sample / _log_pdf   evo_motion_networks/src/functions.cpp:53-68,94-111
//
// One workgroup (4 waves) owns a TM-row tile of the batch and ONE of the two networks (blockIdx.y: 0 actor,
// 1 critic).  Both hidden layers run on v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain): each wave
// owns TM rows x 64 output columns as RT x 2 accumulator tiles.  A operands come from LDS (observations staged
// once, activations in a second tile, both "k-split" so a lane reads four k-steps with one ds_read_b128), B
// operands are read straight from host-packed weights (one global_load_dwordx4 per lane per four k-steps,
// 1 KiB contiguous per wave instruction, L2 resident).  Bias, Mish,
// LayerNorm, the tanh/softplus heads, inverse-CDF sampling and the log-pdf are fused behind the GEMMs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "policy_dev.h"
#include "mlp_tile.h"

namespace evm {

#define RT 1          // 32-row MFMA tiles per wave: a workgroup owns TM = 32 RT rows.  RT = 1 puts the 4096-env batch
                      // of BASELINE configs[2] on 2 x 128 = 256 workgroups, one per CU (RT = 2 left half the chip idle)
#define TM (32 * RT)
#define PARTS (PT / TM)   // threads sharing a row in the row-wise epilogues
#define RUN (256 / PARTS) // stored activations per thread (a contiguous run of the k-split row)
// LDS floats of the two overlaid lives of the tile buffer: observations | activations + head partials + head pre-activations
#define POLICY_TILE_FLOATS (TM * ALD1 > TM * ALD2 + 4 * 32 * 32 + TM * 32 ? TM * ALD1 : TM * ALD2 + 4 * 32 * 32 + TM * 32)

__device__ __forceinline__ float rng_uniform(uint64_t seed, uint64_t counter, uint32_t row, uint32_t dim) {
    // counter-based (stateless) generator: splitmix64 finaliser over (seed, counter, row, dim) -> 24-bit uniform [0,1)
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (counter + 1) + ((uint64_t) row << 20) + dim;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float) (uint32_t) (z >> 40) * (1.0f / 16777216.0f);
}

template <int SPLIT>  // 0: fp32 MFMA layers, 1: six bf16 products per fp32 product (dense_layer_split)
__global__ __launch_bounds__(PT) void k_policy_forward(PolicyDev p, int n, const float *__restrict__ obs,
                                                       const float *__restrict__ uniform, uint64_t seed, uint64_t counter,
                                                       float *action, float *logp, float *value, float *mu_out,
                                                       float *sigma_out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    // one buffer, two lives: the observation tile [TM][ALD1] during layer 1, then the activation tile [TM][ALD2]
    // followed by the head GEMM's partial tiles and the head pre-activations [TM][32]; behind both, the LayerNorm
    // statistics exchange of mish_ln_epilogue
    float *xs = sm;
    float *hb = sm;
    float *red = sm + POLICY_TILE_FLOATS;
    const int net = blockIdx.y;    // 0 actor, 1 critic
    const int row0 = blockIdx.x * TM;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const NetDev &N = net == 0 ? p.actor : p.critic;
#ifdef EVM_PSTAMPS  // diagnostic build (tools/pstamps.py): phase clock stamps of wave 0, written over the mu output
    unsigned long long ps_t[8];
#define PSTAMP(i) ps_t[i] = __builtin_amdgcn_s_memtime();
#else
#define PSTAMP(i)
#endif
    PSTAMP(0)
    stage_rows_ksplit<TM>(xs, obs, row0, n, p.S);
    __syncthreads();
    PSTAMP(1)
    f32x16 acc[RT][2];
    if (SPLIT) dense_layer_split<K1>(xs, ALD1, N.w1s, wave, lane, acc);
    else dense_layer<K1, RT>(xs, ALD1, N.w1t, wave, lane, acc);
    PSTAMP(2)
    mish_ln_epilogue(acc[0], N.b1, N.g1, N.be1, hb, red, wave, lane, row0, n, nullptr, nullptr, nullptr, 0, 0);
    PSTAMP(3)
    if (SPLIT) dense_layer_split<256>(hb, ALD2, N.w2s, wave, lane, acc);
    else dense_layer<256, RT>(hb, ALD2, N.w2t, wave, lane, acc);
    PSTAMP(4)
    mish_ln_epilogue(acc[0], N.b2, N.g2, N.be2, hb, red, wave, lane, row0, n, nullptr, nullptr, nullptr, 0, 0);
    PSTAMP(5)

    // heads: Linear(256, 1) for the critic, Linear(256, A) x 2 (mu, sigma) for the actor
    const int t = threadIdx.x, row = t / PARTS, part = t % PARTS;
    const int gr = row0 + row;
    const int A = p.A;
    const int nout = net == 1 ? 1 : 2 * A;
    // The head GEMM [TM x 256] x [256 x 32] on the matrix pipe too (as a row-wise dot product loop it was a quarter of the
    // kernel): K is split over the four waves, the four partial 32 x 32 tiles meet in LDS.
    static_assert(TM == 32, "the head GEMM is one 32-row MFMA tile");
    float *hs4 = sm + TM * ALD2;     // [4 waves][32 rows][32 cols] partial sums
    float *hs = hs4 + 4 * 32 * 32;   // [TM][32] pre-activations
    head_gemm(hb, N.whp, hs4, wave, lane);
    __syncthreads();
    for (int o = part; o < nout; o += PARTS)
        hs[row * 32 + o] = ((hs4[row * 32 + o] + hs4[(32 + row) * 32 + o]) + (hs4[(64 + row) * 32 + o] + hs4[(96 + row) * 32 + o])) + N.bh[o];
    __syncthreads();
    PSTAMP(6)
#ifdef EVM_PSTAMPS
    if (net == 0 && mu_out && threadIdx.x == 0 && blockIdx.x < 128) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(mu_out) + (size_t) blockIdx.x * 8;
        for (int i = 0; i < 7; i++) o[i] = ps_t[i];
    }
#endif
    if (net == 1) {
        if (part == 0 && gr < n) value[gr] = hs[row * 32];
        return;
    }
    for (int a = part; a < A; a += PARTS) {
        if (gr >= n) continue;
        const float mu = tanhf(hs[row * 32 + a]);
        const float sigma = softplus_f(hs[row * 32 + A + a]);
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
#ifndef EVM_PSTAMPS
        if (mu_out) mu_out[o] = mu;
#endif
        if (sigma_out) sigma_out[o] = sigma;
    }
}

// The same forward on 16-row tiles (v_mfma_f32_16x16x4_f32, mlp_tile.h): twice the workgroups of half the size.  Chosen by
// launch_policy_forward while the 32-row grid has no more than one workgroup per CU (the 4096-row rollout: 256; SAC's
// actor-only act(): 128 workgroups on 256 CUs).  Same operand layouts, same outputs to rounding (another k order inside the
// fp32 sums).
#define TM16 16
#define POLICY_TILE16_FLOATS (TM16 * ALD1 > TM16 * ALD2 + 4 * 16 * 32 + TM16 * 32 ? TM16 * ALD1 : TM16 * ALD2 + 4 * 16 * 32 + TM16 * 32)

__device__ __forceinline__ void sample_action(const PolicyDev &p, const float *hs, int row, int gr, int a,
                                              const float *__restrict__ uniform, uint64_t seed, uint64_t counter, float *action,
                                              float *logp, float *mu_out, float *sigma_out) {
    const int A = p.A;
    const float mu = tanhf(hs[row * 32 + a]);
    const float sigma = softplus_f(hs[row * 32 + A + a]);
    const float ss = fminf(fmaxf(sigma, 1e-6f), 1e6f);
    const float al = fminf(fmaxf((-1.f - mu) / ss, -5.f), 5.f);
    const float be = fminf(fmaxf((1.f - mu) / ss, -5.f), 5.f);
    const float ta = theta_f(al), tb = theta_f(be);
    const float u = uniform ? uniform[(size_t) gr * A + a] : rng_uniform(seed, counter, (uint32_t) gr, (uint32_t) a);
    const float cdf = fminf(fmaxf(ta + u * (tb - ta), 0.f), 1.f);
    const float inv = 1.41421356237309504880f * erfinvf(2.0f * cdf - 1.0f);
    const float act = fminf(fmaxf(inv * ss + mu, -1.f), 1.f);
    const float z = tb - ta;
    const float q = (act - mu) / ss;
    const float lp = -0.91893853320467274178f - logf(ss) - 0.5f * (q * q) - logf(z);
    const size_t o = (size_t) gr * A + a;
    action[o] = act;
    logp[o] = lp;
#ifndef EVM_PSTAMPS
    if (mu_out) mu_out[o] = mu;
#endif
    if (sigma_out) sigma_out[o] = sigma;
}

template <int SPLIT>  // 1: the hidden layers as six bf16 products per fp32 product (dense_layer16_split)
__global__ __launch_bounds__(PT) void k_policy_forward16(PolicyDev p, int n, const float *__restrict__ obs,
                                                         const float *__restrict__ uniform, uint64_t seed, uint64_t counter,
                                                         float *action, float *logp, float *value, float *mu_out,
                                                         float *sigma_out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *xs = sm;
    float *hb = sm;
    float *red = sm + POLICY_TILE16_FLOATS;
    const int net = blockIdx.y;
    const int row0 = blockIdx.x * TM16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const NetDev &N = net == 0 ? p.actor : p.critic;
    // every operand that does not depend on this workgroup's own results is requested a phase ahead of its use: the first
    // weight blocks of a layer before the phase that produces its A tile, the epilogue's vectors before the GEMM
    f32x4 ring[DEPTH16][4];
#ifdef EVM_PSTAMPS
    unsigned long long ps_t[8];
#endif
    // Operands that do not depend on the workgroup's own results (first weight blocks of a layer, the epilogue's vectors,
    // the head weights) can be requested a phase ahead of their use (-DEVM_PREFETCH16).  Measured with two workgroups per CU
    // (4096 rows, both networks): 33.7 us against 32.9 us without — the co-resident workgroup already covers those round
    // trips and the early requests only lengthen register lifetimes; actor only (one workgroup per CU): 20.5 against 20.7 us.
    // Default: every operand where it is used.
#ifdef EVM_PREFETCH16
#define EARLY(x) x
#define LATE(x)
#else
#define EARLY(x)
#define LATE(x) x
#endif
    PSTAMP(0)
    EARLY(dense16_prefetch(N.w1t, wave, lane, ring);)
    stage_rows_ksplit<TM16>(xs, obs, row0, n, p.S);
    LnParams16 P;
    EARLY(P = ln_params16(N.b1, N.g1, N.be1, wave, lane);)
    __syncthreads();
    PSTAMP(1)
    f32x4c acc[4];
    if (SPLIT) dense_layer16_split<K1>(xs, ALD1, N.w1s, wave, lane, acc);
    else {
        LATE(dense16_prefetch(N.w1t, wave, lane, ring);)
        dense_layer16<K1>(xs, ALD1, N.w1t, wave, lane, acc, ring);
    }
    PSTAMP(2)
    EARLY(dense16_prefetch(N.w2t, wave, lane, ring);)
    LATE(P = ln_params16(N.b1, N.g1, N.be1, wave, lane);)
    mish_ln_epilogue16(acc, P, hb, red, wave, lane);
    PSTAMP(3)
    EARLY(P = ln_params16(N.b2, N.g2, N.be2, wave, lane);)
    if (SPLIT) dense_layer16_split<256>(hb, ALD2, N.w2s, wave, lane, acc);
    else {
        LATE(dense16_prefetch(N.w2t, wave, lane, ring);)
        dense_layer16<256>(hb, ALD2, N.w2t, wave, lane, acc, ring);
    }
    PSTAMP(4)
    HeadB16 HB;
    EARLY(HB = head16_prefetch(N.whp, wave, lane);)
    LATE(P = ln_params16(N.b2, N.g2, N.be2, wave, lane);)
    mish_ln_epilogue16(acc, P, hb, red, wave, lane);
    LATE(HB = head16_prefetch(N.whp, wave, lane);)
    PSTAMP(5)
#undef EARLY
#undef LATE
    float *hs4 = sm + TM16 * ALD2;   // [4 waves][16 rows][32 cols] partial sums
    float *hs = hs4 + 4 * 16 * 32;   // [16][32] pre-activations
    head_gemm16(hb, HB, hs4, wave, lane);
    __syncthreads();
    const int A = p.A;
    const int nout = net == 1 ? 1 : 2 * A;
    for (int e = threadIdx.x; e < TM16 * 32; e += PT) {
        const int row = e >> 5, o = e & 31;
        if (o < nout)
            hs[e] = ((hs4[e] + hs4[16 * 32 + e]) + (hs4[2 * 16 * 32 + e] + hs4[3 * 16 * 32 + e])) + N.bh[o];
        (void) row;
    }
    __syncthreads();
    PSTAMP(6)
#ifdef EVM_PSTAMPS
    if (net == 0 && mu_out && threadIdx.x == 0 && blockIdx.x < 256) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(mu_out) + (size_t) blockIdx.x * 8;
        for (int i = 0; i < 7; i++) o[i] = ps_t[i];
    }
#endif
    if (net == 1) {
        if (threadIdx.x < TM16 && row0 + (int) threadIdx.x < n) value[row0 + threadIdx.x] = hs[threadIdx.x * 32];
        return;
    }
    for (int e = threadIdx.x; e < TM16 * A; e += PT) {  // (row, action) pairs dealt flat: 384 over 256 threads at A = 24
        const int row = e / A, a = e - row * A;
        if (row0 + row < n) sample_action(p, hs, row, row0 + row, a, uniform, seed, counter, action, logp, mu_out, sigma_out);
    }
}

size_t policy_lds16_bytes() { return (size_t) (POLICY_TILE16_FLOATS + EVM_RED16_FLOATS) * sizeof(float); }

// flat parameters of one network -> the operand layout of the forward kernel; one thread per source element
__global__ __launch_bounds__(256) void k_policy_pack(NetDev n, int S, int A, int actor, const float *__restrict__ flat) {
    const size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    const size_t total = (size_t) 256 * S + 3 * 256 + 256 * 256 + 3 * 256 + (actor ? (size_t) 2 * A * 256 + 2 * A : 257);
    if (i < total) policy_pack_write(n, S, A, actor, i, flat[i]);
}
hipError_t launch_policy_pack(const NetDev &n, int S, int A, bool actor, const float *flat, hipStream_t s) {
    const size_t total = (size_t) 256 * S + 3 * 256 + 256 * 256 + 3 * 256 + (actor ? (size_t) 2 * A * 256 + 2 * A : 257);
    hipLaunchKernelGGL(k_policy_pack, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, s, n, S, A, actor ? 1 : 0, flat);
    return hipGetLastError();
}

// hipFuncSetAttribute is per DEVICE: a process that creates policies / trainers on a second device must set it there too
static bool &evm_attr_done_for_current_device() {
    static bool done[64] = {};
    int dev = 0;
    (void) hipGetDevice(&dev);
    return done[dev >= 0 && dev < 64 ? dev : 0];
}

size_t policy_lds_bytes() { return (size_t) (POLICY_TILE_FLOATS + EVM_RED_FLOATS) * sizeof(float); }

// Form of a launch.  32-row tiles run their layers as six bf16 MFMA products per fp32 product (dense_layer_split: 28.9 us for both
// networks at 4096 rows against 35.2 us on the fp32 MFMA) unless gemm == 0; 16-row tiles (fp32 MFMA) take over while the 32-row grid
// would leave CUs WITHOUT a workgroup (actor only at 4096 rows: 20.7 against 26.8 us; 2048 rows: 21 us) — with fp32 MFMA layers
// (gemm == 0) already when it has no more than one per CU (33.0 against 35.2 us).  EVM_POLICY_TILE = 16 | 32 (read once) or
// evm_policy_set_tile_rows force a tile height, for measurements and tests.
static int policy_tile_rows(int n, int nets, int asked, int gemm) {
    static int forced = -1, cus[64] = {};
    if (asked == 16 || asked == 32) return asked;
    if (forced < 0) {
        const char *e = getenv("EVM_POLICY_TILE");
        forced = e ? atoi(e) : 0;
    }
    if (forced == 16 || forced == 32) return forced;
    int dev = 0;
    (void) hipGetDevice(&dev);
    dev = dev >= 0 && dev < 64 ? dev : 0;
    if (!cus[dev]) {
        int c = 0;
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c < 1) c = 256;
        cus[dev] = c;
    }
    const int wg32 = ((n + TM - 1) / TM) * nets;
    return (gemm == 1 ? wg32 < cus[dev] : wg32 <= cus[dev]) ? 16 : 32;
}

hipError_t launch_policy_forward(const PolicyDev &p, int n, const float *obs, const float *uniform, uint64_t seed,
                                 uint64_t counter, float *action, float *logp, float *value, float *mu, float *sigma,
                                 hipStream_t s, int tile_rows, int gemm) {
    bool &attr = evm_attr_done_for_current_device();
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_policy_forward<0>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int) policy_lds_bytes());
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_policy_forward<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int) policy_lds_bytes());
        if (e != hipSuccess) return e;
        attr = true;
    }
    const int nets = value ? 2 : 1;  // value == NULL: the critic network is not run (SAC's act)
    if (policy_tile_rows(n, nets, tile_rows, gemm) == 16) {
        // EVM_POLICY_SPLIT16=1 (read once): the 16-row form on bf16-split layers too.  Measured at 4096 rows: actor only 20.2 against
        // 20.6 us, both networks 29.7 against 32.5 us (the 32-row split form: 28.6); 2048 rows, both networks 20.7 against 21.2 us —
        // with 16-row tiles every CU reads every weight plane twice as often and the time left is L2 traffic and the non-GEMM phases.
        // Off by default: half a microsecond does not buy a second default path; goldens hold either way.
        static int split16 = -1;
        if (split16 < 0) { const char *e = getenv("EVM_POLICY_SPLIT16"); split16 = e ? atoi(e) : 0; }
        if (split16 && gemm == 1)
            hipLaunchKernelGGL(k_policy_forward16<1>, dim3((n + TM16 - 1) / TM16, nets), dim3(PT), policy_lds16_bytes(), s, p, n, obs,
                               uniform, seed, counter, action, logp, value, mu, sigma);
        else
            hipLaunchKernelGGL(k_policy_forward16<0>, dim3((n + TM16 - 1) / TM16, nets), dim3(PT), policy_lds16_bytes(), s, p, n, obs,
                               uniform, seed, counter, action, logp, value, mu, sigma);
        return hipGetLastError();
    }
    dim3 grid((n + TM - 1) / TM, nets);
    if (gemm == 1)
        hipLaunchKernelGGL(k_policy_forward<1>, grid, dim3(PT), policy_lds_bytes(), s, p, n, obs, uniform, seed, counter, action,
                           logp, value, mu, sigma);
    else
        hipLaunchKernelGGL(k_policy_forward<0>, grid, dim3(PT), policy_lds_bytes(), s, p, n, obs, uniform, seed, counter, action,
                           logp, value, mu, sigma);
    return hipGetLastError();
}

}  // namespace evm
