// Device-side view of the actor / critic weights used by policy_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>
#include <vector>

namespace evm {

struct NetDev {
    const float *w1t;  // Linear(S, 256) weight packed [K1/8][256][2][4] (zero padded to K1 = 384), see policy_kernels.hip
    const float *b1, *g1, *be1;  // bias, LayerNorm weight / bias
    const float *w2t;  // Linear(256, 256) weight packed [32][256][2][4]
    const float *b2, *g2, *be2;
    const float *wh;   // heads, row major [out][256]: actor = mu rows then sigma rows (2A), critic = 1 row
    const float *bh;
    const float *whp;  // the same head weights as an MFMA B operand, 32 columns (rows of wh, zero beyond): [32][32][2][4]
    // The hidden layers' weights once more as THREE bf16 planes per value (w = p0 + p1 + p2 exactly: the 24-bit significand cut into
    // 8-bit pieces), the B operand of v_mfma_f32_32x32x16_bf16 in dense_layer_split (mlp_tile.h):
    //   [K / 16 blocks][256 cols][2 k groups g][3 planes][8 bf16] with k = 16 b + 2 i + g for the i-th bf16 of group g
    // (the k order of the fp32 k-split activation tile: a lane's eight A values are eight consecutive floats of half g)
    const uint16_t *w1s, *w2s;
};
// fp32 -> three bf16 planes by truncation (exact)
__host__ __device__ inline void bf16_split3(float x, uint16_t &p0, uint16_t &p1, uint16_t &p2) {
    union { float f; uint32_t u; } a, b, c, t;
    a.f = x; t.u = a.u & 0xffff0000u; p0 = (uint16_t) (t.u >> 16);
    b.f = x - t.f; t.u = b.u & 0xffff0000u; p1 = (uint16_t) (t.u >> 16);
    c.f = b.f - t.f; p2 = (uint16_t) (c.u >> 16);
}
__host__ __device__ inline size_t split_index(int col, int k) {  // element offset (uint16) of plane 0; planes are 8 elements apart
    const int b = k >> 4, g = k & 1, i = (k & 15) >> 1;
    return ((((size_t) b * 256 + col) * 2 + g) * 3) * 8 + i;
}
struct PolicyDev {
    int S, A, K1pad;
    NetDev actor, critic;
};

#ifdef __HIPCC__
// One flat parameter (index i of the network's named_parameters() vector, value v) -> every place the forward kernels read it
// from: the k-split GEMM operands w1t / w2t, the bias / LayerNorm vectors, the head weights row major (wh) and as the head
// GEMM's B operand (whp: [32 s4][32 cols][2][4], column = head row).  The same mapping as the host packer in policy_host.cpp.
// Used by k_policy_pack and, fused behind the Adam step, by the trainers (one launch less per network and step).
__device__ __forceinline__ void policy_pack_write(const NetDev &n, int S, int A, int actor, size_t i, float v) {
    const size_t n_w1 = (size_t) 256 * S, n_w2 = 256 * 256;
    size_t o = 0;
    auto wr = [](const float *p) { return const_cast<float *>(p); };
    if (i < n_w1) {  // head.0.weight [256][S]
        const int col = (int) (i / S), k = (int) (i % S);
        const int st = k >> 1, h = k & 1, s4 = st >> 2, t = st & 3;
        wr(n.w1t)[(((size_t) s4 * 256 + col) * 2 + h) * 4 + t] = v;
        if (n.w1s) {
            uint16_t *ws = const_cast<uint16_t *>(n.w1s) + split_index(col, k);
            bf16_split3(v, ws[0], ws[8], ws[16]);
        }
        return;
    }
    o = n_w1;
    if (i < o + 256) { wr(n.b1)[i - o] = v; return; }
    o += 256;
    if (i < o + 256) { wr(n.g1)[i - o] = v; return; }
    o += 256;
    if (i < o + 256) { wr(n.be1)[i - o] = v; return; }
    o += 256;
    if (i < o + n_w2) {
        const size_t j = i - o;
        const int col = (int) (j / 256), k = (int) (j % 256);
        const int st = k >> 1, h = k & 1, s4 = st >> 2, t = st & 3;
        wr(n.w2t)[(((size_t) s4 * 256 + col) * 2 + h) * 4 + t] = v;
        if (n.w2s) {
            uint16_t *ws = const_cast<uint16_t *>(n.w2s) + split_index(col, k);
            bf16_split3(v, ws[0], ws[8], ws[16]);
        }
        return;
    }
    o += n_w2;
    if (i < o + 256) { wr(n.b2)[i - o] = v; return; }
    o += 256;
    if (i < o + 256) { wr(n.g2)[i - o] = v; return; }
    o += 256;
    if (i < o + 256) { wr(n.be2)[i - o] = v; return; }
    o += 256;
    auto head = [&](size_t e, float x) {  // e = row * 256 + k
        wr(n.wh)[e] = x;
        const int row = (int) (e >> 8), k = (int) (e & 255);
        const int st = k >> 1, h = k & 1, s4 = st >> 2, t = st & 3;
        wr(n.whp)[(((size_t) s4 * 32 + row) * 2 + h) * 4 + t] = x;
    };
    if (actor) {
        const size_t hw = (size_t) A * 256;
        if (i < o + hw) { head(i - o, v); return; }                  // mu.0.weight
        o += hw;
        if (i < o + A) { wr(n.bh)[i - o] = v; return; }              // mu.0.bias
        o += A;
        if (i < o + hw) { head(hw + (i - o), v); return; }           // sigma.0.weight
        o += hw;
        if (i < o + A) { wr(n.bh)[A + (i - o)] = v; return; }        // sigma.0.bias
    } else {
        if (i < o + 256) { head(i - o, v); return; }
        o += 256;
        if (i < o + 1) { wr(n.bh)[0] = v; return; }
    }
}
#endif

size_t policy_lds_bytes();
// device-side repack of one network's flat parameters (named_parameters order) into the kernel's operand layout
hipError_t launch_policy_pack(const NetDev &n, int S, int A, bool actor, const float *flat, hipStream_t s);
hipError_t launch_policy_forward(const PolicyDev &p, int n, const float *obs, const float *uniform, uint64_t seed,
                                 uint64_t counter, float *action, float *logp, float *value, float *mu, float *sigma,
                                 hipStream_t s, int tile_rows = 0, int gemm = 0);

}  // namespace evm

// the handle behind evm_policy_* (policy_host.cpp); the PPO trainer (ppo_host.cpp) repacks its weights
struct EvmPolicy {
    int S, A, H, K1pad, device;
    int tile_rows;  // 0 = chosen per launch, 16 / 32 forced (evm_policy_set_tile_rows)
    int gemm;       // 0 = fp32 MFMA, 1 = six bf16 products per fp32 product (32-row form only; EVM_POLICY_SPLIT at creation)
    float *arena;
    size_t arena_floats;
    evm::PolicyDev dev;
    uint64_t counter;
    bool timing;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pairs;
    size_t ev_used;
};
