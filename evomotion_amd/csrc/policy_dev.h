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
};
struct PolicyDev {
    int S, A, K1pad;
    NetDev actor, critic;
};

size_t policy_lds_bytes();
// device-side repack of one network's flat parameters (named_parameters order) into the kernel's operand layout
hipError_t launch_policy_pack(const NetDev &n, int S, int A, bool actor, const float *flat, hipStream_t s);
hipError_t launch_policy_forward(const PolicyDev &p, int n, const float *obs, const float *uniform, uint64_t seed,
                                 uint64_t counter, float *action, float *logp, float *value, float *mu, float *sigma,
                                 hipStream_t s);

}  // namespace evm

// the handle behind evm_policy_* (policy_host.cpp); the PPO trainer (ppo_host.cpp) repacks its weights
struct EvmPolicy {
    int S, A, H, K1pad, device;
    float *arena;
    size_t arena_floats;
    evm::PolicyDev dev;
    uint64_t counter;
    bool timing;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pairs;
    size_t ev_used;
};
