// Device view of the replay ring (replay_kernels.hip / replay_host.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace evm {

struct ReplayDev {
    int C, N, S, A;        // slots, envs, state width, action width
    float *state;          // [C][N][S]
    float *action;         // [C][N][A]
    float *reward;         // [C][N]
    float *done;           // [C][N]   1.0 / 0.0 (the batch tensor of soft_actor_critic.cpp:78-80)
    float *pending;        // [N][S]   next state of the newest slot
    int *valid_idx;        // [C][N]   env indices of the slot's transitions (valid == 1), ascending
    int *slot_count;       // [C]      number of transitions in the slot
    int *plan;             // [max_batch][2] (slot, env) of the current draw
    long long *total;      // [1]      stored transitions, written by the sampling plan (for evm_replay_stats)
};

hipError_t launch_replay_push(const ReplayDev &d, int slot, const float *state, const float *action, const float *reward,
                              const uint8_t *done, const uint8_t *valid, const float *next_state, hipStream_t s);
hipError_t launch_replay_sample(const ReplayDev &d, int head, int live, int batch, uint64_t seed, float *states, float *actions,
                                float *rewards, float *done, float *next_states, int *index, hipStream_t s);

// the keyed permutation of [0, m): shared with the numpy oracle (oracle/replay_oracle.py)
__host__ __device__ inline uint32_t replay_mix(uint32_t x, uint32_t key, uint32_t mask) {
    // three rounds of invertible mixing on the low bits selected by `mask` (a power of two minus one)
    for (int r = 0; r < 3; r++) {
        x = (x * 0x9E3779B1u + key) & mask;   // odd multiplier: a bijection modulo 2^k
        x ^= x >> 7;                          // xorshift: a bijection on k-bit words (shift < k guarded by the mask)
        x &= mask;
        x = (x * 0x85EBCA6Bu + (key >> 16) + r) & mask;
        x ^= x >> 11;
        x &= mask;
    }
    return x;
}
__host__ __device__ inline uint32_t replay_rank(uint32_t b, uint32_t m, uint64_t seed) {
    // b-th element of a permutation of [0, m) (cycle walking over the next power of two); b < m
    uint32_t mask = 1;
    while (mask < m) mask <<= 1;
    mask -= 1;
    const uint32_t key = (uint32_t) (seed ^ (seed >> 32)) * 0x27D4EB2Fu + 0x165667B1u;
    uint32_t x = b;
    do { x = replay_mix(x, key, mask); } while (x >= m);
    return x;
}

}  // namespace evm
