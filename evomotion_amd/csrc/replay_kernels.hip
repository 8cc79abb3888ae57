// Replay ring of the SAC rows (gfx950): streaming copies and gathers — HBM-bound byte work, no arithmetic.
//
// Replaces, for N environments at once (reference paths relative to its repo root):
//   ReplayBuffer::add / update_last    evo_motion_networks/src/replay_buffer.cpp:31-42,146-153
//   AbstractReplayBuffer::sample       evo_motion_networks/src/replay_buffer.cpp:16-28
//   the stacking of the sampled batch  evo_motion_networks/src/agents/soft_actor_critic.cpp:68-87
//
// push:   k_replay_copy: grid-stride float4 copies of the [N, S] / [N, A] blocks into the slot, coalesced; its last block
//         does the ordered compaction of the valid rows of the new slot beside them (one launch)
// sample: k_replay_plan (per workgroup: prefix sum of the per-slot counts over the live slots in age order, then one draw per
//         thread: keyed permutation rank -> (slot, env) by binary search) + k_replay_gather (one wave per drawn row:
//         1 484-byte contiguous reads of state and next state)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "replay_dev.h"

namespace evm {

#define RP_T 1024

// ordered compaction of the slot's transitions: valid_idx[slot][k] = k-th env (ascending) with valid == 1.  One workgroup of NT
// threads (the last block of k_replay_copy: the copies of the other blocks run beside its serial passes).
template <int NT>
__device__ __forceinline__ void replay_index_block(const ReplayDev &d, int slot, const uint8_t *__restrict__ valid) {
    __shared__ int wave_sum[NT / 64];
    __shared__ int base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    int *out = d.valid_idx + (size_t) slot * d.N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i0 = 0; i0 < d.N; i0 += NT) {
        const int i = i0 + (int) threadIdx.x;
        const bool v = i < d.N && (valid ? valid[i] == 1 : true);
        const unsigned long long m = __ballot(v);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_sum[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; w++) off += wave_sum[w];
        if (v) out[off + before] = i;
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
            for (int w = 0; w < NT / 64; w++) t += wave_sum[w];
            base += t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) d.slot_count[slot] = base;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void copy_block(float *dst, const float *src, size_t n, size_t tid, size_t nthreads) {
    // both pointers are 16-byte aligned when n % 4 == 0 blocks start on allocation boundaries; checked by the caller
    const size_t n4 = n >> 2;
    const f32x4 *s4 = reinterpret_cast<const f32x4 *>(src);
    f32x4 *d4 = reinterpret_cast<f32x4 *>(dst);
    for (size_t i = tid; i < n4; i += nthreads) d4[i] = s4[i];
    for (size_t i = (n4 << 2) + tid; i < n; i += nthreads) dst[i] = src[i];
}
__device__ __forceinline__ void copy_block_scalar(float *dst, const float *src, size_t n, size_t tid, size_t nthreads) {
    for (size_t i = tid; i < n; i += nthreads) dst[i] = src[i];
}

__global__ __launch_bounds__(RP_T) void k_replay_copy(ReplayDev d, int slot, const float *__restrict__ state,
                                                     const float *__restrict__ action, const float *__restrict__ reward,
                                                     const uint8_t *__restrict__ done, const float *__restrict__ next_state,
                                                     const uint8_t *__restrict__ valid, int aligned) {
    if (blockIdx.x == gridDim.x - 1) { replay_index_block<RP_T>(d, slot, valid); return; }  // (uniform per block)
    const size_t tid = blockIdx.x * (size_t) blockDim.x + threadIdx.x, nt = (size_t) (gridDim.x - 1) * blockDim.x;
    const size_t ns = (size_t) d.N * d.S, na = (size_t) d.N * d.A;
    if (aligned) {
        copy_block(d.state + (size_t) slot * ns, state, ns, tid, nt);
        copy_block(d.pending, next_state, ns, tid, nt);
        copy_block(d.action + (size_t) slot * na, action, na, tid, nt);
    } else {
        copy_block_scalar(d.state + (size_t) slot * ns, state, ns, tid, nt);
        copy_block_scalar(d.pending, next_state, ns, tid, nt);
        copy_block_scalar(d.action + (size_t) slot * na, action, na, tid, nt);
    }
    for (size_t i = tid; i < (size_t) d.N; i += nt) {
        d.reward[(size_t) slot * d.N + i] = reward[i];
        d.done[(size_t) slot * d.N + i] = done[i] ? 1.f : 0.f;
    }
}

// live slots in age order: j = 0 oldest ... live-1 newest; physical slot = (head - live + j) mod C
__device__ __forceinline__ int phys_slot(int head, int live, int C, int j) { return (head - live + j + C) % C; }

__global__ __launch_bounds__(RP_T) void k_replay_plan(ReplayDev d, int head, int live, int batch, uint64_t seed) {
    extern __shared__ int prefix[];  // [live + 1] exclusive prefix sums of the slot counts in age order
    __shared__ int wave_tot[RP_T / 64];
    __shared__ int carry;
    if (threadIdx.x == 0) { carry = 0; prefix[0] = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j0 = 0; j0 < live; j0 += RP_T) {
        const int j = j0 + (int) threadIdx.x;
        int v = j < live ? d.slot_count[phys_slot(head, live, d.C, j)] : 0;
        int incl = v;  // inclusive scan inside the wave
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int off = carry;
        for (int w = 0; w < wave; w++) off += wave_tot[w];
        if (j < live) prefix[j + 1] = off + incl;
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
            for (int w = 0; w < RP_T / 64; w++) t += wave_tot[w];
            carry += t;
        }
        __syncthreads();
    }
    const int m = prefix[live];
    if (threadIdx.x == 0 && blockIdx.x == 0) d.total[0] = m;
    // one draw per thread and trip; the grid's blocks (each with its own copy of the prefix sums: one pass over the slot counts)
    // share the draws — as ONE block the 4 096 draws of a batch were four dependent trips of rank, binary search and a global
    // read per thread: 13.8 us, longer than the gather they feed
    for (int b = blockIdx.x * RP_T + threadIdx.x; b < batch; b += gridDim.x * RP_T) {
        int slot = -1, env = -1;
        if (m > 0) {
            const uint32_t r = replay_rank((uint32_t) (b % m), (uint32_t) m, seed);
            int lo = 0, hi = live;  // largest j with prefix[j] <= r
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (prefix[mid] <= (int) r) lo = mid; else hi = mid;
            }
            slot = phys_slot(head, live, d.C, lo);
            env = d.valid_idx[(size_t) slot * d.N + ((int) r - prefix[lo])];
        }
        d.plan[2 * b] = slot;
        d.plan[2 * b + 1] = env;
    }
}

// one wave per drawn row
__global__ __launch_bounds__(256) void k_replay_gather(ReplayDev d, int head, int batch, float *__restrict__ states,
                                                       float *__restrict__ actions, float *__restrict__ rewards,
                                                       float *__restrict__ done, float *__restrict__ next_states,
                                                       int *__restrict__ index) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= batch) return;
    const int slot = d.plan[2 * b], env = d.plan[2 * b + 1];
    float *so = states + (size_t) b * d.S, *no = next_states + (size_t) b * d.S, *ao = actions + (size_t) b * d.A;
    if (slot < 0) {  // nothing stored: zeros
        for (int k = lane; k < d.S; k += 64) { so[k] = 0.f; no[k] = 0.f; }
        for (int k = lane; k < d.A; k += 64) ao[k] = 0.f;
        if (lane == 0) { rewards[b] = 0.f; done[b] = 0.f; if (index) { index[2 * b] = -1; index[2 * b + 1] = -1; } }
        return;
    }
    const int newest = (head - 1 + d.C) % d.C;
    const float *sp = d.state + ((size_t) slot * d.N + env) * d.S;
    const float *np = slot == newest ? d.pending + (size_t) env * d.S : d.state + ((size_t) ((slot + 1) % d.C) * d.N + env) * d.S;
    const float *ap = d.action + ((size_t) slot * d.N + env) * d.A;
    // every load of the row in flight before its first store (as a load / store loop the compiler, which cannot rule out that the
    // outputs alias the ring, waits for each pair: six dependent round trips per row instead of one)
    constexpr int RP_KMAX = 8;  // 64 * 8 = 512 floats per row cover S <= 512 in one batch
    float vs[RP_KMAX], vn[RP_KMAX];
    for (int k0 = 0; k0 < d.S; k0 += 64 * RP_KMAX) {
#pragma unroll
        for (int u = 0; u < RP_KMAX; u++) {
            const int k = k0 + 64 * u + lane;
            const int kc = k < d.S ? k : d.S - 1;
            vs[u] = sp[kc]; vn[u] = np[kc];
        }
        float va = 0.f, vr = 0.f, vd = 0.f;
        if (k0 == 0) {
            if (lane < d.A) va = ap[lane];
            if (lane == 0) { vr = d.reward[(size_t) slot * d.N + env]; vd = d.done[(size_t) slot * d.N + env]; }
        }
#pragma unroll
        for (int u = 0; u < RP_KMAX; u++) {
            const int k = k0 + 64 * u + lane;
            if (k < d.S) { so[k] = vs[u]; no[k] = vn[u]; }
        }
        if (k0 == 0) {
            if (lane < d.A) ao[lane] = va;
            if (lane == 0) { rewards[b] = vr; done[b] = vd; }
        }
    }
    for (int k = 64 + lane; k < d.A; k += 64) ao[k] = ap[k];  // (A > 64: not the case of any shipped skeleton)
    if (lane == 0 && index) { index[2 * b] = slot; index[2 * b + 1] = env; }
}

hipError_t launch_replay_push(const ReplayDev &d, int slot, const float *state, const float *action, const float *reward,
                              const uint8_t *done, const uint8_t *valid, const float *next_state, hipStream_t s) {
    const size_t ns = (size_t) d.N * d.S, na = (size_t) d.N * d.A;
    const int aligned = ((uintptr_t) state % 16 == 0) && ((uintptr_t) next_state % 16 == 0) && ((uintptr_t) action % 16 == 0) &&
                        (ns * sizeof(float)) % 16 == 0 && (na * sizeof(float)) % 16 == 0;
    int blocks = (int) ((ns / 4 + RP_T - 1) / RP_T);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    // + 1: the block that compacts the slot's valid rows (replay_index_block)
    hipLaunchKernelGGL(k_replay_copy, dim3(blocks + 1), dim3(RP_T), 0, s, d, slot, state, action, reward, done, next_state, valid, aligned);
    return hipGetLastError();
}

hipError_t launch_replay_sample(const ReplayDev &d, int head, int live, int batch, uint64_t seed, float *states, float *actions,
                                float *rewards, float *done, float *next_states, int *index, hipStream_t s) {
    int plan_blocks = (batch + RP_T - 1) / RP_T;
    if (plan_blocks < 1) plan_blocks = 1;
    if (plan_blocks > 64) plan_blocks = 64;
    hipLaunchKernelGGL(k_replay_plan, dim3(plan_blocks), dim3(RP_T), (size_t) (live + 1) * sizeof(int), s, d, head, live, batch, seed);
    hipLaunchKernelGGL(k_replay_gather, dim3((batch + 3) / 4), dim3(256), 0, s, d, head, batch, states, actions, rewards, done,
                       next_states, index);
    return hipGetLastError();
}

}  // namespace evm
