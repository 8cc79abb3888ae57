// C ABI of the device-resident replay ring (include/evomotion.h, evm_replay_*).
#include <hip/hip_runtime.h>

#include <string>
#include <utility>
#include <vector>

#include "../../include/evomotion.h"
#include "replay_dev.h"

namespace evm { void set_last_error(const std::string &m); }

struct EvmReplay {
    evm::ReplayDev d;
    int device;
    int head;            // next slot to write
    int live;            // slots holding data (<= C)
    long long pushes;
    int max_batch;
    std::vector<void *> allocs;
    bool timing;
    struct Ev { hipEvent_t a, b; int kind; };
    std::vector<Ev> evs;
    size_t ev_used;
};

static int rfail(int code, const std::string &m) { evm::set_last_error(m); return code; }

static int ev_begin(EvmReplay *rb, hipStream_t s, int kind) {
    if (!rb->timing) return EVM_OK;
    if (rb->ev_used == rb->evs.size()) {
        EvmReplay::Ev e;
        if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return rfail(EVM_E_HIP, "hipEventCreate failed");
        rb->evs.push_back(e);
    }
    rb->evs[rb->ev_used].kind = kind;
    (void) hipEventRecord(rb->evs[rb->ev_used].a, s);
    return EVM_OK;
}
static void ev_end(EvmReplay *rb, hipStream_t s) {
    if (!rb->timing) return;
    (void) hipEventRecord(rb->evs[rb->ev_used].b, s);
    rb->ev_used++;
}

extern "C" {

int evm_replay_create(int capacity_slots, int n_envs, int state_dim, int action_dim, int device, EvmReplay **out) {
    if (!out) return rfail(EVM_E_INVALID, "out is null");
    *out = nullptr;
    // k_replay_plan keeps (live slots + 1) prefix sums in dynamic LDS, + 36 bytes of static: stay inside the 64 KB a kernel gets
    // without hipFuncAttributeMaxDynamicSharedMemorySize
    if (capacity_slots < 1 || capacity_slots > 16000) return rfail(EVM_E_INVALID, "capacity_slots must be in [1, 16000]");
    if (n_envs < 1 || state_dim < 1 || action_dim < 1) return rfail(EVM_E_INVALID, "sizes must be positive");
    if (hipSetDevice(device) != hipSuccess) return rfail(EVM_E_HIP, "hipSetDevice failed");
    EvmReplay *rb = new EvmReplay();
    rb->device = device; rb->head = 0; rb->live = 0; rb->pushes = 0; rb->max_batch = 0; rb->timing = false; rb->ev_used = 0;
    evm::ReplayDev &d = rb->d;
    d.C = capacity_slots; d.N = n_envs; d.S = state_dim; d.A = action_dim; d.plan = nullptr;
    const size_t CN = (size_t) capacity_slots * n_envs;
    struct { void **p; size_t bytes; } req[] = {
        {(void **) &d.state, CN * state_dim * 4}, {(void **) &d.action, CN * action_dim * 4}, {(void **) &d.reward, CN * 4},
        {(void **) &d.done, CN * 4}, {(void **) &d.pending, (size_t) n_envs * state_dim * 4}, {(void **) &d.valid_idx, CN * 4},
        {(void **) &d.slot_count, (size_t) capacity_slots * 4}, {(void **) &d.total, 8}};
    for (auto &r : req) {
        if (hipMalloc(r.p, r.bytes) != hipSuccess) {
            for (void *a : rb->allocs) (void) hipFree(a);
            delete rb;
            return rfail(EVM_E_HIP, "hipMalloc failed (replay ring of " + std::to_string(CN * (state_dim + action_dim + 3) * 4 >> 20) + " MiB)");
        }
        rb->allocs.push_back(*r.p);
    }
    (void) hipMemset(d.slot_count, 0, (size_t) capacity_slots * 4);
    (void) hipMemset(d.total, 0, 8);
    *out = rb;
    return EVM_OK;
}

void evm_replay_destroy(EvmReplay *rb) {
    if (!rb) return;
    (void) hipSetDevice(rb->device);
    for (void *a : rb->allocs) (void) hipFree(a);
    if (rb->d.plan) (void) hipFree(rb->d.plan);
    for (auto &e : rb->evs) { (void) hipEventDestroy(e.a); (void) hipEventDestroy(e.b); }
    delete rb;
}

int evm_replay_push(EvmReplay *rb, const float *d_state, const float *d_action, const float *d_reward, const uint8_t *d_done,
                    const uint8_t *d_valid, const float *d_next_state, void *stream) {
    if (!rb || !d_state || !d_action || !d_reward || !d_done || !d_next_state) return rfail(EVM_E_INVALID, "null argument");
    hipStream_t s = (hipStream_t) stream;
    int rc = ev_begin(rb, s, 0);
    if (rc) return rc;
    hipError_t e = evm::launch_replay_push(rb->d, rb->head, d_state, d_action, d_reward, d_done, d_valid, d_next_state, s);
    if (e != hipSuccess) return rfail(EVM_E_HIP, std::string("replay push: ") + hipGetErrorString(e));
    ev_end(rb, s);
    rb->head = (rb->head + 1) % rb->d.C;
    if (rb->live < rb->d.C) rb->live++;
    rb->pushes++;
    return EVM_OK;
}

int evm_replay_sample(EvmReplay *rb, int batch, uint64_t seed, float *d_states, float *d_actions, float *d_rewards, float *d_done,
                      float *d_next_states, int *d_index, void *stream) {
    if (!rb || !d_states || !d_actions || !d_rewards || !d_done || !d_next_states) return rfail(EVM_E_INVALID, "null argument");
    if (batch < 1) return rfail(EVM_E_INVALID, "batch must be >= 1");
    if (rb->live == 0) return rfail(EVM_E_RUNTIME, "the replay memory is empty");
    hipStream_t s = (hipStream_t) stream;
    if (batch > rb->max_batch) {
        if (hipStreamSynchronize(s) != hipSuccess) return rfail(EVM_E_HIP, "stream sync failed");
        if (rb->d.plan) (void) hipFree(rb->d.plan);
        if (hipMalloc((void **) &rb->d.plan, (size_t) batch * 8) != hipSuccess) { rb->d.plan = nullptr; rb->max_batch = 0; return rfail(EVM_E_HIP, "hipMalloc failed"); }
        rb->max_batch = batch;
    }
    int rc = ev_begin(rb, s, 1);
    if (rc) return rc;
    hipError_t e = evm::launch_replay_sample(rb->d, rb->head, rb->live, batch, seed, d_states, d_actions, d_rewards, d_done,
                                             d_next_states, d_index, s);
    if (e != hipSuccess) return rfail(EVM_E_HIP, std::string("replay sample: ") + hipGetErrorString(e));
    ev_end(rb, s);
    return EVM_OK;
}

int evm_replay_stats(EvmReplay *rb, long long *h_out, void *stream) {
    if (!rb || !h_out) return rfail(EVM_E_INVALID, "null argument");
    hipStream_t s = (hipStream_t) stream;
    if (hipStreamSynchronize(s) != hipSuccess) return rfail(EVM_E_HIP, "stream sync failed");
    std::vector<int> cnt(rb->d.C);
    if (hipMemcpy(cnt.data(), rb->d.slot_count, (size_t) rb->d.C * 4, hipMemcpyDeviceToHost) != hipSuccess) return rfail(EVM_E_HIP, "hipMemcpy failed");
    long long tot = 0;
    for (int j = 0; j < rb->live; j++) tot += cnt[(rb->head - rb->live + j + rb->d.C) % rb->d.C];
    h_out[0] = tot; h_out[1] = rb->live; h_out[2] = rb->pushes;
    return EVM_OK;
}

int evm_replay_timing_begin(EvmReplay *rb) {
    if (!rb) return rfail(EVM_E_INVALID, "replay is null");
    rb->timing = true;
    rb->ev_used = 0;
    return EVM_OK;
}
int evm_replay_timing_end(EvmReplay *rb, void *stream, float *ms_push, int *n_push, float *ms_sample, int *n_sample) {
    if (!rb) return rfail(EVM_E_INVALID, "replay is null");
    if (hipStreamSynchronize((hipStream_t) stream) != hipSuccess) return rfail(EVM_E_HIP, "stream sync failed");
    float ms[2] = {0.f, 0.f};
    int n[2] = {0, 0};
    for (size_t i = 0; i < rb->ev_used; i++) {
        float t = 0.f;
        (void) hipEventElapsedTime(&t, rb->evs[i].a, rb->evs[i].b);
        ms[rb->evs[i].kind] += t;
        n[rb->evs[i].kind]++;
    }
    rb->timing = false;
    if (ms_push) *ms_push = ms[0];
    if (n_push) *n_push = n[0];
    if (ms_sample) *ms_sample = ms[1];
    if (n_sample) *n_sample = n[1];
    return EVM_OK;
}

}  // extern "C"
