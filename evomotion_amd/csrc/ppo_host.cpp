// C ABI of the PPO / GAE update (include/evomotion.h, evm_ppo_*): the device-resident replacement of
// PpoGaeAgent::train (evo_motion_networks/src/agents/ppo_gae.cpp:117-190).
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/evomotion.h"
#include "ppo_dev.h"

namespace evm { void set_last_error(const std::string &m); }

struct EvmPpo {
    EvmPolicy *policy;  // borrowed: its packed weights are what the training forward reads, and what the rollout uses
    evm::PpoDev dev;
    std::vector<void *> allocs;
    bool have_params;
    hipEvent_t ev0, ev1;
    bool timing;
    float ms_acc;
    int n_timed;
    size_t staged_rows;  // rows of the observation copy made by the last evm_ppo_grads
    float *own_grads;    // the trainer's own contiguous [actor | critic] gradient vector (unless the caller supplied one)
    // evm_ppo_select_rows (allocated at its first call): row numbers, count, dense copies of the selected rows, a mask of ones
    int *sel_idx, *sel_count;
    float *sel_states, *sel_actions, *sel_logp, *sel_adv, *sel_returns;
    uint8_t *sel_mask;
};

static int qfail(int code, const std::string &m) { evm::set_last_error(m); return code; }

extern "C" {

int evm_ppo_create(EvmPolicy *policy, size_t max_rows, EvmPpo **out) {
    if (!out) return qfail(EVM_E_INVALID, "out is null");
    *out = nullptr;
    if (!policy) return qfail(EVM_E_INVALID, "policy is null");
    if (max_rows < 1 || max_rows > ((size_t) 1 << 30)) return qfail(EVM_E_INVALID, "max_rows out of range");
    if (hipSetDevice(policy->device) != hipSuccess) return qfail(EVM_E_HIP, "hipSetDevice failed");
    EvmPpo *q = new EvmPpo();
    q->policy = policy;
    q->have_params = false;
    q->timing = false; q->ms_acc = 0.f; q->n_timed = 0; q->staged_rows = 0;
    q->ev0 = q->ev1 = nullptr;
    q->sel_idx = q->sel_count = nullptr; q->sel_states = q->sel_actions = q->sel_logp = q->sel_adv = q->sel_returns = nullptr; q->sel_mask = nullptr;
    evm::PpoDev &d = q->dev;
    d.S = policy->S; d.A = policy->A; d.max_rows = max_rows;
    bool ok = true;
    auto alloc = [&](size_t bytes) -> void * {
        void *p = nullptr;
        if (!ok) return nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) { ok = false; return nullptr; }
        (void) hipMemset(p, 0, bytes);
        q->allocs.push_back(p);
        return p;
    };
    size_t na, nc;
    evm_policy_param_counts(policy, &na, &nc);
    const size_t tiles = (max_rows + 31) / 32;
    // both networks' gradients in ONE vector, actor first: a data-parallel caller reduces it with one collective
    const size_t goff = (na + 63) / 64 * 64;  // the critic's part starts on a 256-byte boundary
    q->own_grads = (float *) alloc((goff + nc) * 4);
    auto make = [&](evm::PpoNet &n, size_t np, size_t head_floats) {
        n.n_params = np; n.step = 0;
        n.theta = (float *) alloc(np * 4); n.grad = &n == &d.actor ? q->own_grads : (q->own_grads ? q->own_grads + goff : nullptr);
        n.m = (float *) alloc(np * 4); n.v = (float *) alloc(np * 4);
        n.w2d = (float *) alloc(65536 * 4);
        n.whd = (float *) alloc(8192 * 4);
        n.z1 = (float *) alloc(max_rows * 256 * 4); n.a1 = (float *) alloc(max_rows * 256 * 4);
        n.z2 = (float *) alloc(max_rows * 256 * 4); n.a2 = (float *) alloc(max_rows * 256 * 4);
        n.st = (float *) alloc(max_rows * 4 * 4);
        n.head = (float *) alloc(max_rows * head_floats * 4);
        n.dh = (float *) alloc(max_rows * 32 * 4);
        n.dz1 = (float *) alloc(max_rows * 256 * 4); n.dz2 = (float *) alloc(max_rows * 256 * 4);
        n.colpart = (float *) alloc(tiles * evm::PPO_COLSLOTS * 256 * 4);
        n.colpart2 = (float *) alloc((size_t) 64 * evm::PPO_COLSLOTS * 256 * 4);
        n.wpart = (float *) alloc(evm::ppo_wpart_floats() * 4);
        n.normp = (double *) alloc(evm::PPO_NORM_PARTS * sizeof(double));
    };
    make(d.actor, na, (size_t) 2 * d.A);
    make(d.critic, nc, 1);
    d.xpad = (float *) alloc(max_rows * 384 * 4);
    d.loss = (double *) alloc(2 * sizeof(double));
    d.loss_part = (double *) alloc(((max_rows * (size_t) d.A + 255) / 256 + (max_rows + 255) / 256) * sizeof(double));
    d.gae = (double *) alloc(3 * sizeof(double));
    d.gae_part = (double *) alloc(((max_rows + 255) / 256) * 3 * sizeof(double));
    d.step_dev = (int *) alloc(sizeof(int));
    if (ok && (hipEventCreate(&q->ev0) != hipSuccess || hipEventCreate(&q->ev1) != hipSuccess)) ok = false;
    if (!ok) {
        for (void *p : q->allocs) (void) hipFree(p);
        delete q;
        return qfail(EVM_E_HIP, "hipMalloc failed (PPO trainer buffers)");
    }
    *out = q;
    return EVM_OK;
}

void evm_ppo_destroy(EvmPpo *q) {
    if (!q) return;
    for (void *p : q->allocs) (void) hipFree(p);
    if (q->ev0) (void) hipEventDestroy(q->ev0);
    if (q->ev1) (void) hipEventDestroy(q->ev1);
    delete q;
}

int evm_ppo_set_params(EvmPpo *q, const float *d_actor, const float *d_critic, int reset_optimizer, void *stream) {
    if (!q || !d_actor || !d_critic) return qfail(EVM_E_INVALID, "null argument");
    hipStream_t s = (hipStream_t) stream;
    evm::PpoDev &d = q->dev;
    hipError_t e = hipMemcpyAsync(d.actor.theta, d_actor, d.actor.n_params * 4, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d.critic.theta, d_critic, d.critic.n_params * 4, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess && reset_optimizer) {
        if (e == hipSuccess) e = hipMemsetAsync(d.step_dev, 0, sizeof(int), s);
        for (evm::PpoNet *n : {&d.actor, &d.critic}) {
            n->step = 0;
            if (e == hipSuccess) e = hipMemsetAsync(n->m, 0, n->n_params * 4, s);
            if (e == hipSuccess) e = hipMemsetAsync(n->v, 0, n->n_params * 4, s);
        }
    }
    if (e == hipSuccess) e = evm::launch_policy_pack(q->policy->dev.actor, d.S, d.A, true, d.actor.theta, s);
    if (e == hipSuccess) e = evm::launch_policy_pack(q->policy->dev.critic, d.S, d.A, false, d.critic.theta, s);
    if (e == hipSuccess) e = evm::launch_ppo_pack_w2d(d.actor, d.S, d.A, true, s);
    if (e == hipSuccess) e = evm::launch_ppo_pack_w2d(d.critic, d.S, d.A, false, s);
    if (e != hipSuccess) return qfail(EVM_E_HIP, std::string("ppo set_params: ") + hipGetErrorString(e));
    q->have_params = true;
    return EVM_OK;
}

// what: 0 parameters, 1 gradients, 2 Adam exp_avg, 3 Adam exp_avg_sq; net: 0 actor, 1 critic; to_trainer: 0 copies the
// trainer's vector out to d_buf, 1 copies d_buf in (parameters: use evm_ppo_set_params, which also repacks them)
int evm_ppo_copy(EvmPpo *q, int what, int net, int to_trainer, float *d_buf, void *stream) {
    if (!q || !d_buf || what < 0 || what > 3 || (net != 0 && net != 1)) return qfail(EVM_E_INVALID, "bad argument");
    if (what == 0 && to_trainer) return qfail(EVM_E_INVALID, "parameters are set with evm_ppo_set_params");
    evm::PpoNet &n = net == 0 ? q->dev.actor : q->dev.critic;
    float *own = what == 0 ? n.theta : what == 1 ? n.grad : what == 2 ? n.m : n.v;
    hipError_t e = to_trainer ? hipMemcpyAsync(own, d_buf, n.n_params * 4, hipMemcpyDeviceToDevice, (hipStream_t) stream)
                              : hipMemcpyAsync(d_buf, own, n.n_params * 4, hipMemcpyDeviceToDevice, (hipStream_t) stream);
    if (e != hipSuccess) return qfail(EVM_E_HIP, std::string("ppo copy: ") + hipGetErrorString(e));
    return EVM_OK;
}

// The gradients of both networks as one contiguous DEVICE vector [actor | pad to 64 floats | critic]:
// after evm_ppo_grads a data-parallel caller all-reduces it in place on the launch stream (one collective per epoch, no copy,
// no host synchronisation) and calls evm_ppo_apply.  d_buf != NULL makes the trainer use the caller's buffer from now on.
int evm_ppo_grad_buffer(EvmPpo *q, float *d_buf, float **d_grads, size_t *n_floats, size_t *critic_offset) {
    if (!q) return qfail(EVM_E_INVALID, "trainer is null");
    evm::PpoDev &d = q->dev;
    const size_t goff = (d.actor.n_params + 63) / 64 * 64;
    if (d_buf) {
        if ((uintptr_t) d_buf % 256) return qfail(EVM_E_INVALID, "the gradient buffer must be 256-byte aligned");
        d.actor.grad = d_buf; d.critic.grad = d_buf + goff;
    }
    if (d_grads) *d_grads = d.actor.grad;
    if (n_floats) *n_floats = goff + d.critic.n_params;
    if (critic_offset) *critic_offset = goff;
    return EVM_OK;
}

// (count, mean, M2) of every rank's advantages -> the trainer's statistics, merged on the device in rank order; d_all_stats
// [world][3] is what an all-gather of the d_stats of evm_ppo_gae delivers.  Afterwards evm_ppo_gae_normalize(d_stats = NULL)
// and evm_ppo_grads(n_selected_global < 0) use the merged numbers: no host read anywhere in the update.
int evm_ppo_gae_merge(EvmPpo *q, const double *d_all_stats, int world, double *d_stats, void *stream) {
    if (!q || !d_all_stats || world < 1) return qfail(EVM_E_INVALID, "bad argument");
    hipError_t e = evm::launch_ppo_gae_merge(q->dev, d_all_stats, world, (hipStream_t) stream);
    if (e == hipSuccess && d_stats) e = hipMemcpyAsync(d_stats, q->dev.gae, 3 * sizeof(double), hipMemcpyDeviceToDevice, (hipStream_t) stream);
    if (e != hipSuccess) return qfail(EVM_E_HIP, std::string("ppo gae merge: ") + hipGetErrorString(e));
    return EVM_OK;
}

int evm_ppo_adam_step(EvmPpo *q, int net, int set_step, int *step) {
    if (!q || (net != 0 && net != 1 && net != 2)) return qfail(EVM_E_INVALID, "bad argument");
    if (net == 2) {  // the actor's device-side counter (evm_ppo_actor_apply: SAC's captured update)
        int h = set_step;
        if (set_step >= 0) {
            if (hipMemcpy(q->dev.step_dev, &h, sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return qfail(EVM_E_HIP, "step upload failed");
        } else if (hipMemcpy(&h, q->dev.step_dev, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) {
            return qfail(EVM_E_HIP, "step download failed");
        }
        if (step) *step = h;
        return EVM_OK;
    }
    evm::PpoNet &n = net == 0 ? q->dev.actor : q->dev.critic;
    if (set_step >= 0) n.step = set_step;
    if (step) *step = n.step;
    return EVM_OK;
}

int evm_ppo_gae(EvmPpo *q, int horizon, int n_envs, const float *d_rewards, const uint8_t *d_done, const float *d_curr_values,
                const float *d_next_values, const uint8_t *d_mask, float gamma, float lam, float *d_adv, double *d_stats,
                void *stream) {
    if (!q || !d_rewards || !d_done || !d_curr_values || !d_next_values || !d_mask || !d_adv) return qfail(EVM_E_INVALID, "null argument");
    if (horizon < 1 || n_envs < 1) return qfail(EVM_E_INVALID, "empty rollout");
    hipStream_t s = (hipStream_t) stream;
    hipError_t e = evm::launch_ppo_gae_scan(q->dev, horizon, n_envs, d_rewards, d_done, d_curr_values, d_next_values, d_mask, gamma, lam,
                                            d_adv, s);
    if (e == hipSuccess && d_stats) e = hipMemcpyAsync(d_stats, q->dev.gae, 3 * sizeof(double), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return qfail(EVM_E_HIP, std::string("ppo gae: ") + hipGetErrorString(e));
    return EVM_OK;
}

int evm_ppo_gae_normalize(EvmPpo *q, int horizon, int n_envs, const double *d_stats, const float *d_curr_values, float *d_adv,
                          float *d_returns, void *stream) {
    if (!q || !d_curr_values || !d_adv || !d_returns) return qfail(EVM_E_INVALID, "null argument");
    hipError_t e = evm::launch_ppo_gae_finish(horizon, n_envs, d_stats ? d_stats : q->dev.gae, d_curr_values, nullptr, d_adv, d_returns,
                                              (hipStream_t) stream);
    if (e != hipSuccess) return qfail(EVM_E_HIP, std::string("ppo gae normalize: ") + hipGetErrorString(e));
    return EVM_OK;
}

int evm_ppo_grads(EvmPpo *q, size_t rows, const float *d_states, const float *d_actions, const float *d_logp_old, const float *d_adv,
                  const float *d_returns, const uint8_t *d_mask, double n_selected_global, float epsilon, float entropy_factor,
                  float critic_loss_factor, int states_unchanged, void *stream) {
    if (!q || !d_states || !d_actions || !d_logp_old || !d_adv || !d_returns || !d_mask) return qfail(EVM_E_INVALID, "null argument");
    if (!q->have_params) return qfail(EVM_E_INVALID, "evm_ppo_set_params has not been called");
    if (rows < 1 || rows > q->dev.max_rows) return qfail(EVM_E_INVALID, "rows exceeds the trainer's capacity");
    // n_selected_global < 0: the count is the trainer's own statistic on the device (evm_ppo_gae [+ evm_ppo_gae_merge])
    if (!(n_selected_global >= 1.0) && !(n_selected_global < 0.0)) return qfail(EVM_E_INVALID, "no selected transition");
    hipStream_t s = (hipStream_t) stream;
    if (q->timing) (void) hipEventRecord(q->ev0, s);
    const evm::PolicyDev &p = q->policy->dev;
    hipError_t e = hipSuccess;
    q->dev.dev_count = n_selected_global < 0.0 ? 1 : 0;
    if (!states_unchanged || rows != q->staged_rows) {
        e = evm::launch_ppo_pad(q->dev, rows, d_states, s);
        q->staged_rows = rows;
    }
    if (e == hipSuccess) e = evm::launch_ppo_forward(p, q->dev, rows, d_states, s);
    if (e == hipSuccess) e = evm::launch_ppo_loss(q->dev, rows, d_actions, d_logp_old, d_adv, d_returns, d_mask, n_selected_global < 0.0 ? -1.0 : 1.0 / n_selected_global,
                                                  epsilon, entropy_factor, critic_loss_factor, s);
    if (e == hipSuccess) e = evm::launch_ppo_backward(p, q->dev, rows, s);
    if (e == hipSuccess) e = evm::launch_ppo_wgrads(q->dev, rows, d_states, s);
    if (e != hipSuccess) return qfail(EVM_E_HIP, std::string("ppo grads: ") + hipGetErrorString(e));
    return EVM_OK;
}

// The selected rows of a rollout as dense copies owned by the trainer (valid until the next call): what evm_ppo_grads should be
// given for every epoch of one update — rows = *n_selected, the returned pointers, states_unchanged from the second epoch on.
// Reads the count back (one stream synchronisation per update).  *n_selected == 0: nothing was copied, the caller keeps its
// own buffers (evm_ppo_grads then does its device-side no-op on the all-zero mask).
int evm_ppo_select_rows(EvmPpo *q, size_t rows, const uint8_t *d_mask, const float *d_states, const float *d_actions,
                        const float *d_logp_old, const float *d_adv, const float *d_returns, size_t *n_selected,
                        const float **s_states, const float **s_actions, const float **s_logp_old, const float **s_adv,
                        const float **s_returns, const uint8_t **s_mask, void *stream) {
    if (!q || !d_mask || !d_states || !d_actions || !d_logp_old || !d_adv || !d_returns || !n_selected || !s_states || !s_actions ||
        !s_logp_old || !s_adv || !s_returns || !s_mask)
        return qfail(EVM_E_INVALID, "null argument");
    if (rows < 1 || rows > q->dev.max_rows) return qfail(EVM_E_INVALID, "rows exceeds the trainer's capacity");
    hipStream_t s = (hipStream_t) stream;
    const evm::PpoDev &d = q->dev;
    if (!q->sel_idx) {
        if (hipSetDevice(q->policy->device) != hipSuccess) return qfail(EVM_E_HIP, "hipSetDevice failed");
        bool ok = true;
        auto alloc = [&](size_t bytes) -> void * {
            void *p = nullptr;
            if (!ok) return nullptr;
            if (hipMalloc(&p, bytes) != hipSuccess) { ok = false; return nullptr; }
            q->allocs.push_back(p);
            return p;
        };
        const size_t m = d.max_rows;
        int *idx = (int *) alloc(m * sizeof(int));
        q->sel_count = (int *) alloc(sizeof(int));
        q->sel_states = (float *) alloc(m * d.S * 4); q->sel_actions = (float *) alloc(m * d.A * 4); q->sel_logp = (float *) alloc(m * d.A * 4);
        q->sel_adv = (float *) alloc(m * 4); q->sel_returns = (float *) alloc(m * 4);
        q->sel_mask = (uint8_t *) alloc(m);
        if (!ok || hipMemset(q->sel_mask, 1, m) != hipSuccess) return qfail(EVM_E_HIP, "hipMalloc failed (selected-row buffers)");
        q->sel_idx = idx;
    }
    hipError_t e = evm::launch_ppo_select_scan(rows, d_mask, q->sel_idx, q->sel_count, s);
    int h = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&h, q->sel_count, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess && h > 0)
        e = evm::launch_ppo_select_gather((size_t) h, q->sel_idx, d.S, d.A, d_states, d_actions, d_logp_old, d_adv, d_returns, q->sel_states,
                                          q->sel_actions, q->sel_logp, q->sel_adv, q->sel_returns, s);
    if (e != hipSuccess) return qfail(EVM_E_HIP, std::string("ppo select rows: ") + hipGetErrorString(e));
    *n_selected = (size_t) h;
    *s_states = q->sel_states; *s_actions = q->sel_actions; *s_logp_old = q->sel_logp; *s_adv = q->sel_adv; *s_returns = q->sel_returns;
    *s_mask = q->sel_mask;
    q->staged_rows = 0;  // the observation copy of an earlier update no longer matches
    return EVM_OK;
}

// SAC shares ActorModule with PPO: the same kernels run its actor step, with the loss gradient supplied by the caller.
int evm_ppo_actor_forward(EvmPpo *q, size_t rows, const float *d_states, float *d_mu, float *d_sigma, void *stream) {
    if (!q || !d_states || !d_mu || !d_sigma) return qfail(EVM_E_INVALID, "null argument");
    if (!q->have_params) return qfail(EVM_E_INVALID, "evm_ppo_set_params has not been called");
    if (rows < 1 || rows > q->dev.max_rows) return qfail(EVM_E_INVALID, "rows exceeds the trainer's capacity");
    hipStream_t s = (hipStream_t) stream;
    hipError_t e = evm::launch_ppo_pad(q->dev, rows, d_states, s);
    q->staged_rows = rows;
    if (e == hipSuccess) e = evm::launch_ppo_forward(q->policy->dev, q->dev, rows, d_states, s, 1);
    if (e == hipSuccess) e = evm::launch_actor_head_out(q->dev, rows, d_mu, d_sigma, s);
    if (e != hipSuccess) return qfail(EVM_E_HIP, std::string("actor forward: ") + hipGetErrorString(e));
    return EVM_OK;
}

int evm_ppo_actor_backward(EvmPpo *q, size_t rows, const float *d_dmu, const float *d_dsigma, void *stream) {
    if (!q || !d_dmu || !d_dsigma) return qfail(EVM_E_INVALID, "null argument");
    if (rows < 1 || rows != q->staged_rows) return qfail(EVM_E_INVALID, "rows differs from the preceding evm_ppo_actor_forward");
    hipStream_t s = (hipStream_t) stream;
    hipError_t e = evm::launch_actor_head_grad(q->dev, rows, d_dmu, d_dsigma, s);
    if (e == hipSuccess) e = evm::launch_ppo_backward(q->policy->dev, q->dev, rows, s, 1);
    if (e == hipSuccess) e = evm::launch_ppo_wgrads(q->dev, rows, nullptr, s, 1);
    if (e != hipSuccess) return qfail(EVM_E_HIP, std::string("actor backward: ") + hipGetErrorString(e));
    return EVM_OK;
}

// Adam step of the actor alone (no clipping) with a DEVICE step counter, then the new weights into `policy`: SAC's actor
// step (soft_actor_critic.cpp:144-153), replayable from a captured HIP graph.  reset_optimizer of evm_ppo_set_params zeroes
// the counter.
int evm_ppo_actor_apply(EvmPpo *q, float learning_rate, void *stream) {
    if (!q) return qfail(EVM_E_INVALID, "trainer is null");
    hipError_t e = evm::launch_actor_apply(q->policy->dev, q->dev, learning_rate, (hipStream_t) stream);
    if (e != hipSuccess) return qfail(EVM_E_HIP, std::string("actor apply: ") + hipGetErrorString(e));
    return EVM_OK;
}

int evm_ppo_apply(EvmPpo *q, float learning_rate, float clip_grad_norm, void *stream) {
    if (!q) return qfail(EVM_E_INVALID, "trainer is null");
    hipStream_t s = (hipStream_t) stream;
    hipError_t e = evm::launch_ppo_apply(q->policy->dev, q->dev, learning_rate, clip_grad_norm, s);
    if (q->timing) {
        (void) hipEventRecord(q->ev1, s);
        (void) hipEventSynchronize(q->ev1);
        float ms = 0.f;
        (void) hipEventElapsedTime(&ms, q->ev0, q->ev1);
        q->ms_acc += ms; q->n_timed++;
    }
    if (e != hipSuccess) return qfail(EVM_E_HIP, std::string("ppo apply: ") + hipGetErrorString(e));
    return EVM_OK;
}

int evm_ppo_losses(EvmPpo *q, double *h_actor_loss, double *h_critic_loss, void *stream) {
    if (!q) return qfail(EVM_E_INVALID, "trainer is null");
    double h[2] = {0.0, 0.0};
    if (hipStreamSynchronize((hipStream_t) stream) != hipSuccess || hipMemcpy(h, q->dev.loss, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess)
        return qfail(EVM_E_HIP, "reading the losses failed");
    if (h_actor_loss) *h_actor_loss = h[0];
    if (h_critic_loss) *h_critic_loss = h[1];
    return EVM_OK;
}

int evm_ppo_timing(EvmPpo *q, int enable, float *ms_total, int *n_epochs) {
    if (!q) return qfail(EVM_E_INVALID, "trainer is null");
    if (ms_total) *ms_total = q->ms_acc;
    if (n_epochs) *n_epochs = q->n_timed;
    q->timing = enable != 0;
    q->ms_acc = 0.f; q->n_timed = 0;
    return EVM_OK;
}

}  // extern "C"
