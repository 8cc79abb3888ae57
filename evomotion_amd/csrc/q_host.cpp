// C ABI of the twin-Q trainer of SAC (include/evomotion.h, evm_q_*): critic_1 / critic_2 and their target networks of
// SoftActorCriticAgent (evo_motion_networks/src/agents/soft_actor_critic.cpp:20-45,100-127,166-168) on the device.
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/evomotion.h"
#include "q_dev.h"

namespace evm { void set_last_error(const std::string &m); }

struct EvmQ {
    int device;
    evm::QDev dev;
    std::vector<void *> allocs;
    bool have[4];
};

static int zfail(int code, const std::string &m) { evm::set_last_error(m); return code; }

extern "C" {

int evm_q_create(int state_dim, int action_dim, int hidden_size, size_t max_rows, int device, EvmQ **out) {
    if (!out) return zfail(EVM_E_INVALID, "out is null");
    *out = nullptr;
    if (hidden_size != 256) return zfail(EVM_E_UNSUPPORTED, "the Q kernels are built for hidden_size = 256");
    if (state_dim < 1 || action_dim < 1 || action_dim > 16 || state_dim + action_dim > 384) return zfail(EVM_E_INVALID, "unsupported state / action size");
    if (max_rows < 1 || max_rows > ((size_t) 1 << 30)) return zfail(EVM_E_INVALID, "max_rows out of range");
    if (hipSetDevice(device) != hipSuccess) return zfail(EVM_E_HIP, "hipSetDevice failed");
    EvmQ *q = new EvmQ();
    q->device = device;
    for (bool &h : q->have) h = false;
    evm::QDev &d = q->dev;
    d.S = state_dim; d.A = action_dim; d.max_rows = max_rows;
    bool ok = true;
    auto alloc = [&](size_t bytes) -> void * {
        void *p = nullptr;
        if (!ok) return nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) { ok = false; return nullptr; }
        (void) hipMemset(p, 0, bytes);
        q->allocs.push_back(p);
        return p;
    };
    const size_t SA = (size_t) state_dim + action_dim;
    const size_t tiles = (max_rows + 31) / 32;
    for (int i = 0; i < 4; i++) {
        evm::QNet &n = d.net[i];
        size_t o = 0;
        for (int l = 0; l < evm::Q_LAYERS; l++) {
            n.o_w[l] = o; o += l == 0 ? 256 * SA : 65536;
            n.o_b[l] = o; o += 256;
            n.o_g[l] = o; o += 256;
            n.o_be[l] = o; o += 256;
        }
        n.o_wh = o; o += 256;
        n.o_bh = o; o += 1;
        n.n_params = o;
        n.theta = (float *) alloc(o * 4);
        const bool critic = i < 2;
        n.grad = critic ? (float *) alloc(o * 4) : nullptr;
        n.m = critic ? (float *) alloc(o * 4) : nullptr;
        n.v = critic ? (float *) alloc(o * 4) : nullptr;
        n.step = critic ? (int *) alloc(sizeof(int)) : nullptr;
        n.wt[0] = (float *) alloc((size_t) 384 * 256 * 4);
        n.wd[0] = nullptr;
        for (int l = 1; l < evm::Q_LAYERS; l++) {
            n.wt[l] = (float *) alloc(65536 * 4);
            n.wd[l] = critic ? (float *) alloc(65536 * 4) : nullptr;
        }
        for (int l = 0; l < evm::Q_LAYERS; l++) {
            n.z[l] = critic ? (float *) alloc(max_rows * 256 * 4) : nullptr;
            n.a[l] = critic ? (float *) alloc(max_rows * 256 * 4) : nullptr;
            n.dz[l] = critic ? (float *) alloc(max_rows * 256 * 4) : nullptr;
        }
        n.st = critic ? (float *) alloc(max_rows * 2 * evm::Q_LAYERS * 4) : nullptr;
        n.q = (float *) alloc(max_rows * 4);
        n.dh = critic ? (float *) alloc(max_rows * 32 * 4) : nullptr;
        n.colpart = critic ? (float *) alloc(tiles * evm::Q_COLSLOTS * 256 * 4) : nullptr;
        n.colpart2 = critic ? (float *) alloc((size_t) 64 * evm::Q_COLSLOTS * 256 * 4) : nullptr;
        n.wpart = critic ? (float *) alloc(evm::q_wpart_floats() * 4) : nullptr;
    }
    d.xq = (float *) alloc(max_rows * 384 * 4);
    d.loss = (double *) alloc(2 * sizeof(double));
    if (!ok) {
        for (void *p : q->allocs) (void) hipFree(p);
        delete q;
        return zfail(EVM_E_HIP, "hipMalloc failed (Q trainer buffers)");
    }
    *out = q;
    return EVM_OK;
}

void evm_q_destroy(EvmQ *q) {
    if (!q) return;
    for (void *p : q->allocs) (void) hipFree(p);
    delete q;
}

int evm_q_param_count(const EvmQ *q, size_t *n) {
    if (!q || !n) return zfail(EVM_E_INVALID, "null argument");
    *n = q->dev.net[0].n_params;
    return EVM_OK;
}

// what: 0 parameters, 1 gradients, 2 Adam exp_avg, 3 Adam exp_avg_sq; net: 0 critic_1, 1 critic_2, 2 / 3 their targets
int evm_q_copy(EvmQ *q, int what, int net, int to_trainer, float *d_buf, void *stream) {
    if (!q || !d_buf || what < 0 || what > 3 || net < 0 || net > 3 || (what != 0 && net > 1)) return zfail(EVM_E_INVALID, "bad argument");
    evm::QNet &n = q->dev.net[net];
    float *own = what == 0 ? n.theta : what == 1 ? n.grad : what == 2 ? n.m : n.v;
    hipStream_t s = (hipStream_t) stream;
    hipError_t e = to_trainer ? hipMemcpyAsync(own, d_buf, n.n_params * 4, hipMemcpyDeviceToDevice, s)
                              : hipMemcpyAsync(d_buf, own, n.n_params * 4, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess && to_trainer && what == 0) {
        e = evm::launch_q_pack(q->dev, net, s);
        q->have[net] = true;
    }
    if (e != hipSuccess) return zfail(EVM_E_HIP, std::string("q copy: ") + hipGetErrorString(e));
    return EVM_OK;
}

int evm_q_adam_step(EvmQ *q, int net, int set_step, int *step) {
    if (!q || (net != 0 && net != 1)) return zfail(EVM_E_INVALID, "bad argument");
    int h = 0;
    if (set_step >= 0) {
        h = set_step;
        if (hipMemcpy(q->dev.net[net].step, &h, sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return zfail(EVM_E_HIP, "step upload failed");
    } else if (hipMemcpy(&h, q->dev.net[net].step, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) {
        return zfail(EVM_E_HIP, "step download failed");
    }
    if (step) *step = h;
    return EVM_OK;
}

static int q_ready(const EvmQ *q, unsigned nets) {
    for (int i = 0; i < 4; i++)
        if ((nets & (1u << i)) && !q->have[i]) return zfail(EVM_E_INVALID, "parameters of a selected network have not been set (evm_q_copy)");
    return EVM_OK;
}

// Q(states, actions) of the selected networks (bit i of `nets` = net i) into d_out[i] ([rows] each; entries of unselected
// networks are ignored and may be NULL)
int evm_q_forward(EvmQ *q, unsigned nets, size_t rows, const float *d_states, const float *d_actions, float *const *d_out, void *stream) {
    if (!q || !d_states || !d_actions || !d_out) return zfail(EVM_E_INVALID, "null argument");
    if (rows < 1 || rows > q->dev.max_rows || (nets & ~15u) || !nets) return zfail(EVM_E_INVALID, "bad rows / network mask");
    if (q_ready(q, nets) != EVM_OK) return EVM_E_INVALID;
    hipStream_t s = (hipStream_t) stream;
    hipError_t e = evm::launch_q_concat(q->dev, rows, d_states, d_actions, s);
    if (e == hipSuccess) e = evm::launch_q_forward(q->dev, nets, rows, 0, s);
    for (int i = 0; i < 4 && e == hipSuccess; i++)
        if ((nets & (1u << i)) && d_out[i]) e = hipMemcpyAsync(d_out[i], q->dev.net[i].q, rows * 4, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return zfail(EVM_E_HIP, std::string("q forward: ") + hipGetErrorString(e));
    return EVM_OK;
}

// gradients of mse_loss(critic_i(states, actions), target_q) for both critics (soft_actor_critic.cpp:118-127)
int evm_q_grads(EvmQ *q, size_t rows, const float *d_states, const float *d_actions, const float *d_target_q, void *stream) {
    if (!q || !d_states || !d_actions || !d_target_q) return zfail(EVM_E_INVALID, "null argument");
    if (rows < 1 || rows > q->dev.max_rows) return zfail(EVM_E_INVALID, "rows exceeds the trainer's capacity");
    if (q_ready(q, 3u) != EVM_OK) return EVM_E_INVALID;
    hipStream_t s = (hipStream_t) stream;
    hipError_t e = evm::launch_q_concat(q->dev, rows, d_states, d_actions, s);
    if (e == hipSuccess) e = evm::launch_q_forward(q->dev, 3u, rows, 1, s);
    if (e == hipSuccess) e = evm::launch_q_loss(q->dev, rows, d_target_q, s);
    if (e == hipSuccess) e = evm::launch_q_backward(q->dev, rows, s);
    if (e == hipSuccess) e = evm::launch_q_wgrads(q->dev, rows, s);
    if (e != hipSuccess) return zfail(EVM_E_HIP, std::string("q grads: ") + hipGetErrorString(e));
    return EVM_OK;
}

int evm_q_apply(EvmQ *q, float learning_rate, void *stream) {
    if (!q) return zfail(EVM_E_INVALID, "trainer is null");
    hipError_t e = evm::launch_q_adam(q->dev, learning_rate, (hipStream_t) stream);
    if (e != hipSuccess) return zfail(EVM_E_HIP, std::string("q apply: ") + hipGetErrorString(e));
    return EVM_OK;
}

int evm_q_soft_update(EvmQ *q, float tau, void *stream) {
    if (!q) return zfail(EVM_E_INVALID, "trainer is null");
    if (q_ready(q, 15u) != EVM_OK) return EVM_E_INVALID;
    hipError_t e = evm::launch_q_soft_update(q->dev, tau, (hipStream_t) stream);
    if (e != hipSuccess) return zfail(EVM_E_HIP, std::string("q soft update: ") + hipGetErrorString(e));
    return EVM_OK;
}

// SAC actor step, Q side (soft_actor_critic.cpp:136-140): min(critic_1, critic_2)(states, actions) into d_qmin [rows] and the
// gradient of -mean(min q) w.r.t. the actions into d_dqda [rows][A]
int evm_q_action_grad(EvmQ *q, size_t rows, const float *d_states, const float *d_actions, float *d_qmin, float *d_dqda, void *stream) {
    if (!q || !d_states || !d_actions || !d_qmin || !d_dqda) return zfail(EVM_E_INVALID, "null argument");
    if (rows < 1 || rows > q->dev.max_rows) return zfail(EVM_E_INVALID, "rows exceeds the trainer's capacity");
    if (q_ready(q, 3u) != EVM_OK) return EVM_E_INVALID;
    hipStream_t s = (hipStream_t) stream;
    hipError_t e = evm::launch_q_concat(q->dev, rows, d_states, d_actions, s);
    if (e == hipSuccess) e = evm::launch_q_action_grad(q->dev, rows, d_qmin, d_dqda, s);
    if (e != hipSuccess) return zfail(EVM_E_HIP, std::string("q action grad: ") + hipGetErrorString(e));
    return EVM_OK;
}

// truncated_normal_sample(mu, sigma, -1, 1) with the supplied uniform draws and the summed truncated_normal_log_pdf of the
// sample (functions.cpp:53-68,94-111; soft_actor_critic.cpp:131-135).  [rows][A] inputs, d_action [rows][A], d_logp_sum [rows]
int evm_sac_sample(int rows, int action_dim, const float *d_mu, const float *d_sigma, const float *d_uniform, float *d_action,
                   float *d_logp_sum, void *stream) {
    if (!d_mu || !d_sigma || !d_uniform || !d_action || !d_logp_sum) return zfail(EVM_E_INVALID, "null argument");
    if (rows < 1 || action_dim < 1) return zfail(EVM_E_INVALID, "empty batch");
    hipError_t e = evm::launch_sac_sample(rows, action_dim, d_mu, d_sigma, d_uniform, d_action, d_logp_sum, (hipStream_t) stream);
    if (e != hipSuccess) return zfail(EVM_E_HIP, std::string("sac sample: ") + hipGetErrorString(e));
    return EVM_OK;
}

// gradient of  mean(exp(log_alpha) * logp_sum - min q)  w.r.t. (mu, sigma), the action being the reparameterised sample
// (soft_actor_critic.cpp:129-142); d_dqda from evm_q_action_grad, d_log_alpha a DEVICE scalar
int evm_sac_actor_grad(int rows, int action_dim, const float *d_mu, const float *d_sigma, const float *d_uniform, const float *d_dqda,
                       const float *d_log_alpha, float *d_dmu, float *d_dsigma, void *stream) {
    if (!d_mu || !d_sigma || !d_uniform || !d_dqda || !d_log_alpha || !d_dmu || !d_dsigma) return zfail(EVM_E_INVALID, "null argument");
    if (rows < 1 || action_dim < 1) return zfail(EVM_E_INVALID, "empty batch");
    hipError_t e = evm::launch_sac_actor_grad(rows, action_dim, d_mu, d_sigma, d_uniform, d_dqda, d_log_alpha, d_dmu, d_dsigma,
                                              (hipStream_t) stream);
    if (e != hipSuccess) return zfail(EVM_E_HIP, std::string("sac actor grad: ") + hipGetErrorString(e));
    return EVM_OK;
}

// target_q [rows] = r + (1 - done) * gamma * (min(tq1, tq2) - exp(log_alpha) * sum_a next_logp[., a])   (soft_actor_critic.cpp:108-116)
int evm_sac_target_q(int rows, int action_dim, const float *d_rewards, const float *d_done, const float *d_tq1, const float *d_tq2,
                     const float *d_next_logp, const float *d_log_alpha, float gamma, float *d_target_q, void *stream) {
    if (!d_rewards || !d_done || !d_tq1 || !d_tq2 || !d_next_logp || !d_log_alpha || !d_target_q) return zfail(EVM_E_INVALID, "null argument");
    if (rows < 1 || action_dim < 1) return zfail(EVM_E_INVALID, "empty batch");
    hipError_t e = evm::launch_sac_target(rows, action_dim, d_rewards, d_done, d_tq1, d_tq2, d_next_logp, d_log_alpha, gamma, d_target_q,
                                          (hipStream_t) stream);
    if (e != hipSuccess) return zfail(EVM_E_HIP, std::string("sac target: ") + hipGetErrorString(e));
    return EVM_OK;
}

// Adam step of the entropy parameter on  -mean(log_alpha * (logp_sum + target_entropy))  (soft_actor_critic.cpp:155-164).
// d_log_alpha [1], d_adam_state [2] = exp_avg, exp_avg_sq, d_adam_step [1] (int) live in caller memory on the device;
// d_losses [2] receives the actor loss mean(alpha * logp_sum - qmin) and the entropy loss, both before the step.
int evm_sac_entropy_step(int rows, const float *d_logp_sum, const float *d_qmin, float target_entropy, float learning_rate,
                         float *d_log_alpha, float *d_adam_state, int *d_adam_step, float *d_losses, void *stream) {
    if (!d_logp_sum || !d_qmin || !d_log_alpha || !d_adam_state || !d_adam_step || !d_losses) return zfail(EVM_E_INVALID, "null argument");
    if (rows < 1) return zfail(EVM_E_INVALID, "empty batch");
    hipError_t e = evm::launch_sac_entropy(rows, d_logp_sum, d_qmin, target_entropy, learning_rate, d_log_alpha, d_adam_state, d_adam_step,
                                           d_losses, (hipStream_t) stream);
    if (e != hipSuccess) return zfail(EVM_E_HIP, std::string("sac entropy step: ") + hipGetErrorString(e));
    return EVM_OK;
}

// DEVICE double[2]: the two critics' mse losses of the last evm_q_grads
int evm_q_losses(EvmQ *q, double *d_out, void *stream) {
    if (!q || !d_out) return zfail(EVM_E_INVALID, "null argument");
    if (hipMemcpyAsync(d_out, q->dev.loss, 2 * sizeof(double), hipMemcpyDeviceToDevice, (hipStream_t) stream) != hipSuccess)
        return zfail(EVM_E_HIP, "loss copy failed");
    return EVM_OK;
}

}  // extern "C"
