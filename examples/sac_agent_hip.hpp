// Compiled C++ AGENT side of the boundary for SAC: the reference's SoftActorCriticAgent behind the `Agent` interface of
// ppo_gae_agent_hip.hpp, torch-free over the C ABI (include/evomotion.h).
//
//   ReplayBufferHip            ReplayBuffer (src/replay_buffer.cpp:10-58,146-153): a FIFO of at most `size` transitions (state, action,
//                              reward, done, next_state); add() stores the newest one open (reward 0, done false, next_state = state),
//                              update_last() completes it, sample() never returns it: the indices of all the others are shuffled with
//                              std::mt19937(seed) + std::shuffle — the reference's own generator — and the first batch_size taken.
//   SoftActorCriticAgentHip    SoftActorCriticAgent (src/agents/soft_actor_critic.cpp:47-91,172-180): act(state, reward) with the
//                              previous transition's reward, check_train() inside act() (`epoch` train() calls on `batch_size` sampled
//                              transitions whenever global_curr_step % train_every == train_every - 1 and the buffer holds a batch),
//                              done(state, reward).  Kept from the reference: act() rewrites the newest transition whenever the buffer
//                              is not empty, also right after done().  train() (:93-170) is the device sequence of INTEGRATION.md §6 —
//                              evm_policy_forward (actor only), evm_q_forward on the targets, evm_sac_target_q, evm_q_grads / _apply,
//                              evm_ppo_actor_forward, evm_sac_sample, evm_q_action_grad, evm_sac_actor_grad, evm_ppo_actor_backward /
//                              _actor_apply, evm_sac_entropy_step, evm_q_soft_update — the calls evomotion_amd/sac.py makes, so both
//                              produce the same weights bit for bit (tests/test_gpu_cxx_sac.py).
//
// The two uniform draws of a train() call (the reference's at::rand inside truncated_normal_sample) come from a std::mt19937 on the
// host unless the caller supplies them (tests do: the reference's recorded draws).
#pragma once
#include "ppo_gae_agent_hip.hpp"

namespace evm_adapter {

static __global__ void k_sac_store(float *state, float *action, float *next_state, int slot, int S, int A, const float *s, const float *a) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < S) { state[(size_t) slot * S + k] = s[k]; next_state[(size_t) slot * S + k] = s[k]; }
    if (k < A) action[(size_t) slot * A + k] = a[k];
}
static __global__ void k_sac_set_next(float *next_state, int slot, int S, const float *s) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < S) next_state[(size_t) slot * S + k] = s[k];
}
static __global__ void k_sac_gather(const float *state, const float *action, const float *next_state, const int *slots, int S, int A, float *bs,
                                    float *ba, float *bn) {
    const int row = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x, slot = slots[row];
    if (k < S) { bs[(size_t) row * S + k] = state[(size_t) slot * S + k]; bn[(size_t) row * S + k] = next_state[(size_t) slot * S + k]; }
    if (k < A) ba[(size_t) row * A + k] = action[(size_t) slot * A + k];
}

class ReplayBufferHip {
public:
    ReplayBufferHip(int size, int seed, int S, int A) : size(size), S(S), A(A), cap(size + 1), rand_gen(seed) {  // replay_buffer.cpp:10-14
        if (size < 1) throw std::invalid_argument("replay_buffer_size");
        try {  // (on the CURRENT device: the agent selects its device before this member is built)
            hip_check(hipMalloc(&state, sizeof(float) * (size_t) cap * S), "hipMalloc");
            hip_check(hipMalloc(&next_state, sizeof(float) * (size_t) cap * S), "hipMalloc");
            hip_check(hipMalloc(&action, sizeof(float) * (size_t) cap * A), "hipMalloc");
            hip_check(hipMalloc(&d_slots, sizeof(int) * (size_t) cap), "hipMalloc");
        } catch (...) {
            release();
            throw;
        }
    }
    ReplayBufferHip(const ReplayBufferHip &) = delete;
    ReplayBufferHip &operator=(const ReplayBufferHip &) = delete;
    ~ReplayBufferHip() { release(); }
    bool empty() const { return count == 0; }
    int length() const { return count; }
    void add(const float *d_state, const float *d_action, hipStream_t s) {  // :30-35 (the transition of act(): reward 0, done false, next_state = state)
        hipLaunchKernelGGL(k_sac_store, dim3((S + 255) / 256), dim3(256), 0, s, state, action, next_state, slot(count), S, A, d_state, d_action);
        hip_check(hipGetLastError(), "k_sac_store");
        reward.push_back(0.f); done.push_back(0);
        count++;
        while (count > size) { head = (head + 1) % cap; count--; reward.pop_front(); done.pop_front(); }
    }
    void update_last(float r, const float *d_next_state, bool is_done, hipStream_t s) {  // :37-42,146-153
        reward.back() = r; done.back() = is_done ? 1 : 0;
        hipLaunchKernelGGL(k_sac_set_next, dim3((S + 255) / 256), dim3(256), 0, s, next_state, slot(count - 1), S, d_next_state);
        hip_check(hipGetLastError(), "k_sac_set_next");
    }
    bool has_enough(int batch_size) const { return count - 1 >= batch_size; }  // :49-52
    // :16-28 — the drawn positions (0 = oldest) stay in last_draw; rows of the batch buffers in draw order
    int sample(int batch_size, float *bs, float *ba, float *br, float *bd, float *bn, hipStream_t s) {
        std::vector<int> index(count - 1);
        std::iota(index.begin(), index.end(), 0);
        std::shuffle(index.begin(), index.end(), rand_gen);
        const int n = std::min(batch_size, (int) index.size());
        last_draw.assign(index.begin(), index.begin() + n);
        std::vector<int> slots(n);
        h_r.resize(n); h_d.resize(n);
        for (int i = 0; i < n; i++) { slots[i] = slot(index[i]); h_r[i] = reward[index[i]]; h_d[i] = done[index[i]] ? 1.f : 0.f; }
        hip_check(hipMemcpyAsync(d_slots, slots.data(), sizeof(int) * n, hipMemcpyHostToDevice, s), "upload");
        hip_check(hipMemcpyAsync(br, h_r.data(), sizeof(float) * n, hipMemcpyHostToDevice, s), "upload");
        hip_check(hipMemcpyAsync(bd, h_d.data(), sizeof(float) * n, hipMemcpyHostToDevice, s), "upload");
        hipLaunchKernelGGL(k_sac_gather, dim3((S + 255) / 256, n), dim3(256), 0, s, state, action, next_state, d_slots, S, A, bs, ba, bn);
        hip_check(hipGetLastError(), "k_sac_gather");
        hip_check(hipStreamSynchronize(s), "sync");  // the host vectors above are reused by the next call
        return n;
    }
    // (tests) element 0 of state / next_state of every transition, oldest first
    void debug_first_values(std::vector<float> &s0, std::vector<float> &n0) const {
        s0.resize(count); n0.resize(count);
        for (int i = 0; i < count; i++) {
            hip_check(hipMemcpy(&s0[i], state + (size_t) slot(i) * S, sizeof(float), hipMemcpyDeviceToHost), "download");
            hip_check(hipMemcpy(&n0[i], next_state + (size_t) slot(i) * S, sizeof(float), hipMemcpyDeviceToHost), "download");
        }
    }
    std::deque<float> reward;
    std::deque<uint8_t> done;
    std::vector<int> last_draw;

private:
    void release() {
        (void) hipFree(state); (void) hipFree(next_state); (void) hipFree(action); (void) hipFree(d_slots);
        state = next_state = action = nullptr; d_slots = nullptr;
    }
    int slot(int logical) const { return (head + logical) % cap; }
    int size, S, A, cap, head = 0, count = 0;
    std::mt19937 rand_gen;
    float *state = nullptr, *next_state = nullptr, *action = nullptr;
    int *d_slots = nullptr;
    std::vector<float> h_r, h_d;
};

class SoftActorCriticAgentHip : public Agent {
public:
    // the constructor arguments of SoftActorCriticAgent (soft_actor_critic.cpp:11-45) + the device
    SoftActorCriticAgentHip(int seed, const std::vector<int64_t> &state_space, const std::vector<int64_t> &action_space, int actor_hidden_size,
                            int critic_hidden_size, int batch_size, int epoch, float learning_rate, float gamma, float tau, int replay_buffer_size,
                            int train_every, int device = 0, hipStream_t stream = nullptr)
        : S((int) state_space.at(0)), A((int) action_space.at(0)), B(batch_size), epoch(epoch), train_every(train_every), lr(learning_rate),
          gamma(gamma), tau(tau), seed(seed), stream(stream), device_(use_device(device)),
          replay_buffer(replay_buffer_size, seed, (int) state_space.at(0), (int) action_space.at(0)),
          noise((uint32_t) seed ^ 0x5bd1e995u), actor_loss_meter("actor", 64), critic_1_loss_meter("critic_1", 64), critic_2_loss_meter("critic_2", 64),
          entropy_loss_meter("entropy", 64), episode_steps_meter("steps", 64), rewards_meter("rewards", 64) {
        actor_hidden = actor_hidden_size; critic_hidden = critic_hidden_size;
        check(evm_policy_create(S, A, actor_hidden_size, device, &pol));
        try {
            check(evm_policy_param_counts(pol, &n_actor, &n_critic));
            check(evm_q_create(S, A, critic_hidden_size, (size_t) B, device, &q));
            check(evm_q_param_count(q, &n_q));
            check(evm_ppo_create(pol, (size_t) B, &actor_tr));
            const size_t fl = (size_t) B * (2 * S + 12 * A + 8) + 2 * A + n_actor + n_critic + n_q + 8;  // (what the take() calls below add up to, rounded up)
            hip_check(hipMalloc(&arena, sizeof(float) * fl), "hipMalloc");
            hip_check(hipMemset(arena, 0, sizeof(float) * fl), "hipMemset");
            hip_check(hipMalloc(&d_lq, 2 * sizeof(double)), "hipMalloc");
            float *p = arena;
            auto take = [&](size_t n) { float *r = p; p += n; return r; };
            bs = take((size_t) B * S); bn = take((size_t) B * S); ba = take((size_t) B * A); br = take(B); bd = take(B);
            next_action = take((size_t) B * A); next_logp = take((size_t) B * A); tq2 = take(B); tq3 = take(B); target_q = take(B);
            mu = take((size_t) B * A); sigma = take((size_t) B * A); action = take((size_t) B * A); logp = take(B); qmin = take(B);
            dqda = take((size_t) B * A); dmu = take((size_t) B * A); dsigma = take((size_t) B * A); u_next = take((size_t) B * A); u_curr = take((size_t) B * A);
            d_act = take(A); d_act_logp = take(A); losses = take(2); log_alpha = take(1); ent_state = take(2); ent_step = reinterpret_cast<int *>(take(1));
            d_actor = take(n_actor); d_critic_dummy = take(n_critic); d_q = take(n_q);
            if ((size_t) (p - arena) > fl) throw std::logic_error("arena");
            // init_weights (init.cpp:7-21) from std::mt19937(seed): the same distributions, LibTorch's own stream is not reproduced
            // (set_parameters() / load() for given numbers); hard_update(target, critic) (:40-41); log_alpha = log(1) (entropy.cpp:7-10)
            std::mt19937 g((uint32_t) seed);
            const std::vector<float> a0 = init_actor(g, actor_hidden_size), q1 = init_q(g, critic_hidden_size), q2 = init_q(g, critic_hidden_size);
            set_parameters(a0, {q1, q2, q1, q2});
        } catch (...) {
            release();
            throw;
        }
    }
    SoftActorCriticAgentHip(const SoftActorCriticAgentHip &) = delete;
    SoftActorCriticAgentHip &operator=(const SoftActorCriticAgentHip &) = delete;
    ~SoftActorCriticAgentHip() override { release(); }

    // flat fp32 parameters in named_parameters() order; qnets = {critic_1, critic_2, target_critic_1, target_critic_2}; fresh optimisers
    void set_parameters(const std::vector<float> &actor, const std::vector<std::vector<float>> &qnets) {
        if (actor.size() != n_actor || qnets.size() != 4) throw std::invalid_argument("parameter count");
        hip_check(hipMemcpyAsync(d_actor, actor.data(), sizeof(float) * n_actor, hipMemcpyHostToDevice, stream), "upload");
        check(evm_policy_set_weights_device(pol, d_actor, nullptr, stream));
        check(evm_ppo_set_params(actor_tr, d_actor, d_critic_dummy, 1, stream));  // the trainer's critic slot is unused
        for (int i = 0; i < 4; i++) {
            if (qnets[i].size() != n_q) throw std::invalid_argument("parameter count");
            hip_check(hipMemcpyAsync(d_q, qnets[i].data(), sizeof(float) * n_q, hipMemcpyHostToDevice, stream), "upload");
            check(evm_q_copy(q, 0, i, 1, d_q, stream));
            hip_check(hipStreamSynchronize(stream), "sync");
        }
        const float zero3[4] = {0.f, 0.f, 0.f, 0.f};
        hip_check(hipMemcpyAsync(log_alpha, zero3, sizeof(float), hipMemcpyHostToDevice, stream), "upload");
        hip_check(hipMemcpyAsync(ent_state, zero3, 2 * sizeof(float), hipMemcpyHostToDevice, stream), "upload");
        hip_check(hipMemsetAsync(ent_step, 0, sizeof(int), stream), "memset");
        hip_check(hipStreamSynchronize(stream), "sync");
    }
    // actor | critic_1 | critic_2 | target_critic_1 | target_critic_2 | log_alpha
    std::vector<float> get_parameters() {
        std::vector<float> h(n_actor + 4 * n_q + 1);
        check(evm_ppo_copy(actor_tr, 0, 0, 0, d_actor, stream));
        hip_check(hipMemcpyAsync(h.data(), d_actor, sizeof(float) * n_actor, hipMemcpyDeviceToHost, stream), "download");
        for (int i = 0; i < 4; i++) {
            check(evm_q_copy(q, 0, i, 0, d_q, stream));
            hip_check(hipMemcpyAsync(h.data() + n_actor + i * n_q, d_q, sizeof(float) * n_q, hipMemcpyDeviceToHost, stream), "download");
            hip_check(hipStreamSynchronize(stream), "sync");
        }
        hip_check(hipMemcpyAsync(h.data() + n_actor + 4 * n_q, log_alpha, sizeof(float), hipMemcpyDeviceToHost, stream), "download");
        hip_check(hipStreamSynchronize(stream), "sync");
        return h;
    }

    // ---- Agent ----------------------------------------------------------------------------------------------------------
    const float *act(const float *d_state, float reward) override { return act(d_state, reward, nullptr, nullptr); }
    // soft_actor_critic.cpp:47-62.  d_uniform [A]: the U[0,1) draws of truncated_normal_sample; train_uniforms: for every epoch of the
    // train() calls this act() may trigger, u_next then u_curr ([batch_size, A] each, host), else drawn here
    const float *act(const float *d_state, float reward, const float *d_uniform, const float *train_uniforms) {
        act_calls++;
        check(evm_policy_forward(pol, 1, d_state, d_uniform, ((uint64_t) seed + 7919ull * act_calls) & 0x7FFFFFFFull, d_act, d_act_logp, nullptr, nullptr,
                                 nullptr, stream));
        if (!replay_buffer.empty()) replay_buffer.update_last(reward, d_state, false, stream);
        replay_buffer.add(d_state, d_act, stream);
        trained_last_act = check_train(train_uniforms);
        curr_episode_step++;
        global_curr_step++;
        return d_act;
    }
    void done(const float *d_state, float reward) override {  // :172-180
        if (replay_buffer.empty()) throw std::logic_error("done() before the first act()");
        replay_buffer.update_last(reward, d_state, true, stream);
        rewards_meter.add(reward);
        episode_steps_meter.add((float) curr_episode_step);
        curr_episode_step = 0;
    }
    // Checkpoints: the reference's module files (soft_actor_critic.cpp:182-199 through saver.h:13-39), written and read by
    // th_archive.hpp without LibTorch: actor.th, critic_1.th, target_critic_1.th, critic_2.th, target_critic_2.th, entropy.th with
    // the reference's parameter names.  The five `*_optimizer.th` files are not written: the optimisers restart after a load()
    // (the reference's own load() restores three of the five, :201-215).
    void save(const std::string &folder) override {
        const std::vector<float> p = get_parameters();
        evm_th::save(folder + "/actor.th", evm_th::actor_module(S, A, actor_hidden, p.data()));
        const char *names[4] = {"/critic_1.th", "/critic_2.th", "/target_critic_1.th", "/target_critic_2.th"};
        for (int i = 0; i < 4; i++) evm_th::save(folder + names[i], evm_th::q_module(S, A, critic_hidden, p.data() + n_actor + i * n_q));
        evm_th::save(folder + "/entropy.th", evm_th::entropy_module(&p[n_actor + 4 * n_q], 1));
    }
    void load(const std::string &folder) override {  // weights only (a missing file -> std::runtime_error, saver.h:33-34)
        const std::vector<float> a = evm_th::flat_parameters(evm_th::load(folder + "/actor.th"), n_actor, "actor.th");
        const char *names[4] = {"/critic_1.th", "/critic_2.th", "/target_critic_1.th", "/target_critic_2.th"};
        std::vector<std::vector<float>> qs;
        for (int i = 0; i < 4; i++) qs.push_back(evm_th::flat_parameters(evm_th::load(folder + names[i]), n_q, names[i] + 1));
        const std::vector<float> la = evm_th::flat_parameters(evm_th::load(folder + "/entropy.th"), 1, "entropy.th");
        set_parameters(a, qs);
        hip_check(hipMemcpy(log_alpha, la.data(), sizeof(float), hipMemcpyHostToDevice), "upload");
    }
    std::vector<LossMeterHip> get_metrics() override {  // :223-226
        return {actor_loss_meter, critic_1_loss_meter, critic_2_loss_meter, entropy_loss_meter, episode_steps_meter, rewards_meter};
    }
    void to(int) override {}
    void set_eval(bool) override {}
    int count_parameters() override { return (int) (n_actor + 4 * n_q + 1); }  // :230-236 counts the actor, the four Q networks and the entropy parameter

    ReplayBufferHip &buffer() { return replay_buffer; }
    long curr_train_step = 0, curr_episode_step = 0, global_curr_step = 0;
    int trained_last_act = 0;
    const int S, A, B;
    int actor_hidden = 0, critic_hidden = 0;

private:
    static void linear(std::vector<float> &v, std::mt19937 &g, int out, int in) {
        std::normal_distribution<float> w(0.f, 0.1f * std::sqrt(2.0f / (float) (in + out))), b(0.f, 0.1f);
        for (int i = 0; i < out * in; i++) v.push_back(w(g));
        for (int i = 0; i < out; i++) v.push_back(b(g));
    }
    static void layernorm(std::vector<float> &v, int n) { v.insert(v.end(), n, 1.f); v.insert(v.end(), n, 0.f); }
    std::vector<float> init_actor(std::mt19937 &g, int H) {
        std::vector<float> v;
        linear(v, g, H, S); layernorm(v, H); linear(v, g, H, H); layernorm(v, H); linear(v, g, A, H); linear(v, g, A, H);
        return v;
    }
    std::vector<float> init_q(std::mt19937 &g, int H) {  // q_net.cpp:8-27
        std::vector<float> v;
        linear(v, g, H, S + A); layernorm(v, H); linear(v, g, H, H); layernorm(v, H); linear(v, g, H, H); layernorm(v, H); linear(v, g, 1, H);
        return v;
    }
    // soft_actor_critic.cpp:64-91
    int check_train(const float *train_uniforms) {
        if (!(global_curr_step % train_every == train_every - 1 && replay_buffer.has_enough(B))) return 0;
        for (int e = 0; e < epoch; e++) {
            const int rows = replay_buffer.sample(B, bs, ba, br, bd, bn, stream);
            draws.push_back(replay_buffer.last_draw);
            std::vector<float> h((size_t) 2 * rows * A);
            if (train_uniforms) std::copy(train_uniforms + (size_t) e * 2 * B * A, train_uniforms + (size_t) (e + 1) * 2 * B * A, h.begin());
            else for (float &x : h) x = (float) (noise() & ((1u << 24) - 1u)) * (1.0f / 16777216.0f);
            hip_check(hipMemcpyAsync(u_next, h.data(), sizeof(float) * rows * A, hipMemcpyHostToDevice, stream), "upload");
            hip_check(hipMemcpyAsync(u_curr, h.data() + (size_t) rows * A, sizeof(float) * rows * A, hipMemcpyHostToDevice, stream), "upload");
            train(rows);
            curr_train_step++;
        }
        return epoch;
    }
    // soft_actor_critic.cpp:93-170 on the device
    void train(int rows) {
        // targets (:100-116)
        check(evm_policy_forward(pol, rows, bn, u_next, 0, next_action, next_logp, nullptr, nullptr, nullptr, stream));
        float *outs[4] = {nullptr, nullptr, tq2, tq3};
        check(evm_q_forward(q, (1u << 2) | (1u << 3), (size_t) rows, bn, next_action, outs, stream));
        check(evm_sac_target_q(rows, A, br, bd, tq2, tq3, next_logp, log_alpha, gamma, target_q, stream));
        // critics (:118-127)
        check(evm_q_grads(q, (size_t) rows, bs, ba, target_q, stream));
        check(evm_q_apply(q, lr, stream));
        // actor (:129-153)
        check(evm_ppo_actor_forward(actor_tr, (size_t) rows, bs, mu, sigma, stream));
        check(evm_sac_sample(rows, A, mu, sigma, u_curr, action, logp, stream));
        check(evm_q_action_grad(q, (size_t) rows, bs, action, qmin, dqda, stream));
        check(evm_sac_actor_grad(rows, A, mu, sigma, u_curr, dqda, log_alpha, dmu, dsigma, stream));
        check(evm_ppo_actor_backward(actor_tr, (size_t) rows, dmu, dsigma, stream));
        check(evm_ppo_actor_apply(actor_tr, lr, stream));
        // entropy parameter (:155-164), soft update (:160-161), meters (:164-167)
        check(evm_sac_entropy_step(rows, logp, qmin, -(float) A, lr, log_alpha, ent_state, ent_step, losses, stream));
        check(evm_q_soft_update(q, tau, stream));
        check(evm_q_losses(q, d_lq, stream));
        float hl[2];
        double hq[2];
        hip_check(hipMemcpyAsync(hl, losses, sizeof(hl), hipMemcpyDeviceToHost, stream), "download");
        hip_check(hipMemcpyAsync(hq, d_lq, sizeof(hq), hipMemcpyDeviceToHost, stream), "download");
        hip_check(hipStreamSynchronize(stream), "sync");
        actor_loss_meter.add(hl[0]); critic_1_loss_meter.add((float) hq[0]); critic_2_loss_meter.add((float) hq[1]); entropy_loss_meter.add(hl[1]);
    }
    void release() {
        if (actor_tr) evm_ppo_destroy(actor_tr);
        if (q) evm_q_destroy(q);
        if (pol) evm_policy_destroy(pol);
        actor_tr = nullptr; q = nullptr; pol = nullptr;
        (void) hipFree(arena); (void) hipFree(d_lq);
        arena = nullptr; d_lq = nullptr;
    }

public:
    std::vector<std::vector<int>> draws;  // the buffer positions of every train() call's batch, in draw order

private:
    int epoch, train_every;
    float lr, gamma, tau;
    int seed;
    hipStream_t stream;
    static int use_device(int d) { hip_check(hipSetDevice(d), "hipSetDevice"); return d; }
    int device_;  // (declared before the buffer: selected before the buffer allocates)
    ReplayBufferHip replay_buffer;
    std::mt19937 noise;
    LossMeterHip actor_loss_meter, critic_1_loss_meter, critic_2_loss_meter, entropy_loss_meter, episode_steps_meter, rewards_meter;
    EvmPolicy *pol = nullptr;
    EvmQ *q = nullptr;
    EvmPpo *actor_tr = nullptr;
    size_t n_actor = 0, n_critic = 0, n_q = 0;
    float *arena = nullptr;
    float *bs, *bn, *ba, *br, *bd, *next_action, *next_logp, *tq2, *tq3, *target_q, *mu, *sigma, *action, *logp, *qmin, *dqda, *dmu, *dsigma, *u_next,
        *u_curr, *d_act, *d_act_logp, *losses, *log_alpha, *ent_state, *d_actor, *d_critic_dummy, *d_q;
    int *ent_step = nullptr;
    double *d_lq = nullptr;  // the critics' losses of the last evm_q_grads
    unsigned long long act_calls = 0;
};

// SofActorCriticFactory (agent_factory.cpp:112-125): parameter keys of the reference
class SoftActorCriticHipFactory : public AgentFactoryHip {
public:
    using AgentFactoryHip::AgentFactoryHip;
    std::shared_ptr<Agent> create_agent(const std::vector<int64_t> &state_space, const std::vector<int64_t> &action_space) override {
        const int device = parameters.count("device") ? std::stoi(parameters["device"]) : 0;
        return std::make_shared<SoftActorCriticAgentHip>(get_int("seed"), state_space, action_space, get_int("actor_hidden_size"), get_int("critic_hidden_size"),
                                                         get_int("batch_size"), get_int("epoch"), get_float("learning_rate"), get_float("gamma"), get_float("tau"),
                                                         get_int("replay_buffer_size"), get_int("train_every"), device);
    }
};

}  // namespace evm_adapter
