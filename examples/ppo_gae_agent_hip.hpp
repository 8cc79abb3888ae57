// Compiled C++ AGENT side of the boundary: the reference's `Agent` interface for PPO over the C ABI (include/evomotion.h),
// torch-free, next to the environment adapters of robot_walk_hip.hpp.
//
//   Agent                       evo_motion_networks/include/evo_motion_networks/agent.h:16-35: act(state, reward) -> action,
//                               done(state, reward), save / load(folder), get_metrics(), to(), set_eval(), count_parameters().
//                               Here `state` and the returned action are DEVICE pointers ([state_dim] / [action_dim] floats);
//                               inside the reference the same class takes `state.data_ptr<float>()` and hands out
//                               `torch::from_blob(action, {A}, torch::kCUDA)`.
//   TrajectoryReplayBufferHip   TrajectoryReplayBuffer (src/replay_buffer.cpp:64-146,176-189): a FIFO of whole trajectories,
//                               std::mt19937(seed) + std::shuffle over all but the last trajectory of more than one step —
//                               the reference's own generator, so the draws are the reference's draws.
//   PpoGaeAgentHip              PpoGaeAgent (src/agents/ppo_gae.cpp:29-115): act / done / check_train with the previous
//                               transition's reward, update_last, train() every `train_every` EPISODES on `batch_size`
//                               trajectories padded to the longest (zeros, done = 1, the shifted mask of :127-132).  The forward
//                               pass is evm_policy_forward (one row), train() is evm_ppo_gae / _gae_normalize / `epoch` x
//                               (_grads, _apply) on the time-major [T][B] batch — the calls evomotion_amd/agent.py::PpoGaeAgent
//                               makes, so both produce the same weights bit for bit (tests/test_gpu_cxx_agent.py).
//   RandomAgentHip / ConstantAgentHip   debug_agents.cpp:7-39; the random agent reproduces the reference's torch::rand stream.
//   PpoGaeHipFactory, ...       the factories (agent_factory.cpp:66-80,137-146: the reference's parameter keys, a missing key ->
//                               std::invalid_argument, :25-29); get_agent_factory(name, parameters) is agent_factory_hip.hpp.
//
// The steps of a trajectory live on the device (state, action, log_prob, curr_value, next_value), reward / done on the host
// like the reference's `float reward; bool done;`.  One small kernel per act() appends a step, one per trajectory packs it into
// the padded batch.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <deque>
#include <fstream>
#include <functional>
#include <iomanip>
#include <map>
#include <memory>
#include <numeric>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "robot_walk_hip.hpp"
#include "th_archive.hpp"

namespace evm_adapter {

// LossMeter (metrics.h:45-55, metrics.cpp:12-29): mean over the last `window_size` values, 0 when empty
class LossMeterHip {
public:
    LossMeterHip(std::string name, int window_size) : name_(std::move(name)), window(window_size) {}
    void add(float v) {  // metrics.cpp:17-23: room is made before the value goes in
        while ((int) results.size() >= window) results.pop_front();
        results.push_back(v);
        curr_step++;
    }
    float loss() const { return results.empty() ? 0.f : std::accumulate(results.begin(), results.end(), 0.f) / (float) results.size(); }
    std::string to_string() const {  // metrics.cpp:52-56,70-74
        std::ostringstream s;
        s << name_ << " = " << std::setprecision(6) << std::fixed << loss();
        return s.str();
    }
    const std::string &name() const { return name_; }
    size_t count() const { return results.size(); }
    long curr_step = 0;

private:
    std::string name_;
    int window;
    std::deque<float> results;
};

class Agent {  // agent.h:16-35 with device pointers instead of tensors
public:
    virtual const float *act(const float *d_state, float reward) = 0;
    virtual void done(const float *d_state, float reward) = 0;
    virtual void save(const std::string &output_folder_path) = 0;
    virtual void load(const std::string &input_folder_path) = 0;
    virtual std::vector<LossMeterHip> get_metrics() = 0;
    virtual void to(int device_type) = 0;
    virtual void set_eval(bool eval) = 0;
    virtual int count_parameters() = 0;
    virtual ~Agent() = default;
};

// ---- device side of a trajectory -------------------------------------------------------------------------------------------
struct TrajDev {
    float *state, *action, *logp, *value, *next_value;  // [cap][S], [cap][A], [cap][A], [cap], [cap]
};
// step i of the trajectory: what PpoGaeAgent::act stores (ppo_gae.cpp:40-42); the previous step's next_value becomes this value
static __global__ void k_agent_append(TrajDev t, int i, int S, int A, const float *state, const float *action, const float *logp,
                                      const float *value) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < S) t.state[(size_t) i * S + k] = state[k];
    if (k < A) { t.action[(size_t) i * A + k] = action[k]; t.logp[(size_t) i * A + k] = logp[k]; }
    if (k == 0) {
        const float v = value[0];
        t.value[i] = v;
        t.next_value[i] = v;
        if (i > 0) t.next_value[i - 1] = v;
    }
}
static __global__ void k_agent_set_next(float *next_value, int i, const float *value) { next_value[i] = value[0]; }
// trajectory -> column b of the time-major padded batch (rows L..T-1 stay zero)
static __global__ void k_agent_pack(TrajDev t, int L, int B, int b, int S, int A, float *states, float *actions, float *logp, float *values,
                                    float *next_values) {
    const int row = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= L) return;
    const size_t o = (size_t) row * B + b;
    if (k < S) states[o * S + k] = t.state[(size_t) row * S + k];
    if (k < A) { actions[o * A + k] = t.action[(size_t) row * A + k]; logp[o * A + k] = t.logp[(size_t) row * A + k]; }
    if (k == 0) { values[o] = t.value[row]; next_values[o] = t.next_value[row]; }
}

class TrajectoryHip {  // episode_trajectory<ppo_episode_step> (replay_buffer.h:22-36)
public:
    TrajectoryHip(int S, int A) : S(S), A(A) {}
    TrajectoryHip(const TrajectoryHip &) = delete;
    TrajectoryHip &operator=(const TrajectoryHip &) = delete;
    ~TrajectoryHip() { release(d); }
    int size() const { return (int) reward.size(); }
    void add(const float *state, const float *action, const float *logp, const float *value, hipStream_t s) {
        const int i = size();
        if (i >= cap) grow(s);
        hipLaunchKernelGGL(k_agent_append, dim3((S + 255) / 256), dim3(256), 0, s, d, i, S, A, state, action, logp, value);
        hip_check(hipGetLastError(), "k_agent_append");
        reward.push_back(0.f);
        done.push_back(0);
    }
    // update_last_step (replay_buffer.cpp:176-186); the next_value of a non-terminal step was written by add()'s kernel
    void update_last(float r, bool is_done, const float *terminal_value, hipStream_t s) {
        reward.back() = r;
        done.back() = is_done ? 1 : 0;
        if (terminal_value) {
            hipLaunchKernelGGL(k_agent_set_next, dim3(1), dim3(1), 0, s, d.next_value, size() - 1, terminal_value);
            hip_check(hipGetLastError(), "k_agent_set_next");
        }
    }
    void pack(int B, int b, float *states, float *actions, float *logp, float *values, float *next_values, hipStream_t s) const {
        hipLaunchKernelGGL(k_agent_pack, dim3((S + 255) / 256, size()), dim3(256), 0, s, d, size(), B, b, S, A, states, actions, logp, values,
                           next_values);
        hip_check(hipGetLastError(), "k_agent_pack");
    }
    const TrajDev &dev() const { return d; }
    std::vector<float> reward;
    std::vector<uint8_t> done;

private:
    static void release(TrajDev &t) {
        (void) hipFree(t.state); (void) hipFree(t.action); (void) hipFree(t.logp); (void) hipFree(t.value); (void) hipFree(t.next_value);
        t = TrajDev{nullptr, nullptr, nullptr, nullptr, nullptr};
    }
    void grow(hipStream_t s) {
        const int ncap = cap ? 2 * cap : 64;
        TrajDev n{nullptr, nullptr, nullptr, nullptr, nullptr};
        try {
            hip_check(hipMalloc(&n.state, sizeof(float) * (size_t) ncap * S), "hipMalloc");
            hip_check(hipMalloc(&n.action, sizeof(float) * (size_t) ncap * A), "hipMalloc");
            hip_check(hipMalloc(&n.logp, sizeof(float) * (size_t) ncap * A), "hipMalloc");
            hip_check(hipMalloc(&n.value, sizeof(float) * ncap), "hipMalloc");
            hip_check(hipMalloc(&n.next_value, sizeof(float) * ncap), "hipMalloc");
            if (cap) {
                hip_check(hipMemcpyAsync(n.state, d.state, sizeof(float) * (size_t) cap * S, hipMemcpyDeviceToDevice, s), "copy");
                hip_check(hipMemcpyAsync(n.action, d.action, sizeof(float) * (size_t) cap * A, hipMemcpyDeviceToDevice, s), "copy");
                hip_check(hipMemcpyAsync(n.logp, d.logp, sizeof(float) * (size_t) cap * A, hipMemcpyDeviceToDevice, s), "copy");
                hip_check(hipMemcpyAsync(n.value, d.value, sizeof(float) * cap, hipMemcpyDeviceToDevice, s), "copy");
                hip_check(hipMemcpyAsync(n.next_value, d.next_value, sizeof(float) * cap, hipMemcpyDeviceToDevice, s), "copy");
                hip_check(hipStreamSynchronize(s), "sync");
            }
        } catch (...) {
            release(n);
            throw;
        }
        release(d);
        d = n;
        cap = ncap;
    }
    int S, A, cap = 0;
    TrajDev d{nullptr, nullptr, nullptr, nullptr, nullptr};
};

class TrajectoryReplayBufferHip {
public:
    TrajectoryReplayBufferHip(int size, int seed, int S, int A) : size(size), S(S), A(A), rand_gen(seed) {}  // replay_buffer.cpp:64-66
    bool empty() const { return memory.empty(); }
    bool trajectory_empty() const { return empty() || memory.back()->size() == 0; }
    void new_trajectory() {  // :105-110
        memory.push_back(std::make_shared<TrajectoryHip>(S, A));
        while ((int) memory.size() > size) memory.erase(memory.begin());
    }
    TrajectoryHip &last() { return *memory.back(); }
    std::vector<int> filtered_positions() const {
        std::vector<int> f;
        for (int i = 0; i < (int) memory.size(); i++)
            if (memory[i]->size() > 1) f.push_back(i);
        return f;
    }
    bool enough_trajectory(int batch_size) const { return (int) filtered_positions().size() >= batch_size; }  // :139-146
    // :73-98: the trajectories of more than one step, all but the last of them shuffled, the first batch_size taken; the memory
    // positions drawn are kept in last_draw
    std::vector<std::shared_ptr<TrajectoryHip>> sample(int batch_size) {
        const std::vector<int> filtered = filtered_positions();
        std::vector<int> index(filtered.size() - 1);
        std::iota(index.begin(), index.end(), 0);
        std::shuffle(index.begin(), index.end(), rand_gen);
        std::vector<std::shared_ptr<TrajectoryHip>> result;
        last_draw.clear();
        for (int i = 0; i < batch_size && i < (int) index.size(); i++) {
            result.push_back(memory[filtered[index[i]]]);
            last_draw.push_back(filtered[index[i]]);
        }
        return result;
    }
    std::vector<std::shared_ptr<TrajectoryHip>> memory;
    std::vector<int> last_draw;

private:
    int size, S, A;
    std::mt19937 rand_gen;
};

class PpoGaeAgentHip : public Agent {
public:
    // the constructor arguments of PpoGaeAgent (ppo_gae.h / ppo_gae.cpp:11-27) + the device
    PpoGaeAgentHip(int seed, const std::vector<int64_t> &state_space, const std::vector<int64_t> &action_space, int hidden_size, float gamma,
                   float lambda, float epsilon, float entropy_factor, float critic_loss_factor, int epoch, int batch_size, int train_every,
                   int replay_buffer_size, float learning_rate, float clip_grad_norm, int device = 0, hipStream_t stream = nullptr)
        : S((int) state_space.at(0)), A((int) action_space.at(0)), H(hidden_size), gamma(gamma), lambda(lambda), epsilon(epsilon),
          entropy_factor(entropy_factor), critic_loss_factor(critic_loss_factor), epoch(epoch), batch_size(batch_size), train_every(train_every),
          learning_rate(learning_rate), clip_grad_norm(clip_grad_norm), seed(seed), stream(stream),
          replay_buffer(replay_buffer_size, seed, (int) state_space.at(0), (int) action_space.at(0)), actor_loss_meter("actor_loss", 64),
          critic_loss_meter("critic_loss", 64), episode_steps_meter("steps", 64) {
        hip_check(hipSetDevice(device), "hipSetDevice");
        check(evm_policy_create(S, A, H, device, &pol));
        try {
            check(evm_policy_param_counts(pol, &n_actor, &n_critic));
            hip_check(hipMalloc(&d_out, sizeof(float) * (2 * A + 1)), "hipMalloc");
            hip_check(hipMalloc(&d_params, sizeof(float) * (n_actor + n_critic)), "hipMalloc");
            // init_weights (init.cpp:7-21): xavier_normal_(gain 0.1) weights, N(0, 0.1) biases, LayerNorm ones / zeros — the same
            // distributions from std::mt19937(seed); at::manual_seed's stream belongs to LibTorch, use set_parameters() or load()
            // for the reference's own numbers
            std::mt19937 g((uint32_t) seed);
            set_parameters(init_network(g, true), init_network(g, false));
        } catch (...) {
            release();
            throw;
        }
    }
    PpoGaeAgentHip(const PpoGaeAgentHip &) = delete;
    PpoGaeAgentHip &operator=(const PpoGaeAgentHip &) = delete;
    ~PpoGaeAgentHip() override { release(); }

    // flat fp32 parameters in named_parameters() order (actor: head.0 w b, head.2 w b, head.3 w b, head.5 w b, mu.0 w b, sigma.0 w b;
    // critic: ... , value head); fresh Adam state like a newly constructed torch::optim::Adam (ppo_gae.cpp:22-25)
    void set_parameters(const std::vector<float> &actor, const std::vector<float> &critic) {
        if (actor.size() != n_actor || critic.size() != n_critic) throw std::invalid_argument("parameter count");
        hip_check(hipMemcpyAsync(d_params, actor.data(), sizeof(float) * n_actor, hipMemcpyHostToDevice, stream), "upload");
        hip_check(hipMemcpyAsync(d_params + n_actor, critic.data(), sizeof(float) * n_critic, hipMemcpyHostToDevice, stream), "upload");
        check(evm_policy_set_weights_device(pol, d_params, d_params + n_actor, stream));
        if (trainer) check(evm_ppo_set_params(trainer, d_params, d_params + n_actor, 1, stream));
        hip_check(hipStreamSynchronize(stream), "sync");
    }
    std::vector<float> get_parameters() {  // actor then critic
        std::vector<float> h(n_actor + n_critic);
        if (trainer) {
            check(evm_ppo_copy(trainer, 0, 0, 0, d_params, stream));
            check(evm_ppo_copy(trainer, 0, 1, 0, d_params + n_actor, stream));
        }
        hip_check(hipMemcpyAsync(h.data(), d_params, sizeof(float) * h.size(), hipMemcpyDeviceToHost, stream), "download");
        hip_check(hipStreamSynchronize(stream), "sync");
        return h;
    }

    // ---- Agent ----------------------------------------------------------------------------------------------------------
    // ppo_gae.cpp:29-45.  `reward` is the reward of the PREVIOUS transition (:38).  The returned pointer ([A] floats on the
    // device) stays valid until the next act() / done().
    const float *act(const float *d_state, float reward) override { return act(d_state, reward, nullptr); }
    // d_uniform [A]: the U[0,1) draws of truncated_normal_sample (the reference's at::rand); NULL = the kernel's generator
    const float *act(const float *d_state, float reward, const float *d_uniform) {
        forward(d_state, d_uniform);
        if (replay_buffer.empty()) replay_buffer.new_trajectory();
        if (!replay_buffer.trajectory_empty()) replay_buffer.last().update_last(reward, false, nullptr, stream);
        replay_buffer.last().add(d_state, d_out, d_out + A, d_out + 2 * A, stream);
        curr_episode_step++;
        return d_out;
    }
    // ppo_gae.cpp:47-61: the terminal state, before the environment is reset (src/train.cpp:64-65)
    void done(const float *d_state, float reward) override {
        if (replay_buffer.trajectory_empty()) throw std::logic_error("done() before the episode's first act()");  // (the reference dereferences an empty vector here)
        forward(d_state, nullptr);
        replay_buffer.last().update_last(reward, true, d_out + 2 * A, stream);
        trained_last_done = check_train();
        replay_buffer.new_trajectory();
        global_curr_step++;
        episode_steps_meter.add((float) curr_episode_step);
        curr_episode_step = 0;
    }
    // Checkpoints: the reference's four files (ppo_gae.cpp:192-204 through saver.h:13-39) — actor.th, actor_optimizer.th, critic.th,
    // critic_optimizer.th — written and read by th_archive.hpp without LibTorch: module archives with the reference's parameter
    // names (head.0.weight ... sigma.0.bias / critic.0.weight ... critic.6.bias) and torch::optim::Adam archives (format "1.5.0",
    // one group, step / exp_avg / exp_avg_sq per parameter).  A folder written here loads in the reference's PpoGaeAgent::load
    // and the other way round (tests/test_checkpoint.py, tests/test_gpu_cxx_loop.py).
    void save(const std::string &folder) override {
        ensure_trainer(1);
        for (int net = 0; net < 2; net++) {
            const size_t n = net == 0 ? n_actor : n_critic;
            std::vector<float> h(3 * n);
            for (int what = 0; what < 3; what++) {
                check(evm_ppo_copy(trainer, what == 0 ? 0 : what + 1, net, 0, d_params, stream));
                hip_check(hipMemcpyAsync(h.data() + what * n, d_params, sizeof(float) * n, hipMemcpyDeviceToHost, stream), "download");
                hip_check(hipStreamSynchronize(stream), "sync");
            }
            int step = 0;
            check(evm_ppo_adam_step(trainer, net, -1, &step));
            const evm_th::Node module = net == 0 ? evm_th::actor_module(S, A, H, h.data()) : evm_th::critic_module(S, H, h.data());
            evm_th::save(folder + (net == 0 ? "/actor.th" : "/critic.th"), module);
            evm_th::save(folder + (net == 0 ? "/actor_optimizer.th" : "/critic_optimizer.th"),
                         evm_th::adam_archive(evm_th::adam_params_of(module, step, h.data() + n, h.data() + 2 * n), learning_rate));
        }
    }
    void load(const std::string &folder) override {  // a missing file -> std::runtime_error (saver.h:33-34)
        ensure_trainer(1);
        std::vector<float> w[2], m[2], v[2];
        int64_t steps[2];
        for (int net = 0; net < 2; net++) {
            const size_t n = net == 0 ? n_actor : n_critic;
            const evm_th::Node module = evm_th::load(folder + (net == 0 ? "/actor.th" : "/critic.th"));
            w[net] = evm_th::flat_parameters(module, n, net == 0 ? "actor.th" : "critic.th");
            std::vector<evm_th::AdamParam> ps = evm_th::adam_params_of(module, 0, nullptr, nullptr);
            evm_th::adam_from_archive(evm_th::load(folder + (net == 0 ? "/actor_optimizer.th" : "/critic_optimizer.th")), ps);
            steps[net] = evm_th::flat_adam(ps, m[net], v[net]);
        }
        set_parameters(w[0], w[1]);
        for (int net = 0; net < 2; net++) {
            const size_t n = net == 0 ? n_actor : n_critic;
            for (int what = 1; what < 3; what++) {
                hip_check(hipMemcpyAsync(d_params, (what == 1 ? m[net] : v[net]).data(), sizeof(float) * n, hipMemcpyHostToDevice, stream), "upload");
                check(evm_ppo_copy(trainer, what + 1, net, 1, d_params, stream));
                hip_check(hipStreamSynchronize(stream), "sync");
            }
            int s = 0;
            check(evm_ppo_adam_step(trainer, net, (int) steps[net], &s));
        }
        check(evm_ppo_copy(trainer, 0, 0, 0, d_params, stream));
        check(evm_ppo_copy(trainer, 0, 1, 0, d_params + n_actor, stream));
        hip_check(hipStreamSynchronize(stream), "sync");
    }
    std::vector<LossMeterHip> get_metrics() override { return {actor_loss_meter, critic_loss_meter, episode_steps_meter}; }  // ppo_gae.cpp:205-207
    void to(int) override {}          // the networks never leave the device
    void set_eval(bool) override {}   // no dropout / batch statistics in these modules: eval and train forward agree
    int count_parameters() override { return (int) (n_actor + n_critic); }

    // bookkeeping the tests read
    TrajectoryReplayBufferHip &buffer() { return replay_buffer; }
    long curr_train_step = 0, curr_episode_step = 0, global_curr_step = 0;
    bool trained_last_done = false;
    const int S, A, H;

private:
    void forward(const float *d_state, const float *d_uniform) {
        act_calls++;
        check(evm_policy_forward(pol, 1, d_state, d_uniform, ((uint64_t) seed + 7919ull * act_calls) & 0x7FFFFFFFull, d_out, d_out + A, d_out + 2 * A,
                                 nullptr, nullptr, stream));
    }
    std::vector<float> init_network(std::mt19937 &g, bool actor) {
        std::vector<float> v;
        auto linear = [&](int out, int in) {
            std::normal_distribution<float> w(0.f, 0.1f * std::sqrt(2.0f / (float) (in + out))), b(0.f, 0.1f);
            for (int i = 0; i < out * in; i++) v.push_back(w(g));
            for (int i = 0; i < out; i++) v.push_back(b(g));
        };
        auto layernorm = [&](int n) { v.insert(v.end(), n, 1.f); v.insert(v.end(), n, 0.f); };
        linear(H, S); layernorm(H); linear(H, H); layernorm(H);
        if (actor) { linear(A, H); linear(A, H); }
        else linear(1, H);
        return v;
    }
    // a trainer that holds `rows` transitions; a bigger one takes over the parameters and the optimiser state of a smaller one
    void ensure_trainer(size_t rows) {
        if (trainer && trainer_rows >= rows) return;
        std::vector<float *> keep;
        int steps[2] = {0, 0};
        if (trainer) {
            for (int net = 0; net < 2; net++) {
                const size_t n = net == 0 ? n_actor : n_critic;
                for (int what : {0, 2, 3}) {
                    float *p = nullptr;
                    hip_check(hipMalloc(&p, sizeof(float) * n), "hipMalloc");
                    keep.push_back(p);
                    check(evm_ppo_copy(trainer, what, net, 0, p, stream));
                }
                check(evm_ppo_adam_step(trainer, net, -1, &steps[net]));
            }
            hip_check(hipStreamSynchronize(stream), "sync");
            evm_ppo_destroy(trainer);
            trainer = nullptr;
        }
        check(evm_ppo_create(pol, rows, &trainer));
        trainer_rows = rows;
        if (keep.empty()) check(evm_ppo_set_params(trainer, d_params, d_params + n_actor, 1, stream));
        else {
            check(evm_ppo_set_params(trainer, keep[0], keep[3], 1, stream));
            for (int net = 0; net < 2; net++) {
                check(evm_ppo_copy(trainer, 2, net, 1, keep[3 * net + 1], stream));
                check(evm_ppo_copy(trainer, 3, net, 1, keep[3 * net + 2], stream));
                int s = 0;
                check(evm_ppo_adam_step(trainer, net, steps[net], &s));
            }
        }
        hip_check(hipStreamSynchronize(stream), "sync");
        for (float *p : keep) (void) hipFree(p);
    }
    // ppo_gae.cpp:63-115 + train (:117-190) on the device
    bool check_train() {
        if (!(global_curr_step % train_every == train_every - 1 && replay_buffer.enough_trajectory(batch_size))) return false;
        const auto episodes = replay_buffer.sample(batch_size);
        const int B = (int) episodes.size();
        int T = 0;
        for (const auto &t : episodes) T = std::max(T, t->size());
        const size_t rows = (size_t) T * B;
        ensure_trainer(rows);
        // time-major [T][B]; padding: zeros, done = 1 (:93-103); mask[t] = 1 at t = 0, else 1 - done[t - 1] (:127-132)
        float *fl = nullptr;
        uint8_t *by = nullptr;
        const size_t nfl = rows * (size_t) (S + 2 * A + 5);
        hip_check(hipMalloc(&fl, sizeof(float) * nfl), "hipMalloc");
        hip_check(hipMalloc(&by, 2 * rows), "hipMalloc");
        try {
            hip_check(hipMemsetAsync(fl, 0, sizeof(float) * nfl, stream), "memset");
            float *states = fl, *actions = states + rows * S, *logp = actions + rows * A, *rewards = logp + rows * A, *values = rewards + rows,
                  *next_values = values + rows, *adv = next_values + rows, *ret = adv + rows;
            uint8_t *done_d = by, *mask_d = by + rows;
            std::vector<float> h_rewards(rows, 0.f);
            std::vector<uint8_t> h_done(rows, 1), h_mask(rows, 0);
            for (int b = 0; b < B; b++) {
                const TrajectoryHip &t = *episodes[b];
                t.pack(B, b, states, actions, logp, values, next_values, stream);
                for (int i = 0; i < t.size(); i++) { h_rewards[(size_t) i * B + b] = t.reward[i]; h_done[(size_t) i * B + b] = t.done[i]; }
            }
            for (int i = 0; i < T; i++)
                for (int b = 0; b < B; b++) h_mask[(size_t) i * B + b] = i == 0 ? 1 : (uint8_t) (1 - h_done[(size_t) (i - 1) * B + b]);
            hip_check(hipMemcpyAsync(rewards, h_rewards.data(), sizeof(float) * rows, hipMemcpyHostToDevice, stream), "upload");
            hip_check(hipMemcpyAsync(done_d, h_done.data(), rows, hipMemcpyHostToDevice, stream), "upload");
            hip_check(hipMemcpyAsync(mask_d, h_mask.data(), rows, hipMemcpyHostToDevice, stream), "upload");
            check(evm_ppo_gae(trainer, T, B, rewards, done_d, values, next_values, mask_d, gamma, lambda, adv, nullptr, stream));
            check(evm_ppo_gae_normalize(trainer, T, B, nullptr, values, adv, ret, stream));
            // the padding rows weigh nothing (ppo_gae.cpp:167-168, 178): the epochs run on the selected rows (as FusedPpoTrainer.train)
            size_t nsel = 0;
            const float *u_states = states, *u_actions = actions, *u_logp = logp, *u_adv = adv, *u_ret = ret;
            const uint8_t *u_mask = mask_d;
            {
                const float *c0, *c1, *c2, *c3, *c4; const uint8_t *c5;
                check(evm_ppo_select_rows(trainer, rows, mask_d, states, actions, logp, adv, ret, &nsel, &c0, &c1, &c2, &c3, &c4, &c5, stream));
                if (nsel > 0 && nsel < (size_t) rows) { u_states = c0; u_actions = c1; u_logp = c2; u_adv = c3; u_ret = c4; u_mask = c5; }
                else nsel = (size_t) rows;
            }
            for (int ep = 0; ep < epoch; ep++) {
                check(evm_ppo_grads(trainer, nsel, u_states, u_actions, u_logp, u_adv, u_ret, u_mask, -1.0, epsilon, entropy_factor, critic_loss_factor,
                                    ep > 0 ? 1 : 0, stream));
                check(evm_ppo_apply(trainer, learning_rate, clip_grad_norm, stream));
                // the LossMeter adds of ppo_gae.cpp:185-186, once per epoch (synchronises: the host vectors above may go)
                double la = 0.0, lc = 0.0;
                check(evm_ppo_losses(trainer, &la, &lc, stream));
                actor_loss_meter.add((float) la);
                critic_loss_meter.add((float) lc);
                last_actor_loss = la; last_critic_loss = lc;
            }
        } catch (...) {
            (void) hipStreamSynchronize(stream);
            (void) hipFree(fl); (void) hipFree(by);
            throw;
        }
        (void) hipFree(fl); (void) hipFree(by);
        curr_train_step++;
        return true;
    }
    void release() {
        if (trainer) evm_ppo_destroy(trainer);
        if (pol) evm_policy_destroy(pol);
        trainer = nullptr; pol = nullptr;
        (void) hipFree(d_out); (void) hipFree(d_params);
        d_out = d_params = nullptr;
    }

public:
    double last_actor_loss = 0.0, last_critic_loss = 0.0;

private:
    float gamma, lambda, epsilon, entropy_factor, critic_loss_factor;
    int epoch, batch_size, train_every;
    float learning_rate, clip_grad_norm;
    int seed;
    hipStream_t stream;
    TrajectoryReplayBufferHip replay_buffer;
    LossMeterHip actor_loss_meter, critic_loss_meter, episode_steps_meter;
    EvmPolicy *pol = nullptr;
    EvmPpo *trainer = nullptr;
    size_t trainer_rows = 0, n_actor = 0, n_critic = 0;
    float *d_out = nullptr;     // [A] action | [A] log_prob | [1] value of the last forward
    float *d_params = nullptr;  // staging: actor | critic
    unsigned long long act_calls = 0;
};

// DebugAgent / RandomAgent / ConstantAgent (debug_agents.cpp:7-39): no parameters, no metrics, no-op done / save / load.
// RandomAgent::act is 2 * torch::rand({A}) - 1 from the GLOBAL generator; LibTorch's CPU generator is std::mt19937 and, for the
// fewer than 16 values an action has, torch::rand takes the low 24 bits of one 32-bit output per value — so a std::mt19937(seed)
// here IS the reference's stream after at::manual_seed(seed) (PpoGaeAgent's constructor calls it, ppo_gae.cpp:26): the golden
// actions of the compiled reference come out exactly (tests/test_gpu_cxx_agent.py).
class DebugAgentHip : public Agent {
public:
    explicit DebugAgentHip(const std::vector<int64_t> &action_space, hipStream_t stream = nullptr) : A((int) action_space.at(0)), stream(stream), host(A) {
        if (A < 1) throw std::invalid_argument("action_space");
        hip_check(hipMalloc(&d_action, sizeof(float) * A), "hipMalloc");
    }
    DebugAgentHip(const DebugAgentHip &) = delete;
    DebugAgentHip &operator=(const DebugAgentHip &) = delete;
    ~DebugAgentHip() override { (void) hipFree(d_action); }
    void done(const float *, float) override {}
    void save(const std::string &) override {}
    void load(const std::string &) override {}
    std::vector<LossMeterHip> get_metrics() override { return {}; }
    void to(int) override {}
    void set_eval(bool) override {}
    int count_parameters() override { return 0; }
    const int A;

protected:
    const float *upload() {
        hip_check(hipMemcpyAsync(d_action, host.data(), sizeof(float) * A, hipMemcpyHostToDevice, stream), "upload");
        hip_check(hipStreamSynchronize(stream), "sync");  // `host` is rewritten by the next act()
        return d_action;
    }
    hipStream_t stream;
    std::vector<float> host;
    float *d_action = nullptr;
};
class RandomAgentHip : public DebugAgentHip {
public:
    RandomAgentHip(const std::vector<int64_t> &action_space, uint32_t seed) : DebugAgentHip(action_space), gen(seed) {
        if (A >= 16) throw std::invalid_argument("RandomAgentHip reproduces torch::rand's stream for fewer than 16 values per call");
    }
    const float *act(const float *, float) override {
        for (int i = 0; i < A; i++) host[i] = 2.f * ((float) (gen() & ((1u << 24) - 1u)) * (1.0f / 16777216.0f)) - 1.f;
        return upload();
    }

private:
    std::mt19937 gen;
};
class ConstantAgentHip : public DebugAgentHip {
public:
    ConstantAgentHip(const std::vector<int64_t> &action_space, float action_value) : DebugAgentHip(action_space) { host.assign(A, action_value); }
    const float *act(const float *, float) override { return upload(); }
};

// AgentFactory / PpoGaeFactory / RandomAgentFactory / ConstantAgentFactory / get_agent_factory (agent.h:39-60,
// agent_factory.cpp:22-29,66-80,137-146,186-211)
class AgentFactoryHip {
public:
    explicit AgentFactoryHip(std::map<std::string, std::string> parameters) : parameters(std::move(parameters)) {}
    virtual std::shared_ptr<Agent> create_agent(const std::vector<int64_t> &state_space, const std::vector<int64_t> &action_space) = 0;
    virtual ~AgentFactoryHip() = default;

protected:
    const std::string &raw(const std::string &key) {
        auto it = parameters.find(key);
        if (it == parameters.end()) throw std::invalid_argument(key);  // agent_factory.cpp:27
        return it->second;
    }
    int get_int(const std::string &key) { return std::stoi(raw(key)); }
    float get_float(const std::string &key) { return std::stof(raw(key)); }
    std::map<std::string, std::string> parameters;
};
class PpoGaeHipFactory : public AgentFactoryHip {
public:
    using AgentFactoryHip::AgentFactoryHip;
    std::shared_ptr<Agent> create_agent(const std::vector<int64_t> &state_space, const std::vector<int64_t> &action_space) override {
        const int device = parameters.count("device") ? std::stoi(parameters["device"]) : 0;  // this adapter's own key
        return std::make_shared<PpoGaeAgentHip>(get_int("seed"), state_space, action_space, get_int("hidden_size"), get_float("gamma"),
                                                get_float("lambda"), get_float("epsilon"), get_float("entropy_factor"),
                                                get_float("critic_loss_factor"), get_int("epoch"), get_int("batch_size"), get_int("train_every"),
                                                get_int("replay_buffer_size"), get_float("learning_rate"), get_float("clip_grad_norm"), device);
    }
};
class RandomAgentHipFactory : public AgentFactoryHip {
public:
    using AgentFactoryHip::AgentFactoryHip;
    std::shared_ptr<Agent> create_agent(const std::vector<int64_t> &, const std::vector<int64_t> &action_space) override {
        // "seed" is this adapter's own key: the at::manual_seed the reference's process made before the first act() (default: LibTorch's)
        const uint32_t seed = parameters.count("seed") ? (uint32_t) std::stoul(parameters["seed"]) : 5489u;
        return std::make_shared<RandomAgentHip>(action_space, seed);
    }
};
class ConstantAgentHipFactory : public AgentFactoryHip {
public:
    using AgentFactoryHip::AgentFactoryHip;
    std::shared_ptr<Agent> create_agent(const std::vector<int64_t> &, const std::vector<int64_t> &action_space) override {
        return std::make_shared<ConstantAgentHip>(action_space, get_float("action_value"));  // agent_factory.cpp:76
    }
};
}  // namespace evm_adapter
