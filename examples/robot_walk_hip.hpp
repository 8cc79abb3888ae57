// Compiled C++ side of the boundary: adapters over the C ABI (include/evomotion.h) with the reference's method names.
//
//   RobotWalkHip          one environment, the public surface of `Environment` (evo_motion_model/include/evo_motion_model/
//                         environment.h:56-72): reset() / do_step(action) -> step{state, reward, done}, get_state_space(),
//                         get_action_space(); built by RobotWalkHipFactory::get_env(num_threads, seed) which
//                         get_environment_factory(name, parameters) returns (environment.h:80,96-97; parameter names and
//                         defaults of env_factory.cpp:74-83,91-100; an unknown NAME -> std::invalid_argument, :118; unknown parameter
//                         KEYS are ignored, like EnvironmentFactory::generic_get_value, :22-28).
//   VecRobotWalkHip       N environments behind the same calls plus the train loop's body as one call
//                         (src/train.cpp:61-66: while(!done) do_step; done(); reset()) = step_autoreset().
//
// No torch here: `step::state` is a DEVICE pointer to the observation row(s), the action is a device pointer.  Inside the
// reference the same class hands out `torch::from_blob(state, {state_dim}, torch::kCUDA)` and takes
// `action.data_ptr<float>()` (INTEGRATION.md §1 shows that variant).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/evomotion.h"

namespace evm_adapter {

inline void check(int rc) {
    if (rc == EVM_E_INVALID) throw std::invalid_argument(evm_last_error());  // env_factory.cpp:118
    if (rc != EVM_OK) throw std::runtime_error(evm_last_error());            // skeleton.cpp:46,58
}
inline void hip_check(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

struct step {           // struct step, environment.h:20-24 (state on the device)
    const float *state;
    float reward;
    bool done;
};

class VecRobotWalkHip {
public:
    VecRobotWalkHip(int n_envs, int seed, const std::string &skeleton_json_path, const EvmEnvParams &params, int device = 0,
                    hipStream_t stream = nullptr)
        : n(n_envs), stream_(stream) {
        hip_check(hipSetDevice(device), "hipSetDevice");
        check(evm_env_create(skeleton_json_path.c_str(), n_envs, device, (uint64_t) seed, &params, &env));
        try {
            check(evm_env_spaces(env, &state_dim, &action_dim));
            hip_check(hipMalloc(&d_obs, sizeof(float) * (size_t) n * state_dim), "hipMalloc");
            hip_check(hipMalloc(&d_reward, sizeof(float) * n), "hipMalloc");
            hip_check(hipMalloc(&d_done, n), "hipMalloc");
            hip_check(hipMalloc(&d_valid, n), "hipMalloc");
        } catch (...) {  // a constructor that throws runs no destructor: release what was taken
            release();
            throw;
        }
    }
    VecRobotWalkHip(const VecRobotWalkHip &) = delete;
    VecRobotWalkHip &operator=(const VecRobotWalkHip &) = delete;
    ~VecRobotWalkHip() { release(); }
    // Environment::reset (environment.cpp:45-48) for every env
    void reset() { check(evm_env_reset(env, nullptr, d_obs, d_reward, d_done, stream_)); }
    // Environment::do_step (environment.cpp:33-39) for every env; d_action [n, action_dim] on the device
    void do_step(const float *d_action) { check(evm_env_step(env, d_action, d_obs, d_reward, d_done, stream_)); }
    // the body of train()'s loop: do_step, and for an env whose episode ended the reset() of train.cpp:65 spread over the
    // following calls (valid: 1 = do_step transition, 2 = reset()'s own step, 0 = a settle step inside reset())
    void step_autoreset(const float *d_action) { check(evm_env_step_autoreset(env, d_action, d_obs, d_reward, d_done, d_valid, stream_)); }
    std::vector<int64_t> get_state_space() const { return {state_dim}; }
    std::vector<int64_t> get_action_space() const { return {action_dim}; }
    const float *state() const { return d_obs; }
    const float *reward() const { return d_reward; }
    const uint8_t *done() const { return d_done; }
    const uint8_t *valid() const { return d_valid; }
    EvmEnv *handle() const { return env; }
    hipStream_t stream() const { return stream_; }
    const int n;
    int state_dim = 0, action_dim = 0;

private:
    void release() {
        if (env) evm_env_destroy(env);
        env = nullptr;
        (void) hipFree(d_obs); (void) hipFree(d_reward); (void) hipFree(d_done); (void) hipFree(d_valid);
        d_obs = d_reward = nullptr; d_done = d_valid = nullptr;
    }
    EvmEnv *env = nullptr;
    hipStream_t stream_;
    float *d_obs = nullptr, *d_reward = nullptr;
    uint8_t *d_done = nullptr, *d_valid = nullptr;
};

class RobotWalkHip {  // the public surface of Environment for ONE environment
public:
    RobotWalkHip(int seed, const std::string &skeleton_json_path, const EvmEnvParams &params, int device = 0)
        : v(1, seed, skeleton_json_path, params, device) {}
    step reset() { v.reset(); return out(); }
    step do_step(const float *d_action) { v.do_step(d_action); return out(); }
    std::vector<int64_t> get_state_space() { return v.get_state_space(); }
    std::vector<int64_t> get_action_space() { return v.get_action_space(); }
    void to(int /*device type*/) {}  // Environment::to (environment.h:70): the state never leaves the device here

private:
    step out() {  // reward / done are read back like the reference's `float reward; bool done;` (one small sync per call)
        float r; uint8_t d;
        hip_check(hipMemcpyAsync(&r, v.reward(), sizeof(float), hipMemcpyDeviceToHost, v.stream()), "reward");
        hip_check(hipMemcpyAsync(&d, v.done(), 1, hipMemcpyDeviceToHost, v.stream()), "done");
        hip_check(hipStreamSynchronize(v.stream()), "sync");
        return {v.state(), r, d != 0};
    }
    VecRobotWalkHip v;
};

// EnvironmentFactory (environment.h:76-93) + get_environment_factory (environment.h:96-97, env_factory.cpp:106-120)
class RobotWalkHipFactory {
public:
    RobotWalkHipFactory(const std::string &env_name, std::map<std::string, std::string> parameters) {
        check(evm_env_default_params_for(env_name.c_str(), &prm));  // unknown env name -> std::invalid_argument
        const bool jump = prm.env_kind == 1;
        // typed parameters with defaults (env_factory.cpp:74-83,91-100).  Keys the factory does not know are ignored, as
        // EnvironmentFactory::generic_get_value does (:22-28: it only ever looks up the keys it wants), so one parameter map
        // can be shared between environments; `strict` (this adapter's own switch) turns them into std::invalid_argument.
        const bool strict = parameters.count("strict") && parameters["strict"] != "0";
        for (const auto &kv : parameters) {
            const std::string &k = kv.first;
            if (k == "skeleton_json_path") skeleton = kv.second;
            else if (k == (jump ? "initial_seconds" : "initial_remaining_seconds")) prm.initial_remaining_seconds = std::stof(kv.second);
            else if (k == (jump ? "max_seconds" : "max_episode_seconds")) prm.max_episode_seconds = std::stof(kv.second);
            else if (k == "target_velocity") prm.target_velocity = std::stof(kv.second);
            else if (k == "minimal_velocity") prm.minimal_velocity = std::stof(kv.second);
            else if (!jump && k == "reset_frames") prm.reset_frames = std::stoi(kv.second);
            else if (jump && k == "reset_seconds") prm.reset_frames = (int) (std::stof(kv.second) / (1.f / 60.f));
            else if (k == "self_collision") prm.self_collision = std::stoi(kv.second);  // this path's own switch (evomotion.h)
            else if (k == "strict") continue;
            else if (strict) throw std::invalid_argument(k);
        }
    }
    std::shared_ptr<RobotWalkHip> get_env(int /*num_threads*/, int seed) {
        return std::make_shared<RobotWalkHip>(seed, skeleton, prm);
    }
    EvmEnvParams prm;
    // the reference's default is RESOURCES_PATH/resources/skeleton/new_format_spider.json (env_factory.cpp:79-80,95-96); its
    // decoded copy ships with this package (EVM_DEFAULT_SKELETON: set by the build to evomotion_amd/data/robot_walk_spider.skel)
#ifdef EVM_DEFAULT_SKELETON
    std::string skeleton = EVM_DEFAULT_SKELETON;
#else
    std::string skeleton = "evomotion_amd/data/robot_walk_spider.skel";
#endif
};
inline std::shared_ptr<RobotWalkHipFactory> get_environment_factory(const std::string &env_name, std::map<std::string, std::string> parameters) {
    return std::make_shared<RobotWalkHipFactory>(env_name, std::move(parameters));
}

}  // namespace evm_adapter
