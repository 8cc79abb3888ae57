// Torch-free rollout on the C ABI: the reference's train() loop shape (src/train.cpp:41-66) for N environments.
//
//   rollout_main --skeleton <file> [--envs 4096] [--steps 1024] [--warmup 64] [--seed 1234] [--mode policy|random]
//                [--dump <file> --dump-envs 8]
//
// reset(), then `steps` times: actions (PpoGaeAgent::act for all envs = evm_policy_forward, or a counter-based uniform
// generator = RandomAgent) -> evm_env_step_autoreset.  Prints one line with env-steps/s (do_step transitions delivered, as
// bench.py counts them) — a Python-free timing of the rollout.  --dump writes, for the first --dump-envs environments, the
// observation after reset() and after every step as raw fp32: tests/test_gpu_cxx_host.py compares it bit for bit with the
// same rollout driven through the Python binding.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "robot_walk_hip.hpp"

using namespace evm_adapter;

// deterministic parameters, reproducible from Python (tests): Linear weights = pattern / sqrt(fan_in), biases 0,
// LayerNorm weight 1 / bias 0; named_parameters() order of ActorModule / CriticModule (actor.cpp:9-28, critic.cpp:8-21)
static float pattern(uint32_t k) {
    uint32_t h = k * 2654435761u + 12345u;
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    return (float) (h >> 8) * (1.0f / 16777216.0f) * 2.0f - 1.0f;  // exact fp32 in [-1, 1)
}
static void push_linear(std::vector<float> &v, int out, int in, uint32_t &k) {
    const float s = 1.0f / std::sqrt((float) in);
    for (int i = 0; i < out * in; i++) v.push_back(pattern(k++) * s);
    for (int i = 0; i < out; i++) v.push_back(0.f);
}
static void push_layernorm(std::vector<float> &v, int n) {
    for (int i = 0; i < n; i++) v.push_back(1.f);
    for (int i = 0; i < n; i++) v.push_back(0.f);
}
static std::vector<float> make_params(int S, int A, int H, bool actor, uint32_t base) {
    std::vector<float> v;
    uint32_t k = base;
    push_linear(v, H, S, k); push_layernorm(v, H);
    push_linear(v, H, H, k); push_layernorm(v, H);
    if (actor) { push_linear(v, A, H, k); push_linear(v, A, H, k); }  // mu head, sigma head
    else push_linear(v, 1, H, k);
    return v;
}

__global__ void k_uniform_actions(float *a, int n, uint32_t call) {  // RandomAgent::act for all envs (debug_agents.cpp:28-30)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t h = ((uint32_t) i + call * 0x9E3779B9u) * 2654435761u + 12345u;
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    a[i] = (float) (h >> 8) * (1.0f / 16777216.0f) * 2.0f - 1.0f;
}

int main(int argc, char **argv) {
    std::string skeleton, dump, mode = "policy";
    int n = 4096, steps = 1024, warmup = 64, seed = 1234, dump_envs = 8;
    for (int i = 1; i < argc; i++) {
        auto arg = [&](const char *name) { return !strcmp(argv[i], name) && i + 1 < argc; };
        if (arg("--skeleton")) skeleton = argv[++i];
        else if (arg("--envs")) n = atoi(argv[++i]);
        else if (arg("--steps")) steps = atoi(argv[++i]);
        else if (arg("--warmup")) warmup = atoi(argv[++i]);
        else if (arg("--seed")) seed = atoi(argv[++i]);
        else if (arg("--mode")) mode = argv[++i];
        else if (arg("--dump")) dump = argv[++i];
        else if (arg("--dump-envs")) dump_envs = atoi(argv[++i]);
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if (skeleton.empty()) { fprintf(stderr, "--skeleton <robot_walk skeleton (.skel fixture or the reference's JSON)> is required\n"); return 2; }
    try {
        auto factory = get_environment_factory("robot_walk", {{"skeleton_json_path", skeleton}});
        VecRobotWalkHip env(n, seed, factory->skeleton, factory->prm);
        const int S = env.state_dim, A = env.action_dim;
        float *d_action, *d_logp, *d_value;
        hip_check(hipMalloc(&d_action, sizeof(float) * (size_t) n * A), "hipMalloc");
        hip_check(hipMalloc(&d_logp, sizeof(float) * (size_t) n * A), "hipMalloc");
        hip_check(hipMalloc(&d_value, sizeof(float) * n), "hipMalloc");
        EvmPolicy *pol = nullptr;
        if (mode == "policy") {
            check(evm_policy_create(S, A, 256, 0, &pol));
            const std::vector<float> pa = make_params(S, A, 256, true, 1000u), pc = make_params(S, A, 256, false, 500000u);
            check(evm_policy_set_weights(pol, pa.data(), pa.size(), pc.data(), pc.size()));
        }
        if (dump_envs > n) dump_envs = n;
        FILE *df = dump.empty() ? nullptr : fopen(dump.c_str(), "wb");
        std::vector<float> host((size_t) dump_envs * S);
        auto dump_obs = [&]() {
            if (!df) return;
            hip_check(hipMemcpy(host.data(), env.state(), host.size() * sizeof(float), hipMemcpyDeviceToHost), "dump");
            fwrite(host.data(), sizeof(float), host.size(), df);
        };
        uint32_t call = 0;
        auto one_step = [&]() {
            if (pol) check(evm_policy_forward(pol, n, env.state(), nullptr, (uint64_t) seed, d_action, d_logp, d_value, nullptr, nullptr, env.stream()));
            else hipLaunchKernelGGL(k_uniform_actions, dim3((n * A + 255) / 256), dim3(256), 0, env.stream(), d_action, n * A, call);
            call++;
            env.step_autoreset(d_action);
        };
        env.reset();
        dump_obs();
        for (int t = 0; t < warmup; t++) { one_step(); dump_obs(); }
        hip_check(hipDeviceSynchronize(), "sync");
        check(evm_env_clear_stats(env.handle()));
        const auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < steps; t++) { one_step(); dump_obs(); }
        hip_check(hipDeviceSynchronize(), "sync");
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        long long st[2] = {0, 0};
        check(evm_env_get_stats(env.handle(), st));
        if (df) fclose(df);
        printf("{\"host\": \"c++ (no torch)\", \"mode\": \"%s\", \"envs\": %d, \"steps\": %d, \"ms_per_step\": %.5f, \"env_steps_per_s\": %.1f, "
               "\"physics_steps_per_s\": %.1f, \"resets_started\": %lld}\n",
               mode.c_str(), n, steps, 1e3 * sec / steps, (double) st[0] / sec, (double) n * steps / sec, st[1]);
        if (pol) evm_policy_destroy(pol);
        (void) hipFree(d_action); (void) hipFree(d_logp); (void) hipFree(d_value);
    } catch (const std::exception &e) {
        fprintf(stderr, "rollout_main: %s\n", e.what());
        return 1;
    }
    return 0;
}
