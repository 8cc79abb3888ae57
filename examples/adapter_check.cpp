// Exercises the single-environment adapter the way src/train.cpp:41-66 drives an Environment:
//   env = factory->get_env(num_threads, seed); step = env->reset(); while (!step.done) step = env->do_step(action); reset()
// and the factory's error behaviour (env_factory.cpp:118).  Exit code 0 = every check held.  Needs a GPU.
#include <cstdio>
#include <cstring>

#include "robot_walk_hip.hpp"

using namespace evm_adapter;

#define EXPECT(cond) do { if (!(cond)) { fprintf(stderr, "adapter_check: %s failed (line %d)\n", #cond, __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: adapter_check <skeleton>\n"); return 2; }
    try {
        bool threw = false;
        try { get_environment_factory("cartpole", {}); } catch (const std::invalid_argument &) { threw = true; }
        EXPECT(threw);  // unknown environment name
        threw = false;
        try { get_environment_factory("robot_walk", {{"no_such_parameter", "1"}}); } catch (const std::invalid_argument &) { threw = true; }
        EXPECT(!threw);  // unknown parameter keys are ignored, like EnvironmentFactory::generic_get_value (env_factory.cpp:22-28)
        try { get_environment_factory("robot_walk", {{"no_such_parameter", "1"}, {"strict", "1"}}); } catch (const std::invalid_argument &) { threw = true; }
        EXPECT(threw);   // ... unless the adapter's own `strict` switch is on
        auto factory = get_environment_factory("robot_walk", {{"skeleton_json_path", argv[1]}, {"max_episode_seconds", "2"}});
        auto env = factory->get_env(/*num_threads=*/8, /*seed=*/1234);
        EXPECT(env->get_state_space() == std::vector<int64_t>{371} && env->get_action_space() == std::vector<int64_t>{12});
        float *d_action;
        hip_check(hipMalloc(&d_action, 12 * sizeof(float)), "hipMalloc");
        float h_action[12];
        for (int i = 0; i < 12; i++) h_action[i] = (i % 2) ? 0.5f : -0.5f;
        hip_check(hipMemcpy(d_action, h_action, sizeof(h_action), hipMemcpyHostToDevice), "copy");
        int episodes = 0, calls = 0;
        step s = env->reset();
        EXPECT(s.state != nullptr);
        while (episodes < 2 && calls < 400) {   // train.cpp:61-66
            while (!s.done && calls < 400) { s = env->do_step(d_action); calls++; }
            episodes++;
            s = env->reset();
        }
        EXPECT(episodes == 2 && calls >= 2 && calls <= 2 * 119);  // max_episode_seconds = 2 -> at most 119 steps each
        printf("adapter_check ok: %d episodes, %d do_step calls\n", episodes, calls);
        (void) hipFree(d_action);
    } catch (const std::exception &e) {
        fprintf(stderr, "adapter_check: %s\n", e.what());
        return 1;
    }
    return 0;
}
