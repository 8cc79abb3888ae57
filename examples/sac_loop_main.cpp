// SoftActorCriticAgentHip (sac_agent_hip.hpp) driven as the reference's train loop drives an Agent (src/train.cpp:55-66) over scripted
// episodes — the ones oracle/ref_sac_loop.cpp ran on the compiled reference, so tests/test_gpu_cxx_sac.py can hold this adapter to
// tests/golden/sac_loop_golden.txt and, bit for bit, to the Python SoftActorCriticAgent.
//
//   sac_loop_main --input <script.bin> --dump <out.bin>
//
// script.bin: int32 {S, A, H, batch_size, epoch, replay_buffer_size, train_every, n_episodes, seed}, int32 lengths[n_episodes],
//   int64 {n_actor, n_q}, float actor[n_actor], q[4][n_q] (critic_1, critic_2, target_critic_1, target_critic_2), then per episode
//   L x {state[S], reward, uniform[A], int32 trains, trains x epoch x {u_next[B][A], u_curr[B][A]}} and the terminal {state[S], reward}.
// out.bin: float actions[sum L][A], then actor | critic_1 | critic_2 | target_critic_1 | target_critic_2 | log_alpha.
// stdout: one JSON line (per act: global_curr_step before, buffer size after, train() calls; every batch's draws; the buffer after
//   every act / done as (1000 state[0], reward, done, 1000 next_state[0]) tuples; the meters).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include "agent_factory_hip.hpp"

using namespace evm_adapter;

template <typename X> static std::vector<X> read_n(FILE *f, size_t n) {
    std::vector<X> v(n);
    if (n && fread(v.data(), sizeof(X), n, f) != n) throw std::runtime_error("short script file");
    return v;
}
static std::string buffer_json(SoftActorCriticAgentHip &agent) {
    std::vector<float> s0, n0;
    agent.buffer().debug_first_values(s0, n0);
    std::string js = "[";
    for (size_t i = 0; i < s0.size(); i++) {
        char b[160];
        snprintf(b, sizeof b, "%s[%ld,%.9g,%d,%ld]", i ? "," : "", std::lround(s0[i] * 1000.0f), agent.buffer().reward[i], (int) agent.buffer().done[i],
                 std::lround(n0[i] * 1000.0f));
        js += b;
    }
    return js + "]";
}

int main(int argc, char **argv) {
    std::string input, dump;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--input" && i + 1 < argc) input = argv[++i];
        else if (a == "--dump" && i + 1 < argc) dump = argv[++i];
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    try {
        FILE *f = fopen(input.c_str(), "rb");
        if (!f) throw std::runtime_error("cannot open " + input);
        const auto hdr = read_n<int32_t>(f, 9);
        const int S = hdr[0], A = hdr[1], H = hdr[2], B = hdr[3], epoch = hdr[4], replay = hdr[5], train_every = hdr[6], n_ep = hdr[7], seed = hdr[8];
        const auto lengths = read_n<int32_t>(f, n_ep);
        const auto counts = read_n<int64_t>(f, 2);
        const auto actor = read_n<float>(f, counts[0]);
        std::vector<std::vector<float>> qs;
        for (int i = 0; i < 4; i++) qs.push_back(read_n<float>(f, counts[1]));
        // through the factory, with the reference's parameter keys (agent_factory.cpp:112-125)
        std::map<std::string, std::string> prm = {{"seed", std::to_string(seed)}, {"actor_hidden_size", std::to_string(H)}, {"critic_hidden_size", std::to_string(H)},
                                                  {"batch_size", std::to_string(B)}, {"epoch", std::to_string(epoch)}, {"learning_rate", "1e-3"}, {"gamma", "0.99"},
                                                  {"tau", "0.005"}, {"replay_buffer_size", std::to_string(replay)}, {"train_every", std::to_string(train_every)}};
        std::string missing_key;
        {
            auto less = prm;
            less.erase("tau");
            try { get_agent_factory("soft_actor_critic", less)->create_agent({S}, {A}); } catch (const std::invalid_argument &e) { missing_key = e.what(); }
        }
        auto agent = std::dynamic_pointer_cast<SoftActorCriticAgentHip>(get_agent_factory("soft_actor_critic", prm)->create_agent({S}, {A}));
        agent->set_parameters(actor, qs);

        float *d_state = nullptr, *d_uniform = nullptr;
        hip_check(hipMalloc(&d_state, sizeof(float) * S), "hipMalloc");
        hip_check(hipMalloc(&d_uniform, sizeof(float) * A), "hipMalloc");
        std::vector<float> actions;
        std::string js_act = "[", js_buf_act = "[", js_buf_done = "[";
        int k_act = 0;
        for (int k = 0; k < n_ep; k++) {
            for (int t = 0; t < lengths[k]; t++, k_act++) {
                const auto st = read_n<float>(f, S);
                const auto rw = read_n<float>(f, 1);
                const auto un = read_n<float>(f, A);
                const int trains = read_n<int32_t>(f, 1)[0];
                const auto tu = read_n<float>(f, (size_t) trains * epoch * 2 * B * A);
                hip_check(hipMemcpy(d_state, st.data(), sizeof(float) * S, hipMemcpyHostToDevice), "upload");
                hip_check(hipMemcpy(d_uniform, un.data(), sizeof(float) * A, hipMemcpyHostToDevice), "upload");
                const long before = agent->global_curr_step;
                const float *d_action = agent->act(d_state, rw[0], d_uniform, trains ? tu.data() : nullptr);
                std::vector<float> a(A);
                hip_check(hipMemcpy(a.data(), d_action, sizeof(float) * A, hipMemcpyDeviceToHost), "download");
                actions.insert(actions.end(), a.begin(), a.end());
                js_act += std::string(k_act ? "," : "") + "[" + std::to_string(before) + "," + std::to_string(agent->buffer().length()) + "," +
                          std::to_string(agent->trained_last_act) + "]";
                js_buf_act += (k_act ? "," : "") + buffer_json(*agent);
            }
            const auto st = read_n<float>(f, S);
            const auto rw = read_n<float>(f, 1);
            hip_check(hipMemcpy(d_state, st.data(), sizeof(float) * S, hipMemcpyHostToDevice), "upload");
            agent->done(d_state, rw[0]);
            js_buf_done += (k ? "," : "") + buffer_json(*agent);
        }
        fclose(f);
        hip_check(hipDeviceSynchronize(), "sync");
        std::string js_sample = "[";
        for (size_t i = 0; i < agent->draws.size(); i++) {
            js_sample += i ? ",[" : "[";
            for (size_t j = 0; j < agent->draws[i].size(); j++) js_sample += (j ? "," : "") + std::to_string(agent->draws[i][j]);
            js_sample += "]";
        }
        const std::vector<float> params = agent->get_parameters();
        if (!dump.empty()) {
            FILE *o = fopen(dump.c_str(), "wb");
            if (!o) throw std::runtime_error("cannot write " + dump);
            fwrite(actions.data(), sizeof(float), actions.size(), o);
            fwrite(params.data(), sizeof(float), params.size(), o);
            fclose(o);
        }
        const auto m = agent->get_metrics();
        printf("{\"act\": %s], \"sample\": %s], \"buffer_act\": %s], \"buffer_done\": %s], \"trains\": %ld, \"global_curr_step\": %ld, \"metric_names\": "
               "[\"%s\", \"%s\", \"%s\", \"%s\", \"%s\", \"%s\"], \"steps_meter\": %.9g, \"loss_meter_adds\": %d, \"missing_key\": \"%s\", \"count_parameters\": %d}\n",
               js_act.c_str(), js_sample.c_str(), js_buf_act.c_str(), js_buf_done.c_str(), agent->curr_train_step, agent->global_curr_step, m[0].name().c_str(),
               m[1].name().c_str(), m[2].name().c_str(), m[3].name().c_str(), m[4].name().c_str(), m[5].name().c_str(), m[4].loss(), (int) m[0].count(),
               missing_key.c_str(), agent->count_parameters());
        (void) hipFree(d_state); (void) hipFree(d_uniform);
    } catch (const std::exception &e) {
        fprintf(stderr, "sac_loop_main: %s\n", e.what());
        return 1;
    }
    return 0;
}
