// The reference's train loop as the AGENT sees it (src/train.cpp:55-66: act(state, reward) per step, done(state, reward) at the
// end of an episode) driven on PpoGaeAgentHip (ppo_gae_agent_hip.hpp) over scripted episodes — the ones oracle/ref_loop.cpp ran
// on the compiled reference, so tests/test_gpu_cxx_agent.py can hold this adapter to tests/golden/agent_loop_golden.txt and,
// bit for bit, to the Python PpoGaeAgent.
//
//   agent_loop_main --input <script.bin> --dump <out.bin> [--ckpt <folder>]
//
// script.bin: int32 {S, A, H, epoch, batch_size, train_every, replay_buffer_size, n_episodes, seed}, int32 lengths[n_episodes],
//   int64 {n_actor, n_critic}, float actor[n_actor], critic[n_critic], then per episode L x {state[S], reward, uniform[A]} and the
//   terminal {state[S], reward}.
// out.bin: float actions[sum L][A], params[n_actor + n_critic], then the newest complete trajectory: float L2, reward[L2], done[L2],
//   curr_value[L2], next_value[L2], log_prob[L2][A].
// stdout: one JSON line with the bookkeeping after every done().
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "agent_factory_hip.hpp"

using namespace evm_adapter;

template <typename X> static std::vector<X> read_n(FILE *f, size_t n) {
    std::vector<X> v(n);
    if (n && fread(v.data(), sizeof(X), n, f) != n) throw std::runtime_error("short script file");
    return v;
}
static std::string join(const std::vector<int> &v) {
    std::string s = "[";
    for (size_t i = 0; i < v.size(); i++) s += (i ? "," : "") + std::to_string(v[i]);
    return s + "]";
}
template <typename X> static std::vector<X> from_device(const X *d, size_t n) {
    std::vector<X> h(n);
    hip_check(hipMemcpy(h.data(), d, sizeof(X) * n, hipMemcpyDeviceToHost), "download");
    return h;
}

int main(int argc, char **argv) {
    std::string input, dump, ckpt;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--input" && i + 1 < argc) input = argv[++i];
        else if (a == "--dump" && i + 1 < argc) dump = argv[++i];
        else if (a == "--ckpt" && i + 1 < argc) ckpt = argv[++i];
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    try {
        FILE *f = fopen(input.c_str(), "rb");
        if (!f) throw std::runtime_error("cannot open " + input);
        const auto hdr = read_n<int32_t>(f, 9);
        const int S = hdr[0], A = hdr[1], H = hdr[2], epoch = hdr[3], batch_size = hdr[4], train_every = hdr[5], replay = hdr[6], n_ep = hdr[7],
                  seed = hdr[8];
        const auto lengths = read_n<int32_t>(f, n_ep);
        const auto counts = read_n<int64_t>(f, 2);
        const auto actor = read_n<float>(f, counts[0]), critic = read_n<float>(f, counts[1]);

        // through the factory, with the reference's parameter keys (agent_factory.cpp:137-146)
        std::map<std::string, std::string> prm = {
            {"seed", std::to_string(seed)}, {"hidden_size", std::to_string(H)}, {"gamma", "0.99"}, {"lambda", "0.95"}, {"epsilon", "0.2"},
            {"entropy_factor", "0.01"}, {"critic_loss_factor", "0.5"}, {"epoch", std::to_string(epoch)}, {"batch_size", std::to_string(batch_size)},
            {"train_every", std::to_string(train_every)}, {"replay_buffer_size", std::to_string(replay)}, {"learning_rate", "1e-3"},
            {"clip_grad_norm", "0.5"}};
        std::string missing_key, unknown_name;
        {
            auto less = prm;
            less.erase("gamma");
            try { get_agent_factory("ppo_gae", less)->create_agent({S}, {A}); } catch (const std::invalid_argument &e) { missing_key = e.what(); }
            try { get_agent_factory("no_such_agent", prm); } catch (const std::invalid_argument &e) { unknown_name = e.what(); }
        }
        std::shared_ptr<Agent> base = get_agent_factory("ppo_gae", prm)->create_agent({S}, {A});
        auto agent = std::dynamic_pointer_cast<PpoGaeAgentHip>(base);
        if ((size_t) agent->count_parameters() != actor.size() + critic.size()) throw std::runtime_error("parameter count");
        agent->set_parameters(actor, critic);

        float *d_state = nullptr, *d_uniform = nullptr;
        hip_check(hipMalloc(&d_state, sizeof(float) * S), "hipMalloc");
        hip_check(hipMalloc(&d_uniform, sizeof(float) * A), "hipMalloc");
        std::vector<float> actions;
        std::string js_done = "[", js_buffer = "[", js_sample = "[";
        int trains = 0;
        for (int k = 0; k < n_ep; k++) {
            for (int t = 0; t < lengths[k]; t++) {
                const auto st = read_n<float>(f, S);
                const auto rw = read_n<float>(f, 1);
                const auto un = read_n<float>(f, A);
                hip_check(hipMemcpy(d_state, st.data(), sizeof(float) * S, hipMemcpyHostToDevice), "upload");
                hip_check(hipMemcpy(d_uniform, un.data(), sizeof(float) * A, hipMemcpyHostToDevice), "upload");
                const float *d_action = agent->act(d_state, rw[0], d_uniform);
                const auto a = from_device(d_action, A);
                actions.insert(actions.end(), a.begin(), a.end());
            }
            const auto st = read_n<float>(f, S);
            const auto rw = read_n<float>(f, 1);
            hip_check(hipMemcpy(d_state, st.data(), sizeof(float) * S, hipMemcpyHostToDevice), "upload");
            const std::vector<int> before = {k, (int) agent->global_curr_step, (int) agent->buffer().memory.size(),
                                             (int) agent->buffer().filtered_positions().size()};
            agent->done(d_state, rw[0]);
            js_done += std::string(k ? "," : "") + "[" + std::to_string(before[0]) + "," + std::to_string(before[1]) + "," + std::to_string(before[2]) +
                       "," + std::to_string(before[3]) + "," + (agent->trained_last_done ? "1" : "0") + "]";
            if (agent->trained_last_done) {
                js_sample += (trains ? "," : "") + join(agent->buffer().last_draw);
                trains++;
            }
            std::vector<int> lens;
            for (const auto &t : agent->buffer().memory) lens.push_back(t->size());
            js_buffer += (k ? "," : "") + join(lens);
        }
        fclose(f);
        hip_check(hipDeviceSynchronize(), "sync");

        const std::vector<float> params = agent->get_parameters();
        // save -> load into a second agent: parameters, moments and step counts survive
        bool ckpt_equal = false;
        if (!ckpt.empty()) {
            agent->save(ckpt);
            auto other = std::dynamic_pointer_cast<PpoGaeAgentHip>(get_agent_factory("ppo_gae", prm)->create_agent({S}, {A}));
            other->load(ckpt);
            ckpt_equal = other->get_parameters() == params;
            bool missing_throws = false;
            try { other->load(ckpt + "/no_such_folder"); } catch (const std::runtime_error &) { missing_throws = true; }
            ckpt_equal = ckpt_equal && missing_throws;
        }
        if (!dump.empty()) {
            FILE *o = fopen(dump.c_str(), "wb");
            if (!o) throw std::runtime_error("cannot write " + dump);
            fwrite(actions.data(), sizeof(float), actions.size(), o);
            fwrite(params.data(), sizeof(float), params.size(), o);
            const auto &mem = agent->buffer().memory;
            const TrajectoryHip &last = *mem[mem.size() - 2];
            const int L2 = last.size();
            std::vector<float> tail = {(float) L2};
            tail.insert(tail.end(), last.reward.begin(), last.reward.end());
            for (uint8_t d : last.done) tail.push_back(d ? 1.f : 0.f);
            const auto v = from_device(last.dev().value, L2), nv = from_device(last.dev().next_value, L2), lp = from_device(last.dev().logp, (size_t) L2 * A);
            tail.insert(tail.end(), v.begin(), v.end());
            tail.insert(tail.end(), nv.begin(), nv.end());
            tail.insert(tail.end(), lp.begin(), lp.end());
            fwrite(tail.data(), sizeof(float), tail.size(), o);
            fclose(o);
        }
        // the debug agents through the same factory (debug_agents.cpp:7-39): three actions of the random agent after seed 1234 — the
        // compiled reference's golden (tests/golden/agent_golden.txt: random_agent_actions) — and the constant agent
        std::string js_random = "[";
        {
            auto ra = get_agent_factory("random", {{"seed", "1234"}})->create_agent({S}, {12});
            for (int k = 0; k < 3; k++) {
                const auto a = from_device(ra->act(d_state, 0.f), 12);
                for (int i = 0; i < 12; i++) { char b[32]; snprintf(b, sizeof b, "%s%.9g", (k || i) ? "," : "", a[i]); js_random += b; }
            }
            ra->done(d_state, 0.f); ra->save("/nonexistent"); ra->load("/nonexistent");
            if (ra->count_parameters() != 0 || !ra->get_metrics().empty()) throw std::runtime_error("debug agent interface");
        }
        const auto ca = from_device(get_agent_factory("constant", {{"action_value", "0.25"}})->create_agent({S}, {A})->act(d_state, 0.f), A);
        std::string constant_missing;
        try { get_agent_factory("constant", {})->create_agent({S}, {A}); } catch (const std::invalid_argument &e) { constant_missing = e.what(); }
        const auto metrics = agent->get_metrics();
        // the reference's own known answers for LossMeter (evo_motion_networks/tests/src/test_metrics.cpp:20-25)
        auto meter_case = [](std::vector<float> v, int window) { LossMeterHip m("test", window); for (float x : v) m.add(x); return m.loss(); };
        const float mk[3] = {meter_case({1.f, 2.f, 1.f, 2.f}, 4), meter_case({1.f, 2.f, 1.f, 2.f}, 2), meter_case({1.f, 1.f, 2.f, 2.f}, 2)};
        printf("{\"done\": %s], \"buffer\": %s], \"sample\": %s], \"trains\": %d, \"curr_train_step\": %ld, \"actor_loss\": %.17g, \"critic_loss\": %.17g, "
               "\"metric_names\": [\"%s\", \"%s\", \"%s\"], \"steps_meter\": %.9g, \"missing_key\": \"%s\", \"unknown_name\": \"%s\", \"ckpt_equal\": %s, "
               "\"count_parameters\": %d, \"loss_meter_adds\": %d, \"meter_known_answers\": [%.9g, %.9g, %.9g], \"steps_string\": \"%s\", \"random_actions\": %s], \"constant_action\": [%.9g, %.9g], \"constant_missing\": \"%s\"}\n",
               js_done.c_str(), js_buffer.c_str(), js_sample.c_str(), trains, agent->curr_train_step, agent->last_actor_loss, agent->last_critic_loss,
               metrics[0].name().c_str(), metrics[1].name().c_str(), metrics[2].name().c_str(), metrics[2].loss(), missing_key.c_str(),
               unknown_name.c_str(), ckpt_equal ? "true" : "false", agent->count_parameters(), (int) metrics[0].count(), mk[0], mk[1], mk[2],
               metrics[2].to_string().c_str(), js_random.c_str(), ca[0], ca[A - 1], constant_missing.c_str());
        (void) hipFree(d_state); (void) hipFree(d_uniform);
    } catch (const std::exception &e) {
        fprintf(stderr, "agent_loop_main: %s\n", e.what());
        return 1;
    }
    return 0;
}
