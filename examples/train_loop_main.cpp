// The reference's train() loop, line for line, on the two compiled adapters — no torch, no Python (src/train.cpp:41-83):
//
//     step = env->reset();
//     for s in saves:  for e in episodes:
//         while (!step.done) step = env->do_step(agent->act(step.state, step.reward));
//         agent->done(step.state, step.reward);
//         step = env->reset();
//         metrics = agent->get_metrics();                      // the progress bar's text
//     agent->save(output_path / ("save_" + s));
//
// env = get_environment_factory("robot_walk", ...)->get_env(num_threads, seed)            (robot_walk_hip.hpp)
// agent = get_agent_factory("ppo_gae", ...)->create_agent(state_space, action_space)      (ppo_gae_agent_hip.hpp)
//
//   train_loop_main --skeleton <file> [--episodes 12] [--saves 1] [--seed 1234] [--batch-size 4] [--train-every 4] [--epoch 2]
//                   [--self-collision 1] [--weights <flat actor | critic floats>] [--out <folder>] [--dump <weights file>]
//
// stdout: one JSON line (episode lengths, trains, the drawn trajectories, the meters' text); tests/test_gpu_cxx_loop.py runs the
// same loop through the Python adapters with the same draws and compares the weights bit for bit.
#include <hip/hip_runtime.h>
#include <sys/stat.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "agent_factory_hip.hpp"

using namespace evm_adapter;

static std::string join(const std::vector<int> &v) {
    std::string s = "[";
    for (size_t i = 0; i < v.size(); i++) s += (i ? "," : "") + std::to_string(v[i]);
    return s + "]";
}

int main(int argc, char **argv) {
    std::string skeleton, out_dir, dump, weights;
    int episodes = 12, saves = 1, seed = 1234, batch_size = 4, train_every = 4, epoch = 2, selfcol = 1;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "--skeleton") skeleton = next();
        else if (a == "--episodes") episodes = atoi(next());
        else if (a == "--saves") saves = atoi(next());
        else if (a == "--seed") seed = atoi(next());
        else if (a == "--batch-size") batch_size = atoi(next());
        else if (a == "--train-every") train_every = atoi(next());
        else if (a == "--epoch") epoch = atoi(next());
        else if (a == "--self-collision") selfcol = atoi(next());
        else if (a == "--out") out_dir = next();
        else if (a == "--dump") dump = next();
        else if (a == "--weights") weights = next();
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    try {
        std::map<std::string, std::string> env_prm = {{"self_collision", std::to_string(selfcol)}};
        if (!skeleton.empty()) env_prm["skeleton_json_path"] = skeleton;
        auto env = get_environment_factory("robot_walk", env_prm)->get_env(/*num_threads=*/8, seed);
        std::map<std::string, std::string> agent_prm = {
            {"seed", std::to_string(seed)}, {"hidden_size", "256"}, {"gamma", "0.99"}, {"lambda", "0.95"}, {"epsilon", "0.2"},
            {"entropy_factor", "0.01"}, {"critic_loss_factor", "0.5"}, {"epoch", std::to_string(epoch)}, {"batch_size", std::to_string(batch_size)},
            {"train_every", std::to_string(train_every)}, {"replay_buffer_size", "64"}, {"learning_rate", "1e-3"}, {"clip_grad_norm", "0.5"}};
        std::shared_ptr<Agent> agent = get_agent_factory("ppo_gae", agent_prm)->create_agent(env->get_state_space(), env->get_action_space());
        auto ppo = std::dynamic_pointer_cast<PpoGaeAgentHip>(agent);
        if (!weights.empty()) {  // instead of init_weights (init.cpp:7-21): the caller's flat parameters, actor then critic
            FILE *f = fopen(weights.c_str(), "rb");
            if (!f) throw std::runtime_error("cannot open " + weights);
            std::vector<float> w(agent->count_parameters());
            const size_t got = fread(w.data(), sizeof(float), w.size(), f);
            fclose(f);
            if (got != w.size()) throw std::runtime_error("short weights file");
            const size_t n_actor = (size_t) 256 * env->get_state_space()[0] + 3 * 256 + 256 * 256 + 3 * 256 + 2 * ((size_t) env->get_action_space()[0] * 256 + env->get_action_space()[0]);
            ppo->set_parameters(std::vector<float>(w.begin(), w.begin() + n_actor), std::vector<float>(w.begin() + n_actor, w.end()));
        }
        agent->set_eval(false);

        step st = env->reset();
        std::vector<int> lengths;
        std::string js_sample = "[";
        long trains = 0, total_steps = 0;
        std::string text;
        for (int s = 0; s < saves; s++) {
            for (int e = 0; e < episodes; e++) {
                int len = 0;
                while (!st.done) { st = env->do_step(agent->act(st.state, st.reward)); len++; }
                agent->done(st.state, st.reward);
                if (ppo->trained_last_done) { js_sample += (trains ? "," : "") + join(ppo->buffer().last_draw); trains++; }
                st = env->reset();
                lengths.push_back(len);
                total_steps += len;
                text = "Save " + std::to_string(s - 1);
                for (const auto &m : agent->get_metrics()) text += ", " + m.to_string();
            }
            if (!out_dir.empty()) {
                const std::string folder = out_dir + "/save_" + std::to_string(s);
                mkdir(folder.c_str(), 0755);
                agent->save(folder);
            }
        }
        hip_check(hipDeviceSynchronize(), "sync");
        if (!dump.empty()) {
            const std::vector<float> p = ppo->get_parameters();
            FILE *o = fopen(dump.c_str(), "wb");
            if (!o) throw std::runtime_error("cannot write " + dump);
            fwrite(p.data(), sizeof(float), p.size(), o);
            fclose(o);
        }
        printf("{\"lengths\": %s, \"steps\": %ld, \"trains\": %ld, \"sample\": %s], \"state_space\": %lld, \"action_space\": %lld, \"parameters_count\": %d, "
               "\"metrics\": \"%s\"}\n",
               join(lengths).c_str(), total_steps, trains, js_sample.c_str(), (long long) env->get_state_space()[0], (long long) env->get_action_space()[0],
               agent->count_parameters(), text.c_str());
    } catch (const std::exception &e) {
        fprintf(stderr, "train_loop_main: %s\n", e.what());
        return 1;
    }
    return 0;
}
