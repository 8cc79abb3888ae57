// Golden-vector generator for the Agent SURFACE of the PPO path: PpoGaeAgent::act / done / check_train and its
// TrajectoryReplayBuffer (evo_motion_networks/src/agents/ppo_gae.cpp:29-115, src/replay_buffer.cpp:73-138,176-189) driven
// over scripted episodes.  This file is OURS; it calls the reference's compiled evo_motion_networks library (built by
// oracle/ref_build.sh from the sources where they lie under /root/reference) through its public headers and prints what
// went in and what came out as text.  Only the printed vectors are committed (tests/golden/agent_loop_golden.txt).
//
// What is scripted: pattern weights (the same as ref_golden.cpp), K episodes of given lengths (one of length 1, which the
// buffer never samples), states and rewards from the `pat` hash below (re-implemented in tests/golden_io.py), the uniform
// draws of truncated_normal_sample recorded by re-seeding the global generator around every act().  What is recorded: every
// action, the trajectories the buffer's own std::mt19937 + std::shuffle picked at every check_train() (replayed on a copy of
// its generator right before the call), the buffer's shape after every done(), and the networks after the last episode.
#include <torch/torch.h>

#include <algorithm>
#include <deque>
#include <filesystem>
#include <map>
#include <memory>
#include <numeric>
#include <optional>
#include <random>
#include <string>
#include <tuple>
#include <vector>
#define private public
#define protected public
#include <evo_motion_networks/agents/ppo_gae.h>
#undef private
#undef protected
#include <evo_motion_networks/functions.h>

#include <cstdint>
#include <cstdio>

static float pat(uint32_t tensor, uint32_t k, float scale) {
    uint32_t h = tensor * 2654435761u + k * 40503u + 12345u;
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    return ((float) (h & 0xFFFFFFu) / 16777216.0f - 0.5f) * scale;
}
static void fill_module(const std::shared_ptr<torch::nn::Module> &m, uint32_t base) {
    torch::NoGradGuard g;
    uint32_t t = base;
    for (auto &np : m->named_parameters()) {
        auto p = np.value();
        const std::string &name = np.key();
        float scale, offset = 0.f;
        if (p.dim() == 2) scale = 2.0f / std::sqrt((float) p.size(1));
        else if (name.find(".2.") != std::string::npos || name.find(".5.") != std::string::npos) {
            scale = 0.2f;
            if (name.find("weight") != std::string::npos) offset = 1.f;
        } else scale = 0.2f;
        auto flat = p.view({-1});
        auto acc = flat.accessor<float, 1>();
        for (int64_t k = 0; k < flat.size(0); k++) acc[k] = offset + pat(t, (uint32_t) k, scale);
        t++;
    }
}
static void dump(const char *name, const torch::Tensor &x) {
    auto t = x.detach().to(torch::kFloat32).contiguous();
    printf("tensor %s %d", name, (int) t.dim());
    for (auto s : t.sizes()) printf(" %d", (int) s);
    printf("\n");
    auto f = t.view({-1});
    for (int64_t i = 0; i < f.size(0); i++) printf("%.9g%c", f[i].item<float>(), (i % 8 == 7 || i == f.size(0) - 1) ? '\n' : ' ');
}
static torch::Tensor state_of(int S, int episode, int t) {
    auto x = torch::zeros({S});
    auto a = x.accessor<float, 1>();
    for (int j = 0; j < S; j++) a[j] = pat(31u + (uint32_t) episode, (uint32_t) (t * S + j), 2.0f);
    return x;
}

int main() {
    torch::set_num_threads(1);
    const int S = 371, A = 12, H = 256;
    const int batch_size = 3, train_every = 2, replay_size = 5, epoch = 2;
    PpoGaeAgent agent(1234, {S}, {A}, H, 0.99f, 0.95f, 0.2f, 0.01f, 0.5f, epoch, batch_size, train_every, replay_size, 1e-3f, 0.5f);
    fill_module(agent.actor, 100);
    fill_module(agent.critic, 200);
    const std::vector<int> lengths = {4, 1, 6, 3, 5, 2, 7, 4, 3, 5};
    printf("# PpoGaeAgent act/done over scripted episodes: hidden %d gamma 0.99 lambda 0.95 epsilon 0.2 entropy 0.01 critic 0.5 epoch %d "
           "batch_size %d train_every %d replay_buffer_size %d lr 1e-3 clip 0.5\n", H, epoch, batch_size, train_every, replay_size);
    printf("config %d %d %d %d %d %d %d\n", S, A, H, epoch, batch_size, train_every, replay_size);
    printf("lengths %d", (int) lengths.size());
    for (int l : lengths) printf(" %d", l);
    printf("\n");
    printf("pat_check %.9g %.9g %.9g\n", pat(31, 0, 2.0f), pat(40, 1234, 2.0f), pat(77, 99, 1.0f));
    std::vector<torch::Tensor> all_u, all_actions;
    int trains = 0;
    for (int k = 0; k < (int) lengths.size(); k++) {
        const int L = lengths[k];
        for (int t = 0; t < L; t++) {
            const auto state = state_of(S, k, t);
            const float reward = pat(77u, (uint32_t) (100 * k + t), 1.0f);  // reward of the PREVIOUS transition (ignored at t = 0: the trajectory is empty)
            const uint64_t seed = 5000u + 100u * (uint64_t) k + (uint64_t) t;
            at::manual_seed(seed);
            all_u.push_back(at::rand({A}));
            at::manual_seed(seed);
            all_actions.push_back(agent.act(state, reward).detach().clone());
        }
        const auto terminal = state_of(S, k, L);
        const float last_reward = pat(77u, (uint32_t) (100 * k + L), 1.0f);
        // what check_train() is about to do, replayed on copies (ppo_gae.cpp:63-66, replay_buffer.cpp:73-93,126-133)
        {
            auto &mem = agent.replay_buffer.memory;
            std::vector<int> filtered;
            for (int i = 0; i < (int) mem.size(); i++)
                if (mem[i].trajectory.size() > 1) filtered.push_back(i);
            const bool will_train = (agent.global_curr_step % train_every == train_every - 1) && (int) filtered.size() >= batch_size;
            printf("done %d global_curr_step %ld memory %d filtered %d train %d\n", k, (long) agent.global_curr_step, (int) mem.size(),
                   (int) filtered.size(), will_train ? 1 : 0);
            if (will_train) {
                auto gen = agent.replay_buffer.rand_gen;  // a copy: the buffer's own generator advances identically inside done()
                std::vector<int> index(filtered.size() - 1);
                std::iota(index.begin(), index.end(), 0);
                std::shuffle(index.begin(), index.end(), gen);
                printf("sample %d %d", trains, std::min(batch_size, (int) index.size()));
                for (int i = 0; i < batch_size && i < (int) index.size(); i++) printf(" %d", filtered[index[i]]);  // positions in memory
                printf("\n");
                trains++;
            }
        }
        agent.done(terminal, last_reward);
        printf("buffer %d", (int) agent.replay_buffer.memory.size());
        for (auto &tr : agent.replay_buffer.memory) printf(" %d", (int) tr.trajectory.size());
        printf("\n");
    }
    printf("trains %d curr_train_step %ld\n", trains, (long) agent.curr_train_step);
    dump("uniform", torch::stack(all_u));
    dump("actions", torch::stack(all_actions));
    // the stored fields of the newest complete trajectory (episode 9): rewards / done / values as update_last left them
    {
        auto &mem = agent.replay_buffer.memory;
        auto &tr = mem[mem.size() - 2].trajectory;
        std::vector<float> rw, dn;
        std::vector<torch::Tensor> cv, nv, lp;
        for (auto &st : tr) { rw.push_back(st.reward); dn.push_back(st.done ? 1.f : 0.f); cv.push_back(st.curr_value.view({-1})); nv.push_back(st.next_value.view({-1})); lp.push_back(st.log_prob); }
        dump("last_rewards", torch::tensor(rw)); dump("last_done", torch::tensor(dn));
        dump("last_values", torch::cat(cv)); dump("last_next_values", torch::cat(nv)); dump("last_log_prob", torch::stack(lp));
    }
    agent.set_eval(true);
    const int B = 8;
    auto X = torch::zeros({B, S});
    { auto a = X.accessor<float, 2>(); for (int i = 0; i < B; i++) for (int j = 0; j < S; j++) a[i][j] = pat(7, (uint32_t) (i * S + j), 2.0f); }
    auto [m1, s1] = agent.actor->forward(X);
    auto [v1] = agent.critic->forward(X);
    dump("after_mu", m1); dump("after_sigma", s1); dump("after_value", v1);
    dump("after_actor_w0_row0", agent.actor->named_parameters()["head.0.weight"][0]);
    dump("after_critic_w0_row0", agent.critic->named_parameters()["critic.0.weight"][0]);
    return 0;
}
