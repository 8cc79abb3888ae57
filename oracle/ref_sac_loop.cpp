// Golden-vector generator for the Agent SURFACE of the SAC path: SoftActorCriticAgent::act / done / check_train and its flat
// ReplayBuffer (evo_motion_networks/src/agents/soft_actor_critic.cpp:47-91,172-180, src/replay_buffer.cpp:16-52,146-153) driven over
// scripted episodes.  This file is OURS; it calls the reference's compiled evo_motion_networks library (oracle/ref_build.sh)
// through its public headers and prints what went in and what came out.  Only the printed vectors are committed
// (tests/golden/sac_loop_golden.txt).
//
// Scripted: pattern weights (as ref_sac.cpp), episodes of given lengths, states / rewards from the `pat` hash (tests/golden_io.py).
// Recorded: the uniform draws of every truncated_normal_sample — the one of act() and, when check_train() fires inside that act(),
// the two of every train() — read by re-seeding the global generator around the call; the transitions every train() sampled
// (the buffer's own std::mt19937 + std::shuffle replayed on a copy); the buffer after every call (reward, done and the tags of
// state / next_state of every element: this is where the reference's update_last-after-done() shows); the networks at the end.
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
#include <evo_motion_networks/agents/soft_actor_critic.h>
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
        const bool ln = name.find(".2.") != std::string::npos || name.find(".5.") != std::string::npos ||
                        name.find(".8.") != std::string::npos;
        if (p.dim() == 2) scale = 2.0f / std::sqrt((float) p.size(1));
        else if (ln) { scale = 0.2f; if (name.find("weight") != std::string::npos) offset = 1.f; }
        else scale = 0.2f;
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
// state t of episode k; element 0 carries a tag (100 k + t) / 1000 that identifies it in the buffer dumps
static torch::Tensor state_of(int S, int episode, int t) {
    auto x = torch::zeros({S});
    auto a = x.accessor<float, 1>();
    for (int j = 0; j < S; j++) a[j] = pat(31u + (uint32_t) episode, (uint32_t) (t * S + j), 2.0f);
    a[0] = (float) (100 * episode + t) / 1000.f;
    return x;
}

int main() {
    torch::set_num_threads(1);
    const int S = 371, A = 12, H = 256;
    const int batch_size = 4, epoch = 2, replay_size = 9, train_every = 3;
    SoftActorCriticAgent agent(1234, {S}, {A}, H, H, batch_size, epoch, 1e-3f, 0.99f, 0.005f, replay_size, train_every);
    fill_module(agent.actor, 100);
    fill_module(agent.critic_1, 300);
    fill_module(agent.critic_2, 400);
    fill_module(agent.target_critic_1, 500);
    fill_module(agent.target_critic_2, 600);
    const std::vector<int> lengths = {4, 1, 5, 3, 2, 4};
    printf("# SoftActorCriticAgent act/done over scripted episodes: hidden %d batch_size %d epoch %d lr 1e-3 gamma 0.99 tau 0.005 "
           "replay_buffer_size %d train_every %d\n", H, batch_size, epoch, replay_size, train_every);
    printf("config %d %d %d %d %d %d %d\n", S, A, H, batch_size, epoch, replay_size, train_every);
    printf("lengths %d", (int) lengths.size());
    for (int l : lengths) printf(" %d", l);
    printf("\n");
    std::vector<torch::Tensor> act_u, actions, train_u_next, train_u_curr;
    int trains = 0, call = 0;
    auto dump_buffer = [&](const char *what, int k, int t) {
        printf("buffer %s %d %d size %d :", what, k, t, (int) agent.replay_buffer.memory.size());
        for (auto &e : agent.replay_buffer.memory)
            printf(" (%d,%.9g,%d,%d)", (int) std::lround(e.state[0].item<float>() * 1000.f), e.reward, (int) e.done,
                   (int) std::lround(e.next_state[0].item<float>() * 1000.f));
        printf("\n");
    };
    for (int k = 0; k < (int) lengths.size(); k++) {
        const int L = lengths[k];
        for (int t = 0; t < L; t++, call++) {
            const auto state = state_of(S, k, t);
            const float reward = pat(77u, (uint32_t) (100 * k + t), 1.0f);  // of the previous transition; at t = 0 the reset()'s own
            // what this act() is going to do (soft_actor_critic.cpp:47-91): update_last + add, then check_train on the new size
            const int size_after = std::min((int) agent.replay_buffer.memory.size() + 1, replay_size);
            const bool will_train = (agent.global_curr_step % train_every == train_every - 1) && size_after - 1 >= batch_size;
            const uint64_t seed = 9000u + (uint64_t) call;
            at::manual_seed(seed);
            act_u.push_back(at::rand({A}));
            printf("act %d %d global_curr_step %ld size_after %d train %d\n", k, t, (long) agent.global_curr_step, size_after, will_train ? 1 : 0);
            if (will_train) {
                auto gen = agent.replay_buffer.rand_gen;  // a copy: the buffer's own generator advances identically inside act()
                for (int e = 0; e < epoch; e++) {
                    train_u_next.push_back(at::rand({batch_size, A}));
                    train_u_curr.push_back(at::rand({batch_size, A}));
                    std::vector<int> index(size_after - 1);
                    std::iota(index.begin(), index.end(), 0);
                    std::shuffle(index.begin(), index.end(), gen);
                    printf("sample %d %d", trains, batch_size);
                    for (int i = 0; i < batch_size; i++) printf(" %d", index[i]);  // positions in memory (after this act()'s add)
                    printf("\n");
                    trains++;
                }
            }
            at::manual_seed(seed);
            actions.push_back(agent.act(state, reward).detach().clone());
            dump_buffer("act", k, t);
        }
        agent.done(state_of(S, k, L), pat(77u, (uint32_t) (100 * k + L), 1.0f));
        dump_buffer("done", k, L);
    }
    printf("trains %d curr_train_step %ld global_curr_step %ld\n", trains, (long) agent.curr_train_step, (long) agent.global_curr_step);
    dump("uniform", torch::stack(act_u));
    dump("actions", torch::stack(actions));
    dump("train_u_next", torch::stack(train_u_next));
    dump("train_u_curr", torch::stack(train_u_curr));
    agent.set_eval(true);
    const int B = 8;
    auto X = torch::zeros({B, S}), AC = torch::zeros({B, A});
    for (int i = 0; i < B; i++) {
        for (int j = 0; j < S; j++) X[i][j] = pat(7, (uint32_t) (i * S + j), 2.0f);
        for (int j = 0; j < A; j++) AC[i][j] = pat(22, (uint32_t) (i * A + j), 1.8f);
    }
    { auto [m, s] = agent.actor->forward(X); dump("after_mu", m); dump("after_sigma", s); }
    { auto [q] = agent.critic_1->forward(X, AC); dump("after_q1", q); }
    { auto [q] = agent.critic_2->forward(X, AC); dump("after_q2", q); }
    { auto [q] = agent.target_critic_1->forward(X, AC); dump("after_tq1", q); }
    { auto [q] = agent.target_critic_2->forward(X, AC); dump("after_tq2", q); }
    dump("after_log_alpha", agent.entropy_parameter->log_alpha());
    return 0;
}
