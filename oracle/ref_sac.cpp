// Golden-vector generator for the SAC rows.  This file is OURS; it calls the reference's compiled
// evo_motion_networks library (oracle/ref_build.sh) through its public headers and prints inputs/outputs as text
// (tests/golden/sac_golden.txt).  Nothing of the reference travels.
//   - QNetworkModule forward on fixed inputs with pattern weights (q_net.cpp:8-43)
//   - one SoftActorCriticAgent::train() call (soft_actor_critic.cpp:93-170) on a fixed batch; the two at::rand draws
//     it makes are recorded (the generator is re-seeded to the same state to read them)
//   - ReplayBuffer add / update_last / FIFO eviction / has_enough on a tiny case (replay_buffer.cpp:16-52,146-153)
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
#include <evo_motion_networks/agents/soft_actor_critic.h>
#undef private
#include <evo_motion_networks/functions.h>

#include <cstdint>
#include <cstdio>

static float pat(uint32_t tensor, uint32_t k, float scale) {
    uint32_t h = tensor * 2654435761u + k * 40503u + 12345u;
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    return ((float) (h & 0xFFFFFFu) / 16777216.0f - 0.5f) * scale;
}
// same rule as ref_golden.cpp: Linear weight U(+-1/sqrt(in)), LayerNorm weight 1 + U(+-0.1), other vectors U(+-0.1);
// a tensor is a LayerNorm tensor when it is 1-D and the next-lower index in the Sequential is not a Linear — here
// decided by name: indices 2, 5, 8 of the trunks hold the LayerNorms
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

int main() {
    torch::set_num_threads(1);
    const int S = 371, A = 12, H = 256, B = 8;
    SoftActorCriticAgent agent(1234, {S}, {A}, H, H, /*batch*/ B, /*epoch*/ 1, 1e-3f, 0.99f, 0.005f, /*replay*/ 128, /*train_every*/ 4);
    printf("# parameter order\n");
    for (auto &np : agent.critic_1->named_parameters()) { printf("param q %s", np.key().c_str()); for (auto s : np.value().sizes()) printf(" %d", (int) s); printf("\n"); }
    for (auto &np : agent.entropy_parameter->named_parameters()) { printf("param entropy %s", np.key().c_str()); for (auto s : np.value().sizes()) printf(" %d", (int) s); printf("\n"); }
    printf("scalar count_parameters %d\n", agent.count_parameters());
    printf("scalar target_entropy %.9g\n", agent.target_entropy);
    fill_module(agent.actor, 100);
    fill_module(agent.critic_1, 300);
    fill_module(agent.critic_2, 400);
    fill_module(agent.target_critic_1, 500);
    fill_module(agent.target_critic_2, 600);
    auto X = torch::zeros({B, S}), AC = torch::zeros({B, A}), NX = torch::zeros({B, S}), RW = torch::zeros({B, 1}), DN = torch::zeros({B, 1});
    for (int i = 0; i < B; i++) {
        for (int j = 0; j < S; j++) { X[i][j] = pat(7, (uint32_t) (i * S + j), 2.0f); NX[i][j] = pat(21, (uint32_t) (i * S + j), 2.0f); }
        for (int j = 0; j < A; j++) AC[i][j] = pat(22, (uint32_t) (i * A + j), 1.8f);
        RW[i][0] = pat(23, (uint32_t) i, 1.0f);
        DN[i][0] = (i % 3 == 2) ? 1.f : 0.f;
    }
    agent.set_eval(true);
    dump("sac_states", X); dump("sac_actions", AC); dump("sac_rewards", RW); dump("sac_done", DN); dump("sac_next_states", NX);
    { auto [q] = agent.critic_1->forward(X, AC); dump("q1_before", q); }
    { auto [q] = agent.target_critic_2->forward(X, AC); dump("tq2_before", q); }
    { auto [q] = agent.critic_1->forward(X[0], AC[0]); dump("q1_before_1d", q); }
    // the two uniform draws train() will make, in order: next_action sample, curr_action sample
    at::manual_seed(777);
    dump("sac_u_next", at::rand({B, A}));
    dump("sac_u_curr", at::rand({B, A}));
    at::manual_seed(777);
    agent.set_eval(false);
    agent.train(X, AC, RW, DN, NX);
    agent.set_eval(true);
    { auto [m, s] = agent.actor->forward(X); dump("after_mu", m); dump("after_sigma", s); }
    { auto [q] = agent.critic_1->forward(X, AC); dump("after_q1", q); }
    { auto [q] = agent.critic_2->forward(X, AC); dump("after_q2", q); }
    { auto [q] = agent.target_critic_1->forward(X, AC); dump("after_tq1", q); }
    { auto [q] = agent.target_critic_2->forward(X, AC); dump("after_tq2", q); }
    dump("after_log_alpha", agent.entropy_parameter->log_alpha());
    printf("scalar loss_actor %.9g\nscalar loss_critic_1 %.9g\nscalar loss_critic_2 %.9g\nscalar loss_entropy %.9g\n",
           agent.actor_loss_meter.loss(), agent.critic_1_loss_meter.loss(), agent.critic_2_loss_meter.loss(),
           agent.entropy_loss_meter.loss());

    // flat replay buffer semantics (capacity 4): add / update_last / eviction; contents printed as (tag, reward, done, next tag)
    {
        ReplayBuffer rb(4, 1234);
        auto tag = [](float v) { return torch::full({1}, v); };
        printf("replay empty %d\n", (int) rb.empty());
        for (int k = 0; k < 6; k++) {
            if (!rb.empty()) rb.update_last((float) (10 + k), tag((float) k), k == 3);
            rb.add({tag((float) k), tag((float) (100 + k)), 0.f, false, tag((float) k)});
            printf("replay after_add %d size %d has_enough2 %d has_enough4 %d :", k, (int) rb.memory.size(), (int) rb.has_enough(2), (int) rb.has_enough(4));
            for (auto &e : rb.memory) printf(" (%g,%g,%g,%d,%g)", e.state.item<float>(), e.action.item<float>(), e.reward, (int) e.done, e.next_state.item<float>());
            printf("\n");
        }
        auto smp = rb.sample(3);
        printf("replay sample3 never_the_newest %d count %d\n", (int) std::all_of(smp.begin(), smp.end(), [&](const episode_step &e) { return e.state.item<float>() != 5.f; }), (int) smp.size());
    }
    return 0;
}
