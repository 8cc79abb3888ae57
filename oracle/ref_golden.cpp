// Golden-vector generator for the AGENT half of the path.  This file is OURS; it calls the reference's
// compiled evo_motion_networks library (built by oracle/ref_build.sh from the sources where they lie under
// /root/reference) through its public headers and prints inputs/outputs as text.  Only the printed vectors are
// committed (tests/golden/agent_golden.txt); nothing of the reference travels.
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
// reach PpoGaeAgent::train / actor / critic for the update golden vector (test harness only); every standard and
// torch header is already included above, so the macro only affects the reference's own class definitions
#define private public
#include <evo_motion_networks/agents/ppo_gae.h>
#undef private
#include <evo_motion_networks/agents/debug_agents.h>
#include <evo_motion_networks/functions.h>
#include <evo_motion_networks/networks/actor.h>
#include <evo_motion_networks/networks/critic.h>

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
            scale = 0.2f;                                  // LayerNorm weight / bias
            if (name.find("weight") != std::string::npos) offset = 1.f;
        } else scale = 0.2f;                               // Linear bias
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
    auto actor = std::make_shared<ActorModule>(std::vector<int64_t>{S}, std::vector<int64_t>{A}, H);
    auto critic = std::make_shared<CriticModule>(std::vector<int64_t>{S}, H);
    printf("# parameter order\n");
    for (auto &np : actor->named_parameters()) { printf("param actor %s", np.key().c_str()); for (auto s : np.value().sizes()) printf(" %d", (int) s); printf("\n"); }
    for (auto &np : critic->named_parameters()) { printf("param critic %s", np.key().c_str()); for (auto s : np.value().sizes()) printf(" %d", (int) s); printf("\n"); }
    fill_module(actor, 100);
    fill_module(critic, 200);
    actor->eval(); critic->eval();
    auto X = torch::zeros({B, S});
    { auto a = X.accessor<float, 2>(); for (int i = 0; i < B; i++) for (int j = 0; j < S; j++) a[i][j] = pat(7, (uint32_t) (i * S + j), 2.0f); }
    auto [mu, sigma] = actor->forward(X);
    auto [value] = critic->forward(X);
    dump("X", X); dump("mu", mu); dump("sigma", sigma); dump("value", value);
    // single-state path (Agent::act uses 1-D states)
    auto [mu1, sigma1] = actor->forward(X[0]);
    dump("mu_1d", mu1); dump("sigma_1d", sigma1);

    // truncated normal on a grid (functions.cpp:53-68,94-128)
    std::vector<float> mus = {-0.95f, -0.3f, 0.f, 0.5f, 0.99f}, sig = {1e-7f, 1e-3f, 0.05f, 0.3f, 1.f, 5.f, 1e7f},
                       xs = {-1.f, -0.5f, 0.f, 0.7f, 1.f};
    std::vector<float> gm, gs, gx;
    for (float m : mus) for (float s : sig) for (float x : xs) { gm.push_back(m); gs.push_back(s); gx.push_back(x); }
    auto tm = torch::tensor(gm), ts = torch::tensor(gs), tx = torch::tensor(gx);
    dump("tn_mu", tm); dump("tn_sigma", ts); dump("tn_x", tx);
    dump("tn_log_pdf", truncated_normal_log_pdf(tx, tm, ts, -1.f, 1.f));
    dump("tn_entropy", truncated_normal_entropy(tm, ts, -1.f, 1.f));
    at::manual_seed(4321);
    auto U = at::rand(tm.sizes());
    at::manual_seed(4321);
    auto smp = truncated_normal_sample(tm, ts, -1.f, 1.f);
    dump("tn_u", U); dump("tn_sample", smp);

    // one PpoGaeAgent::train() call on a synthetic padded batch (ppo_gae.cpp:117-190)
    {
        const int Bt = 4, T = 7;
        PpoGaeAgent agent(1234, {S}, {A}, H, 0.99f, 0.95f, 0.2f, 0.01f, 0.5f, /*epoch*/ 2, /*batch*/ Bt, /*train_every*/ 1,
                          /*replay*/ 16, 1e-3f, 0.5f);
        fill_module(agent.actor, 100);
        fill_module(agent.critic, 200);
        auto st = torch::zeros({Bt, T, S}), ac = torch::zeros({Bt, T, A}), rw = torch::zeros({Bt, T, 1}), dn = torch::zeros({Bt, T, 1});
        int lens[4] = {7, 5, 3, 6};
        for (int b = 0; b < Bt; b++) for (int t = 0; t < T; t++) {
            bool pad = t >= lens[b];
            for (int j = 0; j < S; j++) st[b][t][j] = pad ? 0.f : pat(11, (uint32_t) ((b * T + t) * S + j), 2.0f);
            for (int j = 0; j < A; j++) ac[b][t][j] = pad ? 0.f : pat(12, (uint32_t) ((b * T + t) * A + j), 1.8f);
            rw[b][t][0] = pad ? 0.f : pat(13, (uint32_t) (b * T + t), 1.0f);
            dn[b][t][0] = (pad || t == lens[b] - 1) ? 1.f : 0.f;
        }
        torch::Tensor lp, cv, nv;
        {
            torch::NoGradGuard g;
            agent.set_eval(true);
            auto [m0, s0] = agent.actor->forward(st);
            lp = truncated_normal_log_pdf(ac, m0, s0, -1.f, 1.f);
            cv = std::get<0>(std::tuple<torch::Tensor>(agent.critic->forward(st).value));
            nv = torch::cat({cv.slice(1, 1, T), torch::zeros({Bt, 1, 1})}, 1);
            auto valid = (torch::arange(T).view({1, T, 1}) < torch::tensor({7, 5, 3, 6}).view({Bt, 1, 1})).to(torch::kFloat32);
            lp = lp * valid; cv = cv * valid; nv = nv * valid;
        }
        dump("ppo_states", st); dump("ppo_actions", ac); dump("ppo_rewards", rw); dump("ppo_done", dn);
        dump("ppo_log_prob", lp); dump("ppo_curr_values", cv); dump("ppo_next_values", nv);
        agent.train(st, ac, rw, dn, lp, cv, nv);
        agent.set_eval(true);
        auto [m1, s1] = agent.actor->forward(X);
        auto [v1] = agent.critic->forward(X);
        dump("ppo_after_mu", m1); dump("ppo_after_sigma", s1); dump("ppo_after_value", v1);
        dump("ppo_after_actor_w0_row0", agent.actor->named_parameters()["head.0.weight"][0]);
    }
    // RandomAgent::act (debug_agents.cpp:28-30): the first three actions after manual_seed(1234)
    {
        at::manual_seed(1234);
        RandomAgent ra({A});
        std::vector<torch::Tensor> acts;
        for (int i = 0; i < 3; i++) acts.push_back(ra.act(X[0], 0.f));
        dump("random_agent_actions", torch::stack(acts));
    }
    return 0;
}
