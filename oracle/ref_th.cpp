// Checkpoint-format harness (test infrastructure).  This file is OURS; it calls the reference's compiled
// evo_motion_networks library (oracle/ref_build.sh) through its public headers:
//   ref_th save <folder>          ActorModule / CriticModule with the pattern weights of ref_golden.cpp written by the
//                                 reference's own save_torch (saver.h:13-25) -> <folder>/actor.th, critic.th
//   ref_th load <folder> <file> [cpu]   reference's load_torch (saver.h:27-39) into an ActorModule, forward on the fixed
//                                 input pattern, prints mu / sigma as text
// Only printed vectors are committed (tests/golden/th_golden.txt); nothing of the reference travels.
#include <torch/torch.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include <evo_motion_networks/networks/actor.h>
#include <evo_motion_networks/networks/critic.h>
#include <evo_motion_networks/saver.h>

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

int main(int argc, char **argv) {
    torch::set_num_threads(1);
    const int S = 371, A = 12, H = 256, B = 8;
    if (argc >= 3 && !strcmp(argv[1], "save")) {
        auto actor = std::make_shared<ActorModule>(std::vector<int64_t>{S}, std::vector<int64_t>{A}, H);
        auto critic = std::make_shared<CriticModule>(std::vector<int64_t>{S}, H);
        fill_module(actor, 100);
        fill_module(critic, 200);
        save_torch(argv[2], actor, "actor.th");
        save_torch(argv[2], critic, "critic.th");
        return 0;
    }
    if (argc >= 4 && !strcmp(argv[1], "load")) {
        auto actor = std::make_shared<ActorModule>(std::vector<int64_t>{S}, std::vector<int64_t>{A}, H);
        if (argc >= 5 && !strcmp(argv[4], "cpu")) {
            // the checkpoint shipped with the reference holds CUDA tensors; load_torch() has no device argument and
            // cannot open it on a CPU-only LibTorch, so the same two calls are made here with a device
            torch::serialize::InputArchive archive;
            archive.load_from((std::filesystem::path(argv[2]) / argv[3]).string(), torch::Device(torch::kCPU));
            actor->load(archive);
        } else {
            load_torch(argv[2], actor, argv[3]);
        }
        actor->eval();
        auto X = torch::zeros({B, S});
        { auto a = X.accessor<float, 2>(); for (int i = 0; i < B; i++) for (int j = 0; j < S; j++) a[i][j] = pat(7, (uint32_t) (i * S + j), 2.0f); }
        auto [mu, sigma] = actor->forward(X);
        dump("mu", mu); dump("sigma", sigma);
        return 0;
    }
    if (argc >= 3 && !strcmp(argv[1], "saveopt")) {
        // the reference's own optimiser archive: Adam over the pattern actor, two steps on a fixed loss, save_torch (ppo_gae.cpp:194)
        auto actor = std::make_shared<ActorModule>(std::vector<int64_t>{S}, std::vector<int64_t>{A}, H);
        fill_module(actor, 100);
        auto opt = std::make_shared<torch::optim::Adam>(actor->parameters(), 1e-3);
        auto X = torch::zeros({B, S});
        { auto a = X.accessor<float, 2>(); for (int i = 0; i < B; i++) for (int j = 0; j < S; j++) a[i][j] = pat(7, (uint32_t) (i * S + j), 2.0f); }
        for (int it = 0; it < 2; it++) {
            auto [mu, sigma] = actor->forward(X);
            auto loss = (mu * mu).sum() + sigma.sum();
            opt->zero_grad(); loss.backward(); opt->step();
        }
        save_torch(argv[2], opt, "actor_optimizer.th");
        save_torch(argv[2], actor, "actor_after.th");
        return 0;
    }
    if (argc >= 4 && !strcmp(argv[1], "loadopt")) {
        // load_torch (ppo_gae.cpp:201) of an optimiser archive into a fresh Adam; prints what arrived
        auto actor = std::make_shared<ActorModule>(std::vector<int64_t>{S}, std::vector<int64_t>{A}, H);
        auto opt = std::make_shared<torch::optim::Adam>(actor->parameters(), 5e-2);
        load_torch(argv[2], opt, argv[3]);
        auto &g = opt->param_groups()[0];
        auto &o = static_cast<torch::optim::AdamOptions &>(g.options());
        printf("scalar lr %.9g\nscalar beta1 %.9g\nscalar eps %.9g\n", o.lr(), std::get<0>(o.betas()), o.eps());
        int idx = 0;
        for (auto &p : g.params()) {
            auto it = opt->state().find(p.unsafeGetTensorImpl());
            if (it == opt->state().end()) { printf("scalar state_%d_missing 1\n", idx++); continue; }
            auto &st = static_cast<torch::optim::AdamParamState &>(*it->second);
            printf("scalar step_%d %lld\n", idx, (long long) st.step());
            if (idx == 0 || idx == 9) {
                char name[64];
                snprintf(name, sizeof name, "exp_avg_%d_head", idx); dump(name, st.exp_avg().view({-1}).slice(0, 0, 8));
                snprintf(name, sizeof name, "exp_avg_sq_%d_head", idx); dump(name, st.exp_avg_sq().view({-1}).slice(0, 0, 8));
            }
            idx++;
        }
        return 0;
    }
    fprintf(stderr, "usage: ref_th save <folder> | ref_th load <folder> <file> | ref_th saveopt <folder> | ref_th loadopt <folder> <file>\n");
    return 2;
}
