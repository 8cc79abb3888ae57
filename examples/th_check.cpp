// Command-line front of th_archive.hpp for the tests (plain C++17, no HIP, no torch):
//   th_check write-actor  <out.th> S A H <flat.bin>          ActorModule archive from its flat named_parameters() vector
//   th_check write-critic <out.th> S H <flat.bin>            CriticModule archive
//   th_check read <in.th> <flat_out.bin>                     prints one line per parameter (name, shape), writes them flat
//   th_check write-actor-folder <folder> S A H <params.bin> <m.bin> <v.bin> <step> <lr>    actor.th + actor_optimizer.th, as
//                                                             PpoGaeAgentHip::save writes them for its actor
//   th_check read-actor-folder <folder> S A H <out.bin>      prints `step N` and `lr X`, writes [params | exp_avg | exp_avg_sq] flat
#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "th_archive.hpp"

static std::vector<float> read_floats(const char *path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error(std::string("cannot read ") + path);
    std::string s((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    std::vector<float> v(s.size() / 4);
    memcpy(v.data(), s.data(), v.size() * 4);
    return v;
}
static void write_floats(const char *path, const std::vector<float> &v) {
    std::ofstream f(path, std::ios::binary);
    f.write((const char *) v.data(), (std::streamsize) (v.size() * 4));
}

int main(int argc, char **argv) {
    try {
        const std::string cmd = argc > 1 ? argv[1] : "";
        if (cmd == "write-actor" && argc == 7) {
            const auto flat = read_floats(argv[6]);
            evm_th::save(argv[2], evm_th::actor_module(atoll(argv[3]), atoll(argv[4]), atoll(argv[5]), flat.data()));
            return 0;
        }
        if (cmd == "write-critic" && argc == 6) {
            const auto flat = read_floats(argv[5]);
            evm_th::save(argv[2], evm_th::critic_module(atoll(argv[3]), atoll(argv[4]), flat.data()));
            return 0;
        }
        if (cmd == "read" && argc == 4) {
            const evm_th::Node root = evm_th::load(argv[2]);
            std::vector<std::pair<std::string, const evm_th::Node *>> ps;
            evm_th::named_parameters(root, "", ps);
            std::vector<float> flat;
            for (const auto &kv : ps) {
                printf("%s", kv.first.c_str());
                for (int64_t d : kv.second->shape) printf(" %lld", (long long) d);
                printf("\n");
                flat.insert(flat.end(), kv.second->f32.begin(), kv.second->f32.end());
            }
            write_floats(argv[3], flat);
            return 0;
        }
        if (cmd == "write-actor-folder" && argc == 11) {
            const std::string folder = argv[2];
            const int64_t S = atoll(argv[3]), A = atoll(argv[4]), H = atoll(argv[5]);
            const auto w = read_floats(argv[6]), m = read_floats(argv[7]), v = read_floats(argv[8]);
            const evm_th::Node module = evm_th::actor_module(S, A, H, w.data());
            evm_th::save(folder + "/actor.th", module);
            evm_th::save(folder + "/actor_optimizer.th", evm_th::adam_archive(evm_th::adam_params_of(module, atoll(argv[9]), m.data(), v.data()), atof(argv[10])));
            return 0;
        }
        if (cmd == "read-actor-folder" && argc == 7) {
            const std::string folder = argv[2];
            const int64_t S = atoll(argv[3]), A = atoll(argv[4]), H = atoll(argv[5]);
            const size_t n = (size_t) (H * S + H + 2 * H + H * H + H + 2 * H + 2 * (A * H + A));
            const evm_th::Node module = evm_th::load(folder + "/actor.th");
            std::vector<float> out = evm_th::flat_parameters(module, n, "actor.th");
            std::vector<evm_th::AdamParam> ps = evm_th::adam_params_of(module, 0, nullptr, nullptr);
            double lr = 0;
            evm_th::adam_from_archive(evm_th::load(folder + "/actor_optimizer.th"), ps, &lr);
            std::vector<float> m, v;
            const int64_t step = evm_th::flat_adam(ps, m, v);
            printf("step %lld\nlr %.9g\n", (long long) step, lr);
            out.insert(out.end(), m.begin(), m.end());
            out.insert(out.end(), v.begin(), v.end());
            write_floats(argv[6], out);
            return 0;
        }
        fprintf(stderr, "usage: see the head of examples/th_check.cpp\n");
        return 2;
    } catch (const std::exception &e) {
        fprintf(stderr, "th_check: %s\n", e.what());
        return 1;
    }
}
