// Torch-free data-parallel PPO training on the C ABI and RCCL: one process per GPU, the reference's train() loop
// (src/train.cpp:41-83: act -> do_step -> ... -> PpoGaeAgent::train, ppo_gae.cpp:117-190) for N environments per rank.
//
//   RANK=r WORLD_SIZE=w LOCAL_RANK=l MASTER_PORT=p train_main --skeleton <file> [--envs 1024] [--horizon 16] [--iters 4]
//                [--epoch 4] [--seed 1234] [--self-collision 1] [--dump <weights file>]
//
// Per iteration: `horizon` x (evm_policy_forward, evm_env_step_autoreset) straight into the rows of the rollout buffer, then
// evm_ppo_gae -> ncclAllGather of the three advantage statistics -> evm_ppo_gae_merge (device) -> evm_ppo_gae_normalize, then
// `epoch` x (evm_ppo_grads -> ONE in-place ncclAllReduce over the trainer's [actor | critic] gradient buffer ->
// evm_ppo_apply).  Everything is enqueued on one HIP stream; the host never waits inside an iteration and never reads a
// device value (the global count of selected transitions stays on the device).  With WORLD_SIZE unset it is a single process
// whose communicator has one rank: tests/test_gpu_cxx_train.py compares its weights bit for bit with the Python update.
//
// Rendezvous: rank 0 writes the ncclUniqueId to $EVM_NCCL_ID_FILE (default /tmp/evm_nccl_id.<MASTER_PORT>), the others poll it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <signal.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "robot_walk_hip.hpp"

using namespace evm_adapter;

static void nccl_check(ncclResult_t r, const char *what) {
    if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
}

// the deterministic parameters of rollout_main.cpp (reproduced in tests/test_gpu_cxx_host.py)
static float pattern(uint32_t k) {
    uint32_t h = k * 2654435761u + 12345u;
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    return (float) (h >> 8) * (1.0f / 16777216.0f) * 2.0f - 1.0f;
}
static void push_linear(std::vector<float> &v, int out, int in, uint32_t &k) {
    const float s = 1.0f / std::sqrt((float) in);
    for (int i = 0; i < out * in; i++) v.push_back(pattern(k++) * s);
    for (int i = 0; i < out; i++) v.push_back(0.f);
}
static void push_layernorm(std::vector<float> &v, int n) {
    for (int i = 0; i < n; i++) v.push_back(1.f);
    for (int i = 0; i < n; i++) v.push_back(0.f);
}
static std::vector<float> make_params(int S, int A, int H, bool actor, uint32_t base) {
    std::vector<float> v;
    uint32_t k = base;
    push_linear(v, H, S, k); push_layernorm(v, H);
    push_linear(v, H, H, k); push_layernorm(v, H);
    if (actor) { push_linear(v, A, H, k); push_linear(v, A, H, k); }
    else push_linear(v, 1, H, k);
    return v;
}

// mask = (valid == 1): only do_step transitions are trained on; next_values[t] = values[t + 1], the last row from the extra
// forward pass over the observation after the horizon
__global__ void k_rollout_finish(int T, int N, const uint8_t *valid, const float *values, const float *last_value, uint8_t *mask,
                                 float *next_values) {
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t) T * N) return;
    mask[i] = valid[i] == 1 ? 1 : 0;
    next_values[i] = i + N < (size_t) T * N ? values[i + N] : last_value[i - (size_t) (T - 1) * N];
}

template <typename X> static X *dalloc(size_t n) {
    X *p = nullptr;
    hip_check(hipMalloc(&p, n * sizeof(X)), "hipMalloc");
    hip_check(hipMemset(p, 0, n * sizeof(X)), "hipMemset");
    return p;
}

int main(int argc, char **argv) {
    std::string skeleton, dump;
    int n = 1024, T = 16, iters = 4, epoch = 4, seed = 1234, selfcol = 1;
    for (int i = 1; i < argc; i++) {
        auto arg = [&](const char *name) { return !strcmp(argv[i], name) && i + 1 < argc; };
        if (arg("--skeleton")) skeleton = argv[++i];
        else if (arg("--envs")) n = atoi(argv[++i]);
        else if (arg("--horizon")) T = atoi(argv[++i]);
        else if (arg("--iters")) iters = atoi(argv[++i]);
        else if (arg("--epoch")) epoch = atoi(argv[++i]);
        else if (arg("--seed")) seed = atoi(argv[++i]);
        else if (arg("--self-collision")) selfcol = atoi(argv[++i]);
        else if (arg("--dump")) dump = argv[++i];
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if (skeleton.empty()) { fprintf(stderr, "--skeleton <robot_walk skeleton> is required\n"); return 2; }
    auto env_int = [](const char *k, int dflt) { const char *v = getenv(k); return v ? atoi(v) : dflt; };
    const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), local = env_int("LOCAL_RANK", 0);
    try {
        hip_check(hipSetDevice(local), "hipSetDevice");
        // ---- one communicator over all ranks (RCCL over xGMI inside a node)
        ncclUniqueId id;
        const char *idf = getenv("EVM_NCCL_ID_FILE");
        const std::string id_path = idf ? idf : std::string("/tmp/evm_nccl_id.") + std::to_string(env_int("MASTER_PORT", 29500));
        // Rendez-vous through a file.  What can go wrong, and what is done about it:
        //   * a stale id file of a run that crashed on the same MASTER_PORT: rank 0 removes it (and its .tmp) before it asks for
        //     the new id, and a reader only accepts a file written after its own start (mtime), so an old id is never used and
        //     ncclCommInitRank cannot block on mismatched ids;
        //   * a rank 0 that never comes: the readers give up after EVM_RENDEZVOUS_TIMEOUT_S seconds (default 60) with a message;
        //   * a communicator that never forms (a rank died after publishing / reading): a watchdog ends the process with a
        //     message instead of leaving it hung in ncclCommInitRank.
        const int rdv_timeout = env_int("EVM_RENDEZVOUS_TIMEOUT_S", 60);
        const time_t t_start = time(nullptr);
        if (rank == 0) {
            if (world > 1) { unlink(id_path.c_str()); unlink((id_path + ".tmp").c_str()); }
            nccl_check(ncclGetUniqueId(&id), "ncclGetUniqueId");
            if (world > 1) {
                const std::string tmp = id_path + ".tmp";
                FILE *f = fopen(tmp.c_str(), "wb");
                if (!f || fwrite(&id, sizeof(id), 1, f) != 1) throw std::runtime_error("cannot write " + tmp);
                fclose(f);
                if (rename(tmp.c_str(), id_path.c_str())) throw std::runtime_error("cannot publish " + id_path);
            }
        } else {
            bool got = false;
            for (int tries = 0; tries < rdv_timeout * 10 && !got; tries++) {
                struct stat sb;
                if (stat(id_path.c_str(), &sb) == 0 && sb.st_mtime >= t_start - 1 && sb.st_size == (off_t) sizeof(id)) {
                    FILE *f = fopen(id_path.c_str(), "rb");
                    got = f && fread(&id, sizeof(id), 1, f) == 1;
                    if (f) fclose(f);
                }
                if (!got) usleep(100000);
            }
            if (!got) throw std::runtime_error("rank " + std::to_string(rank) + ": no fresh ncclUniqueId at " + id_path + " after " +
                                               std::to_string(rdv_timeout) + " s (is rank 0 running with the same MASTER_PORT / EVM_NCCL_ID_FILE?)");
        }
        ncclComm_t comm;
        {
            static char wd_msg[256];
            snprintf(wd_msg, sizeof(wd_msg), "train_main: rank %d of %d: ncclCommInitRank did not complete within %d s (a peer is missing or "
                     "read another id); giving up\n", rank, world, 2 * rdv_timeout);
            signal(SIGALRM, [](int) { (void) !write(2, wd_msg, strlen(wd_msg)); _exit(3); });
            alarm((unsigned) (2 * rdv_timeout));
            nccl_check(ncclCommInitRank(&comm, world, id, rank), "ncclCommInitRank");
            alarm(0);
            signal(SIGALRM, SIG_DFL);
        }
        hipStream_t s;
        hip_check(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate");

        // ---- environments, policy, trainer
        auto factory = get_environment_factory("robot_walk", {{"skeleton_json_path", skeleton}, {"self_collision", std::to_string(selfcol)}});
        EvmEnv *env = nullptr;
        check(evm_env_create(factory->skeleton.c_str(), n, local, (uint64_t) (seed + rank), &factory->prm, &env));
        int S = 0, A = 0;
        check(evm_env_spaces(env, &S, &A));
        EvmPolicy *pol = nullptr;
        check(evm_policy_create(S, A, 256, local, &pol));
        const std::vector<float> pa = make_params(S, A, 256, true, 1000u), pc = make_params(S, A, 256, false, 500000u);  // the same on every rank
        check(evm_policy_set_weights(pol, pa.data(), pa.size(), pc.data(), pc.size()));
        const size_t rows = (size_t) T * n;
        EvmPpo *tr = nullptr;
        check(evm_ppo_create(pol, rows, &tr));
        float *d_pa = dalloc<float>(pa.size()), *d_pc = dalloc<float>(pc.size());
        hip_check(hipMemcpy(d_pa, pa.data(), pa.size() * 4, hipMemcpyHostToDevice), "upload");
        hip_check(hipMemcpy(d_pc, pc.data(), pc.size() * 4, hipMemcpyHostToDevice), "upload");
        check(evm_ppo_set_params(tr, d_pa, d_pc, 1, s));
        // the gradient buffer is OURS: registered once, all-reduced in place, read by the optimiser kernel
        size_t gn = 0, goff = 0;
        check(evm_ppo_grad_buffer(tr, nullptr, nullptr, &gn, &goff));
        float *d_grads = dalloc<float>(gn);
        check(evm_ppo_grad_buffer(tr, d_grads, nullptr, nullptr, nullptr));

        // ---- rollout buffer, time-major; states has T + 1 slabs: the env writes observation t + 1 into slab t + 1
        float *states = dalloc<float>((rows + n) * S), *actions = dalloc<float>(rows * A), *logp = dalloc<float>(rows * A);
        float *values = dalloc<float>(rows), *next_values = dalloc<float>(rows), *rewards = dalloc<float>(rows);
        float *adv = dalloc<float>(rows), *ret = dalloc<float>(rows);
        uint8_t *done = dalloc<uint8_t>(rows), *valid = dalloc<uint8_t>(rows), *mask = dalloc<uint8_t>(rows);
        float *sc_action = dalloc<float>((size_t) n * A), *sc_logp = dalloc<float>((size_t) n * A), *sc_value = dalloc<float>(n);
        float *r0 = dalloc<float>(n);
        uint8_t *d0 = dalloc<uint8_t>(n);
        double *stats = dalloc<double>(3), *all_stats = dalloc<double>((size_t) 3 * world);
        // weights are the same on every rank, the exploration noise is not (agent.py: noise_seed)
        const uint64_t noise_seed = ((uint64_t) seed ^ ((uint64_t) rank * 0x9E3779B97F4A7C15ull)) & 0x7FFFFFFFull;
        const float gamma = 0.99f, lam = 0.95f, eps = 0.2f, ef = 0.01f, cf = 0.5f, lr = 1e-3f, clip = 0.5f;  // agent_factory.cpp:137-146

        check(evm_env_reset(env, nullptr, states, r0, d0, s));
        hipEvent_t e0, e1, e2;
        hip_check(hipEventCreate(&e0), "event"); hip_check(hipEventCreate(&e1), "event"); hip_check(hipEventCreate(&e2), "event");
        float ms_roll = 0.f, ms_upd = 0.f;
        hip_check(hipStreamSynchronize(s), "sync");
        check(evm_env_clear_stats(env));
        const auto t0 = std::chrono::steady_clock::now();
        for (int it = 0; it < iters; it++) {
            if (it > 0) hip_check(hipMemcpyAsync(states, states + rows * S, (size_t) n * S * 4, hipMemcpyDeviceToDevice, s), "carry");
            hip_check(hipEventRecord(e0, s), "event");
            for (int t = 0; t < T; t++) {
                const size_t o = (size_t) t * n;
                check(evm_policy_forward(pol, n, states + o * S, nullptr, noise_seed, actions + o * A, logp + o * A, values + o, nullptr, nullptr, s));
                check(evm_env_step_autoreset(env, actions + o * A, states + (o + n) * S, rewards + o, done + o, valid + o, s));
            }
            check(evm_policy_forward(pol, n, states + rows * S, nullptr, noise_seed, sc_action, sc_logp, sc_value, nullptr, nullptr, s));
            hipLaunchKernelGGL(k_rollout_finish, dim3((unsigned) ((rows + 255) / 256)), dim3(256), 0, s, T, n, valid, values, sc_value, mask, next_values);
            hip_check(hipEventRecord(e1, s), "event");
            // ---- PpoGaeAgent::train over all ranks
            check(evm_ppo_gae(tr, T, n, rewards, done, values, next_values, mask, gamma, lam, adv, stats, s));
            nccl_check(ncclAllGather(stats, all_stats, 3, ncclDouble, comm, s), "ncclAllGather");            // 24 B per rank
            check(evm_ppo_gae_merge(tr, all_stats, world, nullptr, s));
            check(evm_ppo_gae_normalize(tr, T, n, nullptr, values, adv, ret, s));
            // the epochs run on the selected rows alone (settle calls and reset emissions weigh nothing): the one host read of an
            // iteration, of this rank's count
            size_t nsel = 0;
            const float *u_states = states, *u_actions = actions, *u_logp = logp, *u_adv = adv, *u_ret = ret;
            const uint8_t *u_mask = mask;
            {
                const float *c0, *c1, *c2, *c3, *c4; const uint8_t *c5;
                check(evm_ppo_select_rows(tr, rows, mask, states, actions, logp, adv, ret, &nsel, &c0, &c1, &c2, &c3, &c4, &c5, s));
                if (nsel > 0 && nsel < rows) { u_states = c0; u_actions = c1; u_logp = c2; u_adv = c3; u_ret = c4; u_mask = c5; }
                else nsel = rows;
            }
            for (int ep = 0; ep < epoch; ep++) {
                check(evm_ppo_grads(tr, nsel, u_states, u_actions, u_logp, u_adv, u_ret, u_mask, -1.0 /* count on the device */, eps, ef, cf, ep > 0, s));
                nccl_check(ncclAllReduce(d_grads, d_grads, gn, ncclFloat, ncclSum, comm, s), "ncclAllReduce");  // one collective per epoch
                check(evm_ppo_apply(tr, lr, clip, s));
            }
            hip_check(hipEventRecord(e2, s), "event");
            hip_check(hipEventSynchronize(e2), "sync");  // timing only (once per iteration, after everything is queued)
            float a = 0.f, b = 0.f;
            hip_check(hipEventElapsedTime(&a, e0, e1), "elapsed"); hip_check(hipEventElapsedTime(&b, e1, e2), "elapsed");
            ms_roll += a; ms_upd += b;
        }
        hip_check(hipStreamSynchronize(s), "sync");
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        long long st[2] = {0, 0};
        check(evm_env_get_stats(env, st));
        double la = 0.0, lc = 0.0, h_stats[3] = {0, 0, 0};
        check(evm_ppo_losses(tr, &la, &lc, s));
        hip_check(hipMemcpy(h_stats, all_stats, sizeof(h_stats), hipMemcpyDeviceToHost), "stats");
        int errs[2] = {0, 0};
        check(evm_env_get_errors(env, errs, 0, s));
        if (!dump.empty()) {
            std::vector<float> h(pa.size() + pc.size());
            check(evm_ppo_copy(tr, 0, 0, 0, d_pa, s)); check(evm_ppo_copy(tr, 0, 1, 0, d_pc, s));
            hip_check(hipStreamSynchronize(s), "sync");
            hip_check(hipMemcpy(h.data(), d_pa, pa.size() * 4, hipMemcpyDeviceToHost), "download");
            hip_check(hipMemcpy(h.data() + pa.size(), d_pc, pc.size() * 4, hipMemcpyDeviceToHost), "download");
            const std::string path = world > 1 ? dump + "." + std::to_string(rank) : dump;
            FILE *f = fopen(path.c_str(), "wb");
            if (!f || fwrite(h.data(), 4, h.size(), f) != h.size()) throw std::runtime_error("cannot write " + path);
            fclose(f);
        }
        if (rank == 0)
            printf("{\"host\": \"c++ (no torch), rccl\", \"world\": %d, \"envs_per_rank\": %d, \"horizon\": %d, \"iters\": %d, \"epoch\": %d, "
                   "\"self_collision\": %d, \"ms_rollout_per_iter\": %.4f, \"ms_update_per_iter\": %.4f, \"ms_per_epoch\": %.4f, "
                   "\"allreduce_bytes_per_epoch\": %zu, \"collectives_per_iter\": %d, \"transitions_rank0\": %lld, \"env_steps_per_s_rank0\": %.1f, "
                   "\"rank0_selected_last\": %.0f, \"actor_loss\": %.17g, \"critic_loss\": %.17g, \"env_errors\": [%d, %d]}\n",
                   world, n, T, iters, epoch, selfcol, ms_roll / iters, ms_upd / iters, ms_upd / iters / epoch, gn * 4, 1 + epoch, st[0],
                   (double) st[0] / sec, h_stats[0], la, lc, errs[0], errs[1]);
        evm_ppo_destroy(tr); evm_policy_destroy(pol); evm_env_destroy(env);
        nccl_check(ncclCommDestroy(comm), "ncclCommDestroy");
        if (rank == 0 && world > 1) unlink(id_path.c_str());
    } catch (const std::exception &e) {
        fprintf(stderr, "train_main[%d]: %s\n", rank, e.what());
        return 1;
    }
    return 0;
}
