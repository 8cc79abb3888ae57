// C ABI (include/evomotion.h) for the vectorised robot_walk environment: allocation, launch sequencing,
// state import/export.  All arithmetic of the hot path lives in env_kernels.hip; there is no CPU fallback.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/evomotion.h"
#include "env_dev.h"
#include "skeleton_host.h"

namespace {
thread_local std::string g_err;
int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
}  // namespace
namespace evm {
void set_last_error(const std::string &m) { g_err = m; }
}
namespace {
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(EVM_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
}  // namespace

// element (slot k, env e) of an array with `cnt` slots per env; device layout is [tile][slot][64 lanes]
static inline size_t tix(size_t cnt, size_t k, size_t e) { return ((e >> 6) * cnt + k) * 64 + (e & 63); }

struct EvmEnv {
    EvmSkelC skel;
    evm::EnvDev d;
    EvmEnvParams prm;
    int device;
    void *arena;
    size_t arena_bytes;
    hipEvent_t ev0, ev1;
    int timed_launches;
    bool timing;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pairs;  // one pair per timed launch
    size_t ev_used;
    int split;  // -1: by batch size (default), 1: split pipeline, 0: monolithic step kernel (EVM_MONOLITHIC=0/1 forces: A/B runs)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_sweeps;  // timed launches: around the sweeps kernel of the split pipeline
    void *gsched;  // device copy of the lane-group sweep schedule (EvmGSchedC), or null
    void *spec_mem;   // the speculation blocks' slots (narrow_dev.h), or null
};

#ifndef EVM_MAX_DEVICES
#define EVM_MAX_DEVICES 64
#endif
static const EvmEnv *g_skel_owner[EVM_MAX_DEVICES] = {};

// The skeleton constants live in one __constant__ block PER DEVICE (a __constant__ symbol has one instance on every device of
// the process), so ownership is tracked per device: envs on different GPUs of one process never evict each other.  An env that
// steps after ANOTHER env on the same device (alternating robot_walk / robot_jump, or two batches) re-uploads its block; the
// previous owner's kernels may still be in flight on another stream, so that device is drained first (rare: one process per
// GPU owns one env in every benchmark and in the rollout loops; one host thread per GPU is the ABI's threading rule).
static int ensure_skeleton(EvmEnv *env, hipStream_t s) {
    const EvmEnv *&owner = g_skel_owner[env->device];
    if (owner != env) {
        if (owner != nullptr) HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(evm::upload_skeleton(&env->skel, s));
        owner = env;
    }
    return EVM_OK;
}

extern "C" {

const char *evm_last_error(void) { return g_err.c_str(); }

void evm_env_default_params(EvmEnvParams *p) {
    p->initial_remaining_seconds = 1.f;
    p->max_episode_seconds = 30.f;
    p->target_velocity = 0.5f;
    p->minimal_velocity = 0.1f;
    p->reset_frames = 30;
    p->env_kind = 0;
    p->self_collision = 1;
}

int evm_env_default_params_for(const char *env_name, EvmEnvParams *p) {
    if (!env_name || !p) return fail(EVM_E_INVALID, "null argument");
    evm_env_default_params(p);
    const std::string n = env_name;
    if (n == "robot_walk") return EVM_OK;
    if (n == "robot_jump") {  // RobotJumpFactory defaults, env_factory.cpp:91-100
        p->env_kind = 1;
        p->reset_frames = (int) ((1.f / 6.f) / (1.f / 60.f));  // static_cast<int>(reset_seconds / DELTA_T_MODEL) = 10
        return EVM_OK;
    }
    return fail(EVM_E_INVALID, n);  // std::invalid_argument(env_name), env_factory.cpp:118
}

int evm_env_create(const char *skeleton_path, int n_envs, int device, uint64_t seed, const EvmEnvParams *params,
                   EvmEnv **out) {
    if (!out) return fail(EVM_E_INVALID, "out is null");
    *out = nullptr;
    if (n_envs < 1) return fail(EVM_E_INVALID, "n_envs must be >= 1");
    EvmEnvParams prm;
    evm_env_default_params(&prm);
    if (params) prm = *params;
    EvmEnv *env = new EvmEnv();
    memset(&env->d, 0, sizeof(env->d));
    env->prm = prm;
    env->device = device;
    env->arena = nullptr;
    env->timing = false;
    env->timed_launches = 0;
    env->ev_used = 0;
    { const char *m = getenv("EVM_MONOLITHIC"); env->split = !m ? -1 : (m[0] == '1' ? 0 : 1); }
    std::string err;
    int rc = evm::load_skeleton_constants(skeleton_path, prm, env->skel, err);
    if (rc != EVM_OK) { delete env; return fail(rc, err); }
    const EvmSkelC &S = env->skel;
    // a 64-env tile is 3 KB per body: beyond the CU's 160 KB of LDS the sweeps work on the tile's global staging copy (slower);
    // the member-vs-member mode needs the lane-group kernel's LDS image and stays refused for such a skeleton (below)
    const bool gtile_only = evm::step_lds_bytes(S.nb, S.nscan) > 160 * 1024;
    if (device < 0 || device >= EVM_MAX_DEVICES) { delete env; return fail(EVM_E_INVALID, "device index out of range"); }
    // the narrowphase's big-hull work list packs an entry as (pair << 20) | env (pairs_dev.h): 2^20 environments per instance
    if (S.self_collision && ((size_t) n_envs + 63) / 64 * 64 > ((size_t) 1 << 20)) {
        delete env;
        return fail(EVM_E_UNSUPPORTED, "member-vs-member contacts: at most 1 048 576 environments per instance (work-list entry = pair << 20 | env)");
    }
    hipError_t he = hipSetDevice(device);
    if (he != hipSuccess) { delete env; return fail(EVM_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(he)); }

    const size_t n = ((size_t) n_envs + 63) / 64 * 64;
    env->d.n = (int) n;
    env->d.tile_floats = (int) (evm::step_lds_bytes(S.nb, S.nscan) / 4);
    env->d.n_real = n_envs;
    env->d.npair_host = S.npair;
    env->d.gtile_only = gtile_only ? 1 : 0;
    struct Seg { void **p; size_t count; };
    std::vector<Seg> segs = {
        {(void **) &env->d.pos, 3u * S.nb}, {(void **) &env->d.quat, 4u * S.nb}, {(void **) &env->d.lin, 3u * S.nb},
        {(void **) &env->d.ang, 3u * S.nb}, {(void **) &env->d.hist, 6u * S.nm}, {(void **) &env->d.mfn, 1u * S.nm},
        {(void **) &env->d.mfp, 36u * S.nm}, {(void **) &env->d.target, (size_t) (S.nmus > 0 ? S.nmus : 1)},
        {(void **) &env->d.flags, 1}, {(void **) &env->d.curr_step, 1}, {(void **) &env->d.remaining, 1},
        {(void **) &env->d.settle_left, 1}, {(void **) &env->d.E, 9}, {(void **) &env->d.iinv_stale, 6u * S.nb},
        {(void **) &env->d.mt, 624}, {(void **) &env->d.mt_idx, 1}, {(void **) &env->d.scratch, (size_t) S.sc_total},
        {(void **) &env->d.diag, 2}, {(void **) &env->d.stat, 2}, {(void **) &env->d.stamps, 1}, {(void **) &env->d.resid, 1}, {(void **) &env->d.errs, 1},
        {(void **) &env->d.gtile, evm::step_lds_bytes(S.nb, S.nscan) / 4 / 64}};  // stamps: 16 u64 per tile = 128 B <= 256 B
    if (S.self_collision) {
        // member-vs-member contacts: persistent pair manifolds, the activity words (+ one flag word) and the two-body contact
        // records of a step, one per manifold id (floor manifolds first)
        segs.push_back({(void **) &env->d.pmn, (size_t) (S.npair > 0 ? S.npair : 1)});
        segs.push_back({(void **) &env->d.pmp, (size_t) EVM_PM_STRIDE * (S.npair > 0 ? S.npair : 1)});
        segs.push_back({(void **) &env->d.pact, (size_t) ((S.npair + 31) / 32 + 1)});
        segs.push_back({(void **) &env->d.crec, (size_t) EVM_CR_STRIDE * (S.nm + S.npair)});
        segs.push_back({(void **) &env->d.plist, (size_t) (S.npair > 0 ? S.npair : 1)});
        segs.push_back({(void **) &env->d.blist, (size_t) (S.npair > 0 ? S.npair : 1)});
        segs.push_back({(void **) &env->d.pcount, (size_t) (2 * EVM_PC_STRIDE + 63) / 64});  // >= 2 x EVM_PC_STRIDE ints whatever the batch (n >= 64)
    }
    size_t total = 0;
    for (auto &s : segs) total += s.count * n * 4;
    he = hipMalloc(&env->arena, total);
    if (he != hipSuccess) { delete env; return fail(EVM_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(he)); }
    env->arena_bytes = total;
    hipMemset(env->arena, 0, total);
    char *base = (char *) env->arena;
    for (auto &s : segs) { *s.p = base; base += s.count * n * 4; }
    hipEventCreate(&env->ev0);
    hipEventCreate(&env->ev1);
    env->spec_mem = nullptr;
    { const char *gs = getenv("EVM_GAP_SOON"); env->d.gap_soon = gs ? (float) atof(gs) : 0.f; }
    { const char *ds = getenv("EVM_DEEP_SOON"); env->d.deep_soon = ds ? (float) atof(ds) : EVM_DEEP_SOON_DEFAULT; }   // (scheduling only: A/B runs)
    {
        // member-vs-member contacts: the slots of the urgent list's speculation blocks (EVM_SPECULATE=0 turns them off: A/B runs; the
        // physics is the same bit for bit either way)
        const char *sp = getenv("EVM_SPECULATE");
        if (S.self_collision && !(sp && sp[0] == '0')) {
            const size_t words = (size_t) EVM_SPEC_SLOTS * EVM_SPEC_WORDS;
            he = hipMalloc(&env->spec_mem, words * sizeof(int));
            if (he == hipSuccess) he = hipMemset(env->spec_mem, 0, words * sizeof(int));
            if (he != hipSuccess) { evm_env_destroy(env); return fail(EVM_E_HIP, std::string("speculation slots: ") + hipGetErrorString(he)); }
            env->d.spec = (int *) env->spec_mem;
        }
    }
    // Sweeps kernel of the split pipeline: the lane-group kernel (16-env workgroups, joint records resident in LDS) when the
    // skeleton's records fit its LDS image, else the 64-env tile kernel.  EVM_SWEEPS=tile forces the latter, EVM_G_WAVES=1..4
    // sets the waves per 16-env workgroup (A/B runs).
    env->gsched = nullptr;
    {
        const char *sw = getenv("EVM_SWEEPS");
        const char *gw = getenv("EVM_G_WAVES");
        int nw = gw ? atoi(gw) : EVM_G_MAX_WAVES;  // one wave per SIMD measured best (tools/ab_sweeps.sh)
        if (nw < 1 || nw > EVM_G_MAX_WAVES) nw = EVM_G_MAX_WAVES;
        if (!(sw && sw[0] == 't')) {
            EvmGSchedC *G = new EvmGSchedC();
            std::string gerr;
            if (evm::build_group_schedule(S, nw, *G, gerr) == EVM_OK) {
                he = hipMalloc(&env->gsched, sizeof(EvmGSchedC));
                if (he == hipSuccess) he = hipMemcpy(env->gsched, G, sizeof(EvmGSchedC), hipMemcpyHostToDevice);
                if (he != hipSuccess) { delete G; evm_env_destroy(env); return fail(EVM_E_HIP, std::string("group schedule: ") + hipGetErrorString(he)); }
                env->d.gs = (const EvmGSchedC *) env->gsched;
                env->d.g_waves = G->nwaves;
                env->d.g_lds = G->lds_bytes;
            }
            delete G;
        }
    }
    if (S.self_collision && !env->gsched) {
        evm_env_destroy(env);
        return fail(EVM_E_UNSUPPORTED, "self_collision = 1 needs the lane-group sweeps kernel, whose LDS image this skeleton exceeds (or EVM_SWEEPS=tile rules out); "
                                       "self_collision = 0 runs it on the global-memory tile");
    }
    // a new env always uploads (its address may be a destroyed owner's); the device's previous owner may have kernels in flight
    if (g_skel_owner[device] != nullptr) (void) hipDeviceSynchronize();
    g_skel_owner[device] = nullptr;
    rc = ensure_skeleton(env, 0);
    if (rc != EVM_OK) { evm_env_destroy(env); return rc; }
    he = evm::launch_init(env->d, seed, 0);
    if (he == hipSuccess) he = hipDeviceSynchronize();
    if (he != hipSuccess) { evm_env_destroy(env); return fail(EVM_E_HIP, std::string("init: ") + hipGetErrorString(he)); }
    *out = env;
    return EVM_OK;
}

void evm_env_destroy(EvmEnv *env) {
    if (!env) return;
    if (env->device >= 0 && env->device < EVM_MAX_DEVICES && g_skel_owner[env->device] == env) g_skel_owner[env->device] = nullptr;
    if (env->arena) hipFree(env->arena);
    if (env->gsched) hipFree(env->gsched);
    if (env->spec_mem) hipFree(env->spec_mem);
    (void) hipEventDestroy(env->ev0);
    (void) hipEventDestroy(env->ev1);
    for (auto &pr : env->ev_pairs) { (void) hipEventDestroy(pr.first); (void) hipEventDestroy(pr.second); }
    for (auto &pr : env->ev_sweeps) { (void) hipEventDestroy(pr.first); (void) hipEventDestroy(pr.second); }
    delete env;
}

int evm_env_spaces(const EvmEnv *env, int *state_dim, int *action_dim) {
    if (!env) return fail(EVM_E_INVALID, "env is null");
    if (state_dim) *state_dim = env->skel.obs_dim;
    if (action_dim) *action_dim = env->skel.act_dim;
    return EVM_OK;
}
int evm_env_counts(const EvmEnv *env, int *n_envs, int *n_bodies, int *n_members, int *n_muscles) {
    if (!env) return fail(EVM_E_INVALID, "env is null");
    if (n_envs) *n_envs = env->d.n_real;
    if (n_bodies) *n_bodies = env->skel.nb;
    if (n_members) *n_members = env->skel.nm;
    if (n_muscles) *n_muscles = env->skel.nmus;
    return EVM_OK;
}

static int step_launch(EvmEnv *env, int mode, const float *a, float *obs, float *rew, uint8_t *done, uint8_t *valid,
                       const uint8_t *mask, hipStream_t s) {
    int rc = ensure_skeleton(env, s);
    if (rc != EVM_OK) return rc;
    // timed region: every 4th step is bracketed by an event pair (an event between two launches costs a bubble of a few
    // microseconds on the stream, so the steps are sampled rather than all instrumented)
    const bool sample = env->timing && (env->timed_launches % 4 == 0);
    if (sample) {
        if (env->ev_used == env->ev_pairs.size()) {
            hipEvent_t a0, a1;
            HIP_TRY(hipEventCreate(&a0));
            HIP_TRY(hipEventCreate(&a1));
            env->ev_pairs.push_back({a0, a1});
            hipEvent_t b0, b1;
            HIP_TRY(hipEventCreate(&b0));
            HIP_TRY(hipEventCreate(&b1));
            env->ev_sweeps.push_back({b0, b1});
        }
        HIP_TRY(hipEventRecord(env->ev_pairs[env->ev_used].first, s));
    }
    env->d.pc_cur ^= 1;  // this step's copy of the narrowphase list counters (zeroed by the previous step's first kernel)
    // the speculation slots' epoch (a slot is valid when its word 0 equals the launch's epoch; memory starts as 0)
    if (env->d.spec && env->d.spec_epoch >= 0x1ffffff0) {   // ((epoch << 2) | code must stay positive)
        HIP_TRY(hipMemsetAsync(env->d.spec, 0, (size_t) EVM_SPEC_SLOTS * EVM_SPEC_WORDS * sizeof(int), s));
        env->d.spec_epoch = 0;
    }
    env->d.spec_epoch++;
    HIP_TRY(evm::launch_step(env->d, evm::step_lds_bytes(env->skel.nb, env->skel.nscan), env->split, mode, a, obs, rew, done, valid, mask, s,
                             // every 8th step also brackets its sweeps kernel
                             (sample && env->ev_used % 2 == 0) ? env->ev_sweeps[env->ev_used].first : nullptr,
                             (sample && env->ev_used % 2 == 0) ? env->ev_sweeps[env->ev_used].second : nullptr));
    if (sample) {
        HIP_TRY(hipEventRecord(env->ev_pairs[env->ev_used].second, s));
        env->ev_used++;
    }
    if (env->timing) env->timed_launches++;
    return EVM_OK;
}

int evm_env_reset(EvmEnv *env, const uint8_t *d_mask, float *d_obs, float *d_reward, uint8_t *d_done, void *stream) {
    if (!env || !d_obs || !d_reward || !d_done) return fail(EVM_E_INVALID, "null argument");
    hipStream_t s = (hipStream_t) stream;
    int rc = ensure_skeleton(env, s);
    if (rc != EVM_OK) return rc;
    HIP_TRY(evm::launch_repose(env->d, d_mask, s));
    const int settle = env->skel.settle_steps;
    for (int i = 0; i < settle; i++) {
        const bool last = i == settle - 1;
        rc = step_launch(env, last ? 2 : 0, nullptr, d_obs, d_reward, d_done, nullptr, d_mask, s);
        if (rc != EVM_OK) return rc;
    }
    if (settle == 0) return fail(EVM_E_UNSUPPORTED, "reset_frames == 0 is not supported");
    return EVM_OK;
}

int evm_env_step(EvmEnv *env, const float *d_action, float *d_obs, float *d_reward, uint8_t *d_done, void *stream) {
    if (!env || (!d_action && env->skel.act_dim > 0) || !d_obs || !d_reward || !d_done) return fail(EVM_E_INVALID, "null argument");
    return step_launch(env, 3, d_action, d_obs, d_reward, d_done, nullptr, nullptr, (hipStream_t) stream);
}

int evm_env_step_autoreset(EvmEnv *env, const float *d_action, float *d_obs, float *d_reward, uint8_t *d_done,
                           uint8_t *d_valid, void *stream) {
    if (!env || (!d_action && env->skel.act_dim > 0) || !d_obs || !d_reward || !d_done || !d_valid) return fail(EVM_E_INVALID, "null argument");
    return step_launch(env, 7, d_action, d_obs, d_reward, d_done, d_valid, nullptr, (hipStream_t) stream);
}

int evm_env_get_body_poses(const EvmEnv *env, float *d_pose, void *stream) {
    if (!env || !d_pose) return fail(EVM_E_INVALID, "null argument");
    int rc = ensure_skeleton(const_cast<EvmEnv *>(env), (hipStream_t) stream);
    if (rc != EVM_OK) return rc;
    HIP_TRY(evm::launch_poses(env->d, d_pose, (hipStream_t) stream));
    return EVM_OK;
}

int evm_env_pairs(const EvmEnv *env, int *n_pairs, int *h_pairs) {
    if (!env) return fail(EVM_E_INVALID, "env is null");
    if (n_pairs) *n_pairs = env->skel.npair;
    for (int p = 0; p < env->skel.npair && h_pairs; p++) { h_pairs[2 * p] = env->skel.pair[p].a; h_pairs[2 * p + 1] = env->skel.pair[p].b; }
    return EVM_OK;
}

// diagnostic: the narrowphase work-list sizes of the last step, one per pair (table order)
int evm_env_debug_pair_counts(EvmEnv *env, int *h_out) {
    if (!env || !h_out) return fail(EVM_E_INVALID, "null argument");
    HIP_TRY(hipDeviceSynchronize());
    if (env->skel.npair > 0) {   // the last step's copy; the urgent list's entries (pairs_dev.h) are counted with the flat list
        int urgent = 0;
        const int *cur = env->d.pcount + env->d.pc_cur * EVM_PC_STRIDE;
        HIP_TRY(hipMemcpy(h_out, cur, (env->skel.npair + 1) * sizeof(int), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&urgent, cur + env->skel.npair + 1, sizeof(int), hipMemcpyDeviceToHost));
        h_out[env->skel.npair] += urgent;
    }
    return EVM_OK;
}

int evm_env_debug_reset_begin(EvmEnv *env, const uint8_t *d_mask) {
    if (!env) return fail(EVM_E_INVALID, "env is null");
    int rc = ensure_skeleton(env, 0);
    if (rc != EVM_OK) return rc;
    HIP_TRY(evm::launch_repose(env->d, d_mask, 0));
    HIP_TRY(hipDeviceSynchronize());
    return EVM_OK;
}
int evm_env_debug_physics_steps(EvmEnv *env, int n_steps, const uint8_t *d_mask) {
    if (!env) return fail(EVM_E_INVALID, "env is null");
    for (int i = 0; i < n_steps; i++) {
        int rc = step_launch(env, 0, nullptr, nullptr, nullptr, nullptr, nullptr, d_mask, 0);
        if (rc != EVM_OK) return rc;
    }
    HIP_TRY(hipDeviceSynchronize());
    return EVM_OK;
}

static void body_constants(const EvmSkelC &S, float *out);

// Host-only: run the skeleton loader (no HIP call) and report what it derived.
int evm_skeleton_probe(const char *skeleton_path, int *counts /* nb nm nh nf nmus obs act root */, float *out) {
    EvmEnvParams prm;
    evm_env_default_params(&prm);
    prm.self_collision = 0;  // (the tables these report do not depend on the collision mode)
    EvmSkelC *S = new EvmSkelC();
    std::string err;
    int rc = evm::load_skeleton_constants(skeleton_path, prm, *S, err);
    if (rc != EVM_OK) { delete S; return fail(rc, err); }
    if (counts) {
        counts[0] = S->nb; counts[1] = S->nm; counts[2] = S->nh; counts[3] = S->nf; counts[4] = S->nmus;
        counts[5] = S->obs_dim; counts[6] = S->act_dim; counts[7] = S->root; counts[8] = S->max_steps;
        counts[9] = S->init_remaining;
    }
    if (out) body_constants(*S, out);
    delete S;
    return EVM_OK;
}

// Host-only: FNV-1a digest of everything the loader derived (the whole constant block the kernels read) — two input
// formats describing the same skeleton must give the same digest.
int evm_skeleton_digest(const char *skeleton_path, unsigned long long *h_out) {
    if (!h_out) return fail(EVM_E_INVALID, "h_out is null");
    EvmEnvParams prm;
    evm_env_default_params(&prm);
    prm.self_collision = 0;  // (the tables these report do not depend on the collision mode)
    EvmSkelC *S = new EvmSkelC();
    std::string err;
    int rc = evm::load_skeleton_constants(skeleton_path, prm, *S, err);
    if (rc != EVM_OK) { delete S; return fail(rc, err); }
    unsigned long long h = 1469598103934665603ull;
    const unsigned char *b = reinterpret_cast<const unsigned char *>(S);
    for (size_t i = 0; i < sizeof(EvmSkelC); i++) { h ^= b[i]; h *= 1099511628211ull; }
    delete S;
    *h_out = h;
    return EVM_OK;
}

// Host-only: the sweep schedule the loader derived.  visits [nvisit,4] = (type, a, b, level);
// sched [EVM_NW, cap] entries (visit | 0x8000 on the last entry of a level, 0x7fff = none); returns sizes in dims[4] =
// (nvisit, nlevels, n_waves, cap).
int evm_skeleton_schedule(const char *skeleton_path, int *dims, int *visits, int *sched, int cap) {
    EvmEnvParams prm;
    evm_env_default_params(&prm);
    prm.self_collision = 0;  // (the tables these report do not depend on the collision mode)
    EvmSkelC *S = new EvmSkelC();
    std::string err;
    int rc = evm::load_skeleton_constants(skeleton_path, prm, *S, err);
    if (rc != EVM_OK) { delete S; return fail(rc, err); }
    if (dims) { dims[0] = S->nvisit; dims[1] = S->nlevels; dims[2] = EVM_NW; dims[3] = cap; }
    for (int w = 0; w < EVM_NW && sched; w++) {
        for (int i = 0; i < cap; i++) sched[w * cap + i] = i < S->nsched[w] ? S->sched[w][i] : -1;
    }
    for (int i = 0; i < S->nvisit && visits; i++) {
        visits[4 * i] = S->visit[i].type; visits[4 * i + 1] = S->visit[i].a; visits[4 * i + 2] = S->visit[i].b;
        visits[4 * i + 3] = S->visit[i].need;
    }
    delete S;
    return EVM_OK;
}

int evm_skeleton_group_schedule(const char *skeleton_path, int n_waves, int *dims, int *entries, int cap) {
    return evm_skeleton_group_schedule_ex(skeleton_path, n_waves, 0, dims, entries, cap);
}
int evm_skeleton_group_schedule_ex(const char *skeleton_path, int n_waves, int self_collision, int *dims, int *entries, int cap) {
    EvmEnvParams prm;
    evm_env_default_params(&prm);
    prm.self_collision = self_collision;
    EvmSkelC *S = new EvmSkelC();
    EvmGSchedC *G = new EvmGSchedC();
    std::string err;
    int rc = evm::load_skeleton_constants(skeleton_path, prm, *S, err);
    if (rc == EVM_OK) rc = evm::build_group_schedule(*S, n_waves, *G, err);
    if (rc != EVM_OK) { delete S; delete G; return fail(rc, err); }
    if (dims) { dims[0] = G->total; dims[1] = G->nwaves; dims[2] = G->lds_bytes; dims[3] = (int) G->est_cycles; }
    for (int w = 0; w < G->nwaves && entries; w++)
        for (int k = G->first[w]; k < G->first[w] + G->count[w] && k < cap; k++) {
            int *o = entries + (size_t) k * (2 + 5 * EVM_G_SLOTS);
            o[0] = w | ((G->entry[k].order & 0xffff) << 8); o[1] = G->entry[k].type;
            for (int q = 0; q < EVM_G_SLOTS; q++) {
                const EvmGSlotC &sl = G->slot[k][q];
                o[2 + 5 * q] = sl.rec; o[3 + 5 * q] = sl.a; o[4 + 5 * q] = sl.b; o[5 + 5 * q] = sl.need; o[6 + 5 * q] = sl.ps;
            }
        }
    delete S;
    delete G;
    return EVM_OK;
}

int evm_env_get_body_constants(const EvmEnv *env, float *out) {
    if (!env || !out) return fail(EVM_E_INVALID, "null argument");
    body_constants(env->skel, out);
    return EVM_OK;
}
static void body_constants(const EvmSkelC &S, float *out) {
    int k = 0;
    for (int b = 0; b < S.nb; b++) {
        const EvmBodyC &B = S.body[b];
        out[k++] = B.mass; out[k++] = B.inv_mass;
        out[k++] = B.inv_inertia[0]; out[k++] = B.inv_inertia[1]; out[k++] = B.inv_inertia[2];
        out[k++] = B.friction;
        out[k++] = b < S.nm ? S.member[b].break_thr : 0.f;
        for (int i = 0; i < 9; i++) out[k++] = B.m0[i];
        for (int i = 0; i < 3; i++) out[k++] = B.t0[i];
    }
}

int evm_env_get_diagnostics(const EvmEnv *env, float *d_out, void *stream) {
    if (!env || !d_out) return fail(EVM_E_INVALID, "null argument");
    // diag is [2][n] field-major on the device; hand it out as [n_envs, 2]
    std::vector<float> h(2 * (size_t) env->d.n), o(2 * (size_t) env->d.n_real);
    HIP_TRY(hipStreamSynchronize((hipStream_t) stream));
    HIP_TRY(hipMemcpy(h.data(), env->d.diag, h.size() * 4, hipMemcpyDeviceToHost));
    for (int e = 0; e < env->d.n_real; e++) { o[2 * e] = h[tix(2, 0, e)]; o[2 * e + 1] = h[tix(2, 1, e)]; }
    HIP_TRY(hipMemcpy(d_out, o.data(), o.size() * 4, hipMemcpyHostToDevice));
    return EVM_OK;
}

// ---- canonical state blob -------------------------------------------------------------------
int evm_env_state_size(const EvmEnv *env) {
    if (!env) return fail(EVM_E_INVALID, "env is null");
    const EvmSkelC &S = env->skel;
    return 13 * S.nb + 1 + 9 + 6 * S.nb + 3 * S.nm + 6 * S.nm + 37 * S.nm + S.nmus + 1 + 2 + 49 * S.npair;
}

namespace {
struct HostMirror {
    std::vector<float> pos, quat, lin, ang, hist, mfp, target, E, iinv, scratch_ms, pmp;
    std::vector<int> mfn, flags, curr, rem, pmn;
};
}

int evm_env_get_state(EvmEnv *env, float *h_state) {
    if (!env || !h_state) return fail(EVM_E_INVALID, "null argument");
    const EvmSkelC &S = env->skel;
    const size_t n = env->d.n;
    HIP_TRY(hipDeviceSynchronize());
    auto dl = [&](const void *src, size_t count, std::vector<float> &dst) {
        dst.resize(count * n);
        return hipMemcpy(dst.data(), src, count * n * 4, hipMemcpyDeviceToHost);
    };
    auto dli = [&](const void *src, size_t count, std::vector<int> &dst) {
        dst.resize(count * n);
        return hipMemcpy(dst.data(), src, count * n * 4, hipMemcpyDeviceToHost);
    };
    HostMirror m;
    HIP_TRY(dl(env->d.pos, 3 * S.nb, m.pos)); HIP_TRY(dl(env->d.quat, 4 * S.nb, m.quat));
    HIP_TRY(dl(env->d.lin, 3 * S.nb, m.lin)); HIP_TRY(dl(env->d.ang, 3 * S.nb, m.ang));
    HIP_TRY(dl(env->d.hist, 6 * S.nm, m.hist)); HIP_TRY(dl(env->d.mfp, 36 * S.nm, m.mfp));
    HIP_TRY(dl(env->d.target, S.nmus > 0 ? S.nmus : 1, m.target)); HIP_TRY(dl(env->d.E, 9, m.E));
    HIP_TRY(dl(env->d.iinv_stale, 6 * S.nb, m.iinv));
    m.scratch_ms.resize((size_t) 3 * S.nm * n);
    HIP_TRY(hipMemcpy2D(m.scratch_ms.data(), (size_t) 3 * S.nm * 256, env->d.scratch + (size_t) S.sc_ms * 64,
                        (size_t) S.sc_total * 256, (size_t) 3 * S.nm * 256, n / 64, hipMemcpyDeviceToHost));
    HIP_TRY(dli(env->d.mfn, S.nm, m.mfn)); HIP_TRY(dli(env->d.flags, 1, m.flags));
    HIP_TRY(dli(env->d.curr_step, 1, m.curr)); HIP_TRY(dli(env->d.remaining, 1, m.rem));
    if (S.npair > 0) { HIP_TRY(dl(env->d.pmp, (size_t) EVM_PM_STRIDE * S.npair, m.pmp)); HIP_TRY(dli(env->d.pmn, S.npair, m.pmn)); }
    const int ss = evm_env_state_size(env);
    for (int e = 0; e < env->d.n_real; e++) {
        float *o = h_state + (size_t) e * ss;
        int k = 0;
        const bool pending = (m.flags[e] & EVM_FLAG_PENDING) != 0;
        for (int b = 0; b < S.nb; b++) {
            for (int a = 0; a < 3; a++) o[k++] = m.pos[tix(3 * S.nb, 3 * b + a, e)];
            if (pending) { o[k++] = 0.f; o[k++] = 0.f; o[k++] = 0.f; o[k++] = 1.f; }
            else for (int a = 0; a < 4; a++) o[k++] = m.quat[tix(4 * S.nb, 4 * b + a, e)];
            for (int a = 0; a < 3; a++) o[k++] = m.lin[tix(3 * S.nb, 3 * b + a, e)];
            for (int a = 0; a < 3; a++) o[k++] = m.ang[tix(3 * S.nb, 3 * b + a, e)];
        }
        o[k++] = pending ? 1.f : 0.f;
        for (int a = 0; a < 9; a++) o[k++] = m.E[tix(9, a, e)];
        for (int a = 0; a < 6 * S.nb; a++) o[k++] = m.iinv[tix(6 * S.nb, a, e)];
        for (int a = 0; a < 3 * S.nm; a++) o[k++] = m.scratch_ms[tix(3 * S.nm, a, e)];
        for (int a = 0; a < 6 * S.nm; a++) o[k++] = m.hist[tix(6 * S.nm, a, e)];
        for (int mm = 0; mm < S.nm; mm++) {
            const int cnt = m.mfn[tix(S.nm, mm, e)];
            o[k++] = (float) cnt;
            for (int j = 0; j < 4; j++)
                for (int f = 0; f < 9; f++) o[k++] = j < cnt ? m.mfp[tix(36 * S.nm, (mm * 4 + j) * 9 + f, e)] : 0.f;
        }
        for (int p = 0; p < S.npair; p++) {
            const int cnt = m.pmn[tix(S.npair, p, e)] & 0xff;   // (bit 8: the narrowphase's scheduling hint, not state)
            o[k++] = (float) cnt;
            for (int j = 0; j < 4; j++)
                for (int f = 0; f < 12; f++) o[k++] = j < cnt ? m.pmp[tix((size_t) EVM_PM_STRIDE * S.npair, (p * 4 + j) * 12 + f, e)] : 0.f;
        }
        for (int a = 0; a < S.nmus; a++) o[k++] = m.target[tix((S.nmus > 0 ? S.nmus : 1), a, e)];
        o[k++] = (m.flags[e] & EVM_FLAG_POWERED) ? 1.f : 0.f;
        o[k++] = (float) m.curr[e];
        o[k++] = (float) m.rem[e];
    }
    return EVM_OK;
}

int evm_env_set_state(EvmEnv *env, const float *h_state) {
    if (!env || !h_state) return fail(EVM_E_INVALID, "null argument");
    const EvmSkelC &S = env->skel;
    const size_t n = env->d.n;
    HIP_TRY(hipDeviceSynchronize());
    HostMirror m;
    m.pos.assign(3 * S.nb * n, 0.f); m.quat.assign(4 * S.nb * n, 0.f); m.lin.assign(3 * S.nb * n, 0.f);
    m.ang.assign(3 * S.nb * n, 0.f); m.hist.assign(6 * S.nm * n, 0.f); m.mfp.assign(36 * S.nm * n, 0.f);
    m.target.assign((S.nmus > 0 ? S.nmus : 1) * n, 0.f); m.E.assign(9 * n, 0.f); m.iinv.assign(6 * S.nb * n, 0.f);
    m.scratch_ms.assign(3 * S.nm * n, 0.f);
    m.mfn.assign(S.nm * n, 0); m.flags.assign(n, 0); m.curr.assign(n, 0); m.rem.assign(n, 0);
    m.pmp.assign((size_t) EVM_PM_STRIDE * S.npair * n, 0.f); m.pmn.assign((size_t) S.npair * n, 0);
    // keep the rollout flag and whatever the padded lanes hold
    HIP_TRY(hipMemcpy(m.flags.data(), env->d.flags, n * 4, hipMemcpyDeviceToHost));
    const int ss = evm_env_state_size(env);
    for (int e = 0; e < env->d.n_real; e++) {
        const float *in = h_state + (size_t) e * ss;
        int k = 0;
        for (int b = 0; b < S.nb; b++) {
            for (int a = 0; a < 3; a++) m.pos[tix(3 * S.nb, 3 * b + a, e)] = in[k++];
            for (int a = 0; a < 4; a++) m.quat[tix(4 * S.nb, 4 * b + a, e)] = in[k++];
            for (int a = 0; a < 3; a++) m.lin[tix(3 * S.nb, 3 * b + a, e)] = in[k++];
            for (int a = 0; a < 3; a++) m.ang[tix(3 * S.nb, 3 * b + a, e)] = in[k++];
        }
        const bool pending = in[k++] != 0.f;
        for (int a = 0; a < 9; a++) m.E[tix(9, a, e)] = in[k++];
        for (int a = 0; a < 6 * S.nb; a++) m.iinv[tix(6 * S.nb, a, e)] = in[k++];
        for (int a = 0; a < 3 * S.nm; a++) m.scratch_ms[tix(3 * S.nm, a, e)] = in[k++];
        for (int a = 0; a < 6 * S.nm; a++) m.hist[tix(6 * S.nm, a, e)] = in[k++];
        for (int mm = 0; mm < S.nm; mm++) {
            m.mfn[tix(S.nm, mm, e)] = (int) in[k++];
            for (int j = 0; j < 4; j++)
                for (int f = 0; f < 9; f++) m.mfp[tix(36 * S.nm, (mm * 4 + j) * 9 + f, e)] = in[k++];
        }
        for (int p = 0; p < S.npair; p++) {
            m.pmn[tix(S.npair, p, e)] = (int) in[k++];
            for (int j = 0; j < 4; j++)
                for (int f = 0; f < 12; f++) m.pmp[tix((size_t) EVM_PM_STRIDE * S.npair, (p * 4 + j) * 12 + f, e)] = in[k++];
        }
        for (int a = 0; a < S.nmus; a++) m.target[tix((S.nmus > 0 ? S.nmus : 1), a, e)] = in[k++];
        const bool powered = in[k++] != 0.f;
        m.curr[e] = (int) in[k++];
        m.rem[e] = (int) in[k++];
        int f = m.flags[e] & ~(EVM_FLAG_PENDING | EVM_FLAG_POWERED);
        if (pending) f |= EVM_FLAG_PENDING;
        if (powered) f |= EVM_FLAG_POWERED;
        m.flags[e] = f;
    }
    auto ul = [&](void *dst, const std::vector<float> &src) { return hipMemcpy(dst, src.data(), src.size() * 4, hipMemcpyHostToDevice); };
    auto uli = [&](void *dst, const std::vector<int> &src) { return hipMemcpy(dst, src.data(), src.size() * 4, hipMemcpyHostToDevice); };
    HIP_TRY(ul(env->d.pos, m.pos)); HIP_TRY(ul(env->d.quat, m.quat)); HIP_TRY(ul(env->d.lin, m.lin));
    HIP_TRY(ul(env->d.ang, m.ang)); HIP_TRY(ul(env->d.hist, m.hist)); HIP_TRY(ul(env->d.mfp, m.mfp));
    HIP_TRY(ul(env->d.target, m.target)); HIP_TRY(ul(env->d.E, m.E)); HIP_TRY(ul(env->d.iinv_stale, m.iinv));
    HIP_TRY(hipMemcpy2D(env->d.scratch + (size_t) S.sc_ms * 64, (size_t) S.sc_total * 256, m.scratch_ms.data(),
                        (size_t) 3 * S.nm * 256, (size_t) 3 * S.nm * 256, n / 64, hipMemcpyHostToDevice));
    HIP_TRY(uli(env->d.mfn, m.mfn)); HIP_TRY(uli(env->d.flags, m.flags));
    HIP_TRY(uli(env->d.curr_step, m.curr)); HIP_TRY(uli(env->d.remaining, m.rem));
    if (S.npair > 0) { HIP_TRY(ul(env->d.pmp, m.pmp)); HIP_TRY(uli(env->d.pmn, m.pmn)); }
    return EVM_OK;
}

int evm_env_get_residual(EvmEnv *env, float *h_max, int clear, void *stream) {
    if (!env) return fail(EVM_E_INVALID, "env is null");
    hipStream_t s = (hipStream_t) stream;
    int bits = 0;
    HIP_TRY(hipMemcpyAsync(&bits, env->d.resid, sizeof(int), hipMemcpyDeviceToHost, s));
    if (clear) HIP_TRY(hipMemsetAsync(env->d.resid, 0, sizeof(int), s));
    HIP_TRY(hipStreamSynchronize(s));
    if (h_max) memcpy(h_max, &bits, sizeof(float));
    return EVM_OK;
}

int evm_env_get_errors(EvmEnv *env, int *h_out, int clear, void *stream) {
    if (!env || !h_out) return fail(EVM_E_INVALID, "null argument");
    hipStream_t s = (hipStream_t) stream;
    HIP_TRY(hipMemcpyAsync(h_out, env->d.errs, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    if (clear) HIP_TRY(hipMemsetAsync(env->d.errs, 0, 2 * sizeof(int), s));
    HIP_TRY(hipStreamSynchronize(s));
    return EVM_OK;
}

int evm_env_get_pair_counters(EvmEnv *env, int *h_out, int clear, void *stream) {
    if (!env || !h_out) return fail(EVM_E_INVALID, "null argument");
    hipStream_t s = (hipStream_t) stream;
    HIP_TRY(hipMemcpyAsync(h_out, env->d.errs + 2, 3 * sizeof(int), hipMemcpyDeviceToHost, s));
    if (clear) HIP_TRY(hipMemsetAsync(env->d.errs + 2, 0, 3 * sizeof(int), s));
    HIP_TRY(hipStreamSynchronize(s));
    return EVM_OK;
}

int evm_env_get_speculation_counters(EvmEnv *env, int *h_out, int clear, void *stream) {
    if (!env || !h_out) return fail(EVM_E_INVALID, "null argument");
    hipStream_t s = (hipStream_t) stream;
#ifdef EVM_DIAG_PEN
    const int nspec = 30;   // + [3..8]: solver queries by origin: reset starting, pending, flagged by the previous step, other with / without cached points; in a settle step
#else
    const int nspec = 3;
#endif
    HIP_TRY(hipMemcpyAsync(h_out, env->d.errs + 5, nspec * sizeof(int), hipMemcpyDeviceToHost, s));
    if (clear) HIP_TRY(hipMemsetAsync(env->d.errs + 5, 0, nspec * sizeof(int), s));
    HIP_TRY(hipStreamSynchronize(s));
    return EVM_OK;
}

int evm_env_get_stats(EvmEnv *env, long long *h_out) {
    if (!env || !h_out) return fail(EVM_E_INVALID, "null argument");
    std::vector<int> h(2 * (size_t) env->d.n);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h.data(), env->d.stat, h.size() * 4, hipMemcpyDeviceToHost));
    h_out[0] = h_out[1] = 0;
    for (int e = 0; e < env->d.n_real; e++) { h_out[0] += h[tix(2, 0, e)]; h_out[1] += h[tix(2, 1, e)]; }
    return EVM_OK;
}
int evm_env_get_stamps(EvmEnv *env, unsigned long long *h_out /* [n_tiles, 16] */) {
    if (!env || !h_out) return fail(EVM_E_INVALID, "null argument");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h_out, env->d.stamps, (size_t) (env->d.n / 64) * 16 * 8, hipMemcpyDeviceToHost));
#ifdef EVM_KSTAMPS  // the narrowphase kernel accumulates with atomics: every read starts a new interval
    const unsigned long long reset[64] = {0, 0, 0, 0, 0, 0, ~0ull, 0};
    HIP_TRY(hipMemcpy(env->d.stamps, reset, sizeof(reset), hipMemcpyHostToDevice));
#endif
    return EVM_OK;
}
int evm_env_clear_stats(EvmEnv *env) {
    if (!env) return fail(EVM_E_INVALID, "env is null");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(env->d.stat, 0, 2 * (size_t) env->d.n * 4));
    return EVM_OK;
}

// ---- timing ------------------------------------------------------------------------------------
int evm_env_timing_begin(EvmEnv *env, void *stream) {
    if (!env) return fail(EVM_E_INVALID, "env is null");
    (void) stream;
    env->timing = true;
    env->timed_launches = 0;
    env->ev_used = 0;
    return EVM_OK;
}
int evm_env_timing_end(EvmEnv *env, void *stream, float *ms_total, int *n_launches) {
    if (!env) return fail(EVM_E_INVALID, "env is null");
    HIP_TRY(hipStreamSynchronize((hipStream_t) stream));
    float ms = 0.f;
    for (size_t i = 0; i < env->ev_used; i++) {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, env->ev_pairs[i].first, env->ev_pairs[i].second));
        ms += t;
    }
    if (env->ev_used > 0) ms = ms / (float) env->ev_used * (float) env->timed_launches;  // sampled steps, scaled to all
    env->timing = false;
    if (ms_total) *ms_total = ms;
    if (n_launches) *n_launches = env->timed_launches;
    return EVM_OK;
}
// the same, and in addition the summed duration of the sweeps kernel (k_split_sweeps) of those launches; 0 when the
// monolithic kernel ran (the sweeps are then a phase inside it)
int evm_env_timing_end_detail(EvmEnv *env, void *stream, float *ms_total, int *n_launches, float *ms_sweeps) {
    if (!env) return fail(EVM_E_INVALID, "env is null");
    HIP_TRY(hipStreamSynchronize((hipStream_t) stream));
    float sw = 0.f;
    const bool split = env->split < 0 ? env->d.n / 64 <= 128 : env->split == 1;
    if (split) {
        int ns = 0;
        for (size_t i = 0; i < env->ev_used; i += 2) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, env->ev_sweeps[i].first, env->ev_sweeps[i].second) == hipSuccess) { sw += t; ns++; }
        }
        if (ns > 0) sw = sw / ns * (float) env->timed_launches;  // sampled every 8th step, scaled to all of them
    }
    if (ms_sweeps) *ms_sweeps = sw;
    return evm_env_timing_end(env, stream, ms_total, n_launches);
}

}  // extern "C"
