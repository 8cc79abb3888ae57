// ORACLE — TEST INFRASTRUCTURE ONLY.  C entry points for tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg (ctypes).  Nothing under evomotion_amd/ may load this library.
#include <chrono>
#include <cstdio>
#include <random>
#include <string>

#include "orc_world.h"

using namespace orc;

static std::string g_err;

extern "C" {

const char *orc_last_error() { return g_err.c_str(); }

void *orc_env_create(const char *skel_path, int seed, float initial_remaining_seconds, float max_episode_seconds,
                     float target_velocity, float minimal_velocity, int reset_frames) {
    SkeletonDef s;
    if (!load_skeleton(skel_path, s, g_err)) return nullptr;
    EnvParams p;
    p.initial_remaining_seconds = initial_remaining_seconds;
    p.max_episode_seconds = max_episode_seconds;
    p.target_velocity = target_velocity;
    p.minimal_velocity = minimal_velocity;
    p.reset_frames = reset_frames;
    World *w = new World();
    if (!w->init(s, seed, p, g_err)) { delete w; return nullptr; }
    return w;
}
// the same with the environment kind: 0 robot_walk, 1 robot_jump
void *orc_env_create_kind(const char *skel_path, int seed, float initial_remaining_seconds, float max_episode_seconds,
                          float target_velocity, float minimal_velocity, int reset_frames, int env_kind) {
    SkeletonDef s;
    if (!load_skeleton(skel_path, s, g_err)) return nullptr;
    EnvParams p;
    p.initial_remaining_seconds = initial_remaining_seconds;
    p.max_episode_seconds = max_episode_seconds;
    p.target_velocity = target_velocity;
    p.minimal_velocity = minimal_velocity;
    p.reset_frames = reset_frames;
    p.env_kind = env_kind;
    World *w = new World();
    if (!w->init(s, seed, p, g_err)) { delete w; return nullptr; }
    return w;
}
void orc_env_destroy(void *h) { delete (World *) h; }
int orc_env_obs_dim(void *h) { return ((World *) h)->obs_dim(); }
int orc_env_act_dim(void *h) { return ((World *) h)->act_dim(); }
int orc_env_num_bodies(void *h) { return ((World *) h)->nb(); }
int orc_env_num_members(void *h) { return ((World *) h)->nmember(); }
int orc_env_state_size(void *h) { return ((World *) h)->state_size(); }
void orc_env_reset(void *h, float *obs, float *reward, int *done) { ((World *) h)->reset(obs, reward, done); }
void orc_env_step(void *h, const float *action, float *obs, float *reward, int *done) {
    ((World *) h)->do_step(action, obs, reward, done);
}
void orc_env_reset_begin(void *h) { ((World *) h)->reset_begin(); }
void orc_env_apply_action(void *h, const float *a) { ((World *) h)->apply_action(a); }
void orc_env_physics_step(void *h) { ((World *) h)->physics_step(); }
void orc_env_compute_step(void *h, float *obs, float *reward, int *done) { ((World *) h)->compute_step(obs, reward, done); }
void orc_env_set_counters(void *h, int curr_step, int remaining) {
    ((World *) h)->curr_step = curr_step; ((World *) h)->remaining_steps = remaining;
}
void orc_env_get_counters(void *h, int *out) {
    World *w = (World *) h;
    out[0] = w->curr_step; out[1] = w->remaining_steps; out[2] = w->max_steps; out[3] = w->last_num_contacts;
    out[4] = w->last_num_joint_rows;
}
void orc_env_get_state(void *h, float *out) { ((World *) h)->get_state(out); }
void orc_env_set_state(void *h, const float *in) { ((World *) h)->set_state(in); }
void orc_env_get_poses(void *h, float *out) { ((World *) h)->get_poses(out); }
// constants the loader derived (for cross-checking the product loader): per body
// [mass, inv_mass, invI x y z, friction, break_thr, first_model basis rows (9), origin (3)] = 19 floats
void orc_env_get_body_constants(void *h, float *out) {
    World *w = (World *) h;
    int k = 0;
    for (const Body &b : w->bodies) {
        out[k++] = b.mass; out[k++] = b.inv_mass;
        out[k++] = b.inv_inertia_local.x; out[k++] = b.inv_inertia_local.y; out[k++] = b.inv_inertia_local.z;
        out[k++] = b.friction; out[k++] = b.break_thr;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) out[k++] = b.first_model.b.r[i][j];
        out[k++] = b.first_model.o.x; out[k++] = b.first_model.o.y; out[k++] = b.first_model.o.z;
    }
}

// Self-check of the MT19937 / uniform_real_distribution<float> restatement against libstdc++ itself.
int orc_selftest_rng(int seed, int n) {
    MT19937 m;
    m.seed((uint32_t) seed);
    std::mt19937 ref(seed);
    std::uniform_real_distribution<float> uni(0.f, 1.f);
    for (int i = 0; i < n; i++) {
        float a = m.uniform01();
        float b = uni(ref);
        if (a != b) return i + 1;
    }
    return 0;
}
void orc_rng_draws(int seed, int n, float *out) {
    MT19937 m;
    m.seed((uint32_t) seed);
    for (int i = 0; i < n; i++) out[i] = m.uniform01();
}

// CPU baseline: time `steps` do_step() calls (with reset when done) on one env, one thread. Returns seconds.
double orc_bench_env_steps(void *h, int steps, unsigned action_seed, int *n_resets_out) {
    World *w = (World *) h;
    std::vector<float> obs(w->obs_dim()), act(w->act_dim());
    float reward; int done;
    uint32_t s = action_seed;
    auto t0 = std::chrono::steady_clock::now();
    w->reset(obs.data(), &reward, &done);
    int resets = 1;
    for (int i = 0; i < steps; i++) {
        for (float &a : act) { s = s * 1664525u + 1013904223u; a = ((s >> 8) * (1.0f / 16777216.0f)) * 2.f - 1.f; }
        w->do_step(act.data(), obs.data(), &reward, &done);
        if (done) { w->reset(obs.data(), &reward, &done); resets++; }
    }
    auto t1 = std::chrono::steady_clock::now();
    if (n_resets_out) *n_resets_out = resets;
    return std::chrono::duration<double>(t1 - t0).count();
}
}
