/*
 * evomotion.h — C ABI of libevomotion_hip.so (MI355X / gfx950).
 *
 * Vectorised drop-in for the reference's two abstract C++ interfaces on the robot_walk hot path.  The
 * reference has no FFI layer; these entry points are what a binding for that path would call.  Each one
 * cites the reference interface it replaces (paths relative to the reference repository root).
 *
 * Conventions: every function returns 0 on success or a negative EVM_E_* code and records a message
 * retrievable with evm_last_error(); no C++ exception crosses this boundary.  `d_` pointers are DEVICE
 * pointers owned by the caller; `h_` pointers are host pointers.  All work is enqueued on the stream given
 * (a hipStream_t passed as void*; NULL = the default stream) and is asynchronous with respect to the host
 * unless stated otherwise.  One host thread per GPU; an EvmEnv / EvmPolicy is not re-entrant (like the
 * reference's Environment: evo_motion_model/include/evo_motion_model/environment.h:35-73).
 */
#ifndef EVOMOTION_H
#define EVOMOTION_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EVM_OK 0
#define EVM_E_INVALID (-1)    /* std::invalid_argument in the reference (env_factory.cpp:118) */
#define EVM_E_RUNTIME (-2)    /* std::runtime_error in the reference (skeleton.cpp:46,58) */
#define EVM_E_HIP (-3)        /* a HIP runtime call failed */
#define EVM_E_UNSUPPORTED (-4)

typedef struct EvmEnv EvmEnv;
typedef struct EvmPolicy EvmPolicy;

/* RobotWalkFactory defaults: evo_motion_model/src/env/env_factory.cpp:74-83 */
typedef struct EvmEnvParams {
    float initial_remaining_seconds; /* 1.0  */
    float max_episode_seconds;       /* 30.0 */
    float target_velocity;           /* 0.5  */
    float minimal_velocity;          /* 0.1  */
    int reset_frames;                /* 30   */
    int env_kind;                    /* 0 = robot_walk, 1 = robot_jump (evo_motion_model/src/env/robot_jump.cpp:66-110:
                                        reward max(vy, 0) + vz, fail on remaining < 0, reset angles within pi/3, and
                                        reset_frames = int(reset_seconds / dt) = 10 settle steps in one loop) */
    int self_collision;              /* 1 (default) = member-vs-member contacts as in the reference: every pair of members may
                                        collide except constraint parent / child (evo_motion_model/src/robot/constraint.cpp:65,147;
                                        dispatcher / broadphase evo_motion_model/src/environment.cpp:20-31); 0 = floor contacts only */
} EvmEnvParams;

const char *evm_last_error(void);
void evm_env_default_params(EvmEnvParams *out);   /* robot_walk */
/* get_environment_factory(name, {}) defaults (env_factory.cpp:74-100,109-120): "robot_walk" or "robot_jump";
 * EVM_E_INVALID for any other name (std::invalid_argument, env_factory.cpp:118). */
int evm_env_default_params_for(const char *env_name, EvmEnvParams *out);

/*
 * Replaces get_environment_factory("robot_walk", params)->get_env(num_threads, seed)
 * (evo_motion_model/include/evo_motion_model/environment.h:80,96-97; RobotWalk ctor
 * evo_motion_model/src/env/robot_walk.cpp:17-46) for n_envs independent environments on HIP device
 * `device`.  Environment i is seeded with seed + i (std::mt19937 stream of robot_walk.h:34).
 * skeleton_path: the decoded skeleton fixture (evomotion_amd/data/robot_walk_spider.skel).
 */
int evm_env_create(const char *skeleton_path, int n_envs, int device, uint64_t seed, const EvmEnvParams *params,
                   EvmEnv **out);
void evm_env_destroy(EvmEnv *env);

/* Environment::get_state_space / get_action_space (environment.h:66-67): 371 and 12 for robot_walk. */
int evm_env_spaces(const EvmEnv *env, int *state_dim, int *action_dim);
int evm_env_counts(const EvmEnv *env, int *n_envs, int *n_bodies, int *n_members, int *n_muscles);
/* Member pairs that may collide (EvmEnvParams::self_collision): every pair of members except constraint parent / child
 * (constraint.cpp:65,147), lexicographic — also the order of their contact rows in the solver.  h_pairs (optional)
 * [n_pairs, 2] member indices.  n_pairs = 0 with self_collision = 0. */
int evm_env_pairs(const EvmEnv *env, int *n_pairs, int *h_pairs);

/*
 * Environment::reset() (evo_motion_model/src/environment.cpp:45-48 -> RobotWalk::reset_engine,
 * robot_walk.cpp:76-104) for every env whose d_mask byte is non-zero (NULL = all): 3 RNG draws, rigid
 * re-pose, 2*reset_frames settle steps, then compute_step().  Outputs are written for the masked envs only.
 * d_obs [n_envs, state_dim] f32, d_reward [n_envs] f32, d_done [n_envs] u8.
 */
int evm_env_reset(EvmEnv *env, const uint8_t *d_mask, float *d_obs, float *d_reward, uint8_t *d_done,
                  void *stream);

/*
 * Environment::do_step(action) (evo_motion_model/src/environment.cpp:33-39) for all envs: controllers
 * (muscle_controller.cpp:10-12), one stepSimulation(1/60), compute_step (robot_walk.cpp:56-74).
 * d_action [n_envs, action_dim] f32 is borrowed for the call.  No automatic reset.
 */
int evm_env_step(EvmEnv *env, const float *d_action, float *d_obs, float *d_reward, uint8_t *d_done, void *stream);

/*
 * Rollout form of the reference's train loop body (src/train.cpp:61-66): do_step for running envs; an env
 * whose previous transition was `done` starts reset() instead and spends the next 2*reset_frames calls in
 * its settle steps, one physics step per call (d_valid = 0, outputs untouched), then emits reset()'s own
 * step (d_valid = 2: the first state of the next episode, not a transition).  d_valid = 1 marks a do_step
 * transition.  Every lane does exactly one physics step per call, so the wavefront never diverges on episode
 * boundaries.  The per-env transition sequence is the reference's.
 */
int evm_env_step_autoreset(EvmEnv *env, const float *d_action, float *d_obs, float *d_reward, uint8_t *d_done,
                           uint8_t *d_valid, void *stream);

/* Parity metric helper: world poses [n_envs, n_bodies, 7] = (px py pz qx qy qz qw), body order of
 * Skeleton::get_bodies (evo_motion_model/src/robot/skeleton.cpp:92-103). */
int evm_env_get_body_poses(const EvmEnv *env, float *d_pose, void *stream);

/* Test / checkpoint hooks (synchronous, host buffers).  Canonical per-env state blob, all f32:
 *   nb x (px py pz qx qy qz qw lx ly lz ax ay az)
 *   reset_pending, E[9]                      (row-major reset rotation, used while reset_pending)
 *   nb x inv_inertia_world (xx xy xz yy yz zz) (used by the next step only while reset_pending)
 *   nm x motion_state_origin[3]
 *   nm x (last_lin[3], last_ang[3])          (proprioception history, member array order)
 *   nm x (count, 4 x (localA[3], localB[3], dist, applied, applied_lateral))
 *   with self_collision: n_pairs x (count, 4 x (localA[3], localB[3], normalOnB[3], dist, applied, applied_lateral))
 *   nmuscle x target_velocity, powered, curr_step, remaining_steps
 */
int evm_env_state_size(const EvmEnv *env);
int evm_env_get_state(EvmEnv *env, float *h_state /* [n_envs, state_size] */);
int evm_env_set_state(EvmEnv *env, const float *h_state);
/* Diagnostic (synchronous): per member pair, how many envs the last step's broadphase handed to the narrowphase. */
int evm_env_debug_pair_counts(EvmEnv *env, int *h_out /* [n_pairs + 1]: per small-hull pair; last = all big-hull pairs together */);
/* Low-level pieces of reset()/do_step() for step-by-step parity tests (synchronous). */
int evm_env_debug_reset_begin(EvmEnv *env, const uint8_t *d_mask);
int evm_env_debug_physics_steps(EvmEnv *env, int n_steps, const uint8_t *d_mask);
/* Host-only (no HIP call): parse a skeleton fixture and return what the loader derived.
 * counts[10] = nb nm nhinge nfixed nmuscle state_dim action_dim root max_steps initial_remaining;
 * h_out (may be NULL) = per body the 19 floats of evm_env_get_body_constants; capacity 64 bodies. */
int evm_skeleton_probe(const char *skeleton_path, int *counts, float *h_out);
/* Host-only: digest of every constant the loader derives from a skeleton file.  `skeleton_path` of every evm_* entry
 * point is either the decoded text fixture (*.skel) or the reference's own format: a skeleton JSON
 * (evo_motion_model/src/json_serializer.cpp:113-168; floats as 32-character bit strings, converter.cpp:138-147) whose
 * hulls are read from <json dir>/../obj/<shape>.obj (shapes.cpp:24-56) — the `skeleton_json_path` parameter of
 * RobotWalkFactory (env_factory.cpp:74-83). */
int evm_skeleton_digest(const char *skeleton_path, unsigned long long *h_out);
/* Host-only: the Gauss-Seidel visit list (Bullet order) and the per-wave dataflow schedule derived from it.
 * dims[4] = nvisit, nlevels, n_waves, cap; visits [nvisit,4] = type, body a, body b, need (needA | needB << 16);
 * sched [n_waves, cap] = joint visit index, or 0x4000 | member for that member's contact rows, -1 = past the end. */
int evm_skeleton_schedule(const char *skeleton_path, int *dims, int *visits, int *sched, int cap);
/* Host-only: the lane-group sweep schedule of the split pipeline's sweeps kernel (one wavefront = 4 lane groups x 16 envs,
 * every group on a different constraint of one type): dims[4] = entries, n_waves, LDS bytes, estimated cycles;
 * entries [n, 2 + 5 * 4] = wave | global order << 8, type (0 hinge, 1 fixed, 2 slider, 3 p2p, 4 contact rows), then per slot
 * record (-1 = empty), body a, body b, need (versions a | b << 16), visits per sweep (a | b << 16). */
int evm_skeleton_group_schedule(const char *skeleton_path, int n_waves, int *dims, int *entries, int cap);
/* the same for a collision mode: with self_collision = 1 the schedule holds the joint entries only (the contact rows of
 * floor and member-pair manifolds run as rounds chosen per env at run time: csrc/contact_rounds.h) */
int evm_skeleton_group_schedule_ex(const char *skeleton_path, int n_waves, int self_collision, int *dims, int *entries, int cap);
/* Loader cross-check: per body 19 floats [mass, inv_mass, invI xyz, friction, break_thr, M0 rows(9), t0(3)] */
int evm_env_get_body_constants(const EvmEnv *env, float *h_out);
/* Per-env diagnostics of the last physics step: [max |delta impulse| of the last PGS iteration, contacts] */
int evm_env_get_diagnostics(const EvmEnv *env, float *d_out /* [n_envs, 2] */, void *stream);

/* Convergence diagnostic of the 10-sweep projected Gauss-Seidel solve, reduced over the whole batch on the device (per env
 * in the wave, one atomic max per workgroup): the largest |delta impulse| any constraint row applied in its LAST sweep, maximum
 * over all envs and all physics steps since the last clear.  Synchronises `stream`. */
int evm_env_get_residual(EvmEnv *env, float *h_max_delta_impulse, int clear, void *stream);
/* Things that must not happen, counted on the device since the last clear (either one also poisons the residual above with
 * +inf): h_out[0] = waits of the sweeps kernel's dataflow schedule that timed out (a schedule bug: the physics of that step is
 * wrong), h_out[1] = contact manifolds left out of a step because an env held more than 32 live manifolds or needed more
 * than 31 contact rounds (member-vs-member mode).  Synchronises `stream`. */
int evm_env_get_errors(EvmEnv *env, int *h_out /* [2] */, int clear, void *stream);
/* Member-vs-member mode: h_out[0] = narrowphase queries since the last clear whose cores overlapped and that went through the
 * penetration-depth solver (Bullet: btGjkEpaPenetrationDepthSolver::calcPenDepth, selected by the btDefaultCollisionConfiguration of
 * evo_motion_model/src/environment.cpp:20-31); every physics step counts, reset()'s settle steps too; h_out[1] = how many of them
 * the previous step had predicted (they were worked on first, one per wavefront), h_out[2] = entries of that urgent list (predictions,
 * right or wrong).  Synchronises `stream`. */
int evm_env_get_pair_counters(EvmEnv *env, int *h_out /* [3] */, int clear, void *stream);
/* ... and the speculation blocks of that list: every entry is also taken by a block of its own that runs the pair's penetration query
 * from the first cycle of the launch, beside the pair's set-up and GJK instead of after them (the answer depends on the two transforms
 * only); the entry's owner asks for the answer or calls the run off.  h_out[0] = such runs, h_out[1] = answers that were used,
 * h_out[2] = waits for an answer that ran out (the query was then run in place: slower, same result).  EVM_SPECULATE=0 in the
 * environment turns the blocks off.  No reference counterpart (scheduling only).  Synchronises `stream`. */
int evm_env_get_speculation_counters(EvmEnv *env, int *h_out /* [3] */, int clear, void *stream);
/* Rollout counters since the last clear, summed over envs: h_out[0] = do_step transitions emitted by
 * evm_env_step_autoreset (the reset()'s own step and settle calls are not counted), h_out[1] = resets started. */
int evm_env_get_stats(EvmEnv *env, long long *h_out /* [2] */);
int evm_env_clear_stats(EvmEnv *env);
/* Diagnostic builds (-DEVM_STAMPS) only: s_memtime stamps at the phase boundaries of the last step, [n_tiles, 16]. */
int evm_env_get_stamps(EvmEnv *env, unsigned long long *h_out);

/* Kernel timing for bench.py's roofline line: between _begin and _end every dynamics-kernel launch is bracketed
 * by its own pair of HIP events on the launch stream; _end synchronises the stream and returns the summed
 * duration (ms) and the number of launches. */
int evm_env_timing_begin(EvmEnv *env, void *stream);
int evm_env_timing_end(EvmEnv *env, void *stream, float *ms_total, int *n_launches);
/* A step is a short pipeline of kernels (reset prologue, two setup kernels, the Gauss-Seidel sweeps, integration +
 * observation) up to 8192 envs, one monolithic kernel above; ms_total covers the whole step, ms_sweeps the sweeps kernel
 * alone (the dominant one; 0 for the monolithic form). */
int evm_env_timing_end_detail(EvmEnv *env, void *stream, float *ms_total, int *n_launches, float *ms_sweeps);

/* ---------------------------------------------------------------------------------------------------------
 * Agent side of the rollout: PpoGaeAgent::act for a whole batch
 * (evo_motion_networks/src/agents/ppo_gae.cpp:29-45; ActorModule / CriticModule networks/actor.cpp:9-48,
 *  networks/critic.cpp:8-35; truncated normal functions.cpp:53-68,94-111), hidden_size = 256.
 * ------------------------------------------------------------------------------------------------------- */
int evm_policy_create(int state_dim, int action_dim, int hidden_size, int device, EvmPolicy **out);
void evm_policy_destroy(EvmPolicy *p);
int evm_policy_param_counts(const EvmPolicy *p, size_t *n_actor, size_t *n_critic); /* 168216 / 162305 */
/* Flat fp32 parameters in the reference's named_parameters() order (weights [out,in] row major). */
int evm_policy_set_weights(EvmPolicy *p, const float *h_actor, size_t n_actor, const float *h_critic, size_t n_critic);
/* The same from DEVICE buffers, asynchronous on `stream` (no host round trip after an optimiser step); either pointer may
 * be NULL to leave that network unchanged. */
int evm_policy_set_weights_device(EvmPolicy *p, const float *d_actor, const float *d_critic, void *stream);
/* (mu, sigma) = actor(obs); action = truncated_normal_sample(mu, sigma, -1, 1); logp = truncated_normal_log_pdf;
 * value = critic(obs).  d_uniform [n, A] supplies the U[0,1) draws the reference takes from at::rand (pass NULL to
 * use the built-in counter-based generator keyed by seed and an internal call counter).
 * d_action, d_logp [n, A]; d_value [n]; d_mu, d_sigma [n, A] optional (may be NULL).  d_value == NULL runs the actor
 * only (SoftActorCriticAgent::act, soft_actor_critic.cpp:47-52). */
int evm_policy_forward(EvmPolicy *p, int n, const float *d_obs, const float *d_uniform, uint64_t seed, float *d_action,
                       float *d_logp, float *d_value, float *d_mu, float *d_sigma, void *stream);

/* Rows per workgroup tile of evm_policy_forward: 0 (default) = 32 rows, the hidden layers as six bf16 MFMA products per fp32
 * product (weights and activations cut exactly into three bf16 planes; error of the order of fp32 rounding — the fp32 matrix
 * pipe of gfx950 is its vector pipe; EVM_POLICY_SPLIT=0 at creation: v_mfma_f32_32x32x2_f32 instead), except 16 rows on
 * v_mfma_f32_16x16x4_f32 while 32-row tiles would leave CUs without a workgroup (below 4096 rows with both networks, 8192 actor only);
 * 16 / 32 force one form (measurements, tests).  The two forms agree to fp32 rounding (another k order), not bit for bit. */
int evm_policy_set_tile_rows(EvmPolicy *p, int rows);

int evm_policy_timing_begin(EvmPolicy *p);
int evm_policy_timing_end(EvmPolicy *p, void *stream, float *ms_total, int *n_launches);

/* ---------------------------------------------------------------------------------------------------------
 * Replay memory of the SAC rows: ReplayBuffer add / update_last / sample for N environments at once
 * (evo_motion_networks/src/replay_buffer.cpp:16-52,146-153; used by soft_actor_critic.cpp:47-91).
 * Device-resident ring of `capacity_slots` rollout steps; slot t holds, for every env, the state the agent acted
 * on, its action and what the env returned (reward, done, valid code of evm_env_step_autoreset).  The next state of
 * a transition is the state of the following slot (for the newest slot: the last next_state pushed), so every
 * observation is stored once.  The oldest slot is evicted first (replay_buffer.cpp:31-35).
 * ------------------------------------------------------------------------------------------------------- */
typedef struct EvmReplay EvmReplay;
int evm_replay_create(int capacity_slots, int n_envs, int state_dim, int action_dim, int device, EvmReplay **out);
void evm_replay_destroy(EvmReplay *rb);
/* add + update_last for all envs: rows with d_valid != 1 (settle calls, reset()'s own emission) are kept in the
 * slot but are never sampled.  d_valid == NULL: every row is a transition. */
int evm_replay_push(EvmReplay *rb, const float *d_state, const float *d_action, const float *d_reward,
                    const uint8_t *d_done, const uint8_t *d_valid, const float *d_next_state, void *stream);
/* h_out[0] = stored transitions (valid rows in live slots), h_out[1] = live slots, h_out[2] = pushes so far (sync). */
int evm_replay_stats(EvmReplay *rb, long long *h_out /* [3] */, void *stream);
/* `batch` distinct transitions drawn uniformly from the stored ones (std::shuffle + first batch_size,
 * replay_buffer.cpp:18-28; the draw is a keyed permutation of the stored-transition ranks).  If fewer than `batch`
 * are stored the ranks wrap around.  Outputs [batch, S], [batch, A], [batch], [batch], [batch, S] f32.
 * d_index (optional) [batch, 2] i32 = (slot, env) of each draw. */
int evm_replay_sample(EvmReplay *rb, int batch, uint64_t seed, float *d_states, float *d_actions, float *d_rewards,
                      float *d_done, float *d_next_states, int *d_index, void *stream);
int evm_replay_timing_begin(EvmReplay *rb);
int evm_replay_timing_end(EvmReplay *rb, void *stream, float *ms_push, int *n_push, float *ms_sample, int *n_sample);

/* ---------------------------------------------------------------------------------------------------------
 * PPO / GAE update: PpoGaeAgent::train (evo_motion_networks/src/agents/ppo_gae.cpp:117-190) on
 * the device — GAE scan and normalisation (:127-150), clipped surrogate + entropy bonus and the critic's squared
 * error (:155-179), the backward pass of ActorModule / CriticModule (networks/actor.cpp:9-48, critic.cpp:8-35),
 * clip_grad_norm_ and one Adam step per network (:170-186), all fp32.  The trainer owns the master parameters,
 * gradients and Adam moments as flat vectors in named_parameters() order and writes every new set of weights
 * straight into `policy` (the operand layout of evm_policy_forward), so the next rollout needs no upload.
 * Rollouts are time-major: [horizon][n_envs] (rows = horizon * n_envs).
 * ------------------------------------------------------------------------------------------------------- */
typedef struct EvmPpo EvmPpo;
/* `policy` is borrowed and must outlive the trainer; max_rows bounds `rows` of evm_ppo_grads (activations are kept
 * for the backward pass: about 6.3 KiB of HBM per row and network, plus 1.5 KiB per row for the aligned observations). */
int evm_ppo_create(EvmPolicy *policy, size_t max_rows, EvmPpo **out);
void evm_ppo_destroy(EvmPpo *q);
/* flat DEVICE parameter vectors (evm_policy_param_counts floats each); reset_optimizer != 0 zeroes the Adam moments and
 * step counts (a freshly constructed torch::optim::Adam, ppo_gae.cpp:22-25) */
int evm_ppo_set_params(EvmPpo *q, const float *d_actor, const float *d_critic, int reset_optimizer, void *stream);
/* what: 0 parameters, 1 gradients, 2 Adam exp_avg, 3 Adam exp_avg_sq; net: 0 actor, 1 critic.  to_trainer == 0 copies
 * the trainer's vector to d_buf, 1 copies d_buf into the trainer (gradients after an all-reduce, moments of a loaded
 * checkpoint; parameters go through evm_ppo_set_params). */
int evm_ppo_copy(EvmPpo *q, int what, int net, int to_trainer, float *d_buf, void *stream);
/* The gradients of both networks as ONE contiguous DEVICE vector of *n_floats floats: the actor's at 0, the critic's at
 * *critic_offset (the actor's count rounded up to 64 floats; the gap is never written).  After
 * evm_ppo_grads a data-parallel caller all-reduces it in place on the launch stream — one collective per epoch, no copy, no host
 * synchronisation — then calls evm_ppo_apply.  d_buf != NULL: the trainer uses that caller-owned buffer from now on (it must
 * outlive the trainer, be 256-byte aligned and *n_floats long; zero it once); d_grads / n_floats / critic_offset are optional outputs. */
int evm_ppo_grad_buffer(EvmPpo *q, float *d_buf, float **d_grads, size_t *n_floats, size_t *critic_offset);
/* (count, mean, M2) of every rank's selected advantages -> the trainer's statistics, Chan's merge on the device in rank order;
 * d_all_stats [world][3] = an all-gather of the d_stats of evm_ppo_gae (d_all_stats and d_stats must not overlap).  Afterwards evm_ppo_gae_normalize(d_stats = NULL) and
 * evm_ppo_grads(n_selected_global < 0) use the merged numbers: the whole update runs without a host read
 * (the global statistic of ppo_gae.cpp:148-149 over all GPUs). */
int evm_ppo_gae_merge(EvmPpo *q, const double *d_all_stats, int world, double *d_stats /* optional copy of the result */, void *stream);
/* Adam step count of a network: set when set_step >= 0, returned in *step.  net 2 = the actor's device-side counter that
 * evm_ppo_actor_apply advances (SAC's update inside a captured graph). */
int evm_ppo_adam_step(EvmPpo *q, int net, int set_step, int *step);
/* Raw advantages of ppo_gae.cpp:134-146 into d_adv [horizon][n_envs]; d_mask = 1 for a trained transition (done is
 * taken as 1 outside the mask, exactly like the reference's padding).  d_stats (optional, DEVICE double[3]) receives
 * (count, mean, sum of squared deviations) of the selected advantages, for merging over ranks. */
int evm_ppo_gae(EvmPpo *q, int horizon, int n_envs, const float *d_rewards, const uint8_t *d_done, const float *d_curr_values,
                const float *d_next_values, const uint8_t *d_mask, float gamma, float lam, float *d_adv, double *d_stats,
                void *stream);
/* adv = (adv - mean) / (std + 1e-8) with the unbiased std, returns = adv + V (ppo_gae.cpp:148-150).  d_stats == NULL
 * uses the statistics of this trainer's last evm_ppo_gae. */
int evm_ppo_gae_normalize(EvmPpo *q, int horizon, int n_envs, const double *d_stats, const float *d_curr_values, float *d_adv,
                          float *d_returns, void *stream);
/* The rows with d_mask != 0, in their order, as dense copies owned by the trainer (valid until the next call): both
 * losses are means over masked_select(..., mask) (ppo_gae.cpp:167-168, 178), so the rows outside the mask — the reference's padding,
 * here reset()'s settle calls and emissions — weigh nothing,
 * so after GAE (which needs the time structure) the epochs can run on the selected rows alone: pass *n_selected and the returned
 * pointers to evm_ppo_grads for every epoch of the update.  Same gradients to rounding (the same terms, summed in other tiles).
 * Reads the count back: one stream synchronisation per update.  *n_selected == 0: nothing copied, keep the original buffers. */
int evm_ppo_select_rows(EvmPpo *q, size_t rows, const uint8_t *d_mask, const float *d_states, const float *d_actions,
                        const float *d_logp_old, const float *d_adv, const float *d_returns, size_t *n_selected,
                        const float **s_states, const float **s_actions, const float **s_logp_old, const float **s_adv,
                        const float **s_returns, const uint8_t **s_mask, void *stream);
/* Forward, losses and backward of both networks over `rows` transitions: gradients land in the trainer (evm_ppo_copy).
 * n_selected_global = number of rows with d_mask == 1 over ALL ranks (the losses are means over them, so summing the
 * ranks' gradients gives the global gradient); negative: use the count of the trainer's own statistics on the device
 * (evm_ppo_gae, merged over the ranks by evm_ppo_gae_merge).  d_states [rows, S], d_actions / d_logp_old [rows, A], d_adv /
 * d_returns [rows], d_mask [rows].  states_unchanged != 0: d_states holds what it held in the previous call (the later
 * epochs of one train call) and the trainer's 16-byte aligned copy of it is reused. */
int evm_ppo_grads(EvmPpo *q, size_t rows, const float *d_states, const float *d_actions, const float *d_logp_old, const float *d_adv,
                  const float *d_returns, const uint8_t *d_mask, double n_selected_global, float epsilon, float entropy_factor,
                  float critic_loss_factor, int states_unchanged, void *stream);
/* SAC's actor is the same ActorModule: forward keeping the activations (mu, sigma [rows, A] out), then backward + weight
 * gradients from d loss / d(mu, sigma) supplied by the caller (evm_sac_actor_grad); gradients via evm_ppo_copy(1, 0, ...) */
int evm_ppo_actor_forward(EvmPpo *q, size_t rows, const float *d_states, float *d_mu, float *d_sigma, void *stream);
int evm_ppo_actor_backward(EvmPpo *q, size_t rows, const float *d_dmu, const float *d_dsigma, void *stream);
/* Adam step (no clipping) of the actor alone with a DEVICE step counter (zeroed by evm_ppo_set_params with
 * reset_optimizer), then the new weights into `policy`: replayable from a captured HIP graph */
int evm_ppo_actor_apply(EvmPpo *q, float learning_rate, void *stream);
/* clip_grad_norm_(clip_grad_norm) and Adam(lr, betas 0.9 / 0.999, eps 1e-8) for both networks, then the new weights into
 * `policy` */
int evm_ppo_apply(EvmPpo *q, float learning_rate, float clip_grad_norm, void *stream);
/* this rank's share of the actor / critic loss of the last evm_ppo_grads (synchronises the stream) */
int evm_ppo_losses(EvmPpo *q, double *h_actor_loss, double *h_critic_loss, void *stream);
/* returns the milliseconds (HIP events, evm_ppo_grads .. evm_ppo_apply) and epochs accumulated since the last call, then
 * switches the measurement on or off */
int evm_ppo_timing(EvmPpo *q, int enable, float *ms_total, int *n_epochs);

/* ---------------------------------------------------------------------------------------------------------
 * Twin Q networks of SAC: critic_1 / critic_2 and their target networks of SoftActorCriticAgent
 * (evo_motion_networks/src/agents/soft_actor_critic.cpp:20-45) as fp32 MFMA kernels — QNetworkModule forward
 * (networks/q_net.cpp:8-43), the critics' mse update (:100-127: backward, weight gradients, Adam) and soft_update
 * (functions.cpp:161-171).  Networks are numbered 0 critic_1, 1 critic_2, 2 target_critic_1, 3 target_critic_2;
 * parameters are flat fp32 vectors in named_parameters() order (q_network.0.weight [256][S+A] ... q_network.9.bias).
 * ------------------------------------------------------------------------------------------------------- */
typedef struct EvmQ EvmQ;
int evm_q_create(int state_dim, int action_dim, int hidden_size, size_t max_rows, int device, EvmQ **out);
void evm_q_destroy(EvmQ *q);
int evm_q_param_count(const EvmQ *q, size_t *n); /* 231 681 for robot_walk */
/* what: 0 parameters (any network), 1 gradients, 2 Adam exp_avg, 3 Adam exp_avg_sq (critics only).  to_trainer == 0
 * copies the trainer's vector to d_buf, 1 copies d_buf in (parameters are repacked for the kernels). */
int evm_q_copy(EvmQ *q, int what, int net, int to_trainer, float *d_buf, void *stream);
/* Adam step count of critic `net` (kept on the device): set when set_step >= 0, else read (synchronises) */
int evm_q_adam_step(EvmQ *q, int net, int set_step, int *step);
/* Q(states, actions) of the networks selected by the bit mask `nets` into d_out[net] ([rows] f32 each) */
int evm_q_forward(EvmQ *q, unsigned nets, size_t rows, const float *d_states, const float *d_actions, float *const *d_out, void *stream);
/* gradients of mse_loss(critic_i(states, actions), target_q), i = 1, 2, into the trainer (evm_q_copy what = 1) */
int evm_q_grads(EvmQ *q, size_t rows, const float *d_states, const float *d_actions, const float *d_target_q, void *stream);
/* one Adam step (torch defaults, no clipping) of both critics */
int evm_q_apply(EvmQ *q, float learning_rate, void *stream);
/* target_i <- tau * critic_i + (1 - tau) * target_i */
int evm_q_soft_update(EvmQ *q, float tau, void *stream);
/* SAC actor step, Q side (soft_actor_critic.cpp:136-140): min(critic_1, critic_2)(states, actions) into d_qmin [rows] and the
 * gradient of -mean(min q) w.r.t. the actions into d_dqda [rows, A] (critics' forward, backward to their first layer) */
int evm_q_action_grad(EvmQ *q, size_t rows, const float *d_states, const float *d_actions, float *d_qmin, float *d_dqda, void *stream);
/* truncated_normal_sample(mu, sigma, -1, 1) with the supplied uniform draws and the summed truncated_normal_log_pdf of the
 * sample (functions.cpp:53-68,94-111; soft_actor_critic.cpp:131-135): [rows, A] in, d_action [rows, A], d_logp_sum [rows] */
int evm_sac_sample(int rows, int action_dim, const float *d_mu, const float *d_sigma, const float *d_uniform, float *d_action,
                   float *d_logp_sum, void *stream);
/* gradient of mean(exp(log_alpha) * logp_sum - min q) w.r.t. (mu, sigma), the action being the reparameterised sample
 * (soft_actor_critic.cpp:129-142); d_dqda from evm_q_action_grad, d_log_alpha a DEVICE scalar; outputs [rows, A] */
int evm_sac_actor_grad(int rows, int action_dim, const float *d_mu, const float *d_sigma, const float *d_uniform, const float *d_dqda,
                       const float *d_log_alpha, float *d_dmu, float *d_dsigma, void *stream);
/* target_q [rows] = r + (1 - done) * gamma * (min(tq1, tq2) - exp(log_alpha) * sum_a next_logp[., a])  (soft_actor_critic.cpp:108-116) */
int evm_sac_target_q(int rows, int action_dim, const float *d_rewards, const float *d_done, const float *d_tq1, const float *d_tq2,
                     const float *d_next_logp, const float *d_log_alpha, float gamma, float *d_target_q, void *stream);
/* Adam step of the entropy parameter on -mean(log_alpha * (logp_sum + target_entropy)) (soft_actor_critic.cpp:155-164).
 * d_log_alpha [1], d_adam_state [2] (exp_avg, exp_avg_sq) and d_adam_step [1] (int) are DEVICE memory of the caller;
 * d_losses [2] receives the actor loss mean(alpha * logp_sum - qmin) and the entropy loss, both before the step. */
int evm_sac_entropy_step(int rows, const float *d_logp_sum, const float *d_qmin, float target_entropy, float learning_rate,
                         float *d_log_alpha, float *d_adam_state, int *d_adam_step, float *d_losses, void *stream);
/* DEVICE double[2]: the critics' losses of the last evm_q_grads */
int evm_q_losses(EvmQ *q, double *d_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* EVOMOTION_H */
