// Device-side view of the vectorised environment state.  Every array is [tile][slot][64 lanes] (tile = 64
// consecutive envs = one wavefront): a slot of a tile is one coalesced 256-byte line.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "skel_const.h"

namespace evm {

#define EVM_FLAG_POWERED 1  // Muscle::contract has been called at least once (muscle.cpp:82-85)
#define EVM_FLAG_PENDING 2  // reset() re-posed the bodies; transforms are E*M0 until the next integrate
#define EVM_FLAG_DONE 4     // rollout form: the last emitted transition was terminal

struct EnvDev {
    int n;       // padded env count (multiple of 64)
    int n_real;  // env count
    float *pos, *quat, *lin, *ang;  // [n/64][nb*3|4][64]
    float *hist;                    // [nm*6][n]  last_lin, last_ang (proprioception_state.cpp:33-34)
    int *mfn;                       // [nm][n]    persistent manifold point counts
    float *mfp;                     // [nm*4*9][n] localA3 localB3 dist applied applied_lateral
    int *pmn;                       // [npair][n]  member-vs-member persistent manifold point counts (null unless self_collision)
    float *pmp;                     // [npair*48][n] their points (EVM_PM_STRIDE)
    unsigned *pact;                 // [ceil(npair/32)][n] bit p: pair p holds a point after this step's collision detection
    float *crec;                    // [(nm+npair)*84][n] two-body contact records of the step (EVM_CR_STRIDE), by manifold id
    int *plist;                     // [npair][n] per pair: the envs whose boxes overlap or that hold a cached point (this step)
    int *blist;                     // [n * npair] (pair << 20 | env) entries of the pairs with a big hull, one flat list
    int *pcount;                    // [2][EVM_PC_STRIDE] the lists' lengths, [npair] = the flat list's, [npair + 1] = the urgent list's; two copies: a step appends to
                                    // copy pc_cur and zeroes the other one for the next step (the host flips pc_cur per launch)
    int pc_cur;
    int gtile_only;                 // the 64-env tile (3 KB per body) does not fit the CU's LDS: the tile sweeps kernel works on the
                                    // global staging copy instead (slower), and there is no monolithic kernel
    int npair_host;                 // EvmSkelC::npair, for the launch geometry
    float *target;                  // [nmus][n]  slider target velocity
    int *flags, *curr_step, *remaining, *settle_left;  // [n]
    float *E;                       // [9][n]     reset rotation (rows)
    float *iinv_stale;              // [nb*6][n]  world inverse inertia used by the first step after reset()
    uint32_t *mt;                   // [624][n]   std::mt19937 state
    int *mt_idx;                    // [n]
    float *scratch;                 // [sc_total][n] per-step constraint data
    float *diag;                    // [2][n]
    unsigned long long *stamps;     // [n/64][16] phase clock stamps (diagnostic builds with -DEVM_STAMPS only)
    int *stat;                      // [2][n]     do_step transitions emitted, resets started (rollout form)
    int *resid;                     // [0] = max over the batch and over the launches since the last clear of the PGS residual
                                    // (largest |delta impulse| of a row in the LAST sweep), fp32 bits (>= 0: integer order)
    int *errs;                      // [0] version waits that timed out (a schedule bug), [1] manifolds left out of a step's contact
                                    // rounds (more than 32 live manifolds or 31 rounds in one env); sticky until cleared;
                                    // [2] narrowphase queries that went through the penetration-depth solver, [3] those of them on the urgent list, [4] entries of the urgent list, [5] penetration queries run by speculation blocks, [6] their answers used, [7] waits for an answer that ran out (counters, not errors)
    float *gtile;                   // [n/64][tile_floats] global staging copy of the LDS tile (split pipeline)
    int tile_floats;                // step_lds_bytes / 4
    const EvmGSchedC *gs;           // lane-group sweep schedule (device copy), or null: the 64-env tile sweeps kernel runs
    int g_waves, g_lds;             // waves per 16-env workgroup and dynamic LDS bytes of k_sweeps_g
    int *spec;                      // [EVM_SPEC_SLOTS][EVM_SPEC_WORDS] slots of the urgent list's speculation blocks (narrow_dev.h); null: none
    float gap_soon;                 // core-box separation below which a pair without cached points goes to the urgent list (EVM_GAP_SOON; 0: the boxes overlap)
    float deep_soon;                // distance below which a pair goes to the next step's urgent list (EVM_DEEP_SOON)
    int spec_epoch;                 // this launch's epoch (the host counts the steps; never 0): a slot's answer is for exactly one launch
};

hipError_t upload_skeleton(const EvmSkelC *h, hipStream_t s);
size_t step_lds_bytes(int nb, int nscan);
// split = 1: pre / sweeps / post kernels (the throughput phases spread over the whole chip); 0: one monolithic kernel
// ev_sweeps0/1 (optional): recorded around the sweeps kernel of the split pipeline
hipError_t launch_step(const EnvDev &d, size_t lds_bytes, int split, int mode, const float *action, float *obs, float *reward,
                       uint8_t *done, uint8_t *valid, const uint8_t *mask, hipStream_t s, hipEvent_t ev_sweeps0 = nullptr,
                       hipEvent_t ev_sweeps1 = nullptr);
hipError_t launch_repose(const EnvDev &d, const uint8_t *mask, hipStream_t s);
hipError_t launch_init(const EnvDev &d, uint64_t seed, hipStream_t s);
hipError_t launch_poses(const EnvDev &d, float *out, hipStream_t s);

}  // namespace evm
