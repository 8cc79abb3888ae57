// Device-side buffers and launchers of the PPO update (ppo_kernels.hip; C ABI in ppo_host.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "policy_dev.h"

namespace evm {

// one network's training state.  Parameters, gradients and Adam moments are flat fp32 vectors in the reference's
// named_parameters() order (the layout evm_policy_set_weights takes); activations are row-major [rows][256].
struct PpoNet {
    float *theta, *grad, *m, *v;  // [n_params]
    size_t n_params;
    float *w2d;                   // Linear(256,256) weight packed for the dgrad GEMM: B[k = out j][col = in i]
    float *whd;                   // head weights packed for their dgrad GEMM: B[k = head output o < 32][col = in i]
    float *z1, *a1, *z2, *a2;     // pre-Mish and post-LayerNorm activations of the two hidden layers
    float *st;                    // [rows][4]: LayerNorm mean, rstd of layer 1, of layer 2
    float *head;                  // actor: [rows][2A] = mu, sigma (after tanh / softplus); critic: [rows] values
    float *dh;                    // [rows][32] gradient of the loss w.r.t. the head pre-activations
    float *dz1, *dz2;             // gradients w.r.t. z1, z2
    float *colpart;               // [tiles][7][256] per-tile column sums (LayerNorm / bias gradients)
    float *colpart2;              // [64][7][256] second reduction level
    float *wpart;                 // split-K partials of the weight gradients
    double *normp;                // [PPO_NORM_PARTS] partial sums of squared gradients
    int step;                     // Adam step count
};

struct PpoDev {
    int S, A;
    size_t max_rows;
    PpoNet actor, critic;
    float *xpad;    // [max_rows][384] observations, zero padded (staged once per update)
    double *loss;   // [2] actor, critic loss sums of the last evm_ppo_grads call
    double *loss_part;  // per-workgroup partial sums of the two loss kernels: actor blocks, then critic blocks
    double *gae;    // [3] n, mean, M2 of the raw advantages
    double *gae_part;  // [blocks of 256 envs][3] partial count, sum, M2
    int *step_dev;  // [1] device-side Adam step count of the actor (evm_ppo_actor_apply: SAC's captured update)
    int dev_count;  // the last evm_ppo_grads took the count of selected transitions from gae[0] (n_selected_global < 0): an EMPTY
                    // selection then leaves the weights and moments untouched and reports NaN losses (k_ppo_adam, k_ppo_loss_sum)
};

constexpr int PPO_SK = 64;       // split-K chunks of the weight-gradient GEMMs
constexpr int PPO_NORM_PARTS = 32;
constexpr int PPO_COLSLOTS = 7;  // dgamma2, dbeta2, dbias2, dgamma1, dbeta1, dbias1, dbias_heads

hipError_t launch_ppo_pack_w2d(const PpoNet &n, int S, int A, bool actor, hipStream_t s);  // w2d and whd
hipError_t launch_ppo_pad(const PpoDev &d, size_t rows, const float *states, hipStream_t s);
hipError_t launch_ppo_forward(const PolicyDev &p, const PpoDev &d, size_t rows, const float *states, hipStream_t s, int nets = 2);
hipError_t launch_actor_head_grad(const PpoDev &d, size_t rows, const float *dmu, const float *dsigma, hipStream_t s);
hipError_t launch_actor_head_out(const PpoDev &d, size_t rows, float *mu, float *sigma, hipStream_t s);
hipError_t launch_ppo_loss(const PpoDev &d, size_t rows, const float *actions, const float *logp_old, const float *adv,
                           const float *returns, const uint8_t *mask, double inv_rows, float epsilon, float entropy_factor,
                           float critic_loss_factor, hipStream_t s);
hipError_t launch_ppo_backward(const PolicyDev &p, const PpoDev &d, size_t rows, hipStream_t s, int nets = 2);
hipError_t launch_ppo_wgrads(const PpoDev &d, size_t rows, const float *states, hipStream_t s, int nets = 2);
hipError_t launch_actor_apply(const PolicyDev &p, const PpoDev &d, float lr, hipStream_t s);
hipError_t launch_ppo_apply(const PolicyDev &p, PpoDev &d, float lr, float clip_grad_norm, hipStream_t s);
hipError_t launch_ppo_gae_merge(const PpoDev &d, const double *all, int world, hipStream_t s);
hipError_t launch_ppo_gae_scan(const PpoDev &d, int T, int N, const float *rewards, const uint8_t *done, const float *curr_values,
                               const float *next_values, const uint8_t *mask, float gamma, float lam, float *adv, hipStream_t s);
hipError_t launch_ppo_gae_finish(int T, int N, const double *stats, const float *curr_values, const uint8_t *mask, float *adv,
                                 float *returns, hipStream_t s);
// rows selected by mask (in order) gathered into dense copies; sel_idx / sel_count are trainer scratch
hipError_t launch_ppo_select_scan(size_t rows, const uint8_t *mask, int *sel_idx, int *sel_count, hipStream_t s);
hipError_t launch_ppo_select_gather(size_t n_sel, const int *sel_idx, int S, int A, const float *states, const float *actions,
                                    const float *logp, const float *adv, const float *returns, float *o_states, float *o_actions,
                                    float *o_logp, float *o_adv, float *o_returns, hipStream_t s);
size_t ppo_wpart_floats();
// generic pieces reused by the Q-network trainer (q_kernels.hip)
//   C[i][j] = sum_m P[m][i] Q[m][j] (split-K MFMA, partials in `part`) -> dst[i * dst_ld + j (+ extra for i >= split_row)]
void wgrad_one(const float *P, int ldp, int np, bool pa, const float *Q, int ldq, int nq, bool qa, int M, float *part, int I, int J,
               float *dst, int dst_ld, int split_row, int extra, hipStream_t s);
void wgrad_heads(const float *dh, const float *a_last, float *wpart, int M, int I, float *dst, int split_row, int extra, hipStream_t s);
int launch_colreduce(const float *colpart, int tiles, int W, float *colpart2, hipStream_t s);  // returns the number of groups
hipError_t launch_pad_rows(const float *src, int S, size_t rows, float *dst, hipStream_t s);   // [rows][S] -> [rows][384]

}  // namespace evm
