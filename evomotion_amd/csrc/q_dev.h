// Device-side buffers and launchers of the twin-Q trainer of SAC (q_kernels.hip; C ABI in q_host.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace evm {

constexpr int Q_LAYERS = 3;                    // hidden layers of QNetworkModule (q_net.cpp:8-24)
constexpr int Q_COLSLOTS = 3 * Q_LAYERS + 1;   // per layer dgamma, dbeta, dbias (last layer first) + the head's bias

// one Q network.  theta / grad / m / v: flat fp32 vectors in named_parameters() order
//   q_network.0.{weight [256][S+A], bias}  .2.{weight, bias} (LayerNorm)  .3.{weight [256][256], bias}  .5.{..}
//   .6.{weight, bias}  .8.{..}  .9.{weight [1][256], bias [1]}
struct QNet {
    float *theta, *grad, *m, *v;
    size_t n_params;
    size_t o_w[Q_LAYERS], o_b[Q_LAYERS], o_g[Q_LAYERS], o_be[Q_LAYERS], o_wh, o_bh;  // offsets into the flat vectors
    float *wt[Q_LAYERS];   // forward B operands, k-split (mlp_tile.h); wt[0] is zero padded to K1 = 384 inputs
    float *wd[Q_LAYERS];   // dgrad B operands of layers 1, 2 (wd[0] unused)
    // kept by a training forward (critics only; NULL for the target networks)
    float *z[Q_LAYERS], *a[Q_LAYERS], *st;  // st: [rows][2 * Q_LAYERS] LayerNorm mean, rstd per layer
    float *dz[Q_LAYERS];
    float *q;        // [rows] network output
    float *dh;       // [rows][32] gradient at the head pre-activation (column 0)
    float *colpart, *colpart2, *wpart;
    int *step;       // Adam step count, on the device (the update may be replayed from a HIP graph)
};

struct QDev {
    int S, A;        // state / action widths; the network input is [state, action], S + A <= 384
    size_t max_rows;
    QNet net[4];     // critic_1, critic_2, target_critic_1, target_critic_2
    float *xq;       // [max_rows][384]: [state, action, 0...] rows, 16-byte aligned
    double *loss;    // [2] mse losses of the two critics
};

hipError_t launch_q_pack(const QDev &d, int which, hipStream_t s);
hipError_t launch_q_concat(const QDev &d, size_t rows, const float *states, const float *actions, hipStream_t s);
// nets: bit mask over net[]; keep != 0 stores the activations (training forward).  Outputs land in net[i].q.
hipError_t launch_q_forward(const QDev &d, unsigned nets, size_t rows, int keep, hipStream_t s);
hipError_t launch_q_loss(const QDev &d, size_t rows, const float *target_q, hipStream_t s);
hipError_t launch_q_backward(const QDev &d, size_t rows, hipStream_t s);            // both critics
hipError_t launch_q_wgrads(const QDev &d, size_t rows, hipStream_t s);
hipError_t launch_q_adam(const QDev &d, float lr, hipStream_t s);                   // both critics, then repack
hipError_t launch_q_soft_update(const QDev &d, float tau, hipStream_t s);           // targets, then repack
size_t q_wpart_floats();
hipError_t launch_sac_sample(int rows, int A, const float *mu, const float *sigma, const float *u, float *action, float *logp_sum, hipStream_t s);
hipError_t launch_sac_actor_grad(int rows, int A, const float *mu, const float *sigma, const float *u, const float *dqda,
                                 const float *log_alpha, float *dmu, float *dsigma, hipStream_t s);
hipError_t launch_sac_target(int rows, int A, const float *r, const float *done, const float *tq1, const float *tq2, const float *next_logp,
                             const float *log_alpha, float gamma, float *out, hipStream_t s);
hipError_t launch_sac_entropy(int rows, const float *logp_sum, const float *qmin, float target_entropy, float lr, float *log_alpha,
                              float *state, int *step, float *losses, hipStream_t s);
hipError_t launch_q_action_grad(const QDev &d, size_t rows, float *qmin, float *dqda, hipStream_t s);

}  // namespace evm
