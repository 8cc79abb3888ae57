// Twin Q networks of SAC on fp32 MFMA (gfx950): forward of critics and target critics, the critics' mse update
// (backward, weight gradients, Adam) and the soft target update.  Replaces the PyTorch-ROCm pass over
//   QNetworkModule                           evo_motion_networks/src/networks/q_net.cpp:8-43
//   SoftActorCriticAgent::train (critics)    evo_motion_networks/src/agents/soft_actor_critic.cpp:100-127
//   soft_update                              evo_motion_networks/src/functions.cpp:161-171
// The tiles, epilogues and weight-gradient GEMMs are those of the PPO update (mlp_train.h, ppo_kernels.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mlp_tile.h"
#include "mlp_train.h"
#include "ppo_dev.h"
#include "q_dev.h"

namespace evm {

// ---------------------------------------------------------------------------------------------------------
// operand layouts
// ---------------------------------------------------------------------------------------------------------
// flat parameters -> wt[l] (forward: B[k = in][col = out] = W[out][in]) and wd[l] (dgrad: B[k = out][col = in] = W[out][in])
__global__ __launch_bounds__(256) void k_q_pack(QNet n, int SA) {
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    const size_t n_w0 = (size_t) 256 * SA;
    auto put = [](float *dst, int k, int col, float v) {
        const int st = k >> 1, h = k & 1, s4 = st >> 2, tt = st & 3;
        dst[(((size_t) s4 * 256 + col) * 2 + h) * 4 + tt] = v;
    };
    if (i < n_w0) {
        const int out = (int) (i / SA), in = (int) (i % SA);
        put(n.wt[0], in, out, n.theta[n.o_w[0] + i]);
        return;
    }
    size_t j = i - n_w0;
    for (int l = 1; l < Q_LAYERS; l++) {
        if (j < 65536) {
            const int out = (int) (j >> 8), in = (int) (j & 255);
            const float v = n.theta[n.o_w[l] + j];
            put(n.wt[l], in, out, v);
            if (n.wd[l]) put(n.wd[l], out, in, v);
            return;
        }
        j -= 65536;
    }
}

// One flat parameter (index i, value v) -> the operand layouts, for the kernels that change parameters in place (Adam, soft update):
// the repack costs no launch of its own.
__device__ __forceinline__ void q_pack_write(const QNet &n, int SA, size_t i, float v) {
    auto put = [](float *dst, int k, int col, float x) {
        const int st = k >> 1, h = k & 1, s4 = st >> 2, tt = st & 3;
        dst[(((size_t) s4 * 256 + col) * 2 + h) * 4 + tt] = x;
    };
    if (i >= n.o_w[0] && i < n.o_w[0] + (size_t) 256 * SA) {
        const size_t j = i - n.o_w[0];
        put(n.wt[0], (int) (j % SA), (int) (j / SA), v);
        return;
    }
#pragma unroll
    for (int l = 1; l < Q_LAYERS; l++)
        if (i >= n.o_w[l] && i < n.o_w[l] + 65536) {
            const size_t j = i - n.o_w[l];
            const int out = (int) (j >> 8), in = (int) (j & 255);
            put(n.wt[l], in, out, v);
            if (n.wd[l]) put(n.wd[l], out, in, v);
            return;
        }
}

// [state, action] rows, zero padded to K1 columns
__global__ __launch_bounds__(256) void k_q_concat(const float *__restrict__ states, const float *__restrict__ actions, int S, int A,
                                                  size_t rows, float *__restrict__ dst) {
    const size_t e = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (e >= rows * K1) return;
    const size_t r = e / K1;
    const int c = (int) (e - r * K1);
    dst[e] = c < S ? states[r * S + c] : (c < S + A ? actions[r * A + (c - S)] : 0.f);
}

// ---------------------------------------------------------------------------------------------------------
// forward: three Linear -> Mish -> LayerNorm blocks and Linear(256, 1)
// ---------------------------------------------------------------------------------------------------------
struct QSel { int idx[4]; int count; };
template <int RT>
__global__ __launch_bounds__(PT) void k_q_forward(QDev d, QSel sel, int n, int keep) {
    constexpr int TM = 32 * RT, PARTS = PT / TM, RUN = 256 / PARTS;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *xs = sm, *hb = sm;
    const QNet &N = d.net[sel.idx[blockIdx.y]];
    const int row0 = blockIdx.x * TM;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool kp = keep != 0 && N.z[0] != nullptr;
    stage_padded_ksplit<TM>(xs, d.xq, row0, n);
    __syncthreads();
    f32x16 acc[RT][2];
    float *red = sm + TM * ALD1;  // behind both tiles
    dense_layer<K1, RT>(xs, ALD1, N.wt[0], wave, lane, acc);
    train_epilogue<RT>(acc, N.theta + N.o_b[0], N.theta + N.o_g[0], N.theta + N.o_be[0], hb, red, wave, lane, row0, n, kp ? N.z[0] : nullptr,
                       kp ? N.a[0] : nullptr, kp ? N.st : nullptr, 0, 2 * Q_LAYERS);
#pragma unroll
    for (int l = 1; l < Q_LAYERS; l++) {
        dense_layer<256, RT>(hb, ALD2, N.wt[l], wave, lane, acc);
        train_epilogue<RT>(acc, N.theta + N.o_b[l], N.theta + N.o_g[l], N.theta + N.o_be[l], hb, red, wave, lane, row0, n,
                           kp ? N.z[l] : nullptr, kp ? N.a[l] : nullptr, kp ? N.st : nullptr, 2 * l, 2 * Q_LAYERS);
    }
    // head: one output; thread (row, part) takes its chunks against the weight row (read from the flat parameters: the
    // stored position q of the k-split row is column QCOL(q))
    const int t = threadIdx.x, row = t / PARTS, part = t % PARTS;
    const f32x4 *hr = reinterpret_cast<const f32x4 *>(hb + row * ALD2);
    const float *wh = N.theta + N.o_wh;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < RUN / 4; i++) {
        const f32x4 x = hr[CHUNK(i, part, PARTS)];
#pragma unroll
        for (int u = 0; u < 4; u++) sum += x[u] * wh[QCOL(4 * CHUNK(i, part, PARTS) + u)];
    }
#pragma unroll
    for (int m = 1; m < PARTS; m <<= 1) sum += __shfl_xor(sum, m);
    if (part == 0 && row0 + row < n) N.q[row0 + row] = sum + N.theta[N.o_bh];
}

// mse_loss(q, target) = mean((q - target)^2) (soft_actor_critic.cpp:118-127): gradient 2 (q - target) / rows at the head
__global__ __launch_bounds__(256) void k_q_loss(QDev d, int n, const float *__restrict__ target, float inv_rows) {
    __shared__ float sh[4];
    const size_t row = (size_t) blockIdx.x * 256 + threadIdx.x;
    const QNet &N = d.net[blockIdx.y];
    float lsum = 0.f;
    if (row < (size_t) n) {
        const float diff = N.q[row] - target[row];
        lsum = diff * diff * inv_rows;
        N.dh[row * 32] = 2.f * diff * inv_rows;
    }
    block_accumulate(lsum, d.loss + blockIdx.y, sh);
}

// ---------------------------------------------------------------------------------------------------------
// backward of one 32-row tile through the head and the three blocks
// ---------------------------------------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(PT) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_q_backward(QDev d, int n) {
    constexpr int TM = 32 * RT, PARTS = PT / TM, RUN = 256 / PARTS;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *Tz = sm;
    float *Td = sm + TM * ALD2;
    const QNet &N = d.net[blockIdx.y];
    const int row0 = blockIdx.x * TM;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = threadIdx.x, row = t / PARTS, part = t % PARTS;
    const int gr = row0 + row;
    float colacc[Q_COLSLOTS];
#pragma unroll
    for (int k = 0; k < Q_COLSLOTS; k++) colacc[k] = 0.f;
    tile_load<TM>(Tz, N.z[Q_LAYERS - 1], row0, n);
    // d a3 = dh * w_head (one output)
    const float g = gr < n ? N.dh[(size_t) gr * 32] : 0.f;
    float da[RUN];
    {
        const float *wh = N.theta + N.o_wh;
#pragma unroll
        for (int i = 0; i < RUN; i++) da[i] = g * wh[QCOL(4 * CHUNK(i >> 2, part, PARTS) + (i & 3))];
    }
    {   // head bias gradient: sum of dh over the tile's rows, by thread 0's wave (rows of the tile = 32: one per lane pair)
        float s = (part == 0) ? g : 0.f;
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
        __shared__ float hsum[4];
        if (lane == 0) hsum[wave] = s;
        __syncthreads();
        if (t == 0) colacc[3 * Q_LAYERS] = (hsum[0] + hsum[1]) + (hsum[2] + hsum[3]);
    }
    __syncthreads();
#pragma unroll
    for (int l = Q_LAYERS - 1; l >= 0; l--) {
        const int s0 = 3 * (Q_LAYERS - 1 - l);
        const float mean = gr < n ? N.st[(size_t) gr * 2 * Q_LAYERS + 2 * l] : 0.f;
        const float rstd = gr < n ? N.st[(size_t) gr * 2 * Q_LAYERS + 2 * l + 1] : 0.f;
        // Tz holds z_l, da the gradient w.r.t. the LayerNorm output a_l (for l < last it is already in Td: the dgrad's result)
        if (l == Q_LAYERS - 1) {
#pragma unroll
            for (int i = 0; i < RUN; i++) Td[row * ALD2 + 4 * CHUNK(i >> 2, part, PARTS) + (i & 3)] = da[i];
        }
        ln_mish_backward<RUN, PARTS>(da, Tz, row, part, mean, rstd, N.theta + N.o_g[l]);  // da <- dz_l, Tz <- da * xhat
        __syncthreads();
        colacc[s0] = tile_colsum<TM>(Tz);      // dgamma_l
        colacc[s0 + 1] = tile_colsum<TM>(Td);  // dbeta_l
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RUN; i++) Td[row * ALD2 + 4 * CHUNK(i >> 2, part, PARTS) + (i & 3)] = da[i];
        if (l > 0) tile_load<TM>(Tz, N.z[l - 1], row0, n);
        __syncthreads();
        colacc[s0 + 2] = tile_colsum<TM>(Td);  // dbias_l
        tile_store<TM>(Td, N.dz[l], row0, n);
        if (l > 0) {
            f32x16 acc[RT][2];
            dense_layer<256, RT>(Td, ALD2, N.wd[l], wave, lane, acc);  // d a_{l-1} = dz_l * W_l
            __syncthreads();
            acc_to_tile<RT>(acc, nullptr, Td, wave, lane);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < RUN; i++) da[i] = Td[row * ALD2 + 4 * CHUNK(i >> 2, part, PARTS) + (i & 3)];
        }
    }
    float *cp = N.colpart + (size_t) blockIdx.x * Q_COLSLOTS * 256;
#pragma unroll
    for (int k = 0; k < Q_COLSLOTS - 1; k++) cp[k * 256 + QCOL(t)] = colacc[k];
    cp[(Q_COLSLOTS - 1) * 256 + t] = colacc[Q_COLSLOTS - 1];  // only position 0 is used
}

struct QColSlots { int off[Q_COLSLOTS]; };
__global__ __launch_bounds__(256) void k_q_colfinish(const float *__restrict__ src, int groups, QColSlots cs, float *__restrict__ grad) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= Q_COLSLOTS * 256) return;
    const int slot = e >> 8, c = e & 255;
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= groups; k += 8) {  // eight loads in flight, summed in index order (a handful of workgroups: pure latency otherwise)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = src[(size_t) (k + u) * Q_COLSLOTS * 256 + e];
#pragma unroll
        for (int u = 0; u < 8; u++) s += v[u];
    }
    for (; k < groups; k++) s += src[(size_t) k * Q_COLSLOTS * 256 + e];
    if (slot < Q_COLSLOTS - 1) grad[cs.off[slot] + c] = s;
    else if (c == 0) grad[cs.off[slot]] = s;
}

// torch::optim::Adam (defaults), step count on the device; grid.y = critic
__global__ __launch_bounds__(256) void k_q_adam(QDev d, float lr) {
    const QNet &N = d.net[blockIdx.y];
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= N.n_params) return;
    const int step = N.step[0] + 1;
    const float bc1 = (float) (1.0 - pow(0.9, (double) step));
    const float bc2s = (float) sqrt(1.0 - pow(0.999, (double) step));
    const float g = N.grad[i];
    const float m = N.m[i] + (g - N.m[i]) * 0.1f;
    const float v = N.v[i] * 0.999f + (g * g) * 0.001f;
    N.m[i] = m;
    N.v[i] = v;
    const float th = N.theta[i] - (lr / bc1) * (m / (sqrtf(v) / bc2s + 1e-8f));
    N.theta[i] = th;
    q_pack_write(N, d.S + d.A, i, th);
}
__global__ void k_q_step_inc(QDev d) {
    if (threadIdx.x < 2) d.net[threadIdx.x].step[0] += 1;
}

// to <- tau * from + (1 - tau) * to, each product rounded to fp32 like the reference's tensor expression (functions.cpp:169)
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void k_q_soft(QDev d, float tau, float one_minus_tau) {
    const QNet &T = d.net[2 + blockIdx.y], &F = d.net[blockIdx.y];
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= T.n_params) return;
    const float a = tau * F.theta[i];
    const float b = one_minus_tau * T.theta[i];
    T.theta[i] = a + b;
    q_pack_write(T, d.S + d.A, i, a + b);
}
#pragma clang fp contract(fast)

// ---------------------------------------------------------------------------------------------------------
// SAC actor step (soft_actor_critic.cpp:129-153): reparameterised sample, min of the twin critics, their gradient
// w.r.t. the action, and the gradient of  mean(alpha * sum_d log_pdf - min q)  w.r.t. (mu, sigma)
// ---------------------------------------------------------------------------------------------------------
struct TnTerms {  // truncated normal on [-1, 1] at (mu, sigma) with the uniform draw u (functions.cpp:53-68,94-111)
    float s, is, cs, al, be, ca, cb, pa, pb, Z, w, pw, raw, act, cact;
};
__device__ __forceinline__ TnTerms tn_terms(float mu, float sg, float u) {
    TnTerms t;
    const float INV_SQRT_2PI = 0.39894228040143267794f;
    t.s = fminf(fmaxf(sg, 1e-6f), 1e6f);
    t.cs = (sg >= 1e-6f && sg <= 1e6f) ? 1.f : 0.f;
    t.is = 1.0f / t.s;
    const float ar = (-1.f - mu) / t.s, br = (1.f - mu) / t.s;
    t.al = fminf(fmaxf(ar, -5.f), 5.f);
    t.be = fminf(fmaxf(br, -5.f), 5.f);
    t.ca = (ar >= -5.f && ar <= 5.f) ? 1.f : 0.f;
    t.cb = (br >= -5.f && br <= 5.f) ? 1.f : 0.f;
    const float ta = theta_f(t.al), tb = theta_f(t.be);
    t.Z = tb - ta;
    t.pa = expf(-0.5f * t.al * t.al) * INV_SQRT_2PI;
    t.pb = expf(-0.5f * t.be * t.be) * INV_SQRT_2PI;
    const float cdf = fminf(fmaxf(ta + u * (tb - ta), 0.f), 1.f);
    t.w = 1.41421356237309504880f * erfinvf(2.0f * cdf - 1.0f);  // standard-normal quantile of cdf
    t.pw = expf(-0.5f * t.w * t.w) * INV_SQRT_2PI;
    t.raw = t.w * t.s + mu;
    t.act = fminf(fmaxf(t.raw, -1.f), 1.f);
    t.cact = (t.raw >= -1.f && t.raw <= 1.f) ? 1.f : 0.f;
    return t;
}
// action [rows][A], logp_sum [rows] (one thread per row)
__global__ __launch_bounds__(256) void k_sac_sample(int n, int A, const float *__restrict__ mu, const float *__restrict__ sigma,
                                                    const float *__restrict__ u, float *__restrict__ action, float *__restrict__ logp_sum) {
    const size_t row = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (row >= (size_t) n) return;
    float sum = 0.f;
    for (int o = 0; o < A; o++) {
        const size_t e = row * A + o;
        const TnTerms t = tn_terms(mu[e], sigma[e], u[e]);
        const float q = (t.act - mu[e]) / t.s;
        sum += -0.91893853320467274178f - logf(t.s) - 0.5f * (q * q) - logf(t.Z);
        action[e] = t.act;
    }
    logp_sum[row] = sum;
}
// min(q1, q2) and the head gradients of  -mean(min q): torch.min sends the gradient to the smaller, half each on a tie
__global__ __launch_bounds__(256) void k_q_min_grad(QDev d, int n, float inv_rows, float *__restrict__ qmin) {
    const size_t row = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (row >= (size_t) n) return;
    const float q1 = d.net[0].q[row], q2 = d.net[1].q[row];
    qmin[row] = fminf(q1, q2);
    const float w1 = q1 < q2 ? 1.f : (q1 == q2 ? 0.5f : 0.f);
    d.net[0].dh[row * 32] = -inv_rows * w1;
    d.net[1].dh[row * 32] = -inv_rows * (1.f - w1);
}
// d loss / d action = sum over the critics of dz0 . W0[:, S + a].  A workgroup takes 16 rows: the action columns of both
// first-layer weights are staged in LDS ([critic][a][j]), thread (row, a) walks its row of dz0 against them.
__global__ __launch_bounds__(256) void k_q_input_grad(QDev d, int n, float *__restrict__ dqda) {
    __shared__ float ws[2][16][256];
    const int A = d.A, SA = d.S + d.A;
    for (int e = threadIdx.x; e < 2 * A * 256; e += 256) {
        const int c = e / (A * 256), r = e - c * (A * 256), a = r >> 8, j = r & 255;
        ws[c][a][j] = d.net[c].theta[d.net[c].o_w[0] + (size_t) j * SA + d.S + a];
    }
    __syncthreads();
    const int a = threadIdx.x & 15, lr = threadIdx.x >> 4;
    const size_t row = (size_t) blockIdx.x * 16 + lr;
    if (a >= A || row >= (size_t) n) return;
    float sum = 0.f;
    for (int c = 0; c < 2; c++) {
        const f32x4 *dz = reinterpret_cast<const f32x4 *>(d.net[c].dz[0] + row * 256);
        const f32x4 *w = reinterpret_cast<const f32x4 *>(ws[c][a]);
        float s = 0.f;
#pragma unroll 8
        for (int j = 0; j < 64; j++) {
            const f32x4 x = dz[j], y = w[j];
            s += (x[0] * y[0] + x[1] * y[1]) + (x[2] * y[2] + x[3] * y[3]);
        }
        sum += s;
    }
    dqda[row * A + a] = sum;
}
// d/d(mu, sigma) of  mean_rows(alpha * sum_d logp_d - q),  action = sample(mu, sigma, u) differentiated through
// (dqda already carries the -1/rows of the q term)
__global__ __launch_bounds__(256) void k_sac_actor_grad(int n, int A, const float *__restrict__ mu, const float *__restrict__ sigma,
                                                        const float *__restrict__ u, const float *__restrict__ dqda,
                                                        const float *__restrict__ log_alpha, float inv_rows, float *__restrict__ dmu,
                                                        float *__restrict__ dsigma) {
    const size_t e = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t) n * A) return;
    const float alpha = expf(log_alpha[0]);
    const float m = mu[e];
    const TnTerms t = tn_terms(m, sigma[e], u[e]);
    const float iZ = 1.0f / t.Z;
    const float q = (t.act - m) * t.is;
    // d action / d mu, d sigma (through the clamp to [-1, 1], the quantile w(cdf) and cdf(alpha, beta))
    const float ipw = 1.0f / t.pw;
    const float A1 = (1.f - u[e]) * t.pa * t.ca, B1 = u[e] * t.pb * t.cb;
    const float da_dmu = t.cact * (1.f - (A1 + B1) * ipw);
    const float da_ds = t.cact * (t.w - (A1 * t.al + B1 * t.be) * ipw);
    // log pdf: partials at fixed action, and w.r.t. the action
    const float lp_mu = q * t.is + iZ * t.is * (t.pb * t.cb - t.pa * t.ca);
    const float lp_s = (q * q - 1.f) * t.is + iZ * t.is * (t.pb * t.be * t.cb - t.pa * t.al * t.ca);
    const float lp_a = -q * t.is;
    const float k = alpha * inv_rows;
    const float g_a = k * lp_a + dqda[e];
    dmu[e] = k * lp_mu + g_a * da_dmu;
    dsigma[e] = (k * lp_s + g_a * da_ds) * t.cs;
}

// target_q = r + (1 - done) * gamma * (min(tq1, tq2) - alpha * sum_a next_logp)   (soft_actor_critic.cpp:108-116), each
// operation rounded like the reference's tensor expression
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void k_sac_target(int n, int A, const float *__restrict__ r, const float *__restrict__ done,
                                                   const float *__restrict__ tq1, const float *__restrict__ tq2,
                                                   const float *__restrict__ next_logp, const float *__restrict__ log_alpha, float gamma,
                                                   float *__restrict__ out) {
    const size_t row = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (row >= (size_t) n) return;
    float lp = 0.f;
    for (int a = 0; a < A; a++) lp += next_logp[row * A + a];
    const float alpha = expf(log_alpha[0]);
    const float tv = fminf(tq1[row], tq2[row]) - alpha * lp;
    out[row] = r[row] + ((1.0f - done[row]) * gamma) * tv;
}
#pragma clang fp contract(fast)

// entropy parameter (soft_actor_critic.cpp:155-164): loss = -mean(log_alpha * (logp_sum + target_entropy)), one Adam step on the
// scalar (state: exp_avg, exp_avg_sq, step on the device); also the actor loss value mean(alpha * logp_sum - qmin).  One block.
__global__ __launch_bounds__(1024) void k_sac_entropy(int n, const float *__restrict__ logp_sum, const float *__restrict__ qmin,
                                                      float target_entropy, float lr, float *log_alpha, float *state, int *step,
                                                      float *__restrict__ losses) {
    __shared__ double sh[16];
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) { s1 += (double) logp_sum[i]; s2 += (double) qmin[i]; }
    const double mean_lp = block_sum_double(s1, sh) / n;
    const double mean_q = block_sum_double(s2, sh) / n;
    if (threadIdx.x == 0) {
        const float la = log_alpha[0];
        losses[0] = (float) ((double) expf(la) * mean_lp - mean_q);          // actor loss (with the alpha used in this update)
        losses[1] = (float) (-(double) la * (mean_lp + (double) target_entropy));  // entropy loss
        const float g = (float) (-(mean_lp + (double) target_entropy));
        const int t = step[0] + 1;
        const float bc1 = (float) (1.0 - pow(0.9, (double) t));
        const float bc2s = (float) sqrt(1.0 - pow(0.999, (double) t));
        const float m = state[0] + (g - state[0]) * 0.1f;
        const float v = state[1] * 0.999f + (g * g) * 0.001f;
        state[0] = m;
        state[1] = v;
        step[0] = t;
        log_alpha[0] = la - (lr / bc1) * (m / (sqrtf(v) / bc2s + 1e-8f));
    }
}

// ---------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------
static size_t q_fwd_lds() {
    constexpr int TM = 32 * PRT;
    const size_t a = (size_t) TM * ALD1 + EVM_RED_FLOATS, b = (size_t) TM * ALD2;
    return (a > b ? a : b) * sizeof(float);
}
static size_t q_bwd_lds() {
    constexpr int TM = 32 * PRT;
    return (size_t) 2 * TM * ALD2 * sizeof(float);
}
size_t q_wpart_floats() { return ppo_wpart_floats(); }

// hipFuncSetAttribute is per DEVICE: a process that creates policies / trainers on a second device must set it there too
static bool &evm_attr_done_for_current_device() {
    static bool done[64] = {};
    int dev = 0;
    (void) hipGetDevice(&dev);
    return done[dev >= 0 && dev < 64 ? dev : 0];
}

static hipError_t q_attrs() {
    bool &done = evm_attr_done_for_current_device();
    if (done) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_q_forward<PRT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) q_fwd_lds());
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_q_backward<PRT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) q_bwd_lds());
    done = e == hipSuccess;
    return e;
}

hipError_t launch_q_pack(const QDev &d, int which, hipStream_t s) {
    const QNet &N = d.net[which];
    const size_t total = (size_t) 256 * (d.S + d.A) + (size_t) (Q_LAYERS - 1) * 65536;
    hipLaunchKernelGGL(k_q_pack, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, s, N, d.S + d.A);
    return hipGetLastError();
}
hipError_t launch_q_concat(const QDev &d, size_t rows, const float *states, const float *actions, hipStream_t s) {
    hipLaunchKernelGGL(k_q_concat, dim3((unsigned) ((rows * K1 + 255) / 256)), dim3(256), 0, s, states, actions, d.S, d.A, rows, d.xq);
    return hipGetLastError();
}
hipError_t launch_q_forward(const QDev &d, unsigned nets, size_t rows, int keep, hipStream_t s) {
    hipError_t e = q_attrs();
    if (e != hipSuccess) return e;
    QSel sel;
    sel.count = 0;
    for (int i = 0; i < 4; i++) if (nets & (1u << i)) sel.idx[sel.count++] = i;
    if (!sel.count) return hipSuccess;
    constexpr int TM = 32 * PRT;
    hipLaunchKernelGGL(k_q_forward<PRT>, dim3((unsigned) ((rows + TM - 1) / TM), sel.count), dim3(PT), q_fwd_lds(), s, d, sel, (int) rows, keep);
    return hipGetLastError();
}
hipError_t launch_q_loss(const QDev &d, size_t rows, const float *target_q, hipStream_t s) {
    hipError_t e = hipMemsetAsync(d.loss, 0, 2 * sizeof(double), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_q_loss, dim3((unsigned) ((rows + 255) / 256), 2), dim3(256), 0, s, d, (int) rows, target_q, (float) (1.0 / (double) rows));
    return hipGetLastError();
}
hipError_t launch_q_backward(const QDev &d, size_t rows, hipStream_t s) {
    hipError_t e = q_attrs();
    if (e != hipSuccess) return e;
    constexpr int TM = 32 * PRT;
    hipLaunchKernelGGL(k_q_backward<PRT>, dim3((unsigned) ((rows + TM - 1) / TM), 2), dim3(PT), q_bwd_lds(), s, d, (int) rows);
    return hipGetLastError();
}
hipError_t launch_q_wgrads(const QDev &d, size_t rows, hipStream_t s) {
    constexpr int TM = 32 * PRT;
    const int M = (int) rows, SA = d.S + d.A;
    const int tiles = (int) ((rows + TM - 1) / TM);
    for (int c = 0; c < 2; c++) {
        const QNet &N = d.net[c];
        // layer 0: dz0^T [state, action]; layers 1, 2: dz_l^T a_{l-1}; head: dh^T a_2
        wgrad_one(N.dz[0], 256, 256, true, d.xq, K1, K1, true, M, N.wpart, 256, SA, N.grad + N.o_w[0], SA, 1 << 30, 0, s);
        for (int l = 1; l < Q_LAYERS; l++)
            wgrad_one(N.dz[l], 256, 256, true, N.a[l - 1], 256, 256, true, M, N.wpart, 256, 256, N.grad + N.o_w[l], 256, 1 << 30, 0, s);
        wgrad_heads(N.dh, N.a[Q_LAYERS - 1], N.wpart, M, 1, N.grad + N.o_wh, 1 << 30, 0, s);
        QColSlots cs;
        for (int l = Q_LAYERS - 1; l >= 0; l--) {
            const int s0 = 3 * (Q_LAYERS - 1 - l);
            cs.off[s0] = (int) N.o_g[l]; cs.off[s0 + 1] = (int) N.o_be[l]; cs.off[s0 + 2] = (int) N.o_b[l];
        }
        cs.off[3 * Q_LAYERS] = (int) N.o_bh;
        const int W = Q_COLSLOTS * 256;
        const int groups = launch_colreduce(N.colpart, tiles, W, N.colpart2, s);
        hipLaunchKernelGGL(k_q_colfinish, dim3((W + 255) / 256), dim3(256), 0, s, N.colpart2, groups, cs, N.grad);
    }
    return hipGetLastError();
}
hipError_t launch_q_adam(const QDev &d, float lr, hipStream_t s) {
    hipLaunchKernelGGL(k_q_adam, dim3((unsigned) ((d.net[0].n_params + 255) / 256), 2), dim3(256), 0, s, d, lr);
    hipLaunchKernelGGL(k_q_step_inc, dim3(1), dim3(64), 0, s, d);
    return hipGetLastError();  // k_q_adam repacks what it changes
}
hipError_t launch_q_soft_update(const QDev &d, float tau, hipStream_t s) {
    const float omt = (float) (1.0 - (double) tau);
    hipLaunchKernelGGL(k_q_soft, dim3((unsigned) ((d.net[0].n_params + 255) / 256), 2), dim3(256), 0, s, d, tau, omt);
    return hipGetLastError();  // k_q_soft repacks what it changes
}

hipError_t launch_sac_sample(int rows, int A, const float *mu, const float *sigma, const float *u, float *action, float *logp_sum, hipStream_t s) {
    hipLaunchKernelGGL(k_sac_sample, dim3((rows + 255) / 256), dim3(256), 0, s, rows, A, mu, sigma, u, action, logp_sum);
    return hipGetLastError();
}
hipError_t launch_sac_actor_grad(int rows, int A, const float *mu, const float *sigma, const float *u, const float *dqda,
                                 const float *log_alpha, float *dmu, float *dsigma, hipStream_t s) {
    hipLaunchKernelGGL(k_sac_actor_grad, dim3((unsigned) (((size_t) rows * A + 255) / 256)), dim3(256), 0, s, rows, A, mu, sigma, u, dqda,
                       log_alpha, (float) (1.0 / (double) rows), dmu, dsigma);
    return hipGetLastError();
}
// critics' forward (kept), min, backward to the first layer, gradient w.r.t. the action columns of the input
hipError_t launch_q_action_grad(const QDev &d, size_t rows, float *qmin, float *dqda, hipStream_t s) {
    hipError_t e = launch_q_forward(d, 3u, rows, 1, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_q_min_grad, dim3((unsigned) ((rows + 255) / 256)), dim3(256), 0, s, d, (int) rows, (float) (1.0 / (double) rows), qmin);
    e = launch_q_backward(d, rows, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_q_input_grad, dim3((unsigned) ((rows + 15) / 16)), dim3(256), 0, s, d, (int) rows, dqda);
    return hipGetLastError();
}

hipError_t launch_sac_target(int rows, int A, const float *r, const float *done, const float *tq1, const float *tq2, const float *next_logp,
                             const float *log_alpha, float gamma, float *out, hipStream_t s) {
    hipLaunchKernelGGL(k_sac_target, dim3((rows + 255) / 256), dim3(256), 0, s, rows, A, r, done, tq1, tq2, next_logp, log_alpha, gamma, out);
    return hipGetLastError();
}
hipError_t launch_sac_entropy(int rows, const float *logp_sum, const float *qmin, float target_entropy, float lr, float *log_alpha,
                              float *state, int *step, float *losses, hipStream_t s) {
    hipLaunchKernelGGL(k_sac_entropy, dim3(1), dim3(1024), 0, s, rows, logp_sum, qmin, target_entropy, lr, log_alpha, state, step, losses);
    return hipGetLastError();
}

}  // namespace evm
