// C ABI for the fused actor-critic forward (include/evomotion.h, evm_policy_*).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/evomotion.h"
#include "policy_dev.h"

namespace {
thread_local std::string g_perr;
}
extern "C" const char *evm_last_error(void);
namespace evm { void set_last_error(const std::string &m); }


static int pfail(int code, const std::string &m) { evm::set_last_error(m); return code; }

extern "C" {

int evm_policy_create(int state_dim, int action_dim, int hidden_size, int device, EvmPolicy **out) {
    if (!out) return pfail(EVM_E_INVALID, "out is null");
    *out = nullptr;
    if (hidden_size != 256) return pfail(EVM_E_UNSUPPORTED, "the fused forward is built for hidden_size = 256");
    if (state_dim < 1 || state_dim > 384 || action_dim < 1 || 2 * action_dim > 32) return pfail(EVM_E_INVALID, "unsupported state/action size");
    if (hipSetDevice(device) != hipSuccess) return pfail(EVM_E_HIP, "hipSetDevice failed");
    EvmPolicy *p = new EvmPolicy();
    p->S = state_dim; p->A = action_dim; p->H = hidden_size; p->device = device; p->counter = 0; p->timing = false; p->ev_used = 0; p->tile_rows = 0;
    p->K1pad = 384;  // K1 of policy_kernels.hip
    const size_t per_net = (size_t) p->K1pad * 256 + 3 * 256 + 256 * 256 + 3 * 256;
    const size_t split_floats = ((size_t) p->K1pad + 256) * 256 * 3 / 2;  // three bf16 planes of both hidden layers' weights, as floats
    p->arena_floats = 2 * per_net + ((size_t) 2 * action_dim * 256 + 2 * action_dim) + (256 + 1) + 2 * 8192 + 4 + 2 * split_floats + 8;
    { const char *e = getenv("EVM_POLICY_SPLIT"); p->gemm = e ? (atoi(e) != 0) : 1; }  // EVM_POLICY_SPLIT=0: the fp32 MFMA layers (A/B runs)
    if (hipMalloc((void **) &p->arena, p->arena_floats * 4) != hipSuccess) { delete p; return pfail(EVM_E_HIP, "hipMalloc failed"); }
    if (hipMemset(p->arena, 0, p->arena_floats * 4) != hipSuccess) { hipFree(p->arena); delete p; return pfail(EVM_E_HIP, "hipMemset failed"); }
    float *b = p->arena;
    auto carve = [&](evm::NetDev &n, size_t head_w, size_t head_b) {
        n.w1t = b; b += (size_t) p->K1pad * 256;
        n.b1 = b; b += 256; n.g1 = b; b += 256; n.be1 = b; b += 256;
        n.w2t = b; b += 256 * 256;
        n.b2 = b; b += 256; n.g2 = b; b += 256; n.be2 = b; b += 256;
        n.wh = b; b += head_w; n.bh = b; b += head_b;
    };
    auto carve_packed = [&](evm::NetDev &n) {  // 16-byte aligned: behind everything else
        n.whp = b; b += 8192;
        n.w1s = reinterpret_cast<const uint16_t *>(b); b += (size_t) p->K1pad * 256 * 3 / 2;
        n.w2s = reinterpret_cast<const uint16_t *>(b); b += (size_t) 256 * 256 * 3 / 2;
    };
    carve(p->dev.actor, (size_t) 2 * action_dim * 256, 2 * action_dim);
    carve(p->dev.critic, 256, 1);
    b = p->arena + ((b - p->arena + 3) / 4) * 4;
    carve_packed(p->dev.actor);
    carve_packed(p->dev.critic);
    p->dev.S = state_dim; p->dev.A = action_dim; p->dev.K1pad = p->K1pad;
    *out = p;
    return EVM_OK;
}

void evm_policy_destroy(EvmPolicy *p) {
    if (!p) return;
    if (p->arena) (void) hipFree(p->arena);
    for (auto &pr : p->ev_pairs) { (void) hipEventDestroy(pr.first); (void) hipEventDestroy(pr.second); }
    delete p;
}

int evm_policy_param_counts(const EvmPolicy *p, size_t *n_actor, size_t *n_critic) {
    if (!p) return pfail(EVM_E_INVALID, "policy is null");
    const size_t trunk = (size_t) 256 * p->S + 256 + 256 + 256 + 256 * 256 + 256 + 256 + 256;
    if (n_actor) *n_actor = trunk + 2 * ((size_t) p->A * 256 + p->A);
    if (n_critic) *n_critic = trunk + 256 + 1;
    return EVM_OK;
}

// h_actor / h_critic: flat fp32 parameters in the reference's named_parameters() order
//   actor : head.0.{weight[256,S],bias} head.2.{weight,bias} head.3.{weight[256,256],bias} head.5.{weight,bias}
//           mu.0.{weight[A,256],bias} sigma.0.{weight[A,256],bias}      (actor.cpp:9-28)
//   critic: critic.0 .2 .3 .5 as above, critic.6.{weight[1,256],bias}   (critic.cpp:8-21)
int evm_policy_set_weights(EvmPolicy *p, const float *h_actor, size_t n_actor, const float *h_critic, size_t n_critic) {
    if (!p || !h_actor || !h_critic) return pfail(EVM_E_INVALID, "null argument");
    size_t ea, ec;
    evm_policy_param_counts(p, &ea, &ec);
    if (n_actor != ea || n_critic != ec) return pfail(EVM_E_INVALID, "parameter count mismatch");
    std::vector<float> host(p->arena_floats, 0.f);
    auto pack = [&](const evm::NetDev &n, const float *src, int heads, bool actor) {
        const int S = p->S, A = p->A;
        // packed for the MFMA B operand: Wp[s4][col][h][t] = W[col][k = 2 (4 s4 + t) + h]   (policy_kernels.hip)
        float *w1t = host.data() + (n.w1t - p->arena);
        for (int o = 0; o < 256; o++)
            for (int k = 0; k < S; k++) {
                const int st = k >> 1, h = k & 1, s4 = st >> 2, t = st & 3;
                w1t[(((size_t) s4 * 256 + o) * 2 + h) * 4 + t] = src[(size_t) o * S + k];
                uint16_t *ws = reinterpret_cast<uint16_t *>(host.data() + (reinterpret_cast<const float *>(n.w1s) - p->arena)) + evm::split_index(o, k);
                evm::bf16_split3(src[(size_t) o * S + k], ws[0], ws[8], ws[16]);
            }
        src += (size_t) 256 * S;
        memcpy(host.data() + (n.b1 - p->arena), src, 256 * 4); src += 256;
        memcpy(host.data() + (n.g1 - p->arena), src, 256 * 4); src += 256;
        memcpy(host.data() + (n.be1 - p->arena), src, 256 * 4); src += 256;
        float *w2t = host.data() + (n.w2t - p->arena);
        for (int o = 0; o < 256; o++)
            for (int k = 0; k < 256; k++) {
                const int st = k >> 1, h = k & 1, s4 = st >> 2, t = st & 3;
                w2t[(((size_t) s4 * 256 + o) * 2 + h) * 4 + t] = src[(size_t) o * 256 + k];
                uint16_t *ws = reinterpret_cast<uint16_t *>(host.data() + (reinterpret_cast<const float *>(n.w2s) - p->arena)) + evm::split_index(o, k);
                evm::bf16_split3(src[(size_t) o * 256 + k], ws[0], ws[8], ws[16]);
            }
        src += 256 * 256;
        memcpy(host.data() + (n.b2 - p->arena), src, 256 * 4); src += 256;
        memcpy(host.data() + (n.g2 - p->arena), src, 256 * 4); src += 256;
        memcpy(host.data() + (n.be2 - p->arena), src, 256 * 4); src += 256;
        float *wh = host.data() + (n.wh - p->arena), *bh = host.data() + (n.bh - p->arena);
        if (actor) {
            memcpy(wh, src, (size_t) A * 256 * 4); src += (size_t) A * 256;                 // mu.0.weight
            memcpy(bh, src, A * 4); src += A;                                                // mu.0.bias
            memcpy(wh + (size_t) A * 256, src, (size_t) A * 256 * 4); src += (size_t) A * 256;  // sigma.0.weight
            memcpy(bh + A, src, A * 4); src += A;
        } else {
            memcpy(wh, src, 256 * 4); src += 256;
            bh[0] = src[0];
        }
        // heads once more as the MFMA B operand of the head GEMM: 32 columns = head rows (zero beyond), k-split like w2t
        float *whp = host.data() + (n.whp - p->arena);
        for (int o = 0; o < heads; o++)
            for (int k = 0; k < 256; k++) {
                const int st = k >> 1, h = k & 1, s4 = st >> 2, t = st & 3;
                whp[(((size_t) s4 * 32 + o) * 2 + h) * 4 + t] = wh[(size_t) o * 256 + k];
            }
    };
    pack(p->dev.actor, h_actor, 2 * p->A, true);
    pack(p->dev.critic, h_critic, 1, false);
    if (hipMemcpy(p->arena, host.data(), p->arena_floats * 4, hipMemcpyHostToDevice) != hipSuccess)
        return pfail(EVM_E_HIP, "weight upload failed");
    return EVM_OK;
}

int evm_policy_set_weights_device(EvmPolicy *p, const float *d_actor, const float *d_critic, void *stream) {
    if (!p || (!d_actor && !d_critic)) return pfail(EVM_E_INVALID, "null argument");
    hipStream_t s = (hipStream_t) stream;
    hipError_t e = hipSuccess;
    if (d_actor) e = evm::launch_policy_pack(p->dev.actor, p->S, p->A, true, d_actor, s);
    if (e == hipSuccess && d_critic) e = evm::launch_policy_pack(p->dev.critic, p->S, p->A, false, d_critic, s);
    if (e != hipSuccess) return pfail(EVM_E_HIP, std::string("weight repack: ") + hipGetErrorString(e));
    return EVM_OK;
}

int evm_policy_forward(EvmPolicy *p, int n, const float *d_obs, const float *d_uniform, uint64_t seed, float *d_action,
                       float *d_logp, float *d_value, float *d_mu, float *d_sigma, void *stream) {
    if (!p || !d_obs || !d_action || !d_logp) return pfail(EVM_E_INVALID, "null argument");  // d_value may be NULL: actor only
    if (n < 1) return pfail(EVM_E_INVALID, "n must be >= 1");
    hipStream_t s = (hipStream_t) stream;
    if (p->timing) {
        if (p->ev_used == p->ev_pairs.size()) {
            hipEvent_t a0, a1;
            if (hipEventCreate(&a0) != hipSuccess || hipEventCreate(&a1) != hipSuccess) return pfail(EVM_E_HIP, "hipEventCreate failed");
            p->ev_pairs.push_back({a0, a1});
        }
        (void) hipEventRecord(p->ev_pairs[p->ev_used].first, s);
    }
    hipError_t e = evm::launch_policy_forward(p->dev, n, d_obs, d_uniform, seed, p->counter++, d_action, d_logp, d_value,
                                              d_mu, d_sigma, s, p->tile_rows, p->gemm);
    if (e != hipSuccess) return pfail(EVM_E_HIP, std::string("policy forward: ") + hipGetErrorString(e));
    if (p->timing) { (void) hipEventRecord(p->ev_pairs[p->ev_used].second, s); p->ev_used++; }
    return EVM_OK;
}

int evm_policy_set_tile_rows(EvmPolicy *p, int rows) {
    if (!p) return pfail(EVM_E_INVALID, "policy is null");
    if (rows != 0 && rows != 16 && rows != 32) return pfail(EVM_E_INVALID, "tile rows: 0 (automatic), 16 or 32");
    p->tile_rows = rows;
    return EVM_OK;
}

int evm_policy_timing_begin(EvmPolicy *p) {
    if (!p) return pfail(EVM_E_INVALID, "policy is null");
    p->timing = true;
    p->ev_used = 0;
    return EVM_OK;
}
int evm_policy_timing_end(EvmPolicy *p, void *stream, float *ms_total, int *n_launches) {
    if (!p) return pfail(EVM_E_INVALID, "policy is null");
    if (hipStreamSynchronize((hipStream_t) stream) != hipSuccess) return pfail(EVM_E_HIP, "stream sync failed");
    float ms = 0.f;
    for (size_t i = 0; i < p->ev_used; i++) {
        float t = 0.f;
        (void) hipEventElapsedTime(&t, p->ev_pairs[i].first, p->ev_pairs[i].second);
        ms += t;
    }
    p->timing = false;
    if (ms_total) *ms_total = ms;
    if (n_launches) *n_launches = (int) p->ev_used;
    return EVM_OK;
}

}  // extern "C"
