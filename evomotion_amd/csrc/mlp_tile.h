// Shared device pieces of the actor / critic MLP kernels (rollout forward: policy_kernels.hip; PPO update:
// ppo_kernels.hip): fp32 MFMA dense layer over a k-split LDS tile, Mish, the operand layouts.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace evm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define PT 256        // threads per workgroup (4 waves)
#define K1 384        // padded input width of layer 1 (371 -> 384), fixed by the host packer
#define ALD1 (K1 + 4) // k-split observation tile row stride
#define ALD2 (256 + 4) // k-split activation tile row stride
#define HIDX(row, c) ((row) * ALD2 + ((c) & 1) * 128 + ((c) >> 1))
#define QCOL(q) (2 * ((q) & 127) + ((q) >> 7))  // column held at stored position q of a k-split activation row
// Row phases (LayerNorm, heads): thread (row, part) owns the 16-byte chunks i * PARTS + part (i = 0 .. RUN / 4 - 1) of its
// k-split row, so the PARTS threads of a row read consecutive chunks — a contiguous run per thread would put all of them
// on the same LDS banks (row stride 260 floats, run 32 floats: an 8-way conflict)
#define CHUNK(i, part, PARTS_) ((i) * (PARTS_) + (part))

// Mish(x) = x tanh(softplus(x)).  With n = e^x: tanh(log(1 + n)) = n (n + 2) / (n (n + 2) + 2), all terms positive
// (no cancellation), one exp and one reciprocal instead of exp + log1p + tanh (which cost ~300 VALU instructions per
// value and were 40 % of the kernel).  x > 20: the ratio is 1 to fp32 precision and e^x would overflow at 88.
// The reciprocal is v_rcp_f32 (1 ulp; m + 2 is in [2, 2.4e17], no denormals): __fdividef expands to the full
// div_scale / div_fmas / div_fixup sequence with this compiler, 10 instructions per value.
__device__ __forceinline__ float mish_f(float x) {
    const float n = __expf(fminf(x, 20.f));
    const float m = n * (n + 2.f);
    return x * (m * __builtin_amdgcn_rcpf(m + 2.f));
}
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float theta_f(float x) { return 0.5f * (1.0f + erff(x / 1.41421356237309504880f)); }

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Operand packing for v_mfma_f32_32x32x2_f32: lane l = (j = l & 31, h = l >> 5) needs A[row j][k = 2s + h] and
// B[k = 2s + h][col j] for k-step s.  Both operands are stored "k-split" so that FOUR consecutive k-steps of a
// lane are 16 contiguous bytes:
//   weights (global, packed on the host):  Wp[s4][col 0..255][h 0..1][t 0..3] = W[col][k = 2 (4 s4 + t) + h]
//       -> one global_load_dwordx4 per lane per 4 k-steps, 1 KiB contiguous per wave instruction
//   activations / observations (LDS):      As[row][h][kk] with k = 2 kk + h, row stride ALD floats
//       -> one ds_read_b128 per lane per 4 k-steps
// ALD = 2 * KH + 4 with KH = K/2 (a multiple of 4): the +4 skews rows by one 16-byte slot so that the 16-lane groups
// of ds_read_b128 hit distinct slots.
template <int K, int RT>
__device__ __forceinline__ void dense_layer(const float *as, int ald, const float *__restrict__ Wp, int wave, int lane,
                                            f32x16 (&acc)[RT][2]) {
#pragma unroll
    for (int i = 0; i < RT; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const int aj = lane & 31, ah = lane >> 5;
    const int col = wave * 64 + aj;
    constexpr int KH = K / 2;
    const float *ap = as + aj * ald + ah * KH;
    const float *b0p = Wp + ((size_t) col * 2 + ah) * 4;
    const float *b1p = Wp + ((size_t) (col + 32) * 2 + ah) * 4;
    // Software pipeline, DEPTH-deep register ring: the operands of block s + DEPTH - 1 are requested before the 8 RT MFMAs of
    // block s issue, i.e. (DEPTH - 1) x ~512 cycles ahead — more than an L2 hit under load.  (Left to itself the compiler
    // issued each block's loads right before its own MFMAs and waited on them: the kernel ran at half speed.)
    constexpr int NB = K / 8;
#ifndef EVM_RING
#define EVM_RING 4
#endif
    constexpr int DEPTH = EVM_RING;
    static_assert(NB % DEPTH == 0, "K / 8 must be a multiple of the ring depth");
    f32x4 a[DEPTH][RT], b0[DEPTH], b1[DEPTH];
#define EVM_LOADQ(q, s)                                                                                   \
    {                                                                                                     \
        _Pragma("unroll") for (int i = 0; i < RT; i++) a[q][i] =                                        \
            *reinterpret_cast<const f32x4 *>(ap + i * 32 * ald + 4 * (s));                                \
        b0[q] = *reinterpret_cast<const f32x4 *>(b0p + (size_t) (s) * 2048);                              \
        b1[q] = *reinterpret_cast<const f32x4 *>(b1p + (size_t) (s) * 2048);                              \
    }
#pragma unroll
    for (int q = 0; q < DEPTH - 1; q++) EVM_LOADQ(q, q)
    __builtin_amdgcn_sched_barrier(0);
    for (int s4 = 0; s4 < NB; s4 += DEPTH) {
#pragma unroll
        for (int q = 0; q < DEPTH; q++) {
            const int sn = min(s4 + q + DEPTH - 1, NB - 1);  // the last requests re-read the final block (harmless)
            EVM_LOADQ((q + DEPTH - 1) % DEPTH, sn)
            __builtin_amdgcn_sched_barrier(0);  // keep the requests ahead of this block's MFMAs
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int i = 0; i < RT; i++) {
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][i][t], b0[q][t], acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][i][t], b1[q][t], acc[i][1], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef EVM_LOADQ
}


// ---- the same dense layer off the fp32 MFMA --------------------------------------------------------------------------------------
// On gfx950 v_mfma_f32_32x32x2_f32 runs at the packed-fp32 VALU rate and shares the SIMD's time with the VALU
// (tools/coexec_pad_probe.hip); v_mfma_f32_32x32x16_bf16 does 16 times the FLOP per clock.  An fp32 value is EXACTLY the sum of three
// bf16 values (its 24-bit significand cut into 8-bit pieces by truncation, x = a0 + a1 + a2), and a product of two bf16 values is exact
// in the fp32 accumulator, so
//     a . b = a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0) + [a1 b2 + a2 b1 + a2 b2 <= 3 x 2^-24 |a b|: dropped]
// six bf16 MFMAs per 16 k-steps instead of eight fp32 ones per 16, at a quarter of the clocks each: 6/16 of the pipe time, with an error
// at least as small as the fp32 chain's (tools/split_gemm_probe.hip: 1.3e-7 against 3.6e-7 of sum |a b|).
// The weights are split once by the packers (NetDev::w1s / w2s); the activations stay fp32 in the k-split LDS tile and are split here,
// per lane and block, between the loads and the MFMAs (~50 VALU operations against the ten fp32 MFMAs = 640 clocks they save).
// Lane l = (row or column l & 31, k group g = l >> 5) supplies k = 16 b + 2 i + g, i = 0 .. 7: eight consecutive floats of half g
// of the k-split tile; C/D layout as v_mfma_f32_32x32x2_f32.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
struct SplitA { bf16x8 p0, p1, p2; };
__device__ __forceinline__ SplitA split8(const f32x4 &x0, const f32x4 &x1) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const float x = i < 4 ? x0[i] : x1[i - 4];
        h[i] = __float_as_uint(x) & 0xffff0000u;
        const float r1 = x - __uint_as_float(h[i]);
        m[i] = __float_as_uint(r1) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(m[i]);
        l[i] = __float_as_uint(r2) & 0xffff0000u;
    }
    u32x4_ a, b, c;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        a[i] = (h[2 * i] >> 16) | h[2 * i + 1];
        b[i] = (m[2 * i] >> 16) | m[2 * i + 1];
        c[i] = (l[2 * i] >> 16) | l[2 * i + 1];
    }
    SplitA s;
    s.p0 = __builtin_bit_cast(bf16x8, a); s.p1 = __builtin_bit_cast(bf16x8, b); s.p2 = __builtin_bit_cast(bf16x8, c);
    return s;
}
template <int K>
__device__ __forceinline__ void dense_layer_split(const float *as, int ald, const uint16_t *__restrict__ Ws, int wave, int lane,
                                                  f32x16 (&acc)[1][2]) {
    f32x16 c0[2], c1[2];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) { c0[j][r] = 0.f; c1[j][r] = 0.f; }
    const int rc = lane & 31, g = lane >> 5;
    constexpr int KH = K / 2;
    const float *ap = as + rc * ald + g * KH;
    // Ws[b][col][g][plane][8]: uint16 offset (((b * 256 + col) * 2 + g) * 3) * 8; as 16-byte units: ((b * 256 + col) * 2 + g) * 3
    const u32x4_ *bp0 = reinterpret_cast<const u32x4_ *>(Ws) + ((size_t) (wave * 64 + rc) * 2 + g) * 3;
    const u32x4_ *bp1 = bp0 + (size_t) 32 * 2 * 3;
    constexpr int NB = K / 16, DEPTH = 4, BSTEP = 256 * 2 * 3;  // 16-byte units per block
    static_assert(NB % DEPTH == 0, "K / 16 must be a multiple of the ring depth");
    f32x4 a0[DEPTH], a1[DEPTH];
    u32x4_ w[DEPTH][2][3];
#define EVM_LOADS(q, s)                                                                    \
    {                                                                                      \
        a0[q] = *reinterpret_cast<const f32x4 *>(ap + 8 * (s));                            \
        a1[q] = *reinterpret_cast<const f32x4 *>(ap + 8 * (s) + 4);                        \
        _Pragma("unroll") for (int p = 0; p < 3; p++) {                                  \
            w[q][0][p] = bp0[(size_t) (s) * BSTEP + p];                                    \
            w[q][1][p] = bp1[(size_t) (s) * BSTEP + p];                                    \
        }                                                                                  \
    }
#pragma unroll
    for (int q = 0; q < DEPTH - 1; q++) EVM_LOADS(q, q)
    __builtin_amdgcn_sched_barrier(0);
    for (int s0 = 0; s0 < NB; s0 += DEPTH) {
#pragma unroll
        for (int q = 0; q < DEPTH; q++) {
            const int sn = min(s0 + q + DEPTH - 1, NB - 1);
            EVM_LOADS((q + DEPTH - 1) % DEPTH, sn)
            __builtin_amdgcn_sched_barrier(0);
            const SplitA a = split8(a0[q], a1[q]);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const bf16x8 b0 = __builtin_bit_cast(bf16x8, w[q][j][0]), b1 = __builtin_bit_cast(bf16x8, w[q][j][1]),
                             b2 = __builtin_bit_cast(bf16x8, w[q][j][2]);
                c1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p0, b2, c1[j], 0, 0, 0);
                c1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p1, b1, c1[j], 0, 0, 0);
                c1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p2, b0, c1[j], 0, 0, 0);
                c1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p0, b1, c1[j], 0, 0, 0);
                c1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p1, b0, c1[j], 0, 0, 0);
                c0[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p0, b0, c0[j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef EVM_LOADS
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[0][j][r] = c0[j][r] + c1[j][r];
}

// Workgroup barrier for LDS hand-overs only: waits for this wave's LDS operations, NOT for its global stores.
// __syncthreads() carries a workgroup fence = s_waitcnt vmcnt(0): with activations streaming to HBM from the epilogues every
// barrier then stalls until the wave's stores are acknowledged (~20 k cycles per layer in the training forward).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Global accesses as (uniform base) + (32-bit byte offset): the form the compiler turns into `global_load/store v, v_off, s[base]`
// (+ an immediate for small constant steps).  Indexing with size_t, or with an unsigned ELEMENT index (whose * 4 may wrap), makes a
// 64-bit address pair per access instead: the 64 row addresses of an accumulator-layout tile then cost 128 registers.
__device__ __forceinline__ float ldg_off(const float *base, unsigned byte_off) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ void stg_off(float *base, unsigned byte_off, float v) {
    *reinterpret_cast<float *>(reinterpret_cast<char *>(base) + byte_off) = v;
}

// Sum over the 32 lanes of each half wave, result in every lane: four DPP adds (quad swaps, half-row and row mirrors: no LDS,
// no wait counters) and one ds_swizzle exchange between the two 16-lane rows of the half.
__device__ __forceinline__ float half_wave_sum(float v) {
#define EVM_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false));
    EVM_DPP_ADD(0xB1)   // quad_perm [1,0,3,2]
    EVM_DPP_ADD(0x4E)   // quad_perm [2,3,0,1]
    EVM_DPP_ADD(0x141)  // row_half_mirror
    EVM_DPP_ADD(0x140)  // row_mirror
#undef EVM_DPP_ADD
    return v + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));  // lane ^ 16
}

// Bias + Mish + LayerNorm(256) of a 32-row tile straight from the MFMA accumulators of one dense layer (acc[j]: columns
// wave * 64 + 32 j + (lane & 31), C layout: row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)).  Nothing is transposed for the
// row statistics: each wave reduces its 64 columns inside the 32-lane halves (mean, then the squared deviations from that
// mean), the four waves' (mean, M2) pairs meet in LDS and are combined as M2 = sum M2_w + 64 sum (mean_w - mean)^2, which is
// exact and as well conditioned as the two-pass form; lane (lane & 31) of every wave combines row (lane & 31) — one division
// and square root per lane — and hands (mean, rstd) to the lanes that hold the row through a wave-private LDS line.
// The normalised activations go to the k-split tile hb for the next layer and, for training, to HBM together with the
// pre-activations z and (mean, rstd); half a wave stores 32 consecutive columns of one row = 128 contiguous bytes.
// One barrier inside, one at the end.  The first barrier also covers "every wave has finished reading the tile hb overlays":
// callers need no barrier between the dense layer and this.  red: EVM_RED_FLOATS of LDS outside every tile
// ([32 rows][4 waves][2] partials, then [4 waves][32 rows][2] results).
// The row-wise form this replaces (accumulators -> tile -> (row, part) threads -> tile) cost three barriers and an LDS round
// trip per layer: ~20 k cycles of latency per layer in the training forward (tools/fstamps.py).
__device__ __forceinline__ void mish_ln_epilogue(f32x16 (&acc)[2], const float *__restrict__ bias, const float *__restrict__ gamma,
                                                 const float *__restrict__ beta, float *hb, float *red, int wave, int lane,
                                                 int row0, int n, float *__restrict__ zg, float *__restrict__ ag,
                                                 float *__restrict__ st, int st_off, int st_stride) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    const int cl = lane & 31, hf = lane >> 5;
    const int c0 = wave * 64 + cl, c1 = c0 + 32;
    const float b0 = bias[c0], b1 = bias[c1];
    float g0 = gamma[c0], g1 = gamma[c1], e0 = beta[c0], e1 = beta[c1];
    // have the six parameters in registers before the first store is issued: vector memory operations complete in order, so
    // a load scheduled behind the z stores (where the compiler sinks it, next to its use) waits for all of them
    asm volatile("" : "+v"(g0), "+v"(g1), "+v"(e0), "+v"(e1));
    float s[16], q[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hf;
        const float z0 = acc[0][r] + b0, z1 = acc[1][r] + b1;
        if (zg && row0 + row < n) {
            zg[(size_t) (row0 + row) * 256 + c0] = z0;
            zg[(size_t) (row0 + row) * 256 + c1] = z1;
        }
        acc[0][r] = mish_f(z0);
        acc[1][r] = mish_f(z1);
        s[r] = half_wave_sum(acc[0][r] + acc[1][r]) * (1.0f / 64.0f);
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const float d0 = acc[0][r] - s[r], d1 = acc[1][r] - s[r];
        q[r] = half_wave_sum(d0 * d0 + d1 * d1);
    }
    if (cl == 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hf;
            *reinterpret_cast<f32x2_ *>(red + (row * 4 + wave) * 2) = f32x2_{s[r], q[r]};
        }
    }
    lds_barrier();
    float *mine = red + 256 + wave * 64;  // this wave's [32 rows][mean, rstd]
    {
        const f32x4 p0 = *reinterpret_cast<const f32x4 *>(red + cl * 8), p1 = *reinterpret_cast<const f32x4 *>(red + cl * 8 + 4);
        const float mean = ((p0[0] + p0[2]) + (p1[0] + p1[2])) * 0.25f;
        const float da = p0[0] - mean, db = p0[2] - mean, dc = p1[0] - mean, dd = p1[2] - mean;
        const float m2 = ((p0[1] + p0[3]) + (p1[1] + p1[3])) + 64.0f * ((da * da + db * db) + (dc * dc + dd * dd));
        const float rstd = 1.0f / sqrtf(m2 * (1.0f / 256.0f) + 1e-5f);
        if (hf == 0) *reinterpret_cast<f32x2_ *>(mine + cl * 2) = f32x2_{mean, rstd};
        if (st && wave == 0 && hf == 0 && row0 + cl < n) {
            st[(size_t) (row0 + cl) * st_stride + st_off] = mean;
            st[(size_t) (row0 + cl) * st_stride + st_off + 1] = rstd;
        }
    }
    // (LDS operations of one wave complete in order: the reads below see this wave's line without a barrier)
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const f32x4 ma = *reinterpret_cast<const f32x4 *>(mine + (8 * g + 4 * hf) * 2), mb = *reinterpret_cast<const f32x4 *>(mine + (8 * g + 4 * hf) * 2 + 4);
        const float mean[4] = {ma[0], ma[2], mb[0], mb[2]}, rstd[4] = {ma[1], ma[3], mb[1], mb[3]};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int r = 4 * g + u, row = 8 * g + 4 * hf + u;
            const float y0 = (acc[0][r] - mean[u]) * rstd[u] * g0 + e0, y1 = (acc[1][r] - mean[u]) * rstd[u] * g1 + e1;
            hb[HIDX(row, c0)] = y0;
            hb[HIDX(row, c1)] = y1;
            if (ag && row0 + row < n) {
                ag[(size_t) (row0 + row) * 256 + c0] = y0;
                ag[(size_t) (row0 + row) * 256 + c1] = y1;
            }
        }
    }
    lds_barrier();
}
#define EVM_RED_FLOATS 512  // LDS floats of the row-statistics exchange: [32 rows][4 waves][2] partials + [4 waves][32 rows][2] results

// The head GEMM [32 x 256] x [256 x 32] on the matrix pipe (as a row-wise dot product loop it was a quarter of the forward
// kernel): K is split over the four waves, wave w leaves its partial 32 x 32 tile in hs4[w][row][col]; the caller
// synchronises and sums the four.  hb: k-split activation tile (row stride ALD2); whp: [32 s4][32 cols][2][4].
__device__ __forceinline__ void head_gemm(const float *hb, const float *__restrict__ whp, float *hs4, int wave, int lane) {
    f32x16 hacc;
#pragma unroll
    for (int r = 0; r < 16; r++) hacc[r] = 0.f;
    const int aj = lane & 31, ah = lane >> 5;
    const float *ap = hb + aj * ALD2 + ah * 128 + wave * 32;  // this wave's k range: 64 wave .. 64 wave + 63
    const float *bp = whp + ((size_t) (8 * wave) * 32 + aj) * 8 + ah * 4;
    f32x4 a4[8], b4[8];
#pragma unroll
    for (int b = 0; b < 8; b++) {
        a4[b] = *reinterpret_cast<const f32x4 *>(ap + 4 * b);
        b4[b] = *reinterpret_cast<const f32x4 *>(bp + (size_t) b * 256);
    }
#pragma unroll
    for (int b = 0; b < 8; b++)
#pragma unroll
        for (int tt = 0; tt < 4; tt++) hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[b][tt], b4[b][tt], hacc, 0, 0, 0);
    // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; r++) hs4[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = hacc[r];
}

// ---- 16-row tiles on v_mfma_f32_16x16x4_f32 (rollout forward when 32-row tiles would leave CUs without a workgroup) ----
// Lane l = (c = l & 15, kq = l >> 4) supplies A[row c][k] and B[k][col c] of one k per step.  The k-split operand layouts
// above serve unchanged: a lane's 16-byte slot (s4 = 2 j + (kq >> 1), h = kq & 1) holds k = 8 s4 + h + 2 t for t = 0 .. 3,
// so ONE ds_read_b128 / global_load_dwordx4 per lane feeds four MFMA steps, and step t of block pair j contracts
// k in {16 j + 2 t, + 1, + 8, + 9}: every k once, in an order of its own (fp32 sums agree with the 32-row form to rounding,
// not bit for bit).  C/D layout: col = l & 15, row = 4 (l >> 4) + reg.
typedef float f32x4c __attribute__((ext_vector_type(4)));

// Wave `wave` of four owns 16 rows x 64 columns = 4 accumulator tiles (columns wave * 64 + 16 j + c).
// The weight ring is the caller's: dense16_prefetch requests the first DEPTH16 - 1 block pairs (they depend on nothing the
// workgroup computes), so a caller issues it BEFORE the phase that produces the A tile (staging, the previous layer's
// epilogue) and the layer starts on operands that are already there instead of on a cold L2 round trip.
#define DEPTH16 4
__device__ __forceinline__ const float *dense16_bptr(const float *__restrict__ Wp, int wave, int lane) {
    const int c = lane & 15, kq = lane >> 4;
    return Wp + ((size_t) (kq >> 1) * 256 + wave * 64 + c) * 8 + (kq & 1) * 4;  // Wp[s4][col][h][t]: ((s4 * 256 + col) * 2 + h) * 4
}
__device__ __forceinline__ void dense16_prefetch(const float *__restrict__ Wp, int wave, int lane, f32x4 (&b)[DEPTH16][4]) {
    const float *bp = dense16_bptr(Wp, wave, lane);
#pragma unroll
    for (int q = 0; q < DEPTH16 - 1; q++)
#pragma unroll
        for (int j = 0; j < 4; j++) b[q][j] = *reinterpret_cast<const f32x4 *>(bp + (size_t) q * 4096 + j * 128);
    __builtin_amdgcn_sched_barrier(0);
}
template <int K>
__device__ __forceinline__ void dense_layer16(const float *as, int ald, const float *__restrict__ Wp, int wave, int lane,
                                              f32x4c (&acc)[4], f32x4 (&b)[DEPTH16][4]) {
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc[j][r] = 0.f;
    const int c = lane & 15, kq = lane >> 4;
    constexpr int KH = K / 2;
    const float *ap = as + c * ald + (kq & 1) * KH + 4 * (kq >> 1);
    const float *bp = dense16_bptr(Wp, wave, lane);
    constexpr int NB = K / 16;  // block pairs
    constexpr int DEPTH = DEPTH16;
    static_assert(NB % DEPTH == 0, "K / 16 must be a multiple of the ring depth");
    f32x4 a[DEPTH];
#pragma unroll
    for (int q = 0; q < DEPTH - 1; q++) a[q] = *reinterpret_cast<const f32x4 *>(ap + 8 * q);
    __builtin_amdgcn_sched_barrier(0);
    for (int s0 = 0; s0 < NB; s0 += DEPTH) {
#pragma unroll
        for (int q = 0; q < DEPTH; q++) {
            const int sn = min(s0 + q + DEPTH - 1, NB - 1);
            const int qn = (q + DEPTH - 1) % DEPTH;
            a[qn] = *reinterpret_cast<const f32x4 *>(ap + 8 * sn);
#pragma unroll
            for (int j = 0; j < 4; j++) b[qn][j] = *reinterpret_cast<const f32x4 *>(bp + (size_t) sn * 4096 + j * 128);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][t], b[q][j][t], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// dense_layer16 off the fp32 MFMA (see dense_layer_split): v_mfma_f32_16x16x32_bf16, lane = (row or column c = l & 15, k group
// kg = l >> 4: eight k).  The weight planes of NetDev::w1s / w2s serve as they are: 32-k block bb = their 16-k blocks 2 bb + (kg >> 1),
// group kg & 1; the lane's eight A values are eight consecutive floats of half (kg & 1) of the k-split tile, from 16 bb + 8 (kg >> 1).
template <int K>
__device__ __forceinline__ void dense_layer16_split(const float *as, int ald, const uint16_t *__restrict__ Ws, int wave, int lane,
                                                    f32x4c (&acc)[4]) {
    f32x4c c0[4], c1[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) { c0[j][r] = 0.f; c1[j][r] = 0.f; }
    const int c = lane & 15, kg = lane >> 4;
    constexpr int KH = K / 2;
    const float *ap = as + c * ald + (kg & 1) * KH + 8 * (kg >> 1);
    // 16-byte units: ((b16 * 256 + col) * 2 + g) * 3 + plane, b16 = 2 bb + (kg >> 1)
    const u32x4_ *bp = reinterpret_cast<const u32x4_ *>(Ws) + (((size_t) (kg >> 1) * 256 + wave * 64 + c) * 2 + (kg & 1)) * 3;
    constexpr int NB = K / 32, DEPTH = 2, BSTEP = 2 * 256 * 2 * 3, JSTEP = 16 * 2 * 3;
    static_assert(NB % DEPTH == 0, "K / 32 must be a multiple of the ring depth");
    f32x4 a0[DEPTH], a1[DEPTH];
    u32x4_ w[DEPTH][4][3];
#define EVM_LOADS16(q, s)                                                                  \
    {                                                                                      \
        a0[q] = *reinterpret_cast<const f32x4 *>(ap + 16 * (s));                           \
        a1[q] = *reinterpret_cast<const f32x4 *>(ap + 16 * (s) + 4);                       \
        _Pragma("unroll") for (int j = 0; j < 4; j++)                                    \
            _Pragma("unroll") for (int p = 0; p < 3; p++) w[q][j][p] = bp[(size_t) (s) * BSTEP + j * JSTEP + p]; \
    }
#pragma unroll
    for (int q = 0; q < DEPTH - 1; q++) EVM_LOADS16(q, q)
    __builtin_amdgcn_sched_barrier(0);
    for (int s0 = 0; s0 < NB; s0 += DEPTH) {
#pragma unroll
        for (int q = 0; q < DEPTH; q++) {
            const int sn = min(s0 + q + DEPTH - 1, NB - 1);
            EVM_LOADS16((q + DEPTH - 1) % DEPTH, sn)
            __builtin_amdgcn_sched_barrier(0);
            const SplitA a = split8(a0[q], a1[q]);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bf16x8 b0 = __builtin_bit_cast(bf16x8, w[q][j][0]), b1 = __builtin_bit_cast(bf16x8, w[q][j][1]),
                             b2 = __builtin_bit_cast(bf16x8, w[q][j][2]);
                c1[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p0, b2, c1[j], 0, 0, 0);
                c1[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p1, b1, c1[j], 0, 0, 0);
                c1[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p2, b0, c1[j], 0, 0, 0);
                c1[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p0, b1, c1[j], 0, 0, 0);
                c1[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p1, b0, c1[j], 0, 0, 0);
                c0[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p0, b0, c0[j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef EVM_LOADS16
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc[j][r] = c0[j][r] + c1[j][r];
}

// sum over the 16 lanes of each DPP row, result in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
#define EVM_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false));
    EVM_DPP_ADD(0xB1)
    EVM_DPP_ADD(0x4E)
    EVM_DPP_ADD(0x141)
    EVM_DPP_ADD(0x140)
#undef EVM_DPP_ADD
    return v;
}

// mish_ln_epilogue for the 16-row accumulator layout (rollout only: nothing goes to HBM).  red: 256 floats
// ([16 rows][4 waves][2] partials, then [4 waves][16 rows][2] results).
// this lane's bias / LayerNorm weight / LayerNorm bias of its four columns: requested ahead of the GEMM by the caller
struct LnParams16 { float bi[4], ga[4], be[4]; };
__device__ __forceinline__ LnParams16 ln_params16(const float *__restrict__ bias, const float *__restrict__ gamma,
                                                  const float *__restrict__ beta, int wave, int lane) {
    LnParams16 P;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int col = wave * 64 + 16 * j + (lane & 15);
        P.bi[j] = bias[col]; P.ga[j] = gamma[col]; P.be[j] = beta[col];
    }
    return P;
}
__device__ __forceinline__ void mish_ln_epilogue16(f32x4c (&acc)[4], const LnParams16 &P, float *hb, float *red, int wave, int lane) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    const int c = lane & 15, kq = lane >> 4;
    const float (&bi)[4] = P.bi, (&ga)[4] = P.ga, (&be)[4] = P.be;
    float s[4], q[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int j = 0; j < 4; j++) acc[j][r] = mish_f(acc[j][r] + bi[j]);
        s[r] = row16_sum((acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r])) * (1.0f / 64.0f);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float d0 = acc[0][r] - s[r], d1 = acc[1][r] - s[r], d2 = acc[2][r] - s[r], d3 = acc[3][r] - s[r];
        q[r] = row16_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
    }
    if (c == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) *reinterpret_cast<f32x2_ *>(red + ((4 * kq + r) * 4 + wave) * 2) = f32x2_{s[r], q[r]};
    }
    lds_barrier();
    float *mine = red + 128 + wave * 32;  // this wave's [16 rows][mean, rstd]
    {
        const f32x4 p0 = *reinterpret_cast<const f32x4 *>(red + c * 8), p1 = *reinterpret_cast<const f32x4 *>(red + c * 8 + 4);
        const float mean = ((p0[0] + p0[2]) + (p1[0] + p1[2])) * 0.25f;
        const float da = p0[0] - mean, db = p0[2] - mean, dc = p1[0] - mean, dd = p1[2] - mean;
        const float m2 = ((p0[1] + p0[3]) + (p1[1] + p1[3])) + 64.0f * ((da * da + db * db) + (dc * dc + dd * dd));
        const float rstd = 1.0f / sqrtf(m2 * (1.0f / 256.0f) + 1e-5f);
        if (kq == 0) *reinterpret_cast<f32x2_ *>(mine + c * 2) = f32x2_{mean, rstd};
    }
    // (LDS operations of one wave complete in order: the reads below see this wave's line without a barrier)
    const f32x4 ma = *reinterpret_cast<const f32x4 *>(mine + 8 * kq), mb = *reinterpret_cast<const f32x4 *>(mine + 8 * kq + 4);
    const float mean[4] = {ma[0], ma[2], mb[0], mb[2]}, rstd[4] = {ma[1], ma[3], mb[1], mb[3]};
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            hb[HIDX(4 * kq + r, wave * 64 + 16 * j + c)] = (acc[j][r] - mean[r]) * rstd[r] * ga[j] + be[j];
    lds_barrier();
}
#define EVM_RED16_FLOATS 256

// head GEMM [16 x 256] x [256 x 32] for the 16-row tile: K split over the four waves (64 each), two 16-column tiles;
// wave w leaves its partial in hs4[w][row 0..15][col 0..31]
struct HeadB16 { f32x4 b0[4], b1[4]; };
__device__ __forceinline__ HeadB16 head16_prefetch(const float *__restrict__ whp, int wave, int lane) {
    HeadB16 H;
    const int c = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int s4 = 8 * wave + 2 * j + (kq >> 1);
        H.b0[j] = *reinterpret_cast<const f32x4 *>(whp + ((size_t) (s4 * 32 + c) * 2 + (kq & 1)) * 4);
        H.b1[j] = *reinterpret_cast<const f32x4 *>(whp + ((size_t) (s4 * 32 + 16 + c) * 2 + (kq & 1)) * 4);
    }
    return H;
}
__device__ __forceinline__ void head_gemm16(const float *hb, const HeadB16 &H, float *hs4, int wave, int lane) {
    f32x4c h0 = {0.f, 0.f, 0.f, 0.f}, h1 = {0.f, 0.f, 0.f, 0.f};
    const int c = lane & 15, kq = lane >> 4;
    f32x4 a4[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int s4 = 8 * wave + 2 * j + (kq >> 1);
        a4[j] = *reinterpret_cast<const f32x4 *>(hb + c * ALD2 + (kq & 1) * 128 + 4 * s4);
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int t = 0; t < 4; t++) {
            h0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j][t], H.b0[j][t], h0, 0, 0, 0);
            h1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j][t], H.b1[j][t], h1, 0, 0, 0);
        }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        hs4[(wave * 16 + 4 * kq + r) * 32 + c] = h0[r];
        hs4[(wave * 16 + 4 * kq + r) * 32 + 16 + c] = h1[r];
    }
}

// stage a TM_ x S tile of a row-major [n][S] matrix, zero padded to K1 columns, k-split in LDS (row stride ALD1)
template <int TM_, int NT = PT>
__device__ __forceinline__ void stage_rows_ksplit(float *xs, const float *__restrict__ obs, int row0, int n, int S) {
                const size_t base = (size_t) row0 * S;
        const int tile = TM_ * S;  // floats; rows of the tile are contiguous in memory
        if (row0 + TM_ <= n && (tile & 3) == 0 && ((reinterpret_cast<uintptr_t>(obs + base)) & 15) == 0) {
            // full, 16-byte aligned tile: flat float4 loads, all in flight together
            const f32x4 *src = reinterpret_cast<const f32x4 *>(obs + base);
            const int nq = tile >> 2;
            constexpr int NIT = (TM_ * K1 / 4 + NT - 1) / NT;  // upper bound (S <= K1)
            f32x4 v[NIT];
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int q = it * NT + (int) threadIdx.x;
                v[it] = src[min(q, nq - 1)];
            }
            // (row, column) of a flat element without a division per element: one division for the thread's first element,
            // then a constant (row, column) stride per trip, with carry
            const int e0 = 4 * (int) threadIdx.x;
            int r = e0 / S, k = e0 - r * S;
            const int dr = (4 * NT) / S, dk = 4 * NT - dr * S;
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int q = it * NT + (int) threadIdx.x;
                if (q < nq) {
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const bool wrap = k + u >= S;
                        const int rr = r + (wrap ? 1 : 0), kk = k + u - (wrap ? S : 0);
                        xs[rr * ALD1 + (kk & 1) * (K1 / 2) + (kk >> 1)] = v[it][u];
                    }
                }
                r += dr; k += dk;
                if (k >= S) { k -= S; r++; }
            }
            for (int e = threadIdx.x; e < TM_ * (K1 - S); e += NT) {
                const int r = e / (K1 - S), k = S + e % (K1 - S);
                xs[r * ALD1 + (k & 1) * (K1 / 2) + (k >> 1)] = 0.f;
            }
        } else {
            // ragged last tile or unaligned caller buffer: scalar loads, clamped address + select (no branches)
            const size_t last = (size_t) n * S - 1;
            for (int e = threadIdx.x; e < TM_ * K1; e += NT) {
                const int r = e / K1, k = e % K1;
                const bool ok = row0 + r < n && k < S;
                const size_t g = base + (size_t) r * S + k;
                const float v = obs[g < last ? g : last];
                xs[r * ALD1 + (k & 1) * (K1 / 2) + (k >> 1)] = ok ? v : 0.f;
            }
        }
}

}  // namespace evm
