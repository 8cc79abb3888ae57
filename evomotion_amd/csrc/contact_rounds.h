// Member-vs-member mode (EvmSkelC::self_collision) of the lane-group sweeps kernel: the contact rows of a sweep — floor
// manifolds and member pairs alike — run AFTER the sweep's joint rows as rounds of body-disjoint manifolds, chosen per
// environment at run time.  Included by sweep_groups.h (same translation unit: GCtx, g_load_body, ...).
//
// Order kept: Bullet solves, per iteration, all joint rows, then all contact normal rows, then all friction rows, both in
// manifold order (btSequentialImpulseConstraintSolver::solveSingleIteration).  The manifold order here is: floor-vs-member
// by member index, then the pairs in table order (Bullet's own order — dispatcher array, island sort — cannot be known).
// Two manifolds that share no body commute exactly, so a greedy levelling of that list (a manifold goes to the first round
// after the last round of either of its bodies) reproduces the sequential order bit for bit whatever the round sizes.
//
// Execution: every (wave, lane group) pair is a SLOT; slot s of environment e owns the s-th active manifold of that env
// for the whole step and keeps the manifold's record (4 points x 5 quads) and its accumulated impulses in registers across
// the ten sweeps (16 slots; the spider has been seen with 14 live manifolds in 5 000 random env-steps).  The 17th..32nd live
// manifold of an env go to a second bank that the same slots serve from global memory (g_slow_visit: record re-read at every
// visit, impulses kept in the record) — correct, slow, and rare; beyond that the manifold is left out and counted in
// EnvDev::errs.  Both banks: bodies are read from and written to the workgroup's LDS
// image with per-lane body indices ([quad][16 lanes]: the bank depends on the env column only, so any mix of bodies is
// conflict-free).  A round is closed by a workgroup barrier.
#pragma once

namespace evm {

struct CBank {
    int id;       // manifold id: member m (floor) or nm + pair; -1 = none
    int a, b;     // body0 (-1: the static floor) and body1
    int round;
    int lv;       // bit j: point j is live (jd_n != 0)
    f32x4 q[20];  // the contact record: per point quads 0..4 (skel_const.h, EVM_CR_STRIDE)
    float apn[4], apf[4];  // the accumulated impulses of the sweeps (the record's applied_n / applied_f fields at load time): on
                           // their own, so that a row rewrites one register, not its quad
};
#define K_APN(K, j) ((K).apn[j])
#define K_APF(K, j) ((K).apf[j])
DEV void g_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// body b (per-lane index) as a row operand; b < 0: the static floor (zero inverse mass and inertia, never stored)
DEV BodyD g_body_any(const GCtx &G, int b, const float *imt) {
    const int bb = b < 0 ? 0 : b;
    BodyD k = g_load_body(G, bb, imt[bb]);
    if (b < 0) { k.dl = f3(0, 0, 0); k.da = f3(0, 0, 0); k.I.xx = k.I.xy = k.I.xz = k.I.yy = k.I.yz = k.I.zz = 0.f; k.im = 0.f; }
    return k;
}
// both bodies of a manifold as an (A, B) pair for the packed row arithmetic of the joint rows (row_iter<LIN = true, BOUNDED>:
// one packed instruction serves both sides); a < 0: the A half is the static floor = all zeros
DEV BodyPD g_pair_any(const GCtx &G, int a, int b, const float *imt) {
    const int aa = a < 0 ? b : a;
    BodyPD Q = g_load_pair(G, aa, b, imt[aa], imt[b]);
    if (a < 0) {
        Q.dl.x.x = 0.f; Q.dl.y.x = 0.f; Q.dl.z.x = 0.f; Q.da.x.x = 0.f; Q.da.y.x = 0.f; Q.da.z.x = 0.f;
        Q.I.xx.x = 0.f; Q.I.xy.x = 0.f; Q.I.xz.x = 0.f; Q.I.yy.x = 0.f; Q.I.yz.x = 0.f; Q.I.zz.x = 0.f;
        Q.im.x = 0.f;
    }
    return Q;
}
DEV void g_store_pair_any(const GCtx &G, int a, int b, const BodyPD &Q) {
    f32x4 x;
    if (a >= 0) {
        x[0] = Q.dl.x.x; x[1] = Q.dl.y.x; x[2] = Q.dl.z.x; x[3] = Q.da.x.x; GB(G, 3 * a) = x;
        x[0] = Q.da.y.x; x[1] = Q.da.z.x; x[2] = Q.I.xx.x; x[3] = Q.I.xy.x; GB(G, 3 * a + 1) = x;
    }
    x[0] = Q.dl.x.y; x[1] = Q.dl.y.y; x[2] = Q.dl.z.y; x[3] = Q.da.x.y; GB(G, 3 * b) = x;
    x[0] = Q.da.y.y; x[1] = Q.da.z.y; x[2] = Q.I.xx.y; x[3] = Q.I.xy.y; GB(G, 3 * b + 1) = x;
}
// one two-body row along `dir` (contact normal or friction direction) on scalar bodies: the split-impulse recovery's form
DEV float g_row2(F3 dir, F3 relA, F3 relB, BodyD &A, BodyD &B, float jd, float rhs, float lo, float hi, float &ap) {
    const F3 c1 = cross(relA, dir), c2 = -cross(relB, dir);
    const float d1 = dot(dir, A.dl) + dot(c1, A.da);
    const float d2 = -dot(dir, B.dl) + dot(c2, B.da);
    float dI = rhs;
    dI -= d1 * jd;
    dI -= d2 * jd;
    const float sum = ap + dI;
    if (sum < lo) { dI = lo - ap; ap = lo; }
    else if (sum > hi) { dI = hi - ap; ap = hi; }
    else ap = sum;
    A.dl = A.dl + dir * (A.im * dI);
    A.da = A.da + mul(A.I, c1) * dI;
    B.dl = B.dl - dir * (B.im * dI);
    B.da = B.da + mul(B.I, c2) * dI;
    return dI;
}
// PHASE 0: warm start (convertContact applies the cached impulses x 0.85, per point normal then friction)
//       1: normal rows   2: friction rows (limits +-mu x the point's normal impulse of this sweep)
// A contact row is a two-body LINEAR row (n, relA x n, -(relB x n)) like a joint's: the packed (A, B) row of the joint rows
// does it in half the instructions of two scalar sides.
template <int PHASE>
DEV float g_contact_bank(const GCtx &G, CBank &K, bool on, const float *imt) {
    float res = 0.f;
    if (!__any(on)) return res;
    BodyPD Q;
    if (on) Q = g_pair_any(G, K.a, K.b, imt);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const bool live = on && K.q[5 * j][3] != 0.f;  // jd_n = 1 / denominator > 0 for a live point
        if (!__any(live)) continue;
        if (live) {
            const F3P rel = f3p(p2(K.q[5 * j][0], -K.q[5 * j + 1][0]), p2(K.q[5 * j][1], -K.q[5 * j + 1][1]), p2(K.q[5 * j][2], -K.q[5 * j + 1][2]));  // (relA, -relB)
            const F3 nrm = f3(K.q[5 * j + 2][0], K.q[5 * j + 2][1], K.q[5 * j + 2][2]);
            const F3 lat = f3(K.q[5 * j + 3][0], K.q[5 * j + 3][1], K.q[5 * j + 3][2]);
            if (PHASE == 0) {
                // internalApplyImpulse(n1 * imA, angA, ap) on body0, (-n2 * imB, -angB, -ap) on body1 for the normal row, then
                // the same with the friction direction: delta = (n * im, I (rel x n)) * (ap, -ap)  [im = (imA, -imB), rel = (relA, -relB)]
                const float an = K_APN(K, j), af = K_APF(K, j);
                {
                    const F3P A = f3p(p2(nrm.x, nrm.x), p2(nrm.y, nrm.y), p2(nrm.z, nrm.z));
                    const F3P cc = cross(rel, A);
                    Q.dl = Q.dl + A * (Q.im * an);
                    Q.da = Q.da + mul(Q.I, cc) * an;
                }
                {
                    const F3P A = f3p(p2(lat.x, lat.x), p2(lat.y, lat.y), p2(lat.z, lat.z));
                    const F3P cc = cross(rel, A);
                    Q.dl = Q.dl + A * (Q.im * af);
                    Q.da = Q.da + mul(Q.I, cc) * af;
                }
            }
        }
    }
    if (on) g_store_pair_any(G, K.a, K.b, Q);
    return res;
}
// One bounded two-body linear row on the packed (A, B) pair, branch-free: the arithmetic of row_iter<true, true> op for op; a lane
// with act == false runs it as an exact no-op (zero impulse, accumulated impulse kept), so a round needs no per-lane branches.
DEV float g_crow(F3 ax, const F3P &rel, BodyPD &Q, float jd, float rhs, float lo, float hi, float ap, bool act, float &ap_out) {
    const F3P A = f3p(p2(ax.x, ax.x), p2(ax.y, ax.y), p2(ax.z, ax.z));
    const F3P cc = cross(rel, A);
    const P2 t = dot(A, Q.dl), u = dot(cc, Q.da);
    const P2 d = p2(t.x, -t.y) + u;
    float dI = rhs;
    dI -= d.x * jd;
    dI -= d.y * jd;
    const float sum = ap + dI;
    const bool cl = sum < lo, ch = !cl && sum > hi;
    const float nap = cl ? lo : (ch ? hi : sum);
    dI = (cl || ch) ? nap - ap : dI;
    dI = act ? dI : 0.f;
    ap_out = act ? nap : ap;
    const F3P ang = mul(Q.I, cc);
    Q.dl = Q.dl + A * (Q.im * dI);
    Q.da = Q.da + ang * dI;
    return dI;
}
// PHASE 1: normal rows   2: friction rows (limits +-mu x the point's normal impulse of this sweep) of the manifolds whose round
// it is (`on`); the other lanes ride along with zero bodies and act == false
template <int PHASE>
DEV float g_contact_rows(const GCtx &G, CBank &K, bool on, const float *imt) {
    float res = 0.f;
    if (!__any(on)) return res;
    BodyPD Q;
    Q.dl = f3p(p2(0.f, 0.f), p2(0.f, 0.f), p2(0.f, 0.f)); Q.da = Q.dl;
    Q.I.xx = Q.I.xy = Q.I.xz = Q.I.yy = Q.I.yz = Q.I.zz = p2(0.f, 0.f); Q.im = p2(0.f, 0.f);
    if (on) Q = g_pair_any(G, K.a, K.b, imt);
    const float mu = K.q[4][3];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const bool live = on && ((K.lv >> j) & 1);
        if (!__any(live)) continue;
        const F3P rel = f3p(p2(K.q[5 * j][0], -K.q[5 * j + 1][0]), p2(K.q[5 * j][1], -K.q[5 * j + 1][1]), p2(K.q[5 * j][2], -K.q[5 * j + 1][2]));  // (relA, -relB)
        if (PHASE == 1) {
            const F3 nrm = f3(K.q[5 * j + 2][0], K.q[5 * j + 2][1], K.q[5 * j + 2][2]);
            float ap;
            res = fmaxf(res, fabsf(g_crow(nrm, rel, Q, K.q[5 * j][3], K.q[5 * j + 1][3], 0.f, 1e10f, K_APN(K, j), live, ap)));
            K_APN(K, j) = ap;
        } else {
            const F3 lat = f3(K.q[5 * j + 3][0], K.q[5 * j + 3][1], K.q[5 * j + 3][2]);
            const float apn = K_APN(K, j), lim = mu * apn;  // a point without normal impulse has no friction row this sweep
            float ap;
            res = fmaxf(res, fabsf(g_crow(lat, rel, Q, K.q[5 * j + 3][3], K.q[5 * j + 4][0], -lim, lim, K_APF(K, j), live && apn > 0.f, ap)));
            K_APF(K, j) = ap;
        }
    }
    if (on) g_store_pair_any(G, K.a, K.b, Q);
    return res;
}
// split-impulse recovery of one manifold (resolveSplitPenetrationImpulse): the push / turn velocities of the members live in
// LDS for the phase, ptl[(6 m + k) * 16 + env] (k = 0..2 push, 3..5 turn), per-lane member index
DEV void g_split_bank(const GCtx &G, const CBank &K, bool on, float (&push_ap)[4], const float *imt, float *ptl) {
    bool pen = false;
#pragma unroll
    for (int j = 0; j < 4; j++) pen = pen || (on && K.q[5 * j + 4][2] != 0.f);
    if (!__any(pen)) return;
    if (!pen) return;
    const int ba = K.a < 0 ? 0 : K.a;
    float *pa = ptl + ((6 * ba) << 4) + G.e, *pb = ptl + ((6 * K.b) << 4) + G.e;
    BodyD A = g_body_any(G, K.a, imt), B = g_body_any(G, K.b, imt);  // inertia + inverse mass; dl = push, da = turn below
    A.dl = K.a < 0 ? f3(0, 0, 0) : f3(pa[0], pa[16], pa[32]);
    A.da = K.a < 0 ? f3(0, 0, 0) : f3(pa[48], pa[64], pa[80]);
    B.dl = f3(pb[0], pb[16], pb[32]);
    B.da = f3(pb[48], pb[64], pb[80]);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float rp = K.q[5 * j + 4][2];
        if (rp != 0.f) {
            const F3 relA = f3(K.q[5 * j][0], K.q[5 * j][1], K.q[5 * j][2]);
            const F3 relB = f3(K.q[5 * j + 1][0], K.q[5 * j + 1][1], K.q[5 * j + 1][2]);
            const F3 nrm = f3(K.q[5 * j + 2][0], K.q[5 * j + 2][1], K.q[5 * j + 2][2]);
            g_row2(nrm, relA, relB, A, B, K.q[5 * j][3], rp, 0.f, EVM_INF, push_ap[j]);
        }
    }
    if (K.a >= 0) { pa[0] = A.dl.x; pa[16] = A.dl.y; pa[32] = A.dl.z; pa[48] = A.da.x; pa[64] = A.da.y; pa[80] = A.da.z; }
    pb[0] = B.dl.x; pb[16] = B.dl.y; pb[32] = B.dl.z; pb[48] = B.da.x; pb[64] = B.da.y; pb[80] = B.da.z;
}
// program word: id (9 bits) | (body0 + 1) << 9 (6 bits, 0 = floor) | body1 << 15 (6 bits) | round << 21 (5 bits); ~0 = none
DEV unsigned g_prog_word(int id, int a, int b, int round) {
    return (unsigned) id | ((unsigned) (a + 1) << 9) | ((unsigned) b << 15) | ((unsigned) round << 21);
}
DEV void g_bank_live(CBank &K) {
    K.lv = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        K.lv |= K.q[5 * j][3] != 0.f ? 1 << j : 0;  // jd_n = 1 / denominator > 0 for a live point
        K.apn[j] = K.q[5 * j + 2][3];
        K.apf[j] = K.q[5 * j + 4][1];
    }
}
DEV void g_bank_load(const Ctx &c, unsigned w, CBank &K) {
    if (w == 0xffffffffu) { K.id = -1; K.a = -1; K.b = 0; K.round = -1; }
    else { K.id = (int) (w & 511u); K.a = (int) ((w >> 9) & 63u) - 1; K.b = (int) ((w >> 15) & 63u); K.round = (int) ((w >> 21) & 31u); }
#pragma unroll
    for (int q = 0; q < 20; q++) K.q[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (K.id >= 0) {
        const f32x4 *p = reinterpret_cast<const f32x4 *>(c.t.crec + ((size_t) (K.id * EVM_CR_STRIDE) << 6)) + c.lane;
#pragma unroll
        for (int q = 0; q < 20; q++) K.q[q] = p[q << 6];
    }
    g_bank_live(K);
}
// the accumulated impulses back into the persistent manifold the bank's manifold came from (warm start of the next step)
DEV void g_bank_writeback(const Ctx &c, const CBank &K) {
    if (K.id < 0) return;
    const int nm = c_skel.nm;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (K.q[5 * j][3] == 0.f) continue;
        if (K.id < nm) {
            c.t.mfp[(((K.id * 4 + j) * 9 + 7) << 6) + c.lane] = K_APN(K, j);
            c.t.mfp[(((K.id * 4 + j) * 9 + 8) << 6) + c.lane] = K_APF(K, j);
        } else {
            const int p = K.id - nm;
            c.t.pmp[(((p * 4 + j) * 12 + 10) << 6) + c.lane] = K_APN(K, j);
            c.t.pmp[(((p * 4 + j) * 12 + 11) << 6) + c.lane] = K_APF(K, j);
        }
    }
}

// The overflow bank: one visit of the manifold in program word w (round r, phase 0 warm start / 1 normal rows / 2 friction rows /
// 3 split-impulse recovery) straight from its record in global memory; the accumulated impulses are written back into the
// record's applied fields (the spare quad holds the split-impulse accumulators).  Not inlined: the fast path's register
// allocation must not pay for this one.
__device__ __attribute__((noinline)) float g_slow_visit(int phase, unsigned w, int r, f32x4 *ldsq, int e, const float *imt, float *crec_lane,
                                                         float *ptl) {
    if (w == 0xffffffffu || (int) ((w >> 21) & 31u) != r) return 0.f;
    GCtx G;
    G.qb = ldsq; G.q = ldsq; G.e = e; G.g = 0; G.QR = 0; G.ver = nullptr; G.multi = false;
    CBank K;
    K.id = (int) (w & 511u); K.a = (int) ((w >> 9) & 63u) - 1; K.b = (int) ((w >> 15) & 63u); K.round = r;
    f32x4 *rec = reinterpret_cast<f32x4 *>(crec_lane + ((size_t) (K.id * EVM_CR_STRIDE) << 6));
#pragma unroll
    for (int q = 0; q < 20; q++) K.q[q] = rec[q << 6];
    g_bank_live(K);
    float res = 0.f;
    if (phase == 0) res = g_contact_bank<0>(G, K, true, imt);
    else if (phase == 1) res = g_contact_rows<1>(G, K, true, imt);
    else if (phase == 2) res = g_contact_rows<2>(G, K, true, imt);
    else {
        f32x4 pq = rec[20 << 6];
        float pa[4] = {pq[0], pq[1], pq[2], pq[3]};
        g_split_bank(G, K, true, pa, imt, ptl);
        pq[0] = pa[0]; pq[1] = pa[1]; pq[2] = pa[2]; pq[3] = pa[3];
        rec[20 << 6] = pq;
    }
    if (phase == 1 || phase == 2) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            K.q[5 * j + 2][3] = K.apn[j]; K.q[5 * j + 4][1] = K.apf[j];
            rec[(5 * j + 2) << 6] = K.q[5 * j + 2]; rec[(5 * j + 4) << 6] = K.q[5 * j + 4];
        }
    }
    return res;
}

// Builds the contact program of the workgroup's 16 envs (one wave; lanes of group 0, one env each): walks the manifold ids in
// solver order, gives the k-th active one of an env to slot k % 16 of bank k / 16 and levels it into rounds.
//   prog [2 banks][16 slots][16 envs] words (pre-filled with ~0), meta[0] = rounds used (max over envs), meta[1] = 1 if some
//   env uses the second bank, meta[2] = 1 if some env has a point for the split-impulse recovery
// nn: this env's floor manifold counts; returns the number of manifolds that did not fit (> 32 live manifolds or > 31 rounds)
#define EVM_PACT_WORDS ((EVM_MAX_PAIRS + 31) / 32)
// (the caller reads the env's activity words pw, the flag word and the floor manifold counts nn early, so that their latency
// hides behind the record image's copy)
DEV int g_build_program(const GCtx &G, int nslots, int epw, const int (&nn)[EVM_MAX_MEMBERS], const unsigned (&pw)[EVM_PACT_WORDS], unsigned flags,
                        unsigned *prog, int *meta) {
    const int nm = c_skel.nm, np = c_skel.npair, nwords = (np + 31) >> 5;
    // Everything below lives in registers on purpose: plain scalars and macros, no lambdas, no arrays indexed at run time.  (A
    // first form with reference-capturing lambdas and a four-word array kept its counters in scratch memory — 44 scratch loads,
    // each waited for: 26 k of the kernel's 440 k ticks.)
    static_assert(EVM_MAX_MEMBERS <= 24, "next-free-round table: 5 bits x 12 members per 64-bit word");
    unsigned long long nf0 = 0ull, nf1 = 0ull;  // next free round per member, 5 bits each: members 0..11 in nf0, 12..23 in nf1
    int ord = 0, rmax = 0, left_out = 0;
#define G_NF_GET(m) ((int) ((((m) < 12 ? nf0 : nf1) >> (5 * ((m) < 12 ? (m) : (m) - 12))) & 31ull))
#define G_NF_SET(m, v)                                                                                  \
    {                                                                                                   \
        const int sh_ = 5 * ((m) < 12 ? (m) : (m) - 12);                                                \
        const unsigned long long msk_ = ~(31ull << sh_), val_ = (unsigned long long) (v) << sh_;         \
        if ((m) < 12) nf0 = (nf0 & msk_) | val_; else nf1 = (nf1 & msk_) | val_;                         \
    }
#define G_PLACE(ID, A, B)                                                                               \
    {                                                                                                   \
        const int a_ = (A), b_ = (B);                                                                   \
        const int ra_ = a_ >= 0 ? G_NF_GET(a_ >= 0 ? a_ : 0) : 0, rb_ = G_NF_GET(b_);                   \
        const int r_ = max(ra_, rb_);                                                                   \
        if (r_ > 30 || ord >= 2 * nslots) left_out++;                                                   \
        else {                                                                                          \
            const int bank_ = ord >= nslots ? 1 : 0, slot_ = ord - bank_ * nslots;                      \
            prog[((bank_ * 16 + slot_) << 4) + G.e] = g_prog_word((ID), a_, b_, r_);                    \
            if (a_ >= 0) G_NF_SET(a_ >= 0 ? a_ : 0, r_ + 1)                                             \
            G_NF_SET(b_, r_ + 1)                                                                        \
            rmax = max(rmax, r_ + 1);                                                                   \
            ord++;                                                                                      \
        }                                                                                               \
    }
#pragma unroll
    for (int m = 0; m < EVM_MAX_MEMBERS; m++) {  // (unrolled: nn stays in registers, m is a constant)
        const bool act = m < nm && nn[m] > 0;
        if (!__any(act)) continue;
        if (act) G_PLACE(m, -1, m)
    }
    // The pair table's (body0, body1) as a register table across the lanes (lane l: pairs l, 64 + l, ...; read with v_readlane):
    // the walk below visits the union of the wave's active pairs one after the other, and a scalar load of the constant table
    // per pair was a dependent memory round trip each time.
    const int lane64 = (int) (threadIdx.x & 63);
    static_assert(EVM_MAX_PAIRS <= 64 * 5, "pair table as five registers per lane");
    int ptab[5];
#pragma unroll
    for (int q = 0; q < 5; q++) {
        const int p = 64 * q + lane64;
        ptab[q] = p < np ? ((int) c_skel.pair[p].a | ((int) c_skel.pair[p].b << 16)) : 0;
    }
#pragma unroll
    for (int k = 0; k < EVM_PACT_WORDS; k++) {   // (unrolled to the end: pw[k] is a register; words past the table are skipped)
        if (k < nwords) {
            // pairs of this word active in some env of the wave (wave-uniform walk over their union): the wave's envs sit in its
            // first epw lanes (lane -> env lane % epw), so the union is an OR over those lanes
            unsigned u = 0u;
            for (int e = 0; e < epw; e++) u |= (unsigned) __builtin_amdgcn_readlane((int) pw[k], e);
            unsigned long long any = u;
            while (any) {
                const int bit = __builtin_ctzll(any);
                any &= any - 1;
                const int p = 32 * k + bit;
                const bool act = (pw[k] >> bit) & 1u;
                const int q = p >> 6, pl = p & 63;
                const int ab = __builtin_amdgcn_readlane(q == 0 ? ptab[0] : (q == 1 ? ptab[1] : (q == 2 ? ptab[2] : (q == 3 ? ptab[3] : ptab[4]))), pl);
                const int pa = ab & 0xffff, pb = ab >> 16;
                if (act) G_PLACE(nm + p, pa, pb)
            }
        }
    }
#undef G_PLACE
#undef G_NF_SET
#undef G_NF_GET
    // workgroup-wide facts
    if (rmax > 0) atomicMax(&meta[0], rmax);
    if (ord > nslots) atomicMax(&meta[1], 1);
    if (flags & 1u) atomicMax(&meta[2], 1);
    return left_out;
}

}  // namespace evm
