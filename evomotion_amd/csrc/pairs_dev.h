// Member-vs-member contacts, setup side (included by env_kernels.hip after the split pipeline's helpers).
//
// The reference lets every pair of members collide except constraint parent / child
// (evo_motion_model/src/robot/constraint.cpp:65,147; world, dispatcher and broadphase: evo_motion_model/src/environment.cpp:20-31).
// Per step and pair Bullet runs: AABB overlap -> btConvexConvexAlgorithm::processCollision (one GJK query, narrow_dev.h) ->
// btManifoldResult::addContactPoint into the pair's persistent manifold -> refreshContactPoints; the solver then builds one
// normal and one friction row per cached point (btSequentialImpulseConstraintSolver::convertContact).
//
// k_split_pairs: work item = (pair, 64-env tile) on one wavefront, the pair wave-uniform, one environment per lane.  A
// pair whose boxes are apart in every lane and that holds no cached point costs two box tests and leaves.  The rows of a
// live manifold go to a two-body contact record (skel_const.h, EVM_CR_STRIDE) that the sweeps kernel reads; the same record
// format serves the floor manifolds in this mode (contact_record with hasA = false), so the sweeps have one row routine.
#pragma once
#include "narrow_dev.h"

namespace evm {

#define PMP(p, slot, f) (c.t.pmp[((((p) * 4 + (slot)) * 12 + (f)) << 6) + c.lane])
#define PMN(p) (c.t.pmn[((p) << 6) + c.lane])
// field f of quad q of manifold id's contact record
#define CRQ(id, q) (reinterpret_cast<f32x4 *>(c.t.crec + ((size_t) ((id) * EVM_CR_STRIDE + 4 * (q)) << 6)) + c.lane)

struct MPoint2 {
    F3 la, lb, nb;
    float dist, ap, apl;
};
DEV MPoint2 sel(bool cnd, const MPoint2 &a, const MPoint2 &b) {
    MPoint2 r;
    r.la = f3(cnd ? a.la.x : b.la.x, cnd ? a.la.y : b.la.y, cnd ? a.la.z : b.la.z);
    r.lb = f3(cnd ? a.lb.x : b.lb.x, cnd ? a.lb.y : b.lb.y, cnd ? a.lb.z : b.lb.z);
    r.nb = f3(cnd ? a.nb.x : b.nb.x, cnd ? a.nb.y : b.nb.y, cnd ? a.nb.z : b.nb.z);
    r.dist = cnd ? a.dist : b.dist; r.ap = cnd ? a.ap : b.ap; r.apl = cnd ? a.apl : b.apl;
    return r;
}
DEV MPoint2 load_mp2(const Ctx &c, int p, int slot) {
    MPoint2 m;
    m.la = f3(PMP(p, slot, 0), PMP(p, slot, 1), PMP(p, slot, 2));
    m.lb = f3(PMP(p, slot, 3), PMP(p, slot, 4), PMP(p, slot, 5));
    m.nb = f3(PMP(p, slot, 6), PMP(p, slot, 7), PMP(p, slot, 8));
    m.dist = PMP(p, slot, 9); m.ap = PMP(p, slot, 10); m.apl = PMP(p, slot, 11);
    return m;
}
DEV void store_mp2(const Ctx &c, int p, int slot, const MPoint2 &m) {
    PMP(p, slot, 0) = m.la.x; PMP(p, slot, 1) = m.la.y; PMP(p, slot, 2) = m.la.z;
    PMP(p, slot, 3) = m.lb.x; PMP(p, slot, 4) = m.lb.y; PMP(p, slot, 5) = m.lb.z;
    PMP(p, slot, 6) = m.nb.x; PMP(p, slot, 7) = m.nb.y; PMP(p, slot, 8) = m.nb.z;
    PMP(p, slot, 9) = m.dist; PMP(p, slot, 10) = m.ap; PMP(p, slot, 11) = m.apl;
}

// One manifold's rows (btSequentialImpulseConstraintSolver::convertContact: setupContactConstraint, the velocity-dependent
// friction direction, setupFrictionConstraint) into its contact record.  A = body0 (absent for the static floor), B = body1;
// posA / posB / nrm / dist / ap / apl: the manifold's points as refreshContactPoints left them.
DEV void contact_record(const Ctx &c, int id, int n, bool hasA, const BodyK &A, const BodyK &B, const F3 (&posA)[4], const F3 (&posB)[4],
                        const F3 (&nrm)[4], const float (&dist)[4], const float (&ap)[4], const float (&apl)[4], float mu, bool lead = true) {
    const float invdt = 1.f / DT_F;
    bool deep = false;  // some point waits for the split-impulse recovery
#pragma unroll
    for (int j = 0; j < 4; j++) {
        f32x4 q0, q1, q2, q3, q4;
        q0 = q1 = q2 = q3 = q4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (__any(j < n)) {
            const F3 nj = nrm[j];
            const F3 rel1 = hasA ? posA[j] - A.o : f3(0.f, 0.f, 0.f);
            const F3 rel2 = posB[j] - B.o;
            const F3 vel1 = hasA ? A.v + cross(A.w, rel1) : f3(0.f, 0.f, 0.f);
            const F3 vel2 = B.v + cross(B.w, rel2);
            const F3 vel = vel1 - vel2;
            const float rel_vel = dot(nj, vel);
            const F3 tq0 = cross(rel1, nj), tq1 = cross(rel2, nj);
            const F3 angA = hasA ? mul(A.I, tq0) : f3(0.f, 0.f, 0.f);
            const F3 angB = mul(B.I, -tq1);
            const float denom0 = hasA ? A.im + dot(nj, cross(angA, rel1)) : 0.f;
            const float denom1 = B.im + dot(nj, cross(-angB, rel2));
            const float jd = 1.0f / (denom0 + denom1 + 0.f);
            const float vel1Dotn = hasA ? dot(nj, A.v) + dot(tq0, A.w) : 0.f;
            const float vel2Dotn = dot(-nj, B.v) + dot(-tq1, B.w);
            const float rv = vel1Dotn + vel2Dotn;
            float positionalError = 0.f, velocityError = 0.f - rv;
            const float penetration = dist[j] + 0.f;
            if (penetration > 0.f) velocityError -= penetration * invdt;
            else positionalError = -penetration * ERP_F * invdt;
            const float penImp = positionalError * jd, velImp = velocityError * jd;
            float rhs, rhs_pen;
            if (penetration > SPLIT_THR_F) { rhs = penImp + velImp; rhs_pen = 0.f; }
            else { rhs = velImp; rhs_pen = penImp; }
            F3 lat = vel - nj * rel_vel;
            const float lat2 = len2(lat);
            if (lat2 > EVM_EPS) lat = lat * (1.f / sqrtf(lat2));
            else { F3 d2; plane_space(nj, lat, d2); }
            const F3 fc1 = cross(rel1, lat), fc2 = cross(rel2, -lat);
            const F3 fangA = hasA ? mul(A.I, fc1) : f3(0.f, 0.f, 0.f);
            const F3 fangB = mul(B.I, fc2);
            const float fd0 = hasA ? A.im + dot(lat, cross(fangA, rel1)) : 0.f;
            const float fd1 = B.im + dot(lat, cross(-fangB, rel2));
            const float fjd = 1.0f / (fd0 + fd1);
            const float fv1 = hasA ? dot(lat, A.v) + dot(fc1, A.w_raw) : 0.f;   // no external torque impulse in the friction rows
            const float fv2 = dot(-lat, B.v) + dot(fc2, B.w_raw);
            const float frhs = (0.f - (fv1 + fv2)) * fjd;
            if (j < n) {
                deep = deep || rhs_pen != 0.f;
                q0 = f32x4{rel1.x, rel1.y, rel1.z, jd};
                q1 = f32x4{rel2.x, rel2.y, rel2.z, rhs};
                q2 = f32x4{nj.x, nj.y, nj.z, ap[j] * WARM_F};
                q3 = f32x4{lat.x, lat.y, lat.z, fjd};
                q4 = f32x4{frhs, apl[j] * WARM_F, rhs_pen, mu};
            }
        }
        f32x4 *p = CRQ(id, 5 * j);
        p[0] = q0; p[64] = q1; p[128] = q2; p[192] = q3; p[256] = q4;
    }
    // the flag word behind the pairs' activity words: bit 0 = this env has a point deeper than the split-impulse threshold
    if (deep && lead) atomicOr(&c.t.pact[((((c_skel.npair + 31) >> 5)) << 6) + c.lane], 1u);
}

// floor manifold of member m in the two-body record format (normal on B = (0, -1, 0), body0 = the static floor)
DEV void floor_record(const Ctx &c, int m, int n, const MPoint *pts) {
    const EvmMemberC &MB = c_skel.member[m];
    const BodyK B = load_bodyk(c, m);
    F3 posA[4], posB[4], nrm[4];
    float dist[4], ap[4], apl[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        posA[j] = f3(0.f, 0.f, 0.f);
        posB[j] = mul(B.R, pts[j].lb) + B.o;
        nrm[j] = f3(0.f, -1.f, 0.f);
        dist[j] = pts[j].dist; ap[j] = pts[j].ap; apl[j] = pts[j].apl;
    }
    contact_record(c, m, n, false, B, B, posA, posB, nrm, dist, ap, apl, MB.mu);
}

// world box of member m: btTransformAabb of the cached local box, fattened by the contact breaking threshold
DEV void member_box(const EvmMemberC &MB, const M33 &R, F3 o, F3 &ctr, F3 &ext) {
    const F3 lc = load_f3(MB.aabb_c), h = load_f3(MB.aabb_h);
    ctr = mul(R, lc) + o;
    ext = f3(fabsf(R.r0.x) * h.x + fabsf(R.r0.y) * h.y + fabsf(R.r0.z) * h.z + 0.02f,
             fabsf(R.r1.x) * h.x + fabsf(R.r1.y) * h.y + fabsf(R.r1.z) * h.z + 0.02f,
             fabsf(R.r2.x) * h.x + fabsf(R.r2.y) * h.y + fabsf(R.r2.z) * h.z + 0.02f);
}

// Broadphase of pair p for the 64 envs of a tile (an item of k_split_pre_a): an env whose two boxes overlap, or that still
// holds a cached point, is appended to the pair's work list (one atomic per wavefront).  drop: lanes whose env starts a
// reset with this step (their cached points are discarded: removeRigidBody / addRigidBody).
// the step's work-list counters (EnvDev::pcount: the copy this step appends to)
// counters [0, npair) = the pairs' own lists, [npair] = the flat big-hull list, [npair + 1] = the URGENT list (below)
DEV int *pc_cur(const EnvDev &d) { return d.pcount + d.pc_cur * EVM_PC_STRIDE; }
DEV int *pc_next(const EnvDev &d) { return d.pcount + (d.pc_cur ^ 1) * EVM_PC_STRIDE; }
// A (pair, env) whose query went through the penetration-depth solver in the previous step (bit 8 of its manifold count word)
// will most likely do so again — a pair that interpenetrates is pushed apart over several steps — and such a query is by far
// the longest thing in the kernel (GJK + EPA + the records).  It goes to a list of its own, stored from the END of blist
// downwards, which the narrowphase kernel's FIRST blocks work on, one query per wavefront: it then runs beside the whole
// rest of the kernel instead of starting when some wavefront of 64 other queries happens to find it.
#define EVM_PMN_DEEP 0x100
#define EVM_PMN_COUNT(x) ((x) & 0xff)

// largest separation of the two members' CORE boxes (oriented local boxes of the un-margined hulls) along the six face normals;
// t = centre of B's box - centre of A's.  > 0: that far apart at least; < 0: overlapping along all six
DEV float core_box_gap(const EvmMemberC &MA, const EvmMemberC &MB, const M33 &Ra, const M33 &Rb, F3 t) {
    const F3 ha = f3(MA.aabb_h[0] - 2.f * MARGIN_F, MA.aabb_h[1] - 2.f * MARGIN_F, MA.aabb_h[2] - 2.f * MARGIN_F);  // core half extents
    const F3 hb = f3(MB.aabb_h[0] - 2.f * MARGIN_F, MB.aabb_h[1] - 2.f * MARGIN_F, MB.aabb_h[2] - 2.f * MARGIN_F);
    const F3 a0 = col0(Ra), a1 = col1(Ra), a2 = col2(Ra), b0 = col0(Rb), b1 = col1(Rb), b2 = col2(Rb);
    float gap = -EVM_INF;
    // axes of A: radius of A = ha_i, radius of B = sum_j hb_j |a_i . b_j|
    gap = fmaxf(gap, fabsf(dot(t, a0)) - ha.x - (hb.x * fabsf(dot(a0, b0)) + hb.y * fabsf(dot(a0, b1)) + hb.z * fabsf(dot(a0, b2))));
    gap = fmaxf(gap, fabsf(dot(t, a1)) - ha.y - (hb.x * fabsf(dot(a1, b0)) + hb.y * fabsf(dot(a1, b1)) + hb.z * fabsf(dot(a1, b2))));
    gap = fmaxf(gap, fabsf(dot(t, a2)) - ha.z - (hb.x * fabsf(dot(a2, b0)) + hb.y * fabsf(dot(a2, b1)) + hb.z * fabsf(dot(a2, b2))));
    gap = fmaxf(gap, fabsf(dot(t, b0)) - hb.x - (ha.x * fabsf(dot(b0, a0)) + ha.y * fabsf(dot(b0, a1)) + ha.z * fabsf(dot(b0, a2))));
    gap = fmaxf(gap, fabsf(dot(t, b1)) - hb.y - (ha.x * fabsf(dot(b1, a0)) + ha.y * fabsf(dot(b1, a1)) + ha.z * fabsf(dot(b1, a2))));
    gap = fmaxf(gap, fabsf(dot(t, b2)) - hb.z - (ha.x * fabsf(dot(b2, a0)) + ha.y * fabsf(dot(b2, a1)) + ha.z * fabsf(dot(b2, a2))));
    return gap;
}

// Ra, oa / Rb, ob: the two members' world transforms of this step (k_split_pre_a derives them itself: the bodies are being
// prepared by other waves of the same kernel)
DEV void pair_broadphase(const Ctx &c, int p, bool drop, const M33 &Ra, F3 oa, const M33 &Rb, F3 ob) {
    const EvmPairC &PC = c_skel.pair[p];
    const int a = PC.a, b = PC.b;
    const int nraw = PMN(p);
    int n = EVM_PMN_COUNT(nraw);
    bool urgent = !drop && (nraw & EVM_PMN_DEEP) != 0;
    if (drop && nraw != 0) { PMN(p) = 0; n = 0; }
    F3 ca, ea, cb, eb;
    member_box(c_skel.member[a], Ra, oa, ca, ea);
    member_box(c_skel.member[b], Rb, ob, cb, eb);
    bool near = fabsf(ca.x - cb.x) <= ea.x + eb.x && fabsf(ca.y - cb.y) <= ea.y + eb.y && fabsf(ca.z - cb.z) <= ea.z + eb.z;
    if (__any(near && n == 0)) {
        // Second cull, for pairs without a cached point: the un-margined hulls lie inside their oriented local boxes, so if a
        // face normal of either box separates the boxes by more than the two margins + the breaking threshold, GJK could only
        // report a distance beyond the manifold's reach: same result, no query.  (Four out of five box overlaps end here.)
        // Not applied in the step that follows reset(): that step's bases are not orthonormal.
        const float reach = MARGIN_F + MARGIN_F + PC.thr + 1e-4f;
        const float gap = core_box_gap(c_skel.member[a], c_skel.member[b], Ra, Rb, cb - ca);
        const bool pending = (c.d.flags[c.env] & EVM_FLAG_PENDING) != 0 || drop;
        if (n == 0 && !pending && gap > reach) near = false;
        // ... and the other end of the same test: core boxes that overlap along all six face normals (a pair that starts out
        // interpenetrating, typically in the step after a reset, where no history can flag it) will most likely need the
        // penetration solver: urgent list
        if (n == 0 && near && gap < c.d.gap_soon) urgent = true;
    }
    const bool need_any = n > 0 || near;
    if (!need_any && nraw != 0 && !drop) PMN(p) = 0;   // (no query this step: scheduling hints of the previous one — flag, slot — end here)
    if (need_any && urgent) {   // (a few per step in the whole batch: one atomic each)
        const int at = atomicAdd(&pc_cur(c.d)[c_skel.npair + 1], 1);
        atomicAdd(&c.d.errs[4], 1);
        c.d.blist[(size_t) c_skel.npair * c.d.n - 1 - at] = (p << 20) | c.env;
    }
    const bool need = need_any && !urgent;
    const unsigned long long m = __ballot(need);
    if (m == 0ull) return;
    const int lane = c.lane & 63, leader = (int) __builtin_ctzll(m);
    // a pair with a big hull (the feet) goes to ONE flat list of (pair, env) entries that the quarter-wave narrowphase kernel
    // walks; the others to the pair's own list (one env per lane, the pair wave-uniform)
    const bool big = c_skel.member[a].hull_n > EVM_BIG_HULL || c_skel.member[b].hull_n > EVM_BIG_HULL;
    const int slot = big ? c_skel.npair : p;
    int base = 0;
    if (lane == leader) base = atomicAdd(&pc_cur(c.d)[slot], (int) __popcll(m));
    base = __shfl(base, leader);
    const int at = base + (int) __popcll(m & ((1ull << lane) - 1ull));
    if (need) {
        if (big) c.d.blist[at] = (p << 20) | c.env;
        else c.d.plist[(size_t) p * c.d.n + at] = c.env;
    }
}

// The urgent list's entry (pair p of c's environment) for its speculation block (env_kernels.hip: narrow_block): the two shapes exactly
// as pair_item builds them, then the penetration query into `slot` (gj::speculate_pen_depth).  No box test, no early exit.
DEV void pair_speculate(const Ctx &c, int p, int lds_hull_off, int *slot) {
    const EvmPairC &PC = c_skel.pair[p];
    const int a = PC.a, b = PC.b;
    const EvmMemberC &MA = c_skel.member[a], &MB = c_skel.member[b];
    gj::Shape SA, SB;
    SA.hull_off = MA.hull_off; SA.hull_n = MA.hull_n; SB.hull_off = MB.hull_off; SB.hull_n = MB.hull_n;
    SA.lds_hull_off = SB.lds_hull_off = lds_hull_off;
    SA.pen_count = SB.pen_count = nullptr;
    SA.spec = SB.spec = nullptr; SA.spec_epoch = SB.spec_epoch = 0;
#ifdef EVM_KSTAMPS
    SA.ks = SB.ks = c.d.stamps;
#endif
    SA.o = G3(pos, 3 * a); SB.o = G3(pos, 3 * b);
    SA.R = m33(SC3(c_skel.sc_r + 9 * a), SC3(c_skel.sc_r + 9 * a + 3), SC3(c_skel.sc_r + 9 * a + 6));
    SB.R = m33(SC3(c_skel.sc_r + 9 * b), SC3(c_skel.sc_r + 9 * b + 3), SC3(c_skel.sc_r + 9 * b + 6));
    gj::speculate_pen_depth(SA, SB, lds_hull_off, slot, c.d.spec_epoch);
}

// one pair for one env per lane (any envs: the narrowphase kernel's compacted work list): narrowphase, manifold
// maintenance, rows.  drop: as above.  GROUP: the 16 lanes of a row carry the SAME (pair, env) and share the hull scans
// (every lane computes and stores the same values; lane 0 of the row does the atomics); p may then differ between rows.
// SOLO: all 64 lanes carry the same (pair, env) — the urgent list's blocks: the penetration-depth solver then has the whole
// wavefront for its parallel parts; every lane stores the same values to the same addresses, lane 0 does the atomics.
template <bool GROUP, bool SOLO = false>
DEV void pair_item(const Ctx &c, int p, bool drop, int lds_hull_off = -1, int *spec = nullptr) {
    const bool lead = SOLO ? threadIdx.x == 0 : (!GROUP || (threadIdx.x & 15) == 0);
#ifdef EVM_KSTAMPS  // (tools/kstamps.py) cycles of a working wavefront by phase; a mark waits for the outstanding memory traffic first
    unsigned long long ks_prev = __builtin_amdgcn_s_memtime();
#define KS_MARK(k)                                                                                        \
    {                                                                                                     \
        __builtin_amdgcn_s_waitcnt(0);                                                                    \
        const unsigned long long ks_now = __builtin_amdgcn_s_memtime();                                   \
        if (threadIdx.x == 0) atomicAdd(&c.d.stamps[8 + (GROUP ? 0 : 4) + (k)], ks_now - ks_prev);       \
        ks_prev = ks_now;                                                                                 \
    }
#else
#define KS_MARK(k)
#endif
    const EvmPairC &PC = c_skel.pair[p];
    const int a = PC.a, b = PC.b;
    const EvmMemberC &MA = c_skel.member[a], &MB = c_skel.member[b];
    const int nraw_in = PMN(p);
    int n = drop ? 0 : EVM_PMN_COUNT(nraw_in);
    gj::Shape SA, SB;
    SA.hull_off = MA.hull_off; SA.hull_n = MA.hull_n; SB.hull_off = MB.hull_off; SB.hull_n = MB.hull_n;
    SA.lds_hull_off = SB.lds_hull_off = lds_hull_off;
    SA.pen_count = SB.pen_count = c.d.errs + 2;
    SA.spec = SB.spec = SOLO ? spec : nullptr; SA.spec_epoch = SB.spec_epoch = c.d.spec_epoch;   // (the slot of the entry's speculation block, narrow_dev.h)
#ifdef EVM_KSTAMPS
    SA.ks = SB.ks = c.d.stamps;
#endif
    SA.o = G3(pos, 3 * a); SB.o = G3(pos, 3 * b);
    SA.R = m33(SC3(c_skel.sc_r + 9 * a), SC3(c_skel.sc_r + 9 * a + 3), SC3(c_skel.sc_r + 9 * a + 6));
    SB.R = m33(SC3(c_skel.sc_r + 9 * b), SC3(c_skel.sc_r + 9 * b + 3), SC3(c_skel.sc_r + 9 * b + 6));
    F3 ca, ea, cb, eb;
    member_box(MA, SA.R, SA.o, ca, ea);
    member_box(MB, SB.R, SB.o, cb, eb);
    const bool overlap = fabsf(ca.x - cb.x) <= ea.x + eb.x && fabsf(ca.y - cb.y) <= ea.y + eb.y && fabsf(ca.z - cb.z) <= ea.z + eb.z;
    if (!__any(overlap || n > 0)) {
        if (__any(drop || nraw_in != 0)) PMN(p) = 0;  // n == 0 in every lane here (and a stale deep flag goes)
        return;
    }
    const float thr = PC.thr;
    KS_MARK(0)
    // The closest-point query first, the cached points after it: the query is one long dependent chain at the register limit of
    // two waves per SIMD, and 48 registers of manifold points held across it were spilled inside its loop.
    gj::Result r;
    r.has = false; r.distance = 0.f; r.normalOnB = f3(0.f, 0.f, 0.f); r.pointOnB = f3(0.f, 0.f, 0.f); r.iterations = 0; r.used_pen = false;
    const bool any_overlap = __any(overlap);
    if (SOLO) { UST(2) }
    if (any_overlap) {
        const float md = MARGIN_F + MARGIN_F + thr;
        r = gj::closest_points<GROUP, SOLO>(SA, SB, md * md, overlap);
    }
    if (SOLO) { UST(6) }
    KS_MARK(1)
    MPoint2 p0 = load_mp2(c, p, 0), p1 = load_mp2(c, p, 1), p2 = load_mp2(c, p, 2), p3 = load_mp2(c, p, 3);
    if (any_overlap) {
        const bool add = r.has && !(r.distance > thr);
        if (add) {
            // btManifoldResult::addContactPoint(normalOnBInWorld, pointInWorld, depth)
            const F3 pointA = r.pointOnB + r.normalOnB * r.distance;
            MPoint2 np;
            np.la = tmul(SA.R, pointA - SA.o);       // btTransform::invXform
            np.lb = tmul(SB.R, r.pointOnB - SB.o);
            np.nb = r.normalOnB; np.dist = r.distance; np.ap = 0.f; np.apl = 0.f;
            float shortest = thr * thr;
            int nearest = -1;
            { const F3 d = p0.la - np.la; const float dd = dot(d, d); if (0 < n && dd < shortest) { shortest = dd; nearest = 0; } }
            { const F3 d = p1.la - np.la; const float dd = dot(d, d); if (1 < n && dd < shortest) { shortest = dd; nearest = 1; } }
            { const F3 d = p2.la - np.la; const float dd = dot(d, d); if (2 < n && dd < shortest) { shortest = dd; nearest = 2; } }
            { const F3 d = p3.la - np.la; const float dd = dot(d, d); if (3 < n && dd < shortest) { shortest = dd; nearest = 3; } }
            int ins;
            if (nearest >= 0) {
                ins = nearest;
                const MPoint2 old = sel(ins == 0, p0, sel(ins == 1, p1, sel(ins == 2, p2, p3)));
                np.ap = old.ap; np.apl = old.apl;   // replaceContactPoint keeps the accumulated impulses
            } else if (n == 4) {
                // sortCachedPoints: keep the deepest, maximise the area
                int maxPen = -1;
                float mp = np.dist;
                if (p0.dist < mp) { maxPen = 0; mp = p0.dist; }
                if (p1.dist < mp) { maxPen = 1; mp = p1.dist; }
                if (p2.dist < mp) { maxPen = 2; mp = p2.dist; }
                if (p3.dist < mp) { maxPen = 3; mp = p3.dist; }
                float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
                if (maxPen != 0) r0 = len2(cross(np.la - p1.la, p3.la - p2.la));
                if (maxPen != 1) r1 = len2(cross(np.la - p0.la, p3.la - p2.la));
                if (maxPen != 2) r2 = len2(cross(np.la - p0.la, p3.la - p1.la));
                if (maxPen != 3) r3 = len2(cross(np.la - p0.la, p2.la - p1.la));
                ins = -1;
                float mv = -1e18f;
                if (fabsf(r0) > mv) { ins = 0; mv = fabsf(r0); }
                if (fabsf(r1) > mv) { ins = 1; mv = fabsf(r1); }
                if (fabsf(r2) > mv) { ins = 2; mv = fabsf(r2); }
                if (fabsf(r3) > mv) { ins = 3; mv = fabsf(r3); }
                if (ins < 0) ins = 0;
            } else {
                ins = n;
                n++;
            }
            p0 = sel(ins == 0, np, p0); p1 = sel(ins == 1, np, p1); p2 = sel(ins == 2, np, p2); p3 = sel(ins == 3, np, p3);
        }
    }
    // refreshContactPoints(trA, trB): world points and distance follow the bodies, the stored normal stays; then removal from
    // the last slot down (swap-with-last)
    // (named scalars, selected with sel3: as arrays the compiler turned `last == 0 ? wA[0] : ...` into indexed loads from
    // scratch memory, a dependent round trip per removal)
    F3 wA0, wA1, wA2, wA3, wB0, wB1, wB2, wB3;
#define REFRESH2(I, P)                                    \
    {                                                     \
        wA##I = mul(SA.R, P.la) + SA.o;                   \
        wB##I = mul(SB.R, P.lb) + SB.o;                   \
        P.dist = dot(wA##I - wB##I, P.nb);                \
    }
    REFRESH2(3, p3) REFRESH2(2, p2) REFRESH2(1, p1) REFRESH2(0, p0)
#undef REFRESH2
    auto pick = [](int k, F3 x0, F3 x1, F3 x2, F3 x3) {
        return f3(k == 0 ? x0.x : (k == 1 ? x1.x : (k == 2 ? x2.x : x3.x)), k == 0 ? x0.y : (k == 1 ? x1.y : (k == 2 ? x2.y : x3.y)),
                  k == 0 ? x0.z : (k == 1 ? x1.z : (k == 2 ? x2.z : x3.z)));
    };
#define REMOVE2(I, P)                                                                              \
    if (I < n) {                                                                                   \
        bool rm = !(P.dist <= thr);                                                                \
        if (!rm) {                                                                                 \
            const F3 projected = wA##I - P.nb * P.dist;                                            \
            const F3 diff = wB##I - projected;                                                     \
            rm = dot(diff, diff) > thr * thr;                                                      \
        }                                                                                          \
        if (rm) {                                                                                  \
            const int last = n - 1;                                                                \
            const MPoint2 lp = sel(last == 0, p0, sel(last == 1, p1, sel(last == 2, p2, p3)));     \
            const F3 la_ = pick(last, wA0, wA1, wA2, wA3), lb_ = pick(last, wB0, wB1, wB2, wB3);   \
            P = lp; wA##I = la_; wB##I = lb_;                                                      \
            n--;                                                                                   \
        }                                                                                          \
    }
    REMOVE2(3, p3)
    REMOVE2(2, p2)
    REMOVE2(1, p1)
    REMOVE2(0, p0)
#undef REMOVE2
    const F3 wA[4] = {wA0, wA1, wA2, wA3}, wB[4] = {wB0, wB1, wB2, wB3};
    store_mp2(c, p, 0, p0); store_mp2(c, p, 1, p1); store_mp2(c, p, 2, p2); store_mp2(c, p, 3, p3);
    // (flagged for the next step's urgent list: it took the penetration branch, or its cores are about to touch)
    const bool deep = r.used_pen || (r.has && r.distance < c.d.deep_soon);
#ifdef EVM_DIAG_PEN
    if (r.used_pen && lead) {
        const int fl = c.d.flags[c.env];
        atomicAdd(&c.d.errs[8 + (drop ? 0 : ((fl & EVM_FLAG_PENDING) ? 1 : ((nraw_in & EVM_PMN_DEEP) ? 2 : (EVM_PMN_COUNT(nraw_in) > 0 ? 3 : 4))))], 1);
        if (c.d.settle_left[c.env] > 0) atomicAdd(&c.d.errs[13], 1);
        if (!(nraw_in & EVM_PMN_DEEP)) { const int k = c_skel.settle_steps - c.d.settle_left[c.env]; atomicAdd(&c.d.errs[14 + (k < 0 ? 0 : (k > 9 ? 9 : k))], 1); }
    }
#endif
#ifdef EVM_DIAG_PEN
    if (GROUP && !SOLO && lead && any_overlap && overlap) {   // big-hull queries: GJK iterations, and how well the previous step's count predicts them
        const int it = r.iterations, prev = (nraw_in >> 9) & 31;
        atomicAdd(&c.d.errs[24 + (it / 4 > 7 ? 7 : it / 4)], 1);
        if (it >= 12) { atomicAdd(&c.d.errs[32], 1); if (prev >= 10) atomicAdd(&c.d.errs[33], 1); }
        if (prev >= 10) atomicAdd(&c.d.errs[34], 1);
    }
    PMN(p) = n | (deep ? EVM_PMN_DEEP : 0) | ((r.iterations > 31 ? 31 : r.iterations) << 9);
#else
    PMN(p) = n | (deep ? EVM_PMN_DEEP : 0);
#endif
    if (SOLO) { UST(7) }
    KS_MARK(2)
    if (!__any(n > 0)) return;
    if (n > 0 && lead) atomicOr(&c.t.pact[((p >> 5) << 6) + c.lane], 1u << (p & 31));
    const BodyK A = load_bodyk(c, a), B = load_bodyk(c, b);
    F3 nrm[4] = {p0.nb, p1.nb, p2.nb, p3.nb};
    float dist[4] = {p0.dist, p1.dist, p2.dist, p3.dist}, ap[4] = {p0.ap, p1.ap, p2.ap, p3.ap}, apl[4] = {p0.apl, p1.apl, p2.apl, p3.apl};
    contact_record(c, c_skel.nm + p, n, true, A, B, wA, wB, nrm, dist, ap, apl, PC.mu, lead);
    if (SOLO) {
        UST(8)
#ifdef EVM_KSTAMPS
        __builtin_amdgcn_s_waitcnt(0);
        if (gj::epa::g_ust_on && threadIdx.x == 0 && r.used_pen) {
            atomicAdd(&c.d.stamps[47], 1ull);
            for (int k = 0; k < 8; k++) atomicAdd(&c.d.stamps[48 + k], gj::epa::g_ust[k + 1] - gj::epa::g_ust[k]);
            for (int k = 0; k < 8; k++) atomicAdd(&c.d.stamps[56 + k], gj::epa::g_uph[k]);
        }
#endif
    }
    KS_MARK(3)
}
#undef KS_MARK

}  // namespace evm
