// Convex-convex narrowphase for the member-vs-member pairs (included by env_kernels.hip; one environment per lane, the pair
// — hence both hulls — is wave-uniform).
//
// What it replaces: Bullet3's btConvexConvexAlgorithm::processCollision -> btGjkPairDetector::getClosestPointsNonVirtual
// with btVoronoiSimplexSolver, one query per overlapping pair and step (the reference's dispatcher / broadphase:
// evo_motion_model/src/environment.cpp:20-31; which pairs: evo_motion_model/src/robot/constraint.cpp:65,147; the shapes:
// btConvexHullShape of the OBJ vertices with local scaling, evo_motion_model/src/item.cpp:17-41), and for overlapping
// un-margined cores btGjkEpaPenetrationDepthSolver, the solver the reference's btDefaultCollisionConfiguration selects
// (epa_dev.h: one such query per wavefront at a time).
//
// GJK's exits hang on rounding (is the new support point already in the simplex, did the distance still shrink), and a
// different exit is a different contact point.  So everything in this header is compiled WITHOUT fma contraction and in
// one fixed operation order: identical transforms give identical contact points on every kernel build.
//
// SIMT shape: the simplex (<= 4 vertices x (w, p, q)) lives in registers, slot selection by compare-and-select; hull
// vertices are wave-uniform scalar loads, two vertices per packed instruction; lanes iterate until every lane of the
// wave has left the loop.
#pragma once

namespace evm {
#pragma clang fp contract(off)
namespace gj {

#define GJ_REL_ERROR2 1.0e-6f
#define GJ_PEN_TOLERANCE 0.001f
#define GJ_MAX_ITER 1000
#define GJ_EQUAL_VERTEX 0.0001f
#define GJ_LARGE 1e18f

DEV F3 add(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV F3 sub(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV F3 neg(F3 a) { return f3(-a.x, -a.y, -a.z); }
DEV F3 scl(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
DEV float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV F3 cross(F3 a, F3 b) { return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DEV float len2(F3 a) { return gj::dot(a, a); }
DEV F3 sel3(bool c, F3 a, F3 b) { return f3(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }
// btVector3 * btMatrix3x3 (row vector times matrix = R^T v)
DEV F3 vmul(F3 v, const M33 &m) {
    return f3(m.r0.x * v.x + m.r1.x * v.y + m.r2.x * v.z, m.r0.y * v.x + m.r1.y * v.y + m.r2.y * v.z,
              m.r0.z * v.x + m.r1.z * v.y + m.r2.z * v.z);
}
// btTransform::operator()(x)
DEV F3 xform(const M33 &R, F3 o, F3 x) { return gj::add(f3(gj::dot(R.r0, x), gj::dot(R.r1, x), gj::dot(R.r2, x)), o); }

// btConvexHullShape::localGetSupportingVertexWithoutMargin: first maximum of dot(dir, scaled vertex) over the hull in table
// order (EvmSkelC::hull pair layout: vertex g = 2 P + s at hull[6 P + s], hull[6 P + 2 + s], hull[6 P + 4 + s])
DEV F3 support(int hull_off, int hull_n, F3 dir) {
    const P2 dx = p2(dir.x, dir.x), dy = p2(dir.y, dir.y), dz = p2(dir.z, dir.z);
    const float *hp = c_skel.hull + 3 * hull_off;
    const int np = (hull_n + 1) >> 1;
    P2 best = p2(-GJ_LARGE, -GJ_LARGE);
    int be = 0, bo = 0;
#define GJ_PAIR(H, Q, PI)                                                                            \
    {                                                                                                \
        const P2 x = p2(H[6 * (Q)], H[6 * (Q) + 1]), y = p2(H[6 * (Q) + 2], H[6 * (Q) + 3]),         \
                 z = p2(H[6 * (Q) + 4], H[6 * (Q) + 5]);                                             \
        const P2 d = (dx * x + dy * y) + dz * z;                                                     \
        const bool ce = d.x > best.x, co = d.y > best.y;                                             \
        best = p2(ce ? d.x : best.x, co ? d.y : best.y);                                             \
        be = ce ? (PI) : be;                                                                         \
        bo = co ? (PI) : bo;                                                                         \
    }
    int p = 0;
    for (; p + 4 <= np; p += 4) {
        float h[24];
#pragma unroll
        for (int k = 0; k < 24; k++) h[k] = hp[6 * p + k];
#pragma unroll
        for (int q = 0; q < 4; q++) GJ_PAIR(h, q, p + q)
    }
    for (; p < np; p++) {
        float h[6];
#pragma unroll
        for (int k = 0; k < 6; k++) h[k] = hp[6 * p + k];
        GJ_PAIR(h, 0, p)
    }
#undef GJ_PAIR
    // even and odd vertices kept separate running maxima: the first maximum overall is the larger one, the lower index on a
    // tie (an odd hull repeats its last vertex at an odd position: it never wins a tie)
    int bi = 2 * be;
    if (best.y > best.x || (best.y == best.x && 2 * bo + 1 < bi)) bi = 2 * bo + 1;
    const int g = hull_off + bi, hb = 6 * (g >> 1) + (g & 1);
    return f3(c_skel.hull[hb], c_skel.hull[hb + 2], c_skel.hull[hb + 4]);
}

// The same support vertex found by the 16 lanes of a DPP row together (GROUP mode of the narrowphase: one query per
// quarter wavefront — for the 451-vertex feet, where one lane per query would walk the whole hull alone): lane k of the row
// takes vertices k, k + 16, ..., then a four-step butterfly inside the row (quad permutes, half-row and row mirrors) leaves
// the first maximum (larger value, lower index on a tie) in every lane.  dir, hull_off and hull_n are equal across the row.
DEV int dpp_i(int v, const int ctrl) {
    switch (ctrl) {
        case 0: return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
        case 1: return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
        case 2: return __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false);  // row_half_mirror
        default: return __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false); // row_mirror
    }
}
// The block may hold ONE big hull in LDS (g_lds_hull, [vertex][xyz], filled by the kernel from hull table offset
// g_lds_hull_off): a row's 16 lanes then read 192 contiguous bytes per step instead of gathering from the constant table.
typedef float gj_f4 __attribute__((ext_vector_type(4)));
#define EVM_LDS_HULL_PTS (EVM_MAX_HULL_PTS / 2)
__device__ __shared__ gj_f4 g_lds_hull[EVM_LDS_HULL_PTS];   // [vertex] = (x, y, z, -): one ds_read_b128 per vertex
DEV F3 support_group(int hull_off, int hull_n, F3 dir, int lds_hull_off) {
    const int sub = (int) (threadIdx.x & 15);
    float best = -GJ_LARGE;
    int bi = 0x7fffffff;
    // lds_hull_off == -2: the block holds the WHOLE hull table in LDS (every hull of the skeleton), else the one hull at that offset
    const bool in_lds = lds_hull_off == -2 || hull_off == lds_hull_off;
    const gj_f4 *lh = g_lds_hull + (lds_hull_off == -2 ? hull_off : 0);
    if (in_lds) {
        // four vertices per lane and trip, the next trip's reads in flight while this one's dot products run
        gj_f4 cur[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { const int v = 16 * k + sub; cur[k] = lh[v < hull_n ? v : 0]; }
        for (int v0 = 0; __any(v0 < hull_n); v0 += 64) {
            gj_f4 nxt[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { const int v = v0 + 64 + 16 * k + sub; nxt[k] = lh[v < hull_n ? v : 0]; }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int v = v0 + 16 * k + sub;
                const float d = (dir.x * cur[k][0] + dir.y * cur[k][1]) + dir.z * cur[k][2];
                if (v < hull_n && d > best) { best = d; bi = v; }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) cur[k] = nxt[k];
        }
    } else {
        for (int v = sub; __any(v < hull_n); v += 16) {
            if (v < hull_n) {
                const int g = hull_off + v, hb = 6 * (g >> 1) + (g & 1);
                const float d = (dir.x * c_skel.hull[hb] + dir.y * c_skel.hull[hb + 2]) + dir.z * c_skel.hull[hb + 4];
                if (d > best) { best = d; bi = v; }
            }
        }
    }
#pragma unroll
    for (int st = 0; st < 4; st++) {
        const float ob = __int_as_float(dpp_i(__float_as_int(best), st));
        const int oi = dpp_i(bi, st);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (in_lds) { const gj_f4 w = lh[bi]; return f3(w[0], w[1], w[2]); }
    const int g = hull_off + bi, hb = 6 * (g >> 1) + (g & 1);
    return f3(c_skel.hull[hb], c_skel.hull[hb + 2], c_skel.hull[hb + 4]);
}

// ... and by a FULL wavefront working on one query (the penetration-depth solver, epa_dev.h): lane l takes vertices l, l + 64, ...;
// a butterfly inside every 16-lane row, then the four rows' winners (read from their first lanes) meet in wave-uniform registers.
// First maximum (larger value, lower index on a tie), like support() and support_group().  dir is equal across the wavefront.
DEV F3 support_wave(int hull_off, int hull_n, F3 dir, int lds_hull_off) {
    const int lane = (int) (threadIdx.x & 63);
    float best = -GJ_LARGE;
    int bi = 0x7fffffff;
    const bool in_lds = lds_hull_off == -2 || hull_off == lds_hull_off;
    const gj_f4 *lh = g_lds_hull + (lds_hull_off == -2 ? hull_off : 0);
    if (in_lds) {
        // four vertices per lane and trip, the next trip's reads in flight while this one's dot products run (a 451-vertex foot:
        // two trips; one LDS round trip of latency instead of eight)
        gj_f4 cur[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { const int v = 64 * k + lane; cur[k] = lh[v < hull_n ? v : 0]; }
        for (int v0 = 0; v0 < hull_n; v0 += 256) {   // (hull_n is wave-uniform)
            gj_f4 nxt[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { const int v = v0 + 256 + 64 * k + lane; nxt[k] = lh[v < hull_n ? v : 0]; }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int v = v0 + 64 * k + lane;
                const float d = (dir.x * cur[k][0] + dir.y * cur[k][1]) + dir.z * cur[k][2];
                if (v < hull_n && d > best) { best = d; bi = v; }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) cur[k] = nxt[k];
        }
    } else
    for (int v = lane; v < hull_n; v += 64) {
        const int g = hull_off + v, hb = 6 * (g >> 1) + (g & 1);
        const float x = c_skel.hull[hb], y = c_skel.hull[hb + 2], z = c_skel.hull[hb + 4];
        const float d = (dir.x * x + dir.y * y) + dir.z * z;
        if (d > best) { best = d; bi = v; }
    }
#pragma unroll
    for (int st = 0; st < 4; st++) {
        const float ob = __int_as_float(dpp_i(__float_as_int(best), st));
        const int oi = dpp_i(bi, st);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    float wb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(best), 0));
    int wi = __builtin_amdgcn_readlane(bi, 0);
#pragma unroll
    for (int r = 1; r < 4; r++) {
        const float ob = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(best), 16 * r));
        const int oi = __builtin_amdgcn_readlane(bi, 16 * r);
        if (ob > wb || (ob == wb && oi < wi)) { wb = ob; wi = oi; }
    }
    if (in_lds) { const gj_f4 w = lh[wi]; return f3(w[0], w[1], w[2]); }
    const int g = hull_off + wi, hb = 6 * (g >> 1) + (g & 1);
    return f3(c_skel.hull[hb], c_skel.hull[hb + 2], c_skel.hull[hb + 4]);
}

// Support vertex of a hull held in LDS for a direction of the lane's own (all lanes read the same vertex at a time: a broadcast
// read); first maximum in table order like support() — used by the penetration branch of the grouped form
DEV F3 support_lane_lds(int hull_off, int hull_n, F3 dir) {
    const gj_f4 *lh = g_lds_hull + hull_off;
    float best = -GJ_LARGE;
    int bi = 0;
    for (int v = 0; v < hull_n; v++) {
        const gj_f4 w = lh[v];
        const float d = (dir.x * w[0] + dir.y * w[1]) + dir.z * w[2];
        if (d > best) { best = d; bi = v; }
    }
    const gj_f4 w = lh[bi];
    return f3(w[0], w[1], w[2]);
}

struct Shape {  // one member's hull and world transform (the basis may be non-orthonormal in the step that follows reset())
    int hull_off, hull_n;
    int lds_hull_off;  // GROUP mode: hull table offset of the hull the block holds in LDS (-1: none)
    M33 R;
    F3 o;
    int *pen_count;    // EnvDev::errs + 2: queries that went through the penetration-depth solver (evm_env_get_pair_counters)
    int *spec;         // SOLO form: the slot in which a partner block leaves this query's penetration answer (speculate_pen_depth), or null
    int spec_epoch;
#ifdef EVM_KSTAMPS
    unsigned long long *ks;  // d.stamps: [17] queries that took the penetration branch, [18] wavefronts with such a query, [19] GJK iterations of the slowest lane summed over waves
#endif
};
// w = support_A(-axis) - support_B(axis) in world space, with the two support points.  GROUP: the 16 lanes of a row work
// on one query (identical state in all of them) and share the support scans.
template <bool GROUP>
DEV void minkowski(const Shape &A, F3 oA, const Shape &B, F3 oB, F3 axis, F3 &pW, F3 &qW) {
    const F3 sA = gj::vmul(gj::neg(axis), A.R), sB = gj::vmul(axis, B.R);
    pW = gj::xform(A.R, oA, GROUP ? gj::support_group(A.hull_off, A.hull_n, sA, A.lds_hull_off) : gj::support(A.hull_off, A.hull_n, sA));
    qW = gj::xform(B.R, oB, GROUP ? gj::support_group(B.hull_off, B.hull_n, sB, A.lds_hull_off) : gj::support(B.hull_off, B.hull_n, sB));
}

// ---- btVoronoiSimplexSolver on registers -------------------------------------------------------------------------------
struct Bary {  // btSubSimplexClosestResult
    F3 closest;
    bool uA, uB, uC, uD, degenerate;
    float b0, b1, b2, b3;
};
DEV bool bary_valid(const Bary &r) { return r.b0 >= 0.f && r.b1 >= 0.f && r.b2 >= 0.f && r.b3 >= 0.f; }
DEV void bary_reset(Bary &r) { r.degenerate = false; r.b0 = r.b1 = r.b2 = r.b3 = 0.f; r.uA = r.uB = r.uC = r.uD = false; }

DEV void closest_triangle(F3 p, F3 a, F3 b, F3 c, Bary &r) {
    r.uA = r.uB = r.uC = r.uD = false;
    const F3 ab = gj::sub(b, a), ac = gj::sub(c, a), ap = gj::sub(p, a);
    const float d1 = gj::dot(ab, ap), d2 = gj::dot(ac, ap);
    if (d1 <= 0.f && d2 <= 0.f) { r.closest = a; r.uA = true; r.b0 = 1; r.b1 = 0; r.b2 = 0; r.b3 = 0; return; }
    const F3 bp = gj::sub(p, b);
    const float d3 = gj::dot(ab, bp), d4 = gj::dot(ac, bp);
    if (d3 >= 0.f && d4 <= d3) { r.closest = b; r.uB = true; r.b0 = 0; r.b1 = 1; r.b2 = 0; r.b3 = 0; return; }
    const float vc = d1 * d4 - d3 * d2;
    if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) {
        const float v = d1 / (d1 - d3);
        r.closest = gj::add(a, gj::scl(ab, v)); r.uA = true; r.uB = true; r.b0 = 1 - v; r.b1 = v; r.b2 = 0; r.b3 = 0;
        return;
    }
    const F3 cp = gj::sub(p, c);
    const float d5 = gj::dot(ab, cp), d6 = gj::dot(ac, cp);
    if (d6 >= 0.f && d5 <= d6) { r.closest = c; r.uC = true; r.b0 = 0; r.b1 = 0; r.b2 = 1; r.b3 = 0; return; }
    const float vb = d5 * d2 - d1 * d6;
    if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) {
        const float w = d2 / (d2 - d6);
        r.closest = gj::add(a, gj::scl(ac, w)); r.uA = true; r.uC = true; r.b0 = 1 - w; r.b1 = 0; r.b2 = w; r.b3 = 0;
        return;
    }
    const float va = d3 * d6 - d5 * d4;
    if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {
        const float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        r.closest = gj::add(b, gj::scl(gj::sub(c, b), w)); r.uB = true; r.uC = true; r.b0 = 0; r.b1 = 1 - w; r.b2 = w; r.b3 = 0;
        return;
    }
    const float denom = 1.0f / (va + vb + vc);
    const float v = vb * denom, w = vc * denom;
    r.closest = gj::add(gj::add(a, gj::scl(ab, v)), gj::scl(ac, w));
    r.uA = true; r.uB = true; r.uC = true;
    r.b0 = 1 - v - w; r.b1 = v; r.b2 = w; r.b3 = 0;
}
// -1 degenerate, 0 inside, 1 outside
DEV int outside_of_plane(F3 p, F3 a, F3 b, F3 c, F3 d) {
    const F3 normal = gj::cross(gj::sub(b, a), gj::sub(c, a));
    const float signp = gj::dot(gj::sub(p, a), normal);
    const float signd = gj::dot(gj::sub(d, a), normal);
    if (signd * signd < (1e-4f * 1e-4f)) return -1;
    return signp * signd < 0.f ? 1 : 0;
}
DEV int dpp_i(int v, const int ctrl);
template <bool GROUP>
DEV bool closest_tetrahedron(F3 p, F3 a, F3 b, F3 c, F3 d, Bary &f) {
    Bary t;
    f.closest = p;
    f.uA = f.uB = f.uC = f.uD = true;
    const int oABC = outside_of_plane(p, a, b, c, d), oACD = outside_of_plane(p, a, c, d, b);
    const int oADB = outside_of_plane(p, a, d, b, c), oBDC = outside_of_plane(p, b, d, c, a);
    if (oABC < 0 || oACD < 0 || oADB < 0 || oBDC < 0) { f.degenerate = true; return false; }
    if (!oABC && !oACD && !oADB && !oBDC) return false;
    float best = EVM_INF;
    if (GROUP) {
        // One query per 16-lane row, identical state in its lanes: lane k of every quad takes face k (ABC, ACD, ADB, BDC), then a
        // two-step exchange inside the quad leaves the sequential loop's winner (smallest squared distance, the FIRST face on a
        // tie: `sq < best` is strict) in all of them — one closest_triangle instead of up to four.
        const int fi = (int) (threadIdx.x & 3);
        const F3 x = sel3(fi == 3, b, a);
        const F3 y = sel3(fi == 0, b, sel3(fi == 1, c, d));
        const F3 z = sel3(fi == 0, c, sel3(fi == 1, d, sel3(fi == 2, b, c)));
        const bool mine = (fi == 0 ? oABC : (fi == 1 ? oACD : (fi == 2 ? oADB : oBDC))) != 0;
        closest_triangle(p, x, y, z, t);
        const F3 q = t.closest;
        float sq = mine ? gj::dot(gj::sub(q, p), gj::sub(q, p)) : EVM_INF;
        // the face's result in tetrahedron terms (the four blocks of the sequential form)
        const bool uA = fi != 3 && t.uA, uB = fi == 0 ? t.uB : (fi == 2 ? t.uC : (fi == 3 ? t.uA : false));
        const bool uC = fi == 0 ? t.uC : (fi == 1 ? t.uB : (fi == 3 ? t.uC : false)), uD = fi == 1 ? t.uC : (fi == 0 ? false : t.uB);
        float w0 = fi == 3 ? 0.f : t.b0;
        float w1 = fi == 0 ? t.b1 : (fi == 2 ? t.b2 : (fi == 3 ? t.b0 : 0.f));
        float w2 = fi == 0 ? t.b2 : (fi == 1 ? t.b1 : (fi == 3 ? t.b2 : 0.f));
        float w3 = fi == 1 ? t.b2 : (fi == 0 ? 0.f : t.b1);
        int fl = (uA ? 1 : 0) | (uB ? 2 : 0) | (uC ? 4 : 0) | (uD ? 8 : 0), face = fi;
        F3 cl = q;
#pragma unroll
        for (int st = 0; st < 2; st++) {
            const float osq = __int_as_float(dpp_i(__float_as_int(sq), st));
            const int oface = dpp_i(face, st), ofl = dpp_i(fl, st);
            const float ox = __int_as_float(dpp_i(__float_as_int(cl.x), st)), oy = __int_as_float(dpp_i(__float_as_int(cl.y), st));
            const float oz = __int_as_float(dpp_i(__float_as_int(cl.z), st));
            const float o0 = __int_as_float(dpp_i(__float_as_int(w0), st)), o1 = __int_as_float(dpp_i(__float_as_int(w1), st));
            const float o2 = __int_as_float(dpp_i(__float_as_int(w2), st)), o3 = __int_as_float(dpp_i(__float_as_int(w3), st));
            const bool take = osq < sq || (osq == sq && oface < face);
            sq = take ? osq : sq; face = take ? oface : face; fl = take ? ofl : fl;
            cl = sel3(take, f3(ox, oy, oz), cl);
            w0 = take ? o0 : w0; w1 = take ? o1 : w1; w2 = take ? o2 : w2; w3 = take ? o3 : w3;
        }
        if (sq < best) {
            f.closest = cl; f.uA = (fl & 1) != 0; f.uB = (fl & 2) != 0; f.uC = (fl & 4) != 0; f.uD = (fl & 8) != 0;
            f.b0 = w0; f.b1 = w1; f.b2 = w2; f.b3 = w3;
        }
        return true;
    }
    if (oABC) {
        closest_triangle(p, a, b, c, t);
        const F3 q = t.closest;
        const float sq = gj::dot(gj::sub(q, p), gj::sub(q, p));
        if (sq < best) { best = sq; f.closest = q; f.uA = t.uA; f.uB = t.uB; f.uC = t.uC; f.uD = false; f.b0 = t.b0; f.b1 = t.b1; f.b2 = t.b2; f.b3 = 0; }
    }
    if (oACD) {
        closest_triangle(p, a, c, d, t);
        const F3 q = t.closest;
        const float sq = gj::dot(gj::sub(q, p), gj::sub(q, p));
        if (sq < best) { best = sq; f.closest = q; f.uA = t.uA; f.uB = false; f.uC = t.uB; f.uD = t.uC; f.b0 = t.b0; f.b1 = 0; f.b2 = t.b1; f.b3 = t.b2; }
    }
    if (oADB) {
        closest_triangle(p, a, d, b, t);
        const F3 q = t.closest;
        const float sq = gj::dot(gj::sub(q, p), gj::sub(q, p));
        if (sq < best) { best = sq; f.closest = q; f.uA = t.uA; f.uB = t.uC; f.uC = false; f.uD = t.uB; f.b0 = t.b0; f.b1 = t.b2; f.b2 = 0; f.b3 = t.b1; }
    }
    if (oBDC) {
        closest_triangle(p, b, d, c, t);
        const F3 q = t.closest;
        const float sq = gj::dot(gj::sub(q, p), gj::sub(q, p));
        if (sq < best) { best = sq; f.closest = q; f.uA = false; f.uB = t.uA; f.uC = t.uC; f.uD = t.uB; f.b0 = 0; f.b1 = t.b0; f.b2 = t.b2; f.b3 = t.b1; }
    }
    return true;
}

struct Simplex {
    int n;
    F3 W[4], P[4], Q[4];
    F3 cP1, cP2, cV, lastW;
    bool cachedValid;
};
DEV void sx_reset(Simplex &s) {
    s.n = 0; s.cachedValid = false;
    s.lastW = f3(GJ_LARGE, GJ_LARGE, GJ_LARGE);
#pragma unroll
    for (int i = 0; i < 4; i++) { s.W[i] = f3(0, 0, 0); s.P[i] = f3(0, 0, 0); s.Q[i] = f3(0, 0, 0); }
    s.cP1 = s.cP2 = s.cV = f3(0, 0, 0);
}
DEV void sx_add(Simplex &s, F3 w, F3 p, F3 q) {
    s.lastW = w;
#pragma unroll
    for (int i = 0; i < 4; i++) { const bool h = s.n == i; s.W[i] = sel3(h, w, s.W[i]); s.P[i] = sel3(h, p, s.P[i]); s.Q[i] = sel3(h, q, s.Q[i]); }
    s.n++;
}
// reduceVertices: removeVertex(3), (2), (1), (0) for the unused ones, in that order, each moving the LAST vertex into the freed
// slot.  The four conditional removals are simulated on slot indices (four 2-bit fields) and the vertices gathered once:
// as four select cascades over the 36 simplex registers each, they were a sixth of a GJK iteration.
DEV void sx_reduce(Simplex &s, const Bary &u) {
    int n = s.n;
    unsigned idx = 0xE4u;  // slot k holds original vertex (idx >> 2k) & 3: identity 3,2,1,0
    auto rem = [&](int i) {
        n--;
        const unsigned last = (idx >> (2 * n)) & 3u;
        idx = (idx & ~(3u << (2 * i))) | (last << (2 * i));
    };
    if (n >= 4 && !u.uD) rem(3);
    if (n >= 3 && !u.uC) rem(2);
    if (n >= 2 && !u.uB) rem(1);
    if (n >= 1 && !u.uA) rem(0);
    s.n = n;
    const F3 W0 = s.W[0], W1 = s.W[1], W2 = s.W[2], W3 = s.W[3], P0 = s.P[0], P1 = s.P[1], P2 = s.P[2], P3 = s.P[3];
    const F3 Q0 = s.Q[0], Q1 = s.Q[1], Q2 = s.Q[2], Q3 = s.Q[3];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int j = (int) ((idx >> (2 * k)) & 3u);
        s.W[k] = sel3(j == 0, W0, sel3(j == 1, W1, sel3(j == 2, W2, W3)));
        s.P[k] = sel3(j == 0, P0, sel3(j == 1, P1, sel3(j == 2, P2, P3)));
        s.Q[k] = sel3(j == 0, Q0, sel3(j == 1, Q1, sel3(j == 2, Q2, Q3)));
    }
}
DEV bool sx_in_simplex(const Simplex &s, F3 w) {
    bool found = false;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const F3 d = gj::sub(s.W[i], w);
        if (i < s.n && !found && gj::dot(d, d) <= GJ_EQUAL_VERTEX) found = true;
    }
    if (w.x == s.lastW.x && w.y == s.lastW.y && w.z == s.lastW.z) return true;
    return found;
}
// updateClosestVectorAndPoints (called right after every addVertex, so the cached values are never stale)
template <bool GROUP>
DEV bool sx_closest(Simplex &s, F3 &v) {
    Bary bc;
    bary_reset(bc);
    const F3 zero = f3(0.f, 0.f, 0.f);
    if (s.n == 1) {
        s.cP1 = s.P[0]; s.cP2 = s.Q[0]; s.cV = gj::sub(s.cP1, s.cP2);
        bc.b0 = 1;
        s.cachedValid = bary_valid(bc);
    } else if (s.n == 2) {
        const F3 from = s.W[0], to = s.W[1];
        F3 diff = gj::sub(zero, from);
        const F3 vv = gj::sub(to, from);
        float t = gj::dot(vv, diff);
        if (t > 0.f) {
            const float dotVV = gj::dot(vv, vv);
            if (t < dotVV) { t /= dotVV; diff = gj::sub(diff, gj::scl(vv, t)); bc.uA = true; bc.uB = true; }
            else { t = 1; diff = gj::sub(diff, vv); bc.uB = true; }
        } else { t = 0; bc.uA = true; }
        bc.b0 = 1 - t; bc.b1 = t;
        s.cP1 = gj::add(s.P[0], gj::scl(gj::sub(s.P[1], s.P[0]), t));
        s.cP2 = gj::add(s.Q[0], gj::scl(gj::sub(s.Q[1], s.Q[0]), t));
        s.cV = gj::sub(s.cP1, s.cP2);
        sx_reduce(s, bc);
        s.cachedValid = bary_valid(bc);
    } else if (s.n == 3) {
        closest_triangle(zero, s.W[0], s.W[1], s.W[2], bc);
        s.cP1 = gj::add(gj::add(gj::scl(s.P[0], bc.b0), gj::scl(s.P[1], bc.b1)), gj::scl(s.P[2], bc.b2));
        s.cP2 = gj::add(gj::add(gj::scl(s.Q[0], bc.b0), gj::scl(s.Q[1], bc.b1)), gj::scl(s.Q[2], bc.b2));
        s.cV = gj::sub(s.cP1, s.cP2);
        sx_reduce(s, bc);
        s.cachedValid = bary_valid(bc);
    } else if (s.n == 4) {
        const bool sep = closest_tetrahedron<GROUP>(zero, s.W[0], s.W[1], s.W[2], s.W[3], bc);
        if (sep) {
            s.cP1 = gj::add(gj::add(gj::add(gj::scl(s.P[0], bc.b0), gj::scl(s.P[1], bc.b1)), gj::scl(s.P[2], bc.b2)), gj::scl(s.P[3], bc.b3));
            s.cP2 = gj::add(gj::add(gj::add(gj::scl(s.Q[0], bc.b0), gj::scl(s.Q[1], bc.b1)), gj::scl(s.Q[2], bc.b2)), gj::scl(s.Q[3], bc.b3));
            s.cV = gj::sub(s.cP1, s.cP2);
            sx_reduce(s, bc);
            s.cachedValid = bary_valid(bc);
        } else if (bc.degenerate) s.cachedValid = false;
        else { s.cachedValid = true; s.cV = zero; }
    } else s.cachedValid = false;
    v = s.cV;
    return s.cachedValid;
}

struct Result {
    bool has;        // a point for btManifoldResult::addContactPoint
    F3 normalOnB, pointOnB;
    float distance;
    int iterations;
    bool used_pen;
};

// One btGjkPairDetector::getClosestPointsNonVirtual run for the lanes in `active`.  oA / oB: the two origins already shifted by
// positionOffset.  Without the penetration branch (that is the caller's, between two runs).  Outputs per lane.
struct RunOut {
    bool isValid;
    float distance, squaredDistance;
    F3 normalInB, pointOnA, pointOnB, axis;
    int degenerate, iters;
};
template <bool GROUP>
DEV RunOut gjk_run(const Shape &A, F3 oA, const Shape &B, F3 oB, float max_dist2, bool active, float marginA, float marginB) {
    RunOut o;
    o.isValid = false; o.distance = 0.f; o.squaredDistance = GJ_LARGE; o.degenerate = 0; o.iters = 0;
    o.normalInB = f3(0, 0, 0); o.pointOnA = f3(0, 0, 0); o.pointOnB = f3(0, 0, 0);
    F3 axis = f3(0.f, 1.f, 0.f);
    bool checkSimplex = false, running = active;
    float squaredDistance = GJ_LARGE;
    const float margin = marginA + marginB;
    Simplex sx;
    sx_reset(sx);
    int cur_iter = 0, degenerate = 0;
#ifdef EVM_KSTAMPS
    unsigned long long ks_sup = 0, ks_rest = 0, ks_trips = 0;
#endif
    while (__any(running)) {
        F3 pW, qW;
#ifdef EVM_KSTAMPS
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long ks_a = __builtin_amdgcn_s_memtime();
#endif
        minkowski<GROUP>(A, oA, B, oB, axis, pW, qW);  // (lanes that have left keep their last axis: harmless, results unused)
#ifdef EVM_KSTAMPS
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long ks_b = __builtin_amdgcn_s_memtime();
#endif
        if (running) {
            const F3 w = gj::sub(pW, qW);
            const float delta = gj::dot(axis, w);
            const float f0 = squaredDistance - delta, f1 = squaredDistance * GJ_REL_ERROR2;
            if (delta > 0.f && delta * delta > squaredDistance * max_dist2) { degenerate = 10; checkSimplex = true; running = false; }
            else if (sx_in_simplex(sx, w)) { degenerate = 1; checkSimplex = true; running = false; }
            else if (f0 <= f1) { degenerate = f0 <= 0.f ? 2 : 11; checkSimplex = true; running = false; }
            else {
                sx_add(sx, w, pW, qW);
                F3 newAxis;
                if (!sx_closest<GROUP>(sx, newAxis)) { degenerate = 3; checkSimplex = true; running = false; }
                else if (gj::len2(newAxis) < GJ_REL_ERROR2) { axis = newAxis; degenerate = 6; checkSimplex = true; running = false; }
                else {
                    const float prev = squaredDistance;
                    squaredDistance = gj::len2(newAxis);
                    if (prev - squaredDistance <= EVM_EPS * prev) { checkSimplex = true; degenerate = 12; running = false; }
                    else {
                        axis = newAxis;
                        if (cur_iter++ > GJ_MAX_ITER) running = false;
                        else if (sx.n == 4) { degenerate = 13; running = false; }
                    }
                }
            }
        }
#ifdef EVM_KSTAMPS
        __builtin_amdgcn_s_waitcnt(0);
        {
            const unsigned long long ks_c = __builtin_amdgcn_s_memtime();
            ks_sup += ks_b - ks_a; ks_rest += ks_c - ks_b; ks_trips++;
        }
#endif
    }
#ifdef EVM_KSTAMPS
    if (threadIdx.x == 0) { atomicAdd(&A.ks[20 + (GROUP ? 0 : 4)], ks_sup); atomicAdd(&A.ks[21 + (GROUP ? 0 : 4)], ks_rest); atomicAdd(&A.ks[22 + (GROUP ? 0 : 4)], ks_trips); }
#endif
    if (checkSimplex) {
        F3 pointOnA = sx.cP1, pointOnB = sx.cP2;  // compute_points: the cached pair of the last closest()
        F3 normalInB = axis;
        const float lenSqr = gj::len2(axis);
        if (lenSqr < GJ_REL_ERROR2) degenerate = 5;
        if (lenSqr > EVM_EPS * EVM_EPS) {
            const float rlen = 1.0f / sqrtf(lenSqr);
            normalInB = gj::scl(normalInB, rlen);
            const float s = sqrtf(squaredDistance);
            pointOnA = gj::sub(pointOnA, gj::scl(axis, marginA / s));
            pointOnB = gj::add(pointOnB, gj::scl(axis, marginB / s));
            o.distance = (1.0f / rlen) - margin;
            o.isValid = true;
            o.normalInB = normalInB;
            o.pointOnA = pointOnA; o.pointOnB = pointOnB;
        }
    }
    o.axis = axis;
    o.degenerate = degenerate;
    o.squaredDistance = squaredDistance;
    o.iters = cur_iter;
    return o;
}

// The penetration solver sits behind a real call (epa::calc_pen_depth_call): inlined, its registers competed with the common path's
// around the call site and every launch paid for the spills (no-EPA launches 70-80 us -> 85-95 us).
#define EPA_NOINLINE 1
#include "epa_dev.h"

// ---- penetration queries BESIDE the pair's own query ----------------------------------------------------------------------------
// Even first in line (the urgent list, pairs_dev.h) a query that goes through the solver is ~75 us of one wavefront's dependent
// instructions AFTER ~25 us of set-up and GJK, and ~15 us of normal check, manifold and contact record follow: longer than everything
// else in the kernel (~80 us).  But calcPenDepth's answer depends on the two shapes and transforms only (its nine guess vectors come from
// the origins, not from the GJK's axis).  So every entry of the urgent list is taken TWICE: by its urgent block, as before, and by a
// SPECULATION block (the launch's very first blocks, env_kernels.hip: narrow_block) that runs calcPenDepth at once and leaves the
// answer in the entry's SLOT (EVM_SPEC_WORDS ints):
//   [0]       E once the answer is in place (E = the launch's epoch: the host counts the steps, so nothing is ever reset)
//   [1..10]   epa::PenOut
//   [11]      the owner's word to the speculative run: (E << 2) | 1 "wanted" (the run moves to the front of its SIMD's issue; until
//             then it has the ordinary priority, so a run that turns out unneeded never starves a neighbour), | 2 "not wanted" (the
//             run stops at its next EPA round: nine runs out of ten)
// Same kernel, same state, same function, same arguments: the answer is the one the call in place would give, bit for bit.  The
// wait is bounded (the speculation blocks have the lower block indices, so they are running or done; but nothing here may hang):
// past the bound the query is run in place, as it always was.
DEV void speculate_pen_depth(const Shape &A, const Shape &B, int lds_hull_off, int *slot, int epoch) {
    const bool lead = (threadIdx.x & 63) == 0;
    const F3 positionOffset = gj::scl(gj::add(A.o, B.o), 0.5f);
    const F3 oA = gj::sub(A.o, positionOffset), oB = gj::sub(B.o, positionOffset);
#ifdef EVM_KSTAMPS
    void *const ksp = (void *) A.ks;
#else
    void *const ksp = nullptr;
#endif
    const epa::PenOut po = epa::calc_pen_depth_call<true>(A.hull_off, A.hull_n, B.hull_off, B.hull_n, lds_hull_off, A.R.r0.x, A.R.r0.y, A.R.r0.z, A.R.r1.x, A.R.r1.y,
                                                          A.R.r1.z, A.R.r2.x, A.R.r2.y, A.R.r2.z, oA.x, oA.y, oA.z, B.R.r0.x, B.R.r0.y, B.R.r0.z, B.R.r1.x, B.R.r1.y,
                                                          B.R.r1.z, B.R.r2.x, B.R.r2.y, B.R.r2.z, oB.x, oB.y, oB.z, ksp, slot + 11, epoch);
    __builtin_amdgcn_s_setprio(0);
    if (lead && !(po.flags & 4)) {
        float *w = reinterpret_cast<float *>(slot);
        w[1] = po.vx; w[2] = po.vy; w[3] = po.vz; w[4] = po.ax; w[5] = po.ay; w[6] = po.az; w[7] = po.bx; w[8] = po.by; w[9] = po.bz;
        slot[10] = po.flags;
        __hip_atomic_store(slot, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

DEV bool ub1(bool c) { return __builtin_amdgcn_readfirstlane(c ? 1 : 0) != 0; }
DEV int ui(int v) { return __builtin_amdgcn_readfirstlane(v); }
// btGjkPairDetector::getClosestPoints for the lanes in `active` (the others return has = false)
// SOLO (grouped form only): all 64 lanes of the wavefront carry the SAME query (the urgent list's blocks, pairs_dev.h)
template <bool GROUP, bool SOLO = false>
DEV Result closest_points(const Shape &A, const Shape &B, float max_dist2, bool active) {
    const float marginA = MARGIN_F, marginB = MARGIN_F, margin = marginA + marginB;
    Result out;
    out.has = false; out.normalOnB = f3(0, 0, 0); out.pointOnB = f3(0, 0, 0); out.distance = 0.f; out.iterations = 0; out.used_pen = false;
    const F3 positionOffset = gj::scl(gj::add(A.o, B.o), 0.5f);
    const F3 oA = gj::sub(A.o, positionOffset), oB = gj::sub(B.o, positionOffset);
    const RunOut r = gjk_run<GROUP>(A, oA, B, oB, max_dist2, active, marginA, marginB);
    if (SOLO) { UST(3) }
    bool isValid = r.isValid;
    float distance = r.distance;
    F3 normalInB = r.normalInB, pointOnB = r.pointOnB;
    const F3 orgNormalInB = r.isValid ? r.normalInB : f3(0.f, 0.f, 0.f);
    out.iterations = r.iters;
    const bool catchDegenerate = r.degenerate != 0 && (distance + margin) < GJ_PEN_TOLERANCE;
    const bool need_pen = active && (!isValid || catchDegenerate);
#ifdef EVM_KSTAMPS
    {
        const unsigned long long m = __ballot(need_pen);
        int it = r.iters;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) it = max(it, __shfl_xor(it, o));
        if (threadIdx.x == 0) {
            if (m) { atomicAdd(&A.ks[17], (unsigned long long) __popcll(m) / (GROUP ? 16 : 1)); atomicAdd(&A.ks[18], 1ull); }
            atomicAdd(&A.ks[19], (unsigned long long) it);
        }
    }
#endif
    if (SOLO && A.spec != nullptr && !__any(need_pen) && (threadIdx.x & 63) == 0)   // the speculative run's answer is not wanted: call it off
        __hip_atomic_store(A.spec + 11, (A.spec_epoch << 2) | 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__any(need_pen)) {
        // btGjkEpaPenetrationDepthSolver::calcPenDepth (epa_dev.h): the wavefront takes its queries that need it one at a time and
        // works on each together — the query's transforms broadcast, the answer handed back to the lane(s) that own it.
        const int lane = (int) (threadIdx.x & 63);
        unsigned long long todo = __ballot(need_pen);
        if (SOLO) todo &= 1ull;   // (one query in the whole wavefront)
        while (todo) {
            const int src = (int) __builtin_ctzll(todo);
            todo &= GROUP ? ~(0xFFFFull << (src & ~15)) : todo - 1;   // (grouped: the 16 lanes of a row are one query)
            auto bc = [&](float x) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), src)); };
            auto bc3 = [&](F3 x) { return f3(bc(x.x), bc(x.y), bc(x.z)); };
            Shape As = A, Bs = B;
            As.hull_off = __builtin_amdgcn_readlane(A.hull_off, src); As.hull_n = __builtin_amdgcn_readlane(A.hull_n, src);   // (grouped: rows hold different pairs)
            Bs.hull_off = __builtin_amdgcn_readlane(B.hull_off, src); Bs.hull_n = __builtin_amdgcn_readlane(B.hull_n, src);
            As.R = m33(bc3(A.R.r0), bc3(A.R.r1), bc3(A.R.r2)); Bs.R = m33(bc3(B.R.r0), bc3(B.R.r1), bc3(B.R.r2));
            const F3 oAs = bc3(oA), oBs = bc3(oB);
            F3 sep, tmpA, tmpB;
            bool has_v;
#ifdef EPA_NOINLINE
#ifdef EVM_KSTAMPS
            void *const ksp = (void *) A.ks;
#else
            void *const ksp = nullptr;
#endif
            epa::PenOut po;
            bool answered = false;
            if (SOLO && A.spec != nullptr) {
                // the entry's speculation block has been on this very query since the launch began (above): tell it the answer is
                // wanted and wait for it.  (The polls are relaxed loads a microsecond apart — an acquire in a loop would invalidate
                // caches under everybody else's feet — with one acquire fence once the value is seen.)
                const int E = A.spec_epoch;
                if (lane == 0) __hip_atomic_store(A.spec + 11, (E << 2) | 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int w0 = 0;
                for (int tries = 0; tries < 4096; tries++) {
                    w0 = ui(__hip_atomic_load(A.spec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if (w0 == E) break;
                    __builtin_amdgcn_s_sleep(16);
                }
                answered = w0 == E;
                if (lane == 0 && A.pen_count != nullptr) atomicAdd(A.pen_count + (answered ? 4 : 5), 1);   // errs[6] answers used, [7] waits that ran out
                if (answered) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    const float *w = reinterpret_cast<const float *>(A.spec);
                    po.vx = w[1]; po.vy = w[2]; po.vz = w[3]; po.ax = w[4]; po.ay = w[5]; po.az = w[6]; po.bx = w[7]; po.by = w[8]; po.bz = w[9];
                    po.flags = A.spec[10];
                } else if (lane == 0) __hip_atomic_store(A.spec + 11, (E << 2) | 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (!answered)
                po = epa::calc_pen_depth_call<GROUP>(As.hull_off, As.hull_n, Bs.hull_off, Bs.hull_n, A.lds_hull_off, As.R.r0.x, As.R.r0.y, As.R.r0.z,
                                                     As.R.r1.x, As.R.r1.y, As.R.r1.z, As.R.r2.x, As.R.r2.y, As.R.r2.z, oAs.x, oAs.y, oAs.z, Bs.R.r0.x,
                                                     Bs.R.r0.y, Bs.R.r0.z, Bs.R.r1.x, Bs.R.r1.y, Bs.R.r1.z, Bs.R.r2.x, Bs.R.r2.y, Bs.R.r2.z, oBs.x,
                                                     oBs.y, oBs.z, ksp, nullptr, 0);
            sep = f3(po.vx, po.vy, po.vz); tmpA = f3(po.ax, po.ay, po.az); tmpB = f3(po.bx, po.by, po.bz);
            has_v = (po.flags & 2) != 0;
            const bool isValid2 = (po.flags & 1) != 0;
#else
            bool called_off;
            const bool isValid2 = epa::calc_pen_depth<GROUP>(As, oAs, Bs, oBs, sep, tmpA, tmpB, has_v, nullptr, 0, called_off);
#endif
            __builtin_amdgcn_s_setprio(0);
            const bool mine = SOLO ? true : (GROUP ? (lane >> 4) == (src >> 4) : lane == src);
            if (mine) {
                out.used_pen = true;
                if (SOLO ? lane == 0 : (!GROUP || (lane & 15) == 0)) { atomicAdd(A.pen_count, 1); if (SOLO) atomicAdd(A.pen_count + 1, 1); }   // [+1]: ... of them predicted (urgent list)
                if (has_v && gj::len2(sep) != 0.f) {
                    if (isValid2) {
                        F3 tmpN = gj::sub(tmpB, tmpA);
                        float lenSqr = gj::len2(tmpN);
                        if (lenSqr <= EVM_EPS * EVM_EPS) { tmpN = sep; lenSqr = gj::len2(sep); }
                        if (lenSqr > EVM_EPS * EVM_EPS) {
                            tmpN = gj::scl(tmpN, 1.0f / sqrtf(lenSqr));
                            const float distance2 = -sqrtf(gj::len2(gj::sub(tmpA, tmpB)));
                            if (!isValid || distance2 < distance) { distance = distance2; pointOnB = tmpB; normalInB = tmpN; isValid = true; }
                        }
                    } else {
                        // EPA found no overlap of the margin-inflated shapes, Distance() a positive distance of the cores
                        const float distance2 = sqrtf(gj::len2(gj::sub(tmpA, tmpB))) - margin;
                        if (!isValid || distance2 < distance) {
                            distance = distance2;
                            pointOnB = gj::add(tmpB, gj::scl(sep, marginB));
                            normalInB = gj::scl(sep, 1.0f / sqrtf(gj::len2(sep)));
                            isValid = true;
                        }
                    }
                }
            }
        }
    }
    if (active && isValid && (distance < 0.f || distance * distance < max_dist2)) out.has = true;
    if (__any(out.has && out.used_pen)) {
        // The normal check at the end of getClosestPointsNonVirtual: the candidate normal n, its opposite and the plain GJK
        // normal are compared by the separation they give along themselves (d0, d1, d2).  After a plain GJK result n IS the
        // GJK normal (d2 = d0 exactly) and d0 + d1 = -(width of A + width of B along n) - 2 margins - ... < 0 with d0 >= -margin >
        // d1: the check cannot change anything, so it only runs for lanes that went through the penetration branch (the
        // oracle evaluates it for every query and gets the same answer).
        F3 pW, qW;
        minkowski<GROUP>(A, oA, B, oB, normalInB, pW, qW);
        const float d0 = gj::dot(normalInB, gj::sub(pW, qW)) - margin;
        const F3 nn = gj::neg(normalInB);
        minkowski<GROUP>(A, oA, B, oB, nn, pW, qW);
        const float d1 = gj::dot(nn, gj::sub(pW, qW)) - margin;
        minkowski<GROUP>(A, oA, B, oB, orgNormalInB, pW, qW);
        const float d2 = gj::dot(orgNormalInB, gj::sub(pW, qW)) - margin;
        if (out.has && out.used_pen) {
            if (d1 > d0) normalInB = nn;
            if (gj::len2(orgNormalInB) != 0.f) {
                if (d2 > d0 && d2 > d1 && d2 > distance) { normalInB = orgNormalInB; distance = d2; }
            }
        }
    }
    out.normalOnB = normalInB;
    out.pointOnB = gj::add(pointOnB, positionOffset);
    out.distance = distance;
    return out;
}

}  // namespace gj
#pragma clang fp contract(fast)
}  // namespace evm
