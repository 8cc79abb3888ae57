// Penetration depth of two overlapping hulls: the expanding polytope algorithm, one query per WAVEFRONT.
// (Included by narrow_dev.h inside namespace evm::gj, in its no-contraction region, after Shape and the support routines.)
//
// What it replaces: the reference's world (evo_motion_model/src/environment.cpp:20-31) is built from
// btDefaultCollisionConfiguration, so every btConvexConvexAlgorithm resolves overlapping cores with
// btGjkEpaPenetrationDepthSolver::calcPenDepth -> btGjkEpaSolver2::Penetration / Distance (Bullet3's btGjkEpa2.cpp: its own GJK
// on the margin-inflated shapes in A's local frame, then EPA with at most 128 support points, 256 faces, 255 rounds).  Rounds
// 2-3 of this repository used the sampled-direction solver here instead; that deviation is gone.
//
// The branch is rare (a few of the ~18 000 queries of a 4096-env step take it) and, as a lane-private loop, hopeless: the
// polytope does not fit a lane's registers.  So the wavefront that finds such a query works on it TOGETHER: the query's two
// transforms are broadcast, every lane then runs the same scalar programme on the same values (the polytope, face list and the
// GJK's simplex live in LDS, all lanes writing identical values to identical addresses), and the data-parallel parts are dealt to
// the lanes: hull support scans (support_group's 16-lane rows for the 451-vertex feet), the visibility of every face from the
// new support point, and the construction of the horizon's new faces (one per lane).
//
// Bit-for-bit the sequential algorithm (oracle/orc_epa.cpp restates it with Bullet's own pointer structure):
//   * the face stock is a LIFO in the original too (newface takes the list root, a retired face becomes the root): a stack here;
//   * the hull list only matters through findbest's tie rule (strict `<` while walking from the most recently appended face):
//     faces carry their append sequence number and the winner is (smallest d^2, LARGEST sequence number);
//   * expand()'s recursion only needs each face's visibility (a pure function of the face and the new point, evaluated for
//     all faces up front) and adjacency; it is replayed as an explicit-stack walk that records the horizon edges in creation
//     order and pops / pushes the stock in the original's order, so every new face gets the original's index;
//   * any failure inside expand() (a face met twice, a degenerate or non-convex new face, no face left) ends the expansion with
//     the last `outer` face in the original, whatever was built until then: the geometry of the new faces can therefore be
//     checked after the walk.
#pragma once

namespace epa {

#ifdef EVM_KSTAMPS
// the urgent list's blocks (one query per wavefront): s_memtime at the way-points of one query, kept in LDS (no atomics, no waits in
// between) and flushed by the block when the query did go through the penetration solver: d.stamps[48..] (tools/kstamps.py)
__device__ __shared__ unsigned long long g_ust[12];
__device__ __shared__ int g_ust_on;
__device__ __shared__ unsigned long long g_uph[8];   // per EPA round phase, summed over the query's rounds: support, visibility, walk, new faces, findbest + reload; [5..7] hull scans of A, of B, scans
#define UPH_T0() unsigned long long uph_t = __builtin_amdgcn_s_memtime();
#define UPH(k) { const unsigned long long uph_n = __builtin_amdgcn_s_memtime(); if (::evm::gj::epa::g_ust_on && threadIdx.x == 0) ::evm::gj::epa::g_uph[k] += uph_n - uph_t; uph_t = uph_n; }
#define UST(k) { if (::evm::gj::epa::g_ust_on && threadIdx.x == 0) ::evm::gj::epa::g_ust[k] = __builtin_amdgcn_s_memtime(); }
#else
#define UST(k)
#define UPH_T0()
#define UPH(k)
#endif


#define EPA_MAXV 128
#define EPA_MAXF 256
#define EPA_NV (EPA_MAXV + 4)   // + the four vertices of the GJK's own store (ids EPA_MAXV..)
#define EPA_GJK_MAX_ITER 128
#define EPA_GJK_ACCURACY 0.0001f
#define EPA_GJK_MIN_DISTANCE 0.0001f
#define EPA_GJK_DUP_EPS 0.0001f
#define EPA_MAX_ITER 255
#define EPA_ACCURACY 0.0001f
#define EPA_PLANE_EPS 0.00001f

struct State {
    gj_f4 svd[EPA_NV];              // support direction of vertex id
    gj_f4 svw[EPA_NV];              // the vertex of the Minkowski difference
    gj_f4 fnd[EPA_MAXF];            // face normal, distance
    unsigned fadj[EPA_MAXF];        // neighbours f0 | f1 << 8 | f2 << 16, their edges e0 << 24 | e1 << 26 | e2 << 28, and in bits 30-31
                                    // this round's state: 0 beyond the horizon, 1 visible from w, 2 visible and visited (pass == pass)
    unsigned fc[EPA_MAXF];          // vertices c0 | c1 << 8 | c2 << 16
    float fkey[EPA_MAXF];           // d^2 of a face in the hull list, +inf otherwise (findbest's key)
    unsigned short fseq[EPA_MAXF];  // hull append sequence number
    unsigned char stock[EPA_MAXF];  // free faces, top of the stack = the stock list's root
    unsigned short frame[EPA_MAXF]; // walk frames f | e << 8 | stage << 10
    unsigned short hz[EPA_MAXF];    // horizon edges in creation order: f | e << 8
    unsigned char hzn[EPA_MAXF];    // ... and the face made for each
    int sc[2][4];                   // the GJK's two simplices: vertex slots,
    float sp[2][4];                 //   weights,
    int srank[2];                   //   ranks
    int gfree[4];                   // its free vertex slots
    int bad;
};
__device__ __shared__ State g_epa;

// All lanes run the same programme on the same values and write identical values to identical LDS addresses, so a lane only ever
// depends on its OWN earlier writes: no synchronisation is needed — except after the three places where the lanes write DIFFERENT
// entries (the stock fill, the visibility pass, the horizon's new faces).
#define EPA_SYNC()                                                 \
    {                                                              \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     \
        __builtin_amdgcn_wave_barrier();                           \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");     \
    }
DEV bool ub(bool c) { return __builtin_amdgcn_readfirstlane(c ? 1 : 0) != 0; }   // (every lane holds the same value: scalar control flow)
DEV int ui(int v) { return __builtin_amdgcn_readfirstlane(v); }
DEV float uf(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
DEV F3 uf3(F3 v) { return f3(uf(v.x), uf(v.y), uf(v.z)); }
DEV F3 ld3(const gj_f4 &q) { return f3(q[0], q[1], q[2]); }
DEV float flen(F3 a) { return sqrtf(gj::dot(a, a)); }
DEV F3 mmul(const M33 &m, F3 v) { return f3(gj::dot(m.r0, v), gj::dot(m.r1, v), gj::dot(m.r2, v)); }
// btMatrix3x3::transposeTimes: a^T * b
DEV M33 tmm(const M33 &a, const M33 &b) {
    const F3 a0 = f3(a.r0.x, a.r1.x, a.r2.x), a1 = f3(a.r0.y, a.r1.y, a.r2.y), a2 = f3(a.r0.z, a.r1.z, a.r2.z);
#define EPA_ROW(r) f3(b.r0.x * r.x + b.r1.x * r.y + b.r2.x * r.z, b.r0.y * r.x + b.r1.y * r.y + b.r2.y * r.z, b.r0.z * r.x + b.r1.z * r.y + b.r2.z * r.z)
    return m33(EPA_ROW(a0), EPA_ROW(a1), EPA_ROW(a2));
#undef EPA_ROW
}
DEV float det3(F3 a, F3 b, F3 c) {
    return a.y * b.z * c.x + a.z * b.x * c.y - a.x * b.z * c.y - a.y * b.x * c.z + a.x * b.y * c.z - a.z * b.y * c.x;
}

// A small hull held in the registers of a FULL wavefront (a member box: eight vertices, lane l keeps vertex l).  The narrowphase's big-hull
// wavefronts read their hulls from LDS all the time; under that load an LDS round trip of this wave took ~1 k cycles and the hull
// scans — two per support point, reading the vertices and then the winner — were 40 % of a penetration query.  From registers
// a scan is dot products, a DPP butterfly per row, four v_readlane for the rows' winners and three more for the winning vertex.
struct RegHull {   // a hull of at most 64 vertices: lane l keeps vertex l (n = 0: not in registers, scanned from LDS by support_wave)
    float x, y, z;
    int n;
};
DEV RegHull reg_hull_load(int hull_off, int hull_n, int lds_hull_off) {
    RegHull h;
    h.n = hull_n <= 64 ? hull_n : 0;
    h.x = h.y = h.z = 0.f;
    const int v = (int) (threadIdx.x & 63);
    if (v < h.n) {
        const bool in_lds = lds_hull_off == -2 || hull_off == lds_hull_off;
        if (in_lds) { const gj_f4 w = (g_lds_hull + (lds_hull_off == -2 ? hull_off : 0))[v]; h.x = w[0]; h.y = w[1]; h.z = w[2]; }
        else { const int g = hull_off + v, hb = 6 * (g >> 1) + (g & 1); h.x = c_skel.hull[hb]; h.y = c_skel.hull[hb + 2]; h.z = c_skel.hull[hb + 4]; }
    }
    return h;
}
// first maximum of dot(dir, vertex) (larger value, lower index on a tie), like support(); dir is equal across the wavefront
DEV F3 reg_hull_support(const RegHull &h, F3 dir) {
    const int lane = (int) (threadIdx.x & 63);
    const float d = (dir.x * h.x + dir.y * h.y) + dir.z * h.z;
    float best = lane < h.n ? d : -GJ_LARGE;
    int bi = lane < h.n ? lane : 0x7fffffff;
#pragma unroll
    for (int st = 0; st < 4; st++) {
        const float ob = __int_as_float(gj::dpp_i(__float_as_int(best), st));
        const int oi = gj::dpp_i(bi, st);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    float wb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(best), 0));
    int wi = __builtin_amdgcn_readlane(bi, 0);
    if (h.n > 16) {   // (wave-uniform; a box fits one row)
#pragma unroll
        for (int r = 1; r < 4; r++) {
            const float ob = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(best), 16 * r));
            const int oi = __builtin_amdgcn_readlane(bi, 16 * r);
            if (ob > wb || (ob == wb && oi < wi)) { wb = ob; wi = oi; }
        }
    }
    return f3(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(h.x), wi)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h.y), wi)),
              __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h.z), wi)));
}

// gjkepa2_impl::MinkowskiDiff, in A's local frame
template <bool GROUP>
struct Mink {
    Shape A, B;
    M33 toshape1, t0b;
    F3 t0o;
    bool margins;
    bool full;   // every lane of the wavefront is here: both hulls sit in its registers
    const RegHull *ha, *hb;
    DEV F3 hull(const Shape &S, bool first, F3 d) const {
        if (full) {
            const RegHull &h = first ? *ha : *hb;
            if (h.n > 0) return reg_hull_support(h, d);
            return gj::support_wave(S.hull_off, S.hull_n, d, GROUP ? A.lds_hull_off : -1);
        }
        return GROUP ? gj::support_group(S.hull_off, S.hull_n, d, A.lds_hull_off) : gj::support(S.hull_off, S.hull_n, d);
    }
    DEV F3 ls(const Shape &S, bool first, F3 d) const {
        if (!margins) return uf3(hull(S, first, d));
        F3 n = d;                                         // btConvexShape::localGetSupportVertexNonVirtual
        if (ub(gj::len2(n) < EVM_EPS * EVM_EPS)) n = f3(-1.f, -1.f, -1.f);
        n = gj::scl(n, 1.0f / flen(n));
        UPH_T0()
        const F3 hv = uf3(hull(S, first, n));
#ifdef EVM_KSTAMPS
        if (first) UPH(5) else UPH(6)
        if (g_ust_on && threadIdx.x == 0) g_uph[7] += 1ull;
#endif
        return gj::add(hv, gj::scl(n, MARGIN_F));
    }
    DEV F3 support0(F3 d) const { return ls(A, true, d); }
    DEV F3 support1(F3 d) const { return gj::xform(t0b, t0o, ls(B, false, mmul(toshape1, d))); }
    DEV F3 support(F3 d) const { return gj::sub(support0(d), support1(gj::neg(d))); }
};
template <bool GROUP>
DEV Mink<GROUP> mink_init(const Shape &A, F3 oA, const Shape &B, F3 oB, bool margins, const RegHull *ha, const RegHull *hb, bool full) {
    Mink<GROUP> m;
    m.A = A; m.B = B;
    m.toshape1 = tmm(B.R, A.R);
    m.t0b = tmm(A.R, B.R);
    m.t0o = gj::vmul(gj::sub(oB, oA), A.R);
    m.margins = margins;
    m.full = full;
    m.ha = ha; m.hb = hb;
    return m;
}

// ---- gjkepa2_impl::GJK ---------------------------------------------------------------------------------------------------
DEV float project2(F3 a, F3 b, float *w, unsigned &m) {
    const F3 d = gj::sub(b, a);
    const float l = gj::len2(d);
    if (l > 0.f) {
        const float t = l > 0.f ? -gj::dot(a, d) / l : 0.f;
        if (t >= 1.f) { w[0] = 0.f; w[1] = 1.f; m = 2; return gj::len2(b); }
        else if (t <= 0.f) { w[0] = 1.f; w[1] = 0.f; m = 1; return gj::len2(a); }
        else { w[0] = 1.f - (w[1] = t); m = 3; return gj::len2(gj::add(a, gj::scl(d, t))); }
    }
    return -1.f;
}
DEV float project3(F3 a, F3 b, F3 c, float *w, unsigned &m) {
    const F3 vt[3] = {a, b, c};
    const F3 dl[3] = {gj::sub(a, b), gj::sub(b, c), gj::sub(c, a)};
    const F3 n = gj::cross(dl[0], dl[1]);
    const float l = gj::len2(n);
    if (l > 0.f) {
        float mindist = -1.f;
        float subw[2] = {0.f, 0.f};
        unsigned subm = 0;
#pragma unroll
        for (unsigned i = 0; i < 3; ++i) {
            const unsigned j = (i + 1) % 3, k = (j + 1) % 3;
            if (gj::dot(vt[i], gj::cross(dl[i], n)) > 0.f) {
                const float subd = project2(vt[i], vt[j], subw, subm);
                if (mindist < 0.f || subd < mindist) {
                    mindist = subd;
                    m = ((subm & 1) ? 1u << i : 0u) + ((subm & 2) ? 1u << j : 0u);
                    w[i] = subw[0];
                    w[j] = subw[1];
                    w[k] = 0.f;
                }
            }
        }
        if (mindist < 0.f) {
            const float d = gj::dot(a, n);
            const float s = sqrtf(l);
            const F3 p = gj::scl(n, d / l);
            mindist = gj::len2(p);
            m = 7;
            w[0] = flen(gj::cross(dl[1], gj::sub(b, p))) / s;
            w[1] = flen(gj::cross(dl[2], gj::sub(c, p))) / s;
            w[2] = 1.f - (w[0] + w[1]);
        }
        return mindist;
    }
    return -1.f;
}
DEV float project4(F3 a, F3 b, F3 c, F3 d, float *w, unsigned &m) {
    const F3 vt[3] = {a, b, c};
    const F3 dl[3] = {gj::sub(a, d), gj::sub(b, d), gj::sub(c, d)};
    const float vl = det3(dl[0], dl[1], dl[2]);
    const bool ng = (vl * gj::dot(a, gj::cross(gj::sub(b, c), gj::sub(a, b)))) <= 0.f;
    if (ng && fabsf(vl) > 0.f) {
        float mindist = -1.f;
        float subw[3] = {0.f, 0.f, 0.f};
        unsigned subm = 0;
#pragma unroll
        for (unsigned i = 0; i < 3; ++i) {
            const unsigned j = (i + 1) % 3, k = (j + 1) % 3;
            const float s = vl * gj::dot(d, gj::cross(dl[i], dl[j]));
            if (s > 0.f) {
                const float subd = project3(vt[i], vt[j], d, subw, subm);
                if (mindist < 0.f || subd < mindist) {
                    mindist = subd;
                    m = ((subm & 1) ? 1u << i : 0u) + ((subm & 2) ? 1u << j : 0u) + ((subm & 4) ? 8u : 0u);
                    w[i] = subw[0];
                    w[j] = subw[1];
                    w[k] = 0.f;
                    w[3] = subw[2];
                }
            }
        }
        if (mindist < 0.f) {
            mindist = 0.f;
            m = 15;
            w[0] = det3(c, b, d) / vl;
            w[1] = det3(a, c, d) / vl;
            w[2] = det3(b, a, d) / vl;
            w[3] = 1.f - (w[0] + w[1] + w[2]);
        }
        return mindist;
    }
    return -1.f;
}

// The GJK's simplices, weights and free list live in LDS (g_epa.sc / sp / srank / gfree; every lane reads and writes the same
// values): kept in registers they cost the kernel 24 more live registers around the call site and the COMMON path of the
// narrowphase paid for the spills (measured: + 15 us on every launch).
template <bool GROUP>
struct Gjk2 {
    Mink<GROUP> shape;
    F3 ray;
    int nfree, current, status;   // status: 0 Valid, 1 Inside, 2 Failed
    float distance;

    DEV F3 W(int simplex, int i) const { return ld3(g_epa.svw[EPA_MAXV + g_epa.sc[simplex][i]]); }
    // the final simplex, as the polytope's start and Distance()'s witnesses read it
    DEV int rk() const { return ui(g_epa.srank[current]); }
    DEV int slot(int i) const { return ui(g_epa.sc[current][i]); }
    DEV F3 W(int i) const { return W(current, i); }
    DEV float P(int i) const { return uf(g_epa.sp[current][i]); }
    DEV F3 Dslot(int sl) const { return ld3(g_epa.svd[EPA_MAXV + sl]); }
    DEV void swap01() {
        const int s = current;
        const int tc = ui(g_epa.sc[s][0]), tc1 = ui(g_epa.sc[s][1]);
        const float tp = uf(g_epa.sp[s][0]), tp1 = uf(g_epa.sp[s][1]);
        g_epa.sc[s][0] = tc1; g_epa.sc[s][1] = tc;
        g_epa.sp[s][0] = tp1; g_epa.sp[s][1] = tp;
    }
    // getsupport(d, store[slot])
    DEV void getsupport(F3 d, int id) {
        const F3 dn = gj::scl(d, 1.0f / flen(d));
        const F3 w = shape.support(dn);
        g_epa.svd[id] = gj_f4{dn.x, dn.y, dn.z, 0.f};
        g_epa.svw[id] = gj_f4{w.x, w.y, w.z, 0.f};
    }
    DEV void removevertice(int s) {
        const int r = ui(g_epa.srank[s]) - 1;
        g_epa.srank[s] = r;
        g_epa.gfree[nfree++] = ui(g_epa.sc[s][r]);
    }
    DEV void appendvertice(int s, F3 v) {
        const int r = ui(g_epa.srank[s]);
        const int slot = ui(g_epa.gfree[--nfree]);
        g_epa.sp[s][r] = 0.f;
        g_epa.sc[s][r] = slot;
        g_epa.srank[s] = r + 1;
        getsupport(v, EPA_MAXV + slot);
    }
    DEV int evaluate(const Mink<GROUP> &shapearg, F3 guess) {
        unsigned iterations = 0;
        float sqdist = 0.f, alpha = 0.f;
        F3 lw0, lw1, lw2, lw3;
        unsigned clastw = 0;
        g_epa.gfree[0] = 0; g_epa.gfree[1] = 1; g_epa.gfree[2] = 2; g_epa.gfree[3] = 3;
        nfree = 4;
        current = 0;
        status = 0;
        shape = shapearg;
        distance = 0.f;
        g_epa.srank[0] = 0;
        ray = guess;
        const float sqrl = gj::len2(ray);
        appendvertice(0, ub(sqrl > 0.f) ? gj::neg(ray) : f3(1.f, 0.f, 0.f));
        g_epa.sp[0][0] = 1.f;
        ray = W(0, 0);
        sqdist = sqrl;
        lw0 = lw1 = lw2 = lw3 = ray;
        do {
            const int next = 1 - current, cs = current;
            const float rl = flen(ray);
            if (ub(rl < EPA_GJK_MIN_DISTANCE)) { status = 1; break; }
            appendvertice(cs, gj::neg(ray));
            const int rank = ui(g_epa.srank[cs]);
            const F3 w = W(cs, rank - 1);
            const bool found = gj::len2(gj::sub(w, lw0)) < EPA_GJK_DUP_EPS || gj::len2(gj::sub(w, lw1)) < EPA_GJK_DUP_EPS ||
                               gj::len2(gj::sub(w, lw2)) < EPA_GJK_DUP_EPS || gj::len2(gj::sub(w, lw3)) < EPA_GJK_DUP_EPS;
            if (ub(found)) { removevertice(cs); break; }
            clastw = (clastw + 1) & 3;
            lw0 = gj::sel3(clastw == 0, w, lw0); lw1 = gj::sel3(clastw == 1, w, lw1);
            lw2 = gj::sel3(clastw == 2, w, lw2); lw3 = gj::sel3(clastw == 3, w, lw3);
            const float omega = gj::dot(ray, w) / rl;
            alpha = omega > alpha ? omega : alpha;
            if (ub(((rl - alpha) - (EPA_GJK_ACCURACY * rl)) <= 0.f)) { removevertice(cs); break; }
            float weights[4] = {0.f, 0.f, 0.f, 0.f};
            unsigned mask = 0;
            if (rank == 2) sqdist = project2(W(cs, 0), W(cs, 1), weights, mask);
            else if (rank == 3) sqdist = project3(W(cs, 0), W(cs, 1), W(cs, 2), weights, mask);
            else sqdist = project4(W(cs, 0), W(cs, 1), W(cs, 2), W(cs, 3), weights, mask);
            mask = (unsigned) ui((int) mask);
            if (ub(sqdist >= 0.f)) {
                int nr = 0;
                ray = f3(0.f, 0.f, 0.f);
                current = next;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < rank) {
                        const int slot = ui(g_epa.sc[cs][i]);
                        if (mask & (1u << i)) {
                            g_epa.sc[next][nr] = slot;
                            g_epa.sp[next][nr] = weights[i];
                            nr++;
                            ray = gj::add(ray, gj::scl(ld3(g_epa.svw[EPA_MAXV + slot]), weights[i]));
                        } else {
                            g_epa.gfree[nfree++] = slot;
                        }
                    }
                }
                g_epa.srank[next] = nr;
                ray = uf3(ray);
                if (mask == 15) status = 1;
            } else {
                removevertice(cs);
                break;
            }
            status = ((++iterations) < (unsigned) EPA_GJK_MAX_ITER) ? status : 2;
        } while (status == 0);
        if (status == 0) distance = flen(ray);
        else if (status == 1) distance = 0.f;
        return status;
    }
    // EncloseOrigin.  The original recurses over the simplex rank (1 -> 2 -> 3 -> 4), trying at rank 1 the six signed axes, at rank
    // 2 the cross products of the segment with the axes (both signs, skipped when zero), at rank 3 the two signed triangle normals,
    // each time appending the support vertex and descending.  The same search as ONE loop with a candidate counter per level and a
    // single appendvertice site: inlined as nested calls it put 22 copies of the support code into the kernel, and this code runs
    // once per query — every instruction of it fetched cold.
    DEV bool enclose4() {
        const int s = current;
        return ub(fabsf(det3(gj::sub(W(s, 0), W(s, 3)), gj::sub(W(s, 1), W(s, 3)), gj::sub(W(s, 2), W(s, 3)))) > 0.f);
    }
    DEV bool enclose_origin() {
        const int s = current;
        const int base = ui(g_epa.srank[s]);
        int lvl = base, k1 = 0, k2 = 0, k3 = 0;
        if (lvl < 1 || lvl > 4) return false;
#pragma nounroll
        for (;;) {
            if (lvl == 4) {
                if (enclose4()) return true;
                if (base == 4) return false;
                removevertice(s);
                lvl = 3;
                continue;
            }
            const int ncand = lvl == 3 ? 2 : 6;
            int k = lvl == 1 ? k1 : (lvl == 2 ? k2 : k3);
            bool pushed = false;
            F3 dir = f3(0.f, 0.f, 0.f);
#pragma nounroll
            while (k < ncand && !pushed) {
                const int c = k++;
                const int i = c >> 1;
                const F3 axis = f3(i == 0 ? 1.f : 0.f, i == 1 ? 1.f : 0.f, i == 2 ? 1.f : 0.f);
                F3 v;
                if (lvl == 1) v = axis;
                else if (lvl == 2) v = gj::cross(gj::sub(W(s, 1), W(s, 0)), axis);
                else v = gj::cross(gj::sub(W(s, 1), W(s, 0)), gj::sub(W(s, 2), W(s, 0)));
                if (lvl == 1 || ub(gj::len2(v) > 0.f)) { dir = (c & 1) ? gj::neg(v) : v; pushed = true; }
            }
            if (lvl == 1) k1 = k; else if (lvl == 2) k2 = k; else k3 = k;
            if (pushed) {
                appendvertice(s, dir);
                lvl++;
                if (lvl == 2) k2 = 0;
                if (lvl == 3) k3 = 0;
                continue;
            }
            if (lvl == base) return false;   // this level is exhausted: back to the parent, which takes its vertex away and goes on
            removevertice(s);
            lvl--;
        }
    }
};

// ---- gjkepa2_impl::EPA ---------------------------------------------------------------------------------------------------
#ifdef EVM_KSTAMPS   // (tools/kstamps.py) where a penetration query's cycles go: d.stamps[32..]
#define EPA_T0() unsigned long long epa_t = __builtin_amdgcn_s_memtime();
#define EPA_MARK(KS, k)                                                                    \
    {                                                                                      \
        __builtin_amdgcn_s_waitcnt(0);                                                     \
        const unsigned long long epa_n = __builtin_amdgcn_s_memtime();                     \
        if (threadIdx.x == __builtin_amdgcn_readfirstlane(threadIdx.x)) atomicAdd(&(KS)[k], epa_n - epa_t); \
        epa_t = epa_n;                                                                     \
    }
#define EPA_COUNT(KS, k, v) { if (threadIdx.x == __builtin_amdgcn_readfirstlane(threadIdx.x)) atomicAdd(&(KS)[k], (unsigned long long) (v)); }
#else
#define EPA_T0()
#define EPA_MARK(KS, k)
#define EPA_COUNT(KS, k, v)
#endif
struct EpaOut {
    int status;       // EPA::eStatus (9 = Failed)
    F3 normal;
    float depth;
    int rank, c[3];
    float p[3];
};
struct NewFace {
    F3 n;
    float d;
    bool ok;
};
// the geometry half of EPA::newface (getedgedist included)
DEV bool edgedist(F3 fn, F3 a, F3 b, float &dist) {
    const F3 ba = gj::sub(b, a);
    const F3 n_ab = gj::cross(ba, fn);
    const float a_dot_nab = gj::dot(a, n_ab);
    if (a_dot_nab < 0.f) {
        const float ba_l2 = gj::len2(ba);
        const float a_dot_ba = gj::dot(a, ba);
        const float b_dot_ba = gj::dot(b, ba);
        if (a_dot_ba > 0.f) dist = flen(a);
        else if (b_dot_ba < 0.f) dist = flen(b);
        else {
            const float a_dot_b = gj::dot(a, b);
            const float q = (gj::len2(a) * gj::len2(b) - a_dot_b * a_dot_b) / ba_l2;
            dist = sqrtf(q > 0.f ? q : 0.f);
        }
        return true;
    }
    return false;
}
DEV NewFace face_geometry(F3 a, F3 b, F3 c, bool forced) {
    NewFace r;
    r.n = gj::cross(gj::sub(b, a), gj::sub(c, a));
    r.d = 0.f;
    const float l = flen(r.n);
    r.ok = false;
    if (l > EPA_ACCURACY) {
        float d = 0.f;
        if (!(edgedist(r.n, a, b, d) || edgedist(r.n, b, c, d) || edgedist(r.n, c, a, d))) d = gj::dot(a, r.n) / l;
        r.d = d;
        r.n = gj::scl(r.n, 1.0f / l);
        r.ok = forced || r.d >= -EPA_PLANE_EPS;
    }
    return r;
}

template <bool GROUP>
// ctl / ctl_epoch (speculative runs, narrow_dev.h: speculate_pen_depth; else null): the word through which the query's owner says, once it
// knows, whether the answer is wanted — (epoch << 2) | 1: wanted, the wave moves to the front of its SIMD's issue; | 2: not wanted,
// the expansion stops (status 10).  Read once per round, the load in flight while the round runs.
DEV EpaOut epa_evaluate(Gjk2<GROUP> &gjk, F3 guess, const int *ctl = nullptr, int ctl_epoch = 0) {
    State &S = g_epa;
    EpaOut out;
    out.status = 9; out.normal = f3(0.f, 0.f, 0.f); out.depth = 0.f; out.rank = 0;
    out.c[0] = out.c[1] = out.c[2] = 0; out.p[0] = out.p[1] = out.p[2] = 0.f;
    const int lane = (int) (threadIdx.x & 63);
    const unsigned long long act = __ballot(true);
    const int nact = (int) __popcll(act), rank = (int) __popcll(act & ((1ull << lane) - 1ull));
    // rows_ok: every 16-lane row of the wavefront is either whole or absent (always so in the grouped form; a full wavefront in
    // the one-query-per-lane form): findbest may then reduce across a row with DPP
    const bool rows_ok = (((act & 0xFFFFull) == 0ull) || ((act & 0xFFFFull) == 0xFFFFull)) && ((((act >> 16) & 0xFFFFull) == 0ull) || (((act >> 16) & 0xFFFFull) == 0xFFFFull)) &&
                         ((((act >> 32) & 0xFFFFull) == 0ull) || (((act >> 32) & 0xFFFFull) == 0xFFFFull)) && ((((act >> 48) & 0xFFFFull) == 0ull) || (((act >> 48) & 0xFFFFull) == 0xFFFFull));
    if (gjk.rk() > 1 && gjk.enclose_origin()) {
        // stock = every face, root = face 0; hull empty
        for (int i = rank; i < EPA_MAXF; i += nact) { S.stock[i] = (unsigned char) (EPA_MAXF - 1 - i); S.fkey[i] = EVM_INF; }
        EPA_SYNC()
        int nstock = EPA_MAXF, seq = 0, hi = 0, nhull = 0;
        int nextsv = 0;
        // orient the simplex
        if (ub(det3(gj::sub(gjk.W(0), gjk.W(3)), gj::sub(gjk.W(1), gjk.W(3)), gj::sub(gjk.W(2), gjk.W(3))) < 0.f)) {
            gjk.swap01();
        }
        const int g0 = EPA_MAXV + gjk.slot(0), g1 = EPA_MAXV + gjk.slot(1), g2 = EPA_MAXV + gjk.slot(2), g3 = EPA_MAXV + gjk.slot(3);
        int tetra[4] = {-1, -1, -1, -1};
#pragma nounroll
        for (int k = 0; k < 4; k++) {   // newface(a, b, c, forced = true): (g0 g1 g2) (g1 g0 g3) (g2 g1 g3) (g0 g2 g3)
            const int va = k == 0 ? g0 : (k == 1 ? g1 : (k == 2 ? g2 : g0)), vb = k == 0 ? g1 : (k == 1 ? g0 : (k == 2 ? g1 : g2)), vc = k == 0 ? g2 : g3;
            const int fi = ui(S.stock[nstock - 1]);
            const NewFace nf = face_geometry(ld3(S.svw[va]), ld3(S.svw[vb]), ld3(S.svw[vc]), true);
            if (ub(nf.ok)) {
                nstock--;
                S.fnd[fi] = gj_f4{nf.n.x, nf.n.y, nf.n.z, nf.d};
                S.fc[fi] = (unsigned) va | ((unsigned) vb << 8) | ((unsigned) vc << 16);
                S.fseq[fi] = (unsigned short) seq++;
                S.fkey[fi] = nf.d * nf.d;
                tetra[0] = k == 0 ? fi : tetra[0]; tetra[1] = k == 1 ? fi : tetra[1]; tetra[2] = k == 2 ? fi : tetra[2]; tetra[3] = k == 3 ? fi : tetra[3];
                nhull++;
                hi = fi + 1 > hi ? fi + 1 : hi;
            }
        }
        if (nhull == 4) {
            // bind(tetra[0],0,tetra[1],0) (0,1,2,0) (0,2,3,0) (1,1,3,2) (1,2,2,1) (2,2,3,1)
            S.fadj[tetra[0]] = (unsigned) tetra[1] | ((unsigned) tetra[2] << 8) | ((unsigned) tetra[3] << 16) | (0u << 24) | (0u << 26) | (0u << 28);
            S.fadj[tetra[1]] = (unsigned) tetra[0] | ((unsigned) tetra[3] << 8) | ((unsigned) tetra[2] << 16) | (0u << 24) | (2u << 26) | (1u << 28);
            S.fadj[tetra[2]] = (unsigned) tetra[0] | ((unsigned) tetra[1] << 8) | ((unsigned) tetra[3] << 16) | (1u << 24) | (2u << 26) | (1u << 28);
            S.fadj[tetra[3]] = (unsigned) tetra[0] | ((unsigned) tetra[2] << 8) | ((unsigned) tetra[1] << 16) | (2u << 24) | (2u << 26) | (1u << 28);
            // FULL wavefront (the urgent list's blocks; a full wave of the other forms): face f < 64 also lives in lane f's registers —
            // normal, distance, adjacency, key, sequence number — re-read from LDS (the authoritative copy: every write below goes
            // through to it) after every batch of new faces.  Visibility then is a ballot (two scalar masks), the horizon walk reads a
            // face's adjacency with v_readlane and runs on the scalar unit, findbest is a reduction over the lanes: none of them
            // touches LDS.  A polytope that grows past 64 faces simply goes on with the LDS forms below.
            bool regs = act == ~0ull;
            F3 rn = f3(0.f, 0.f, 0.f);
            float rd = 0.f, rkey = EVM_INF;
            unsigned radj = 0u;
            int rseq = -1;
            auto reload = [&]() {
                if (!regs) return;
                if (hi > 64) { regs = false; return; }
                const gj_f4 nd = S.fnd[lane];
                rn = f3(nd[0], nd[1], nd[2]); rd = nd[3];
                radj = S.fadj[lane];
                rkey = lane < hi ? S.fkey[lane] : EVM_INF;
                rseq = (int) S.fseq[lane];
            };
            reload();
            // findbest: smallest d^2, the most recently appended face on a tie
            auto findbest = [&]() {
                int bf = 0, bs = -1;
                float bd = EVM_INF;
                if (regs) {
                    bf = lane; bd = rkey; bs = rkey < EVM_INF ? rseq : -1;
#pragma unroll
                    for (int st = 0; st < 4; st++) {
                        const float ok = __int_as_float(gj::dpp_i(__float_as_int(bd), st));
                        const int oq = gj::dpp_i(bs, st), of = gj::dpp_i(bf, st);
                        if (ok < bd || (ok == bd && oq > bs)) { bd = ok; bs = oq; bf = of; }
                    }
                    float wd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bd), 0));
                    int ws = __builtin_amdgcn_readlane(bs, 0), wf = __builtin_amdgcn_readlane(bf, 0);
#pragma unroll
                    for (int r = 1; r < 4; r++) {
                        const float ok = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bd), 16 * r));
                        const int oq = __builtin_amdgcn_readlane(bs, 16 * r), of = __builtin_amdgcn_readlane(bf, 16 * r);
                        if (ok < wd || (ok == wd && oq > ws)) { wd = ok; ws = oq; wf = of; }
                    }
                    return wf;
                }
                if (rows_ok) {   // the row's 16 lanes take a face each (+16, +32, ...), a four-step butterfly inside the row finishes
                    for (int f = lane & 15; f < hi; f += 16) {
                        const float k = S.fkey[f];
                        const int q = (int) S.fseq[f];
                        if (k < EVM_INF && (k < bd || (k == bd && q > bs))) { bf = f; bd = k; bs = q; }
                    }
#pragma unroll
                    for (int st = 0; st < 4; st++) {
                        const float ok = __int_as_float(gj::dpp_i(__float_as_int(bd), st));
                        const int oq = gj::dpp_i(bs, st), of = gj::dpp_i(bf, st);
                        if (ok < bd || (ok == bd && oq > bs)) { bd = ok; bs = oq; bf = of; }
                    }
                    return ui(bf);
                }
                for (int f0 = 0; f0 < hi; f0 += 8) {     // (unconditional loads, eight in flight: the scan is latency, not work)
                    float k[8];
                    int q[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) { k[u] = S.fkey[f0 + u]; q[u] = (int) S.fseq[f0 + u]; }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const bool in = f0 + u < hi && k[u] < EVM_INF;
                        if (in && (k[u] < bd || (k[u] == bd && q[u] > bs))) { bf = f0 + u; bd = k[u]; bs = q[u]; }
                    }
                }
                return ui(bf);
            };
            int best = findbest();
            gj_f4 outer_nd = S.fnd[best];
            unsigned outer_c = S.fc[best];
            int status = 0;   // Valid
            unsigned iterations = 0;
            for (; iterations < (unsigned) EPA_MAX_ITER; ++iterations) {
                if (ctl != nullptr) {
                    const int cw = ui(__hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if ((cw >> 2) == ctl_epoch) {
                        if ((cw & 3) == 2) { status = 10; break; }
                        if ((cw & 3) == 1) __builtin_amdgcn_s_setprio(3);
                    }
                }
                if (nextsv < EPA_MAXV) {
                    const int w = nextsv++;
                    UPH_T0()
                    F3 bn;
                    float bdist;
                    if (regs) {
                        auto rl = [&](float x) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), best)); };
                        bn = f3(rl(rn.x), rl(rn.y), rl(rn.z)); bdist = rl(rd);
                    } else {
                        const gj_f4 bnd = S.fnd[best];
                        bn = uf3(f3(bnd[0], bnd[1], bnd[2]));
                        bdist = uf(bnd[3]);
                    }
                    gjk.getsupport(bn, w);
                    const F3 ww = ld3(S.svw[w]);
                    const float wdist = gj::dot(bn, ww) - bdist;
                    UPH(0)
                    EPA_COUNT(gjk.shape.A.ks, 36, 1)
                    if (ub(wdist > EPA_ACCURACY)) {
                        // visibility of every hull face from w (best counts as visited: its pass is the current one)
                        unsigned long long vmask = 0ull, dmask = 0ull;   // register form: faces visible from w / visited in this round
                        if (regs) {
                            const bool beyond = (gj::dot(rn, ww) - rd) < -EPA_PLANE_EPS;
                            vmask = __ballot(rkey < EVM_INF && !beyond);
                            dmask = 1ull << best;
                        } else
                        for (int f = rank; f < hi; f += nact) {
                            const gj_f4 nd = S.fnd[f];
                            const bool beyond = (gj::dot(f3(nd[0], nd[1], nd[2]), ww) - nd[3]) < -EPA_PLANE_EPS;
                            S.fadj[f] = (S.fadj[f] & 0x3FFFFFFFu) | ((f == best ? 2u : (beyond ? 0u : 1u)) << 30);
                        }
                        EPA_SYNC()
                        UPH(1)
                        // expand(pass, w, best->f[j], best->e[j], horizon), j = 0..2, as one explicit-stack walk
                        bool valid = true;
                        int nh = 0, sp = 0;
                        const unsigned badj = (unsigned) ui((int) S.fadj[best]);
                        for (int j = 0; j < 3 && valid; j++) {
                            int cf = (int) ((badj >> (8 * j)) & 255u), ce = (int) ((badj >> (24 + 2 * j)) & 3u);
                            bool calling = true, ret = false;
                            for (int guard = 0;; guard++) {
                                if (guard > 8 * EPA_MAXF) { valid = false; break; }   // (cannot happen: every trip visits a face or pops a frame)
                                if (calling) {
                                    const unsigned adj = regs ? (unsigned) __builtin_amdgcn_readlane((int) radj, cf) : (unsigned) ui((int) S.fadj[cf]);
                                    const int vis = regs ? (((dmask >> cf) & 1ull) ? 2 : (int) ((vmask >> cf) & 1ull)) : (int) (adj >> 30);
                                    if (vis >= 2) ret = false;
                                    else if (vis == 0) {
                                        if (nstock == 0) ret = false;
                                        else {
                                            const int nf = ui((int) S.stock[--nstock]);
                                            S.hz[nh] = (unsigned short) (cf | (ce << 8));
                                            S.hzn[nh] = (unsigned char) nf;
                                            nh++;
                                            hi = nf + 1 > hi ? nf + 1 : hi;
                                            ret = true;
                                        }
                                    } else {
                                        if (regs) dmask |= 1ull << cf;
                                        else S.fadj[cf] = adj | (2u << 30);     // (1 -> 3: bit 31 = visited)
                                        S.frame[sp++] = (unsigned short) (cf | (ce << 8));
                                        const int e1 = (ce + 1) % 3;
                                        cf = (int) ((adj >> (8 * e1)) & 255u); ce = (int) ((adj >> (24 + 2 * e1)) & 3u);
                                        continue;
                                    }
                                    calling = false;
                                }
                                // a call returned `ret`
                                if (!ret) { valid = false; break; }
                                if (sp == 0) break;
                                const int fr = ui((int) S.frame[sp - 1]);
                                const int f = fr & 255, e = (fr >> 8) & 3, stage = fr >> 10;
                                if (stage == 0) {
                                    S.frame[sp - 1] = (unsigned short) (fr | (1 << 10));
                                    const unsigned adj = regs ? (unsigned) __builtin_amdgcn_readlane((int) radj, f) : (unsigned) ui((int) S.fadj[f]);
                                    const int e2 = (e + 2) % 3;
                                    cf = (int) ((adj >> (8 * e2)) & 255u); ce = (int) ((adj >> (24 + 2 * e2)) & 3u);
                                    calling = true;
                                } else {
                                    sp--;
                                    S.fkey[f] = EVM_INF;                // remove(m_hull, f); append(m_stock, f)
                                    if (regs && lane == f) rkey = EVM_INF;
                                    S.stock[nstock++] = (unsigned char) f;
                                    ret = true;
                                }
                            }
                        }
                        UPH(2)
                        if (valid && nh >= 3) {
                            // the horizon's faces, one per lane: newface(f->c[e1], f->c[e], w, false), bind(nf, 0, f, e), the fan's
                            // bind(prev, 1, nf, 2) and the closing bind(last, 1, first, 2)
                            S.bad = 0;
                            EPA_SYNC()
                            for (int i = rank; i < nh; i += nact) {
                                const int he = (int) S.hz[i], f = he & 255, e = he >> 8, e1 = (e + 1) % 3;
                                const int nf = (int) S.hzn[i];
                                const int nxt = (int) S.hzn[i + 1 < nh ? i + 1 : 0], prv = (int) S.hzn[i > 0 ? i - 1 : nh - 1];
                                const unsigned fcw = S.fc[f];
                                const int ca = (int) ((fcw >> (8 * e1)) & 255u), cb = (int) ((fcw >> (8 * e)) & 255u);
                                const NewFace g = face_geometry(ld3(S.svw[ca]), ld3(S.svw[cb]), ww, false);
                                if (!g.ok) S.bad = 1;
                                S.fnd[nf] = gj_f4{g.n.x, g.n.y, g.n.z, g.d};
                                S.fc[nf] = (unsigned) ca | ((unsigned) cb << 8) | ((unsigned) w << 16);
                                S.fadj[nf] = (unsigned) f | ((unsigned) nxt << 8) | ((unsigned) prv << 16) | ((unsigned) e << 24) | (2u << 26) | (1u << 28);
                                S.fseq[nf] = (unsigned short) (seq + i);
                                S.fkey[nf] = g.d * g.d;
                                // f's edge e now borders nf's edge 0 (each lane its own byte / bit field of f's word: by atomics)
                                atomicAnd(&S.fadj[f], ~((255u << (8 * e)) | (3u << (24 + 2 * e))));
                                atomicOr(&S.fadj[f], ((unsigned) nf << (8 * e)));
                            }
                            EPA_SYNC()
                            UPH(3)
                            if (ub(S.bad != 0)) { status = 4; break; }   // InvalidHull
                            seq += nh;
                            S.fkey[best] = EVM_INF;                         // remove(m_hull, best); append(m_stock, best)
                            S.stock[nstock++] = (unsigned char) best;
                            reload();
                            best = findbest();
                            outer_nd = S.fnd[best];
                            outer_c = S.fc[best];
                            UPH(4)
                        } else { status = 4; break; }
                    } else { status = 7; break; }   // AccuraryReached
                } else { status = 6; break; }       // OutOfVertices
            }
            const F3 on = uf3(f3(outer_nd[0], outer_nd[1], outer_nd[2]));
            const float od = uf(outer_nd[3]);
            outer_c = (unsigned) ui((int) outer_c);
            const F3 projection = gj::scl(on, od);
            out.normal = on;
            out.depth = od;
            out.rank = 3;
            out.c[0] = (int) (outer_c & 255u); out.c[1] = (int) ((outer_c >> 8) & 255u); out.c[2] = (int) ((outer_c >> 16) & 255u);
            const F3 w0 = ld3(S.svw[out.c[0]]), w1 = ld3(S.svw[out.c[1]]), w2 = ld3(S.svw[out.c[2]]);
            out.p[0] = flen(gj::cross(gj::sub(w1, projection), gj::sub(w2, projection)));
            out.p[1] = flen(gj::cross(gj::sub(w2, projection), gj::sub(w0, projection)));
            out.p[2] = flen(gj::cross(gj::sub(w0, projection), gj::sub(w1, projection)));
            const float sum = out.p[0] + out.p[1] + out.p[2];
            out.p[0] /= sum; out.p[1] /= sum; out.p[2] /= sum;
            out.status = status;
            return out;
        }
    }
    // fallback
    out.status = 8;
    out.normal = gj::neg(guess);
    const float nl = flen(out.normal);
    if (ub(nl > 0.f)) out.normal = gj::scl(out.normal, 1.0f / nl);
    else out.normal = f3(1.f, 0.f, 0.f);
    out.depth = 0.f;
    out.rank = 1;
    out.c[0] = EPA_MAXV + gjk.slot(0);
    out.p[0] = 1.f;
    return out;
}

// btGjkEpaPenetrationDepthSolver::calcPenDepth for ONE query whose transforms every lane holds (wave-uniform values).
// A.o / B.o unused: oA, oB are the origins (already shifted by the detector's positionOffset).  Returns Penetration()'s verdict;
// has_v: m_cachedSeparatingAxis was set.
#ifndef EPA_CALL
#define EPA_CALL DEV
#endif
template <bool GROUP>
EPA_CALL bool calc_pen_depth(const Shape &A, F3 oA, const Shape &B, F3 oB, F3 &v, F3 &wa, F3 &wb, bool &has_v, const int *ctl, int ctl_epoch, bool &cancelled) {
    has_v = false;
    cancelled = false;
    v = f3(0.f, 0.f, 0.f); wa = f3(0.f, 0.f, 0.f); wb = f3(0.f, 0.f, 0.f);
    // the kernel's longest dependent chain: first in line at its SIMD's issue (lowered again by the caller); a speculative run only once
    // its answer is known to be wanted (epa_evaluate)
    if (ctl == nullptr) __builtin_amdgcn_s_setprio(3);
    const bool full = __ballot(true) == ~0ull;
    RegHull ha, hb;
    ha.n = hb.n = 0;
    if (full) { ha = reg_hull_load(A.hull_off, A.hull_n, GROUP ? A.lds_hull_off : -1); hb = reg_hull_load(B.hull_off, B.hull_n, GROUP ? A.lds_hull_off : -1); }
    EPA_COUNT(A.ks, 32, 1)
#pragma nounroll
    for (int gi = 0; gi < 9; gi++) {
        EPA_COUNT(A.ks, 42, 1)
        F3 guess;
        if (gi < 2) {   // (B - A).safeNormalize(), (A - B).safeNormalize()
            const F3 d = gi == 0 ? gj::sub(oB, oA) : gj::sub(oA, oB);
            const float l2 = gj::len2(d);
            guess = ub(l2 >= EVM_EPS * EVM_EPS) ? gj::scl(d, 1.0f / sqrtf(l2)) : f3(1.f, 0.f, 0.f);
        } else {
            const int k = gi - 2;   // (0,0,1) (0,1,0) (1,0,0) (1,1,0) (1,1,1) (0,1,1) (1,0,1)
            guess = f3((k == 2 || k == 3 || k == 4 || k == 6) ? 1.f : 0.f, (k == 1 || k == 3 || k == 4 || k == 5) ? 1.f : 0.f,
                       (k == 0 || k == 4 || k == 5 || k == 6) ? 1.f : 0.f);
        }
#pragma nounroll
        for (int phase = 0; phase < 2; phase++) {   // 0: btGjkEpaSolver2::Penetration (margins), 1: ::Distance (cores) — one evaluate() site
            if (ctl != nullptr) {
                const int cw = ui(__hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if ((cw >> 2) == ctl_epoch && (cw & 3) == 2) { cancelled = true; return false; }
                if ((cw >> 2) == ctl_epoch && (cw & 3) == 1) __builtin_amdgcn_s_setprio(3);
            }
            const Mink<GROUP> shape = mink_init<GROUP>(A, oA, B, oB, phase == 0, &ha, &hb, full);
            Gjk2<GROUP> gjk;
            EPA_T0()
            const int st = gjk.evaluate(shape, phase == 0 ? gj::neg(guess) : guess);
            if (phase == 0) {
                EPA_MARK(A.ks, 34)
                if (gi == 0) { UST(4) }
                if (st == 1) {
                    const EpaOut e = epa_evaluate<GROUP>(gjk, gj::neg(guess), ctl, ctl_epoch);
                    EPA_MARK(A.ks, 35)
                    UST(5)
                    if (e.status == 10) { cancelled = true; return false; }
                    if (e.status != 9) {
                        F3 w0 = f3(0.f, 0.f, 0.f);
#pragma nounroll
                        for (int i = 0; i < e.rank; ++i) w0 = gj::add(w0, gj::scl(shape.support0(ld3(g_epa.svd[e.c[i]])), e.p[i]));
                        wa = gj::xform(A.R, oA, w0);
                        wb = gj::xform(A.R, oA, gj::sub(w0, gj::scl(e.normal, e.depth)));
                        v = gj::neg(e.normal);
                        has_v = true;
                        return true;
                    }
                }
            } else if (st == 0) {
                F3 w0 = f3(0.f, 0.f, 0.f), w1 = f3(0.f, 0.f, 0.f);
                const int grank = gjk.rk();
#pragma nounroll
                for (int i = 0; i < grank; ++i) {
                    const float p = gjk.P(i);
                    const F3 d = gjk.Dslot(gjk.slot(i));
                    w0 = gj::add(w0, gj::scl(shape.support0(d), p));
                    w1 = gj::add(w1, gj::scl(shape.support1(gj::neg(d)), p));
                }
                wa = gj::xform(A.R, oA, w0);
                wb = gj::xform(A.R, oA, w1);
                F3 n = gj::sub(w0, w1);
                const float dist = flen(n);
                v = gj::scl(n, 1.0f / (dist > EPA_GJK_MIN_DISTANCE ? dist : 1.f));
                has_v = true;
                return false;
            }
        }
    }
    return false;
}

// The same behind a real call (EPA_NOINLINE): the solver's register allocation, and whatever it spills, then stays out of the
// narrowphase's common path.  Everything crosses the call by value in registers: no reference parameter, no stack object.
struct PenOut { float vx, vy, vz, ax, ay, az, bx, by, bz; int flags; };   // flags: bit 0 Penetration()'s verdict, bit 1 v was set, bit 2 a speculative run was called off
template <bool GROUP>
__device__ __attribute__((noinline)) PenOut calc_pen_depth_call(int hoA, int hnA, int hoB, int hnB, int ldsoff, float a0, float a1, float a2, float a3,
                                                                float a4, float a5, float a6, float a7, float a8, float oax, float oay, float oaz, float b0,
                                                                float b1, float b2, float b3, float b4, float b5, float b6, float b7, float b8, float obx,
                                                                float oby, float obz, void *ks, const int *ctl, int ctl_epoch) {
    Shape A, B;
    A.hull_off = hoA; A.hull_n = hnA; B.hull_off = hoB; B.hull_n = hnB;
    A.lds_hull_off = B.lds_hull_off = ldsoff;
    A.R = m33(f3(a0, a1, a2), f3(a3, a4, a5), f3(a6, a7, a8)); B.R = m33(f3(b0, b1, b2), f3(b3, b4, b5), f3(b6, b7, b8));
    A.o = f3(oax, oay, oaz); B.o = f3(obx, oby, obz);
    A.pen_count = B.pen_count = nullptr;
    A.spec = B.spec = nullptr; A.spec_epoch = B.spec_epoch = 0;
#ifdef EVM_KSTAMPS
    A.ks = B.ks = (unsigned long long *) ks;
#else
    (void) ks;
#endif
    F3 v, wa, wb;
    bool has_v, cancelled;
    const bool ok = calc_pen_depth<GROUP>(A, A.o, B, B.o, v, wa, wb, has_v, ctl, ctl_epoch, cancelled);
    PenOut o;
    o.vx = v.x; o.vy = v.y; o.vz = v.z; o.ax = wa.x; o.ay = wa.y; o.az = wa.z; o.bx = wb.x; o.by = wb.y; o.bz = wb.z;
    o.flags = (ok ? 1 : 0) | (has_v ? 2 : 0) | (cancelled ? 4 : 0);
    return o;
}

}  // namespace epa
