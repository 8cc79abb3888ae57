// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_narrow.h for scope, citations and the stated deviations).
#include "orc_narrow.h"
#include "orc_epa.h"

namespace orc {

int g_penetration_solver = 0;
int g_ccd_pretest = 0;

static constexpr float REL_ERROR2 = 1.0e-6f;                    // btGjkPairDetector.cpp, single precision
static constexpr float GJK_EPA_PENETRATION_TOLERANCE = 0.001f;  // gGjkEpaPenetrationTolerance
static constexpr int GJK_MAX_ITER = 1000;                       // gGjkMaxIter
static constexpr float EQUAL_VERTEX_THRESHOLD = 0.0001f;        // VORONOI_DEFAULT_EQUAL_VERTEX_THRESHOLD

// btConvexHullShape::localGetSupportingVertexWithoutMargin (see the header for the product order)
V3 local_support(const ConvexView &S, const V3 &dir) {
    float best = -BT_LARGE_FLOAT;
    V3 bp(0.f, 0.f, 0.f);
    for (int i = 0; i < S.n; i++) {
        const V3 p = S.pts[i] * S.scale;
        const float d = dot(dir, p);
        if (d > best) { best = d; bp = p; }
    }
    return bp;
}

void world_aabb(const ConvexView &S, float contact_threshold, V3 &mn, V3 &mx) {
    // btPolyhedralConvexAabbCachingShape::recalcLocalAabb: supports along +-axes, +- margin
    V3 lmax(-BT_LARGE_FLOAT, -BT_LARGE_FLOAT, -BT_LARGE_FLOAT), lmin(BT_LARGE_FLOAT, BT_LARGE_FLOAT, BT_LARGE_FLOAT);
    for (int i = 0; i < S.n; i++) {
        const V3 p = S.pts[i] * S.scale;
        for (int k = 0; k < 3; k++) {
            if (p[k] > lmax[k]) lmax.at(k) = p[k];
            if (p[k] < lmin[k]) lmin.at(k) = p[k];
        }
    }
    lmax += V3(S.margin, S.margin, S.margin);
    lmin -= V3(S.margin, S.margin, S.margin);
    // btTransformAabb(localMin, localMax, margin, trans)
    V3 half = 0.5f * (lmax - lmin);
    half += V3(S.margin, S.margin, S.margin);
    const V3 lc = 0.5f * (lmax + lmin);
    const V3 c = S.xf(lc);
    V3 ext;
    for (int k = 0; k < 3; k++) {
        const V3 r = S.xf.b.r[k];
        ext.at(k) = half.x * std::fabs(r.x) + half.y * std::fabs(r.y) + half.z * std::fabs(r.z);
    }
    // btCollisionWorld::updateSingleAabb: contactThreshold = gContactBreakingThreshold
    const V3 ct(contact_threshold, contact_threshold, contact_threshold);
    mn = c - ext - ct;
    mx = c + ext + ct;
}

// ------------------------------------------------------------------------------------------------
// btVoronoiSimplexSolver
// ------------------------------------------------------------------------------------------------
namespace {

struct SubSimplexClosest {
    V3 closest;
    bool usedA = false, usedB = false, usedC = false, usedD = false;
    float bary[4] = {0, 0, 0, 0};
    bool degenerate = false;
    void reset_used() { usedA = usedB = usedC = usedD = false; }
    void set_bary(float a, float b, float c, float d) { bary[0] = a; bary[1] = b; bary[2] = c; bary[3] = d; }
    void reset() { degenerate = false; set_bary(0, 0, 0, 0); reset_used(); }
    bool valid() const { return bary[0] >= 0.f && bary[1] >= 0.f && bary[2] >= 0.f && bary[3] >= 0.f; }
};

struct Simplex {
    int n = 0;
    V3 W[5], P[5], Qp[5];
    V3 cachedP1, cachedP2, cachedV, lastW;
    bool cachedValid = false, needsUpdate = true;
    SubSimplexClosest bc;

    void reset() {
        cachedValid = false; n = 0; needsUpdate = true;
        lastW = V3(BT_LARGE_FLOAT, BT_LARGE_FLOAT, BT_LARGE_FLOAT);
        bc.reset();
    }
    void add(const V3 &w, const V3 &p, const V3 &q) {
        lastW = w; needsUpdate = true;
        W[n] = w; P[n] = p; Qp[n] = q;
        n++;
    }
    void remove(int i) {
        n--;
        W[i] = W[n]; P[i] = P[n]; Qp[i] = Qp[n];
    }
    void reduce(const SubSimplexClosest &u) {
        if (n >= 4 && !u.usedD) remove(3);
        if (n >= 3 && !u.usedC) remove(2);
        if (n >= 2 && !u.usedB) remove(1);
        if (n >= 1 && !u.usedA) remove(0);
    }
    bool full() const { return n == 4; }
    bool in_simplex(const V3 &w) const {
        bool found = false;
        for (int i = 0; i < n; i++) {
            const V3 d = W[i] - w;
            if (dot(d, d) <= EQUAL_VERTEX_THRESHOLD) { found = true; break; }
        }
        if (w.x == lastW.x && w.y == lastW.y && w.z == lastW.z) return true;
        return found;
    }

    static bool closest_triangle(const V3 &p, const V3 &a, const V3 &b, const V3 &c, SubSimplexClosest &r) {
        r.reset_used();
        const V3 ab = b - a, ac = c - a, ap = p - a;
        const float d1 = dot(ab, ap), d2 = dot(ac, ap);
        if (d1 <= 0.f && d2 <= 0.f) { r.closest = a; r.usedA = true; r.set_bary(1, 0, 0, 0); return true; }
        const V3 bp = p - b;
        const float d3 = dot(ab, bp), d4 = dot(ac, bp);
        if (d3 >= 0.f && d4 <= d3) { r.closest = b; r.usedB = true; r.set_bary(0, 1, 0, 0); return true; }
        const float vc = d1 * d4 - d3 * d2;
        if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) {
            const float v = d1 / (d1 - d3);
            r.closest = a + v * ab; r.usedA = true; r.usedB = true; r.set_bary(1 - v, v, 0, 0);
            return true;
        }
        const V3 cp = p - c;
        const float d5 = dot(ab, cp), d6 = dot(ac, cp);
        if (d6 >= 0.f && d5 <= d6) { r.closest = c; r.usedC = true; r.set_bary(0, 0, 1, 0); return true; }
        const float vb = d5 * d2 - d1 * d6;
        if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) {
            const float w = d2 / (d2 - d6);
            r.closest = a + w * ac; r.usedA = true; r.usedC = true; r.set_bary(1 - w, 0, w, 0);
            return true;
        }
        const float va = d3 * d6 - d5 * d4;
        if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {
            const float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
            r.closest = b + w * (c - b); r.usedB = true; r.usedC = true; r.set_bary(0, 1 - w, w, 0);
            return true;
        }
        const float denom = 1.0f / (va + vb + vc);
        const float v = vb * denom, w = vc * denom;
        r.closest = a + ab * v + ac * w;
        r.usedA = true; r.usedB = true; r.usedC = true;
        r.set_bary(1 - v - w, v, w, 0);
        return true;
    }
    // -1: degenerate, 0: inside, 1: outside
    static int outside_of_plane(const V3 &p, const V3 &a, const V3 &b, const V3 &c, const V3 &d) {
        const V3 normal = cross(b - a, c - a);
        const float signp = dot(p - a, normal);
        const float signd = dot(d - a, normal);
        if (signd * signd < (1e-4f * 1e-4f)) return -1;
        return signp * signd < 0.f ? 1 : 0;
    }
    static bool closest_tetrahedron(const V3 &p, const V3 &a, const V3 &b, const V3 &c, const V3 &d, SubSimplexClosest &f) {
        SubSimplexClosest t;
        f.closest = p;
        f.reset_used();
        f.usedA = f.usedB = f.usedC = f.usedD = true;
        const int oABC = outside_of_plane(p, a, b, c, d), oACD = outside_of_plane(p, a, c, d, b);
        const int oADB = outside_of_plane(p, a, d, b, c), oBDC = outside_of_plane(p, b, d, c, a);
        if (oABC < 0 || oACD < 0 || oADB < 0 || oBDC < 0) { f.degenerate = true; return false; }
        if (!oABC && !oACD && !oADB && !oBDC) return false;
        float best = SIMD_INFINITY;
        if (oABC) {
            closest_triangle(p, a, b, c, t);
            const V3 q = t.closest;
            const float sq = dot(q - p, q - p);
            if (sq < best) {
                best = sq; f.closest = q; f.reset_used();
                f.usedA = t.usedA; f.usedB = t.usedB; f.usedC = t.usedC;
                f.set_bary(t.bary[0], t.bary[1], t.bary[2], 0);
            }
        }
        if (oACD) {
            closest_triangle(p, a, c, d, t);
            const V3 q = t.closest;
            const float sq = dot(q - p, q - p);
            if (sq < best) {
                best = sq; f.closest = q; f.reset_used();
                f.usedA = t.usedA; f.usedC = t.usedB; f.usedD = t.usedC;
                f.set_bary(t.bary[0], 0, t.bary[1], t.bary[2]);
            }
        }
        if (oADB) {
            closest_triangle(p, a, d, b, t);
            const V3 q = t.closest;
            const float sq = dot(q - p, q - p);
            if (sq < best) {
                best = sq; f.closest = q; f.reset_used();
                f.usedA = t.usedA; f.usedB = t.usedC; f.usedD = t.usedB;
                f.set_bary(t.bary[0], t.bary[2], 0, t.bary[1]);
            }
        }
        if (oBDC) {
            closest_triangle(p, b, d, c, t);
            const V3 q = t.closest;
            const float sq = dot(q - p, q - p);
            if (sq < best) {
                best = sq; f.closest = q; f.reset_used();
                f.usedB = t.usedA; f.usedC = t.usedC; f.usedD = t.usedB;
                f.set_bary(0, t.bary[0], t.bary[2], t.bary[1]);
            }
        }
        return true;
    }

    bool update() {  // updateClosestVectorAndPoints
        if (!needsUpdate) return cachedValid;
        bc.reset();
        needsUpdate = false;
        switch (n) {
            case 0: cachedValid = false; break;
            case 1:
                cachedP1 = P[0]; cachedP2 = Qp[0]; cachedV = cachedP1 - cachedP2;
                bc.reset(); bc.set_bary(1, 0, 0, 0);
                cachedValid = bc.valid();
                break;
            case 2: {
                const V3 from = W[0], to = W[1];
                V3 diff = V3(0, 0, 0) - from;
                const V3 v = to - from;
                float t = dot(v, diff);
                if (t > 0.f) {
                    const float dotVV = dot(v, v);
                    if (t < dotVV) { t /= dotVV; diff -= t * v; bc.usedA = true; bc.usedB = true; }
                    else { t = 1; diff -= v; bc.usedB = true; }
                } else { t = 0; bc.usedA = true; }
                bc.set_bary(1 - t, t, 0, 0);
                cachedP1 = P[0] + t * (P[1] - P[0]);
                cachedP2 = Qp[0] + t * (Qp[1] - Qp[0]);
                cachedV = cachedP1 - cachedP2;
                reduce(bc);
                cachedValid = bc.valid();
                break;
            }
            case 3: {
                closest_triangle(V3(0, 0, 0), W[0], W[1], W[2], bc);
                cachedP1 = P[0] * bc.bary[0] + P[1] * bc.bary[1] + P[2] * bc.bary[2];
                cachedP2 = Qp[0] * bc.bary[0] + Qp[1] * bc.bary[1] + Qp[2] * bc.bary[2];
                cachedV = cachedP1 - cachedP2;
                reduce(bc);
                cachedValid = bc.valid();
                break;
            }
            case 4: {
                const bool sep = closest_tetrahedron(V3(0, 0, 0), W[0], W[1], W[2], W[3], bc);
                if (sep) {
                    cachedP1 = P[0] * bc.bary[0] + P[1] * bc.bary[1] + P[2] * bc.bary[2] + P[3] * bc.bary[3];
                    cachedP2 = Qp[0] * bc.bary[0] + Qp[1] * bc.bary[1] + Qp[2] * bc.bary[2] + Qp[3] * bc.bary[3];
                    cachedV = cachedP1 - cachedP2;
                    reduce(bc);
                } else {
                    if (bc.degenerate) cachedValid = false;
                    else { cachedValid = true; cachedV = V3(0, 0, 0); }  // the origin is inside the tetrahedron
                    break;
                }
                cachedValid = bc.valid();
                break;
            }
            default: cachedValid = false;
        }
        return cachedValid;
    }
    bool closest(V3 &v) { const bool ok = update(); v = cachedV; return ok; }
    void compute_points(V3 &p1, V3 &p2) { update(); p1 = cachedP1; p2 = cachedP2; }
};

struct InnerResult {  // btDiscreteCollisionDetectorInterface::Result of one detector run
    bool has = false;
    V3 normalOnB, pointOnB;
    float depth = 0;
};

// btMinkowskiPenetrationDepthSolver::getPenetrationDirections (NUM_UNITSPHERE_POINTS = 42)
static const float kPenDirs[42][3] = {
    {0.000000f, -0.000000f, -1.000000f}, {0.723608f, -0.525725f, -0.447219f}, {-0.276388f, -0.850649f, -0.447219f},
    {-0.894426f, -0.000000f, -0.447216f}, {-0.276388f, 0.850649f, -0.447220f}, {0.723608f, 0.525725f, -0.447219f},
    {0.276388f, -0.850649f, 0.447220f}, {-0.723608f, -0.525725f, 0.447219f}, {-0.723608f, 0.525725f, 0.447219f},
    {0.276388f, 0.850649f, 0.447219f}, {0.894426f, 0.000000f, 0.447216f}, {-0.000000f, 0.000000f, 1.000000f},
    {0.425323f, -0.309011f, -0.850654f}, {-0.162456f, -0.499995f, -0.850654f}, {0.262869f, -0.809012f, -0.525738f},
    {0.425323f, 0.309011f, -0.850654f}, {0.850648f, -0.000000f, -0.525736f}, {-0.525730f, -0.000000f, -0.850652f},
    {-0.688190f, -0.499997f, -0.525736f}, {-0.162456f, 0.499995f, -0.850654f}, {-0.688190f, 0.499997f, -0.525736f},
    {0.262869f, 0.809012f, -0.525738f}, {0.951058f, 0.309013f, 0.000000f}, {0.951058f, -0.309013f, 0.000000f},
    {0.587786f, -0.809017f, 0.000000f}, {0.000000f, -1.000000f, 0.000000f}, {-0.587786f, -0.809017f, 0.000000f},
    {-0.951058f, -0.309013f, -0.000000f}, {-0.951058f, 0.309013f, -0.000000f}, {-0.587786f, 0.809017f, -0.000000f},
    {-0.000000f, 1.000000f, -0.000000f}, {0.587786f, 0.809017f, -0.000000f}, {0.688190f, -0.499997f, 0.525736f},
    {-0.262869f, -0.809012f, 0.525738f}, {-0.850648f, 0.000000f, 0.525736f}, {-0.262869f, 0.809012f, 0.525738f},
    {0.688190f, 0.499997f, 0.525736f}, {0.525730f, 0.000000f, 0.850652f}, {0.162456f, -0.499995f, 0.850654f},
    {-0.425323f, -0.309011f, 0.850654f}, {-0.425323f, 0.309011f, 0.850654f}, {0.162456f, 0.499995f, 0.850654f}};


// ------------------------------------------------------------------------------------------------
// The libccd-derived intersection pre-test at the top of btGjkPairDetector::getClosestPointsNonVirtual (bullet3 >= 2.88:
// btComputeSupport / btDoSimplex2,3,4 / btVec3PointSegmentDist2 / btVec3PointTriDist2).  It decides `status`: 0 = the un-margined
// cores intersect (the penetration branch is then forced whatever the Voronoi loop found), -1 = they do not.
// ------------------------------------------------------------------------------------------------
struct CcdSupport { V3 v, v1, v2; };
struct CcdSimplex {
    CcdSupport ps[4];
    int last = -1;
    int size() const { return last + 1; }
    void add(const CcdSupport &s) { ps[++last] = s; }
    void set(int pos, const CcdSupport &s) { ps[pos] = s; }
    void set_size(int n) { last = n - 1; }
};
static bool fuzzy_zero(float x) { return std::fabs(x) < SIMD_EPSILON; }
static bool fuzzy_zero(double x) { return std::fabs(x) < (double) SIMD_EPSILON; }   // btFuzzyZero(btScalar) on a double argument converts: see below
static int ccd_eq(float _a, float _b) {
    const float ab = std::fabs(_a - _b);
    if (std::fabs(ab) < SIMD_EPSILON) return 1;
    const float a = std::fabs(_a), b = std::fabs(_b);
    if (b > a) return ab < SIMD_EPSILON * b;
    return ab < SIMD_EPSILON * a;
}
static int ccd_sign(float v) { if (fuzzy_zero(v)) return 0; return v < 0.f ? -1 : 1; }
static bool ccd_vec_eq(const V3 &a, const V3 &b) { return ccd_eq(a.x, b.x) && ccd_eq(a.y, b.y) && ccd_eq(a.z, b.z); }
static float ccd_dist2(const V3 &a, const V3 &b) { const V3 ab = a - b; return dot(ab, ab); }
static V3 triple_cross(const V3 &a, const V3 &b, const V3 &c) { return cross(cross(a, b), c); }
static float point_segment_dist2(const V3 &P, const V3 &x0, const V3 &b, V3 *witness) {
    float dist, t;
    V3 d = b - x0;
    const V3 a = x0 - P;
    t = -1.f * dot(a, d);
    t /= dot(d, d);
    if (t < 0.f || fuzzy_zero(t)) {
        dist = ccd_dist2(x0, P);
        if (witness) *witness = x0;
    } else if (t > 1.f || ccd_eq(t, 1.f)) {
        dist = ccd_dist2(b, P);
        if (witness) *witness = b;
    } else {
        if (witness) {
            *witness = d;
            *witness = *witness * t;
            *witness = *witness + x0;
            dist = ccd_dist2(*witness, P);
        } else {
            d = d * t;
            d = d + a;
            dist = dot(d, d);
        }
    }
    return dist;
}
static float point_tri_dist2(const V3 &P, const V3 &x0, const V3 &B, const V3 &C, V3 *witness) {
    // (the original keeps u..t and the distances in double)
    V3 d1 = B - x0, d2 = C - x0;
    const V3 a = x0 - P;
    const double u = dot(a, a), v = dot(d1, d1), w = dot(d2, d2), p = dot(a, d1), q = dot(a, d2), r = dot(d1, d2);
    const double s = (q * r - w * p) / (w * v - r * r);
    const double t = (-s * r - q) / w;
    double dist, dist2;
    V3 witness2;
    // btFuzzyZero / ccdEq take btScalar: the doubles are converted to float at the call
    const float sf = (float) s, tf = (float) t, tsf = (float) (t + s);
    if ((fuzzy_zero(sf) || s > 0.0) && (ccd_eq(sf, 1.f) || s < 1.0) && (fuzzy_zero(tf) || t > 0.0) && (ccd_eq(tf, 1.f) || t < 1.0) &&
        (ccd_eq(tsf, 1.f) || t + s < 1.0)) {
        if (witness) {
            d1 = d1 * (float) s;
            d2 = d2 * (float) t;
            *witness = x0;
            *witness = *witness + d1;
            *witness = *witness + d2;
            dist = ccd_dist2(*witness, P);
        } else {
            dist = s * s * v;
            dist += t * t * w;
            dist += 2.0 * s * t * r;
            dist += 2.0 * s * p;
            dist += 2.0 * t * q;
            dist += u;
        }
    } else {
        dist = point_segment_dist2(P, x0, B, witness);
        dist2 = point_segment_dist2(P, x0, C, &witness2);
        if (dist2 < dist) { dist = dist2; if (witness) *witness = witness2; }
        dist2 = point_segment_dist2(P, B, C, &witness2);
        if (dist2 < dist) { dist = dist2; if (witness) *witness = witness2; }
    }
    return (float) dist;
}
static int do_simplex2(CcdSimplex &sx, V3 &dir) {
    const CcdSupport A = sx.ps[sx.last], B = sx.ps[0];
    const V3 AB = B.v - A.v;
    const V3 AO = A.v * -1.f;
    const float d = dot(AB, AO);
    const V3 tmp = cross(AB, AO);
    if (fuzzy_zero(dot(tmp, tmp)) && d > 0.f) return 1;
    if (fuzzy_zero(d) || d < 0.f) {
        sx.set(0, A);
        sx.set_size(1);
        dir = AO;
    } else {
        dir = triple_cross(AB, AO, AB);
    }
    return 0;
}
static int do_simplex3(CcdSimplex &sx, V3 &dir) {
    const CcdSupport A = sx.ps[sx.last], B = sx.ps[1], C = sx.ps[0];
    const V3 origin(0, 0, 0);
    const float dist = point_tri_dist2(origin, A.v, B.v, C.v, nullptr);
    if (fuzzy_zero(dist)) return 1;
    if (ccd_vec_eq(A.v, B.v) || ccd_vec_eq(A.v, C.v)) return -1;
    const V3 AO = A.v * -1.f;
    const V3 AB = B.v - A.v, AC = C.v - A.v;
    const V3 ABC = cross(AB, AC);
    V3 tmp = cross(ABC, AC);
    float d = dot(tmp, AO);
    auto label45 = [&]() {
        d = dot(AB, AO);
        if (fuzzy_zero(d) || d > 0.f) {
            sx.set(0, B);
            sx.set(1, A);
            sx.set_size(2);
            dir = triple_cross(AB, AO, AB);
        } else {
            sx.set(0, A);
            sx.set_size(1);
            dir = AO;
        }
    };
    if (fuzzy_zero(d) || d > 0.f) {
        d = dot(AC, AO);
        if (fuzzy_zero(d) || d > 0.f) {
            sx.set(1, A);
            sx.set_size(2);
            dir = triple_cross(AC, AO, AC);
        } else {
            label45();
        }
    } else {
        tmp = cross(AB, ABC);
        d = dot(tmp, AO);
        if (fuzzy_zero(d) || d > 0.f) {
            label45();
        } else {
            d = dot(ABC, AO);
            if (fuzzy_zero(d) || d > 0.f) {
                dir = ABC;
            } else {
                const CcdSupport Ctmp = C;
                sx.set(0, B);
                sx.set(1, Ctmp);
                dir = ABC * -1.f;
            }
        }
    }
    return 0;
}
static int do_simplex4(CcdSimplex &sx, V3 &dir) {
    const CcdSupport A = sx.ps[sx.last], B = sx.ps[2], C = sx.ps[1], D = sx.ps[0];
    const V3 origin(0, 0, 0);
    float dist = point_tri_dist2(A.v, B.v, C.v, D.v, nullptr);
    if (fuzzy_zero(dist)) return -1;
    dist = point_tri_dist2(origin, A.v, B.v, C.v, nullptr);
    if (fuzzy_zero(dist)) return 1;
    dist = point_tri_dist2(origin, A.v, C.v, D.v, nullptr);
    if (fuzzy_zero(dist)) return 1;
    dist = point_tri_dist2(origin, A.v, B.v, D.v, nullptr);
    if (fuzzy_zero(dist)) return 1;
    dist = point_tri_dist2(origin, B.v, C.v, D.v, nullptr);
    if (fuzzy_zero(dist)) return 1;
    const V3 AO = A.v * -1.f;
    const V3 AB = B.v - A.v, AC = C.v - A.v, AD = D.v - A.v;
    const V3 ABC = cross(AB, AC), ACD = cross(AC, AD), ADB = cross(AD, AB);
    const int B_on_ACD = ccd_sign(dot(ACD, AB)), C_on_ADB = ccd_sign(dot(ADB, AC)), D_on_ABC = ccd_sign(dot(ABC, AD));
    const bool AB_O = ccd_sign(dot(ACD, AO)) == B_on_ACD, AC_O = ccd_sign(dot(ADB, AO)) == C_on_ADB, AD_O = ccd_sign(dot(ABC, AO)) == D_on_ABC;
    if (AB_O && AC_O && AD_O) return 1;
    if (!AB_O) {
        sx.set(2, A);
        sx.set_size(3);
    } else if (!AC_O) {
        sx.set(1, D);
        sx.set(0, B);
        sx.set(2, A);
        sx.set_size(3);
    } else {
        sx.set(0, C);
        sx.set(1, B);
        sx.set(2, A);
        sx.set_size(3);
    }
    return do_simplex3(sx, dir);
}
static int do_simplex(CcdSimplex &sx, V3 &dir) {
    if (sx.size() == 2) return do_simplex2(sx, dir);
    if (sx.size() == 3) return do_simplex3(sx, dir);
    return do_simplex4(sx, dir);
}
static void ccd_support(const ConvexView &A, const Xf &trA, const ConvexView &B, const Xf &trB, const V3 &dir, CcdSupport &out) {
    const V3 sepInA = dir * trA.b, sepInB = (-dir) * trB.b;
    out.v1 = trA(local_support(A, sepInA));
    out.v2 = trB(local_support(B, sepInB));
    out.v = out.v1 - out.v2;
}
// returns status: 0 intersect, -1 not; iterations for the diagnostics
static int ccd_intersect(const ConvexView &A, const Xf &trA, const ConvexView &B, const Xf &trB, int &iterations) {
    int status = -2;
    CcdSimplex sx;
    V3 dir(1, 0, 0);
    CcdSupport last;
    ccd_support(A, trA, B, trB, dir, last);
    sx.add(last);
    dir = -last.v;
    for (iterations = 0; iterations < GJK_MAX_ITER; iterations++) {
        ccd_support(A, trA, B, trB, dir, last);
        const float delta = dot(last.v, dir);
        if (delta < 0) { status = -1; break; }
        sx.add(last);
        const int res = do_simplex(sx, dir);
        if (res == 1) { status = 0; break; }
        else if (res == -1) { status = -1; break; }
        if (fuzzy_zero(dot(dir, dir))) status = -1;
        if (length2(dir) < SIMD_EPSILON) { status = -1; break; }
        if (length2(dir) < SIMD_EPSILON * SIMD_EPSILON) { status = -1; break; }
    }
    return status;
}

struct Detector {
    const ConvexView &A, &B;
    bool with_penetration;  // the nested detector of the penetration solver has none (no recursion)
    int cur_iter = 0, degenerate = 0, last_method = -1, ccd_status = -2, ccd_iters = 0;
    bool used_pen = false;
    Detector(const ConvexView &a, const ConvexView &b, bool pen) : A(a), B(b), with_penetration(pen) {}

    bool pen_depth(const Xf &transA, const Xf &transB, V3 &v, V3 &pa, V3 &pb);

    // btGjkPairDetector::getClosestPointsNonVirtual
    InnerResult run(const Xf &transA_in, const Xf &transB_in, float max_dist2) {
        InnerResult out;
        float distance = 0.f;
        V3 normalInB(0, 0, 0), pointOnA, pointOnB;
        Xf localA = transA_in, localB = transB_in;
        const V3 positionOffset = (localA.o + localB.o) * 0.5f;
        localA.o -= positionOffset;
        localB.o -= positionOffset;
        const float marginA = A.margin, marginB = B.margin;
        cur_iter = 0;
        V3 axis(0, 1, 0);  // m_cachedSeparatingAxis
        bool isValid = false, checkSimplex = false;
        const bool checkPenetration = true;
        degenerate = 0;
        last_method = -1;
        V3 orgNormalInB(0, 0, 0);
        const float margin = marginA + marginB;
        float squaredDistance = BT_LARGE_FLOAT, delta = 0.f;
        int status = -2;
        if (g_ccd_pretest && with_penetration) status = ccd_intersect(A, localA, B, localB, ccd_iters);
        ccd_status = status;
        Simplex sx;
        sx.reset();
        for (;;) {
            const V3 sepInA = (-axis) * localA.b;
            const V3 sepInB = axis * localB.b;
            const V3 pInA = local_support(A, sepInA);
            const V3 qInB = local_support(B, sepInB);
            const V3 pWorld = localA(pInA);
            const V3 qWorld = localB(qInB);
            const V3 w = pWorld - qWorld;
            delta = dot(axis, w);
            if (delta > 0.f && delta * delta > squaredDistance * max_dist2) { degenerate = 10; checkSimplex = true; break; }
            if (sx.in_simplex(w)) { degenerate = 1; checkSimplex = true; break; }
            const float f0 = squaredDistance - delta, f1 = squaredDistance * REL_ERROR2;
            if (f0 <= f1) { degenerate = f0 <= 0.f ? 2 : 11; checkSimplex = true; break; }
            sx.add(w, pWorld, qWorld);
            V3 newAxis;
            if (!sx.closest(newAxis)) { degenerate = 3; checkSimplex = true; break; }
            if (length2(newAxis) < REL_ERROR2) { axis = newAxis; degenerate = 6; checkSimplex = true; break; }
            const float prev = squaredDistance;
            squaredDistance = length2(newAxis);
            if (prev - squaredDistance <= SIMD_EPSILON * prev) { checkSimplex = true; degenerate = 12; break; }
            axis = newAxis;
            if (cur_iter++ > GJK_MAX_ITER) break;
            if (sx.full()) { degenerate = 13; break; }
        }
        if (checkSimplex) {
            sx.compute_points(pointOnA, pointOnB);
            normalInB = axis;
            const float lenSqr = length2(axis);
            if (lenSqr < REL_ERROR2) degenerate = 5;
            if (lenSqr > SIMD_EPSILON * SIMD_EPSILON) {
                const float rlen = 1.0f / std::sqrt(lenSqr);
                normalInB *= rlen;
                const float s = std::sqrt(squaredDistance);
                pointOnA -= axis * (marginA / s);
                pointOnB += axis * (marginB / s);
                distance = (1.0f / rlen) - margin;
                isValid = true;
                orgNormalInB = normalInB;
                last_method = 1;
            } else last_method = 2;
        }
        const bool catchDegenerate = with_penetration && degenerate != 0 && (distance + margin) < GJK_EPA_PENETRATION_TOLERANCE;
        if ((checkPenetration && (!isValid || catchDegenerate)) || status == 0) {
            if (with_penetration) {
                V3 tmpA, tmpB;
                axis = V3(0, 0, 0);
                used_pen = true;
                const bool isValid2 = pen_depth(localA, localB, axis, tmpA, tmpB);
                if (length2(axis) != 0.f) {
                    if (isValid2) {
                        V3 tmpN = tmpB - tmpA;
                        float lenSqr = length2(tmpN);
                        if (lenSqr <= SIMD_EPSILON * SIMD_EPSILON) { tmpN = axis; lenSqr = length2(axis); }
                        if (lenSqr > SIMD_EPSILON * SIMD_EPSILON) {
                            tmpN = tmpN / std::sqrt(lenSqr);
                            const float distance2 = -length(tmpA - tmpB);
                            last_method = 3;
                            if (!isValid || distance2 < distance) {
                                distance = distance2; pointOnA = tmpA; pointOnB = tmpB; normalInB = tmpN; isValid = true;
                            } else last_method = 8;
                        } else last_method = 9;
                    } else {
                        // the sampled directions found no overlap but the nested run returned a positive distance
                        if (length2(axis) > 0.f) {
                            const float distance2 = length(tmpA - tmpB) - margin;
                            if (!isValid || distance2 < distance) {
                                distance = distance2; pointOnA = tmpA; pointOnB = tmpB;
                                pointOnA -= axis * marginA;
                                pointOnB += axis * marginB;
                                normalInB = normalized(axis);
                                isValid = true;
                                last_method = 6;
                            } else last_method = 5;
                        }
                    }
                }
            }
        }
        if (isValid && (distance < 0.f || distance * distance < max_dist2)) {
            // the normal check at the end of getClosestPointsNonVirtual: the candidate normal, its opposite and the plain
            // GJK normal are compared by the separation they give along themselves
            auto sep_along = [&](const V3 &nrm) {
                const V3 sA = (-nrm) * localA.b, sB = nrm * localB.b;
                const V3 pW = localA(local_support(A, sA)), qW = localB(local_support(B, sB));
                return pW - qW;
            };
            float d2 = 0.f;
            { const V3 w = sep_along(orgNormalInB); d2 = dot(orgNormalInB, w) - margin; }
            float d1 = 0.f;
            { const V3 w = sep_along(-normalInB); d1 = dot(-normalInB, w) - margin; }
            float d0 = 0.f;
            { const V3 w = sep_along(normalInB); d0 = dot(normalInB, w) - margin; }
            if (d1 > d0) { last_method = 10; normalInB *= -1.f; }
            if (length2(orgNormalInB) != 0.f) {
                if (d2 > d0 && d2 > d1 && d2 > distance) { normalInB = orgNormalInB; distance = d2; }
            }
            out.has = true;
            out.normalOnB = normalInB;
            out.pointOnB = pointOnB + positionOffset;
            out.depth = distance;
        }
        return out;
    }
};

// m_penetrationDepthSolver->calcPenDepth: btGjkEpaPenetrationDepthSolver (the reference's configuration) unless the legacy
// switch asks for btMinkowskiPenetrationDepthSolver::calcPenDepth (convex hulls add no preferred directions)
bool Detector::pen_depth(const Xf &transA, const Xf &transB, V3 &v, V3 &pa, V3 &pb) {
    if (g_penetration_solver == 0) return epa_calc_pen_depth(A, B, transA, transB, v, pa, pb);
    float minProj = BT_LARGE_FLOAT;
    V3 minNorm(0, 0, 0), minA, minB;
    for (int i = 0; i < 42; i++) {
        const V3 norm(kPenDirs[i][0], kPenDirs[i][1], kPenDirs[i][2]);
        const V3 sepInA = (-norm) * transA.b;
        const V3 sepInB = norm * transB.b;
        const V3 pWorld = transA(local_support(A, sepInA));
        const V3 qWorld = transB(local_support(B, sepInB));
        const V3 w = qWorld - pWorld;
        const float delta = dot(norm, w);
        if (delta < minProj) { minProj = delta; minNorm = norm; minA = pWorld; minB = qWorld; }
    }
    minA += minNorm * A.margin;
    minB -= minNorm * B.margin;
    if (minProj < 0.f) return false;
    const float extraSeparation = 0.5f;
    minProj += extraSeparation + (A.margin + B.margin);
    const V3 offset = minNorm * minProj;
    Xf displaced = transA;
    displaced.o = transA.o + offset;
    Detector nested(A, B, false);
    const InnerResult res = nested.run(displaced, transB, BT_LARGE_FLOAT);
    const float correctedMinNorm = minProj - res.depth;
    if (res.has) {
        pa = res.pointOnB - minNorm * correctedMinNorm;
        pb = res.pointOnB;
        v = minNorm;
    }
    return res.has;
}

}  // namespace

ClosestResult gjk_closest_points(const ConvexView &A, const ConvexView &B, float max_dist2) {
    Detector det(A, B, true);
    const InnerResult r = det.run(A.xf, B.xf, max_dist2);
    ClosestResult o;
    o.has = r.has;
    o.normalOnB = r.normalOnB;
    o.pointOnB = r.pointOnB;
    o.distance = r.depth;
    o.iterations = det.cur_iter;
    o.degenerate = det.degenerate;
    o.method = det.last_method;
    o.used_penetration = det.used_pen;
    o.ccd_status = det.ccd_status;
    o.ccd_iterations = det.ccd_iters;
    return o;
}

}  // namespace orc
