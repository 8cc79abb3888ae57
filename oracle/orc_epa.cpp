// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_epa.h for scope and citations).
#include "orc_epa.h"

namespace orc {
namespace {

// btConvexShape::localGetSupportVertexNonVirtual: core support + margin along the normalised direction
V3 support_with_margin(const ConvexView &S, const V3 &dir) {
    V3 n = dir;
    if (length2(n) < SIMD_EPSILON * SIMD_EPSILON) n = V3(-1.f, -1.f, -1.f);
    n = normalized(n);
    return local_support(S, n) + S.margin * n;
}

// gjkepa2_impl::MinkowskiDiff
struct MinkowskiDiff {
    const ConvexView *s0, *s1;
    M3 toshape1;  // wtrs1.basis^T * wtrs0.basis
    Xf toshape0;  // wtrs0^-1 * wtrs1
    bool margins;
    V3 ls(const ConvexView &S, const V3 &d) const { return margins ? support_with_margin(S, d) : local_support(S, d); }
    V3 support0(const V3 &d) const { return ls(*s0, d); }
    V3 support1(const V3 &d) const { return toshape0(ls(*s1, toshape1 * d)); }
    V3 support(const V3 &d) const { return support0(d) - support1(-d); }
    V3 support(const V3 &d, int index) const { return index ? support1(d) : support0(d); }
};

void initialize(const ConvexView &A, const Xf &wtrs0, const ConvexView &B, const Xf &wtrs1, EpaResults &results, MinkowskiDiff &shape,
                bool withmargins) {
    results.witnesses[0] = results.witnesses[1] = V3(0, 0, 0);
    results.status = EpaResults::Separated;
    shape.s0 = &A;
    shape.s1 = &B;
    shape.toshape1 = wtrs1.b.transpose() * wtrs0.b;  // btMatrix3x3::transposeTimes
    // btTransform::inverseTimes: basis^T * t.basis, (t.origin - origin) * basis
    shape.toshape0.b = wtrs0.b.transpose() * wtrs1.b;
    shape.toshape0.o = (wtrs1.o - wtrs0.o) * wtrs0.b;
    shape.margins = withmargins;
}

struct SV { V3 d, w; };

float det3(const V3 &a, const V3 &b, const V3 &c) {
    return a.y * b.z * c.x + a.z * b.x * c.y - a.x * b.z * c.y - a.y * b.x * c.z + a.x * b.y * c.z - a.z * b.y * c.x;
}

// gjkepa2_impl::GJK
struct GJK2 {
    enum Status { Valid, Inside, Failed };
    struct Simplex { int c[4]; float p[4]; int rank; };
    MinkowskiDiff shape;
    V3 ray;
    float distance = 0.f;
    Simplex simplices[2];
    SV store[4];
    int free_[4];
    int nfree = 0, current = 0;
    Simplex *simplex = nullptr;
    Status status = Failed;
    int iterations_done = 0;

    void getsupport(const V3 &d, SV &sv) const {
        sv.d = d / length(d);
        sv.w = shape.support(sv.d);
    }
    void removevertice(Simplex &s) { free_[nfree++] = s.c[--s.rank]; }
    void appendvertice(Simplex &s, const V3 &v) {
        s.p[s.rank] = 0;
        s.c[s.rank] = free_[--nfree];
        getsupport(v, store[s.c[s.rank++]]);
    }
    const V3 &W(const Simplex &s, int i) const { return store[s.c[i]].w; }

    static float projectorigin(const V3 &a, const V3 &b, float *w, unsigned &m) {
        const V3 d = b - a;
        const float l = length2(d);
        if (l > GJK2_SIMPLEX2_EPS) {
            const float t = l > 0 ? -dot(a, d) / l : 0;
            if (t >= 1) { w[0] = 0; w[1] = 1; m = 2; return length2(b); }
            else if (t <= 0) { w[0] = 1; w[1] = 0; m = 1; return length2(a); }
            else { w[0] = 1 - (w[1] = t); m = 3; return length2(a + d * t); }
        }
        return -1;
    }
    static float projectorigin(const V3 &a, const V3 &b, const V3 &c, float *w, unsigned &m) {
        static const unsigned imd3[] = {1, 2, 0};
        const V3 *vt[] = {&a, &b, &c};
        const V3 dl[] = {a - b, b - c, c - a};
        const V3 n = cross(dl[0], dl[1]);
        const float l = length2(n);
        if (l > GJK2_SIMPLEX3_EPS) {
            float mindist = -1;
            float subw[2] = {0.f, 0.f};
            unsigned subm = 0;
            for (unsigned i = 0; i < 3; ++i) {
                if (dot(*vt[i], cross(dl[i], n)) > 0) {
                    const unsigned j = imd3[i];
                    const float subd = projectorigin(*vt[i], *vt[j], subw, subm);
                    if (mindist < 0 || subd < mindist) {
                        mindist = subd;
                        m = ((subm & 1) ? 1u << i : 0u) + ((subm & 2) ? 1u << j : 0u);
                        w[i] = subw[0];
                        w[j] = subw[1];
                        w[imd3[j]] = 0;
                    }
                }
            }
            if (mindist < 0) {
                const float d = dot(a, n);
                const float s = std::sqrt(l);
                const V3 p = n * (d / l);
                mindist = length2(p);
                m = 7;
                w[0] = length(cross(dl[1], b - p)) / s;
                w[1] = length(cross(dl[2], c - p)) / s;
                w[2] = 1 - (w[0] + w[1]);
            }
            return mindist;
        }
        return -1;
    }
    static float projectorigin(const V3 &a, const V3 &b, const V3 &c, const V3 &d, float *w, unsigned &m) {
        static const unsigned imd3[] = {1, 2, 0};
        const V3 *vt[] = {&a, &b, &c, &d};
        const V3 dl[] = {a - d, b - d, c - d};
        const float vl = det3(dl[0], dl[1], dl[2]);
        const bool ng = (vl * dot(a, cross(b - c, a - b))) <= 0;
        if (ng && std::fabs(vl) > GJK2_SIMPLEX4_EPS) {
            float mindist = -1;
            float subw[3] = {0.f, 0.f, 0.f};
            unsigned subm = 0;
            for (unsigned i = 0; i < 3; ++i) {
                const unsigned j = imd3[i];
                const float s = vl * dot(d, cross(dl[i], dl[j]));
                if (s > 0) {
                    const float subd = projectorigin(*vt[i], *vt[j], d, subw, subm);
                    if (mindist < 0 || subd < mindist) {
                        mindist = subd;
                        m = ((subm & 1) ? 1u << i : 0u) + ((subm & 2) ? 1u << j : 0u) + ((subm & 4) ? 8u : 0u);
                        w[i] = subw[0];
                        w[j] = subw[1];
                        w[imd3[j]] = 0;
                        w[3] = subw[2];
                    }
                }
            }
            if (mindist < 0) {
                mindist = 0;
                m = 15;
                w[0] = det3(c, b, d) / vl;
                w[1] = det3(a, c, d) / vl;
                w[2] = det3(b, a, d) / vl;
                w[3] = 1 - (w[0] + w[1] + w[2]);
            }
            return mindist;
        }
        return -1;
    }

    Status evaluate(const MinkowskiDiff &shapearg, const V3 &guess) {
        unsigned iterations = 0;
        float sqdist = 0, alpha = 0;
        V3 lastw[4];
        unsigned clastw = 0;
        free_[0] = 0; free_[1] = 1; free_[2] = 2; free_[3] = 3;
        nfree = 4;
        current = 0;
        status = Valid;
        shape = shapearg;
        distance = 0;
        simplices[0].rank = 0;
        ray = guess;
        const float sqrl = length2(ray);
        appendvertice(simplices[0], sqrl > 0 ? -ray : V3(1, 0, 0));
        simplices[0].p[0] = 1;
        ray = W(simplices[0], 0);
        sqdist = sqrl;
        lastw[0] = lastw[1] = lastw[2] = lastw[3] = ray;
        do {
            const int next = 1 - current;
            Simplex &cs = simplices[current];
            Simplex &ns = simplices[next];
            const float rl = length(ray);
            if (rl < GJK2_MIN_DISTANCE) { status = Inside; break; }  // touching or inside
            appendvertice(cs, -ray);
            const V3 w = W(cs, cs.rank - 1);
            bool found = false;
            for (unsigned i = 0; i < 4; ++i) {
                if (length2(w - lastw[i]) < GJK2_DUPLICATED_EPS) { found = true; break; }
            }
            if (found) { removevertice(simplices[current]); break; }  // return old simplex
            lastw[clastw = (clastw + 1) & 3] = w;
            const float omega = dot(ray, w) / rl;
            alpha = omega > alpha ? omega : alpha;  // btMax(omega, alpha)
            if (((rl - alpha) - (GJK2_ACCURACY * rl)) <= 0) { removevertice(simplices[current]); break; }
            float weights[4];
            unsigned mask = 0;
            switch (cs.rank) {
                case 2: sqdist = projectorigin(W(cs, 0), W(cs, 1), weights, mask); break;
                case 3: sqdist = projectorigin(W(cs, 0), W(cs, 1), W(cs, 2), weights, mask); break;
                case 4: sqdist = projectorigin(W(cs, 0), W(cs, 1), W(cs, 2), W(cs, 3), weights, mask); break;
            }
            if (sqdist >= 0) {
                ns.rank = 0;
                ray = V3(0, 0, 0);
                current = next;
                for (unsigned i = 0, ni = cs.rank; i < ni; ++i) {
                    if (mask & (1u << i)) {
                        ns.c[ns.rank] = cs.c[i];
                        ns.p[ns.rank++] = weights[i];
                        ray += store[cs.c[i]].w * weights[i];
                    } else {
                        free_[nfree++] = cs.c[i];
                    }
                }
                if (mask == 15) status = Inside;
            } else {
                removevertice(simplices[current]);
                break;
            }
            status = ((++iterations) < (unsigned) GJK2_MAX_ITERATIONS) ? status : Failed;
        } while (status == Valid);
        simplex = &simplices[current];
        switch (status) {
            case Valid: distance = length(ray); break;
            case Inside: distance = 0; break;
            default: break;
        }
        iterations_done = (int) iterations;
        return status;
    }

    bool enclose_origin() {
        Simplex &s = *simplex;
        switch (s.rank) {
            case 1:
                for (unsigned i = 0; i < 3; ++i) {
                    V3 axis(0, 0, 0);
                    axis.at(i) = 1;
                    appendvertice(s, axis);
                    if (enclose_origin()) return true;
                    removevertice(s);
                    appendvertice(s, -axis);
                    if (enclose_origin()) return true;
                    removevertice(s);
                }
                break;
            case 2: {
                const V3 d = W(s, 1) - W(s, 0);
                for (unsigned i = 0; i < 3; ++i) {
                    V3 axis(0, 0, 0);
                    axis.at(i) = 1;
                    const V3 p = cross(d, axis);
                    if (length2(p) > 0) {
                        appendvertice(s, p);
                        if (enclose_origin()) return true;
                        removevertice(s);
                        appendvertice(s, -p);
                        if (enclose_origin()) return true;
                        removevertice(s);
                    }
                }
                break;
            }
            case 3: {
                const V3 n = cross(W(s, 1) - W(s, 0), W(s, 2) - W(s, 0));
                if (length2(n) > 0) {
                    appendvertice(s, n);
                    if (enclose_origin()) return true;
                    removevertice(s);
                    appendvertice(s, -n);
                    if (enclose_origin()) return true;
                    removevertice(s);
                }
                break;
            }
            case 4:
                if (std::fabs(det3(W(s, 0) - W(s, 3), W(s, 1) - W(s, 3), W(s, 2) - W(s, 3))) > 0) return true;
                break;
        }
        return false;
    }
};

// gjkepa2_impl::EPA
struct EPA {
    enum Status { Valid, Touching, Degenerated, NonConvex, InvalidHull, OutOfFaces, OutOfVertices, AccuraryReached, FallBack, Failed };
    struct Face {
        V3 n;
        float d;
        int c[3];   // vertices (>= 0: sv_store index; < 0: -(1 + index) into the GJK's own four-vertex store)
        int f[3];   // adjacent faces
        int l[2];   // list links (prev, next), -1 = none
        unsigned char e[3], pass;
    };
    struct List { int root = -1; unsigned count = 0; };
    struct Horizon { int cf = -1, ff = -1; unsigned nf = 0; };

    GJK2 &gjk;
    Status status = Failed;
    struct { int c[3]; float p[3]; int rank; } result;
    V3 normal;
    float depth = 0;
    SV sv_store[EPA_MAX_VERTICES];
    Face fc_store[EPA_MAX_FACES];
    unsigned nextsv = 0;
    List hull, stock;
    int iterations_done = 0;

    explicit EPA(GJK2 &g) : gjk(g) {
        normal = V3(0, 0, 0);
        for (int i = 0; i < EPA_MAX_FACES; ++i) append(stock, EPA_MAX_FACES - i - 1);
    }
    const SV &sv(int id) const { return id >= 0 ? sv_store[id] : gjk.store[-1 - id]; }

    void bind(int fa, unsigned ea, int fb, unsigned eb) {
        fc_store[fa].e[ea] = (unsigned char) eb; fc_store[fa].f[ea] = fb;
        fc_store[fb].e[eb] = (unsigned char) ea; fc_store[fb].f[eb] = fa;
    }
    void append(List &list, int face) {
        fc_store[face].l[0] = -1;
        fc_store[face].l[1] = list.root;
        if (list.root >= 0) fc_store[list.root].l[0] = face;
        list.root = face;
        ++list.count;
    }
    void remove(List &list, int face) {
        Face &F = fc_store[face];
        if (F.l[1] >= 0) fc_store[F.l[1]].l[0] = F.l[0];
        if (F.l[0] >= 0) fc_store[F.l[0]].l[1] = F.l[1];
        if (face == list.root) list.root = F.l[1];
        --list.count;
    }
    bool getedgedist(const Face &face, const SV &a, const SV &b, float &dist) const {
        const V3 ba = b.w - a.w;
        const V3 n_ab = cross(ba, face.n);  // outward edge normal in the triangle's plane
        const float a_dot_nab = dot(a.w, n_ab);
        if (a_dot_nab < 0) {  // outside of edge a->b
            const float ba_l2 = length2(ba);
            const float a_dot_ba = dot(a.w, ba);
            const float b_dot_ba = dot(b.w, ba);
            if (a_dot_ba > 0) dist = length(a.w);
            else if (b_dot_ba < 0) dist = length(b.w);
            else {
                const float a_dot_b = dot(a.w, b.w);
                const float q = (length2(a.w) * length2(b.w) - a_dot_b * a_dot_b) / ba_l2;
                dist = std::sqrt(q > 0.f ? q : 0.f);  // btMax(q, 0)
            }
            return true;
        }
        return false;
    }
    int newface(int a, int b, int c, bool forced) {
        if (stock.root >= 0) {
            const int fi = stock.root;
            remove(stock, fi);
            append(hull, fi);
            Face &face = fc_store[fi];
            face.pass = 0;
            face.c[0] = a; face.c[1] = b; face.c[2] = c;
            const SV &A = sv(a), &B = sv(b), &C = sv(c);
            face.n = cross(B.w - A.w, C.w - A.w);
            const float l = length(face.n);
            const bool v = l > EPA_ACCURACY;
            if (v) {
                if (!(getedgedist(face, A, B, face.d) || getedgedist(face, B, C, face.d) || getedgedist(face, C, A, face.d))) {
                    face.d = dot(A.w, face.n) / l;  // the origin projects into the triangle: distance to its plane
                }
                face.n = face.n / l;
                if (forced || face.d >= -EPA_PLANE_EPS) return fi;
                status = NonConvex;
            } else {
                status = Degenerated;
            }
            remove(hull, fi);
            append(stock, fi);
            return -1;
        }
        status = stock.root >= 0 ? OutOfVertices : OutOfFaces;
        return -1;
    }
    int findbest() const {
        int minf = hull.root;
        float mind = fc_store[minf].d * fc_store[minf].d;
        for (int f = fc_store[minf].l[1]; f >= 0; f = fc_store[f].l[1]) {
            const float sqd = fc_store[f].d * fc_store[f].d;
            if (sqd < mind) { minf = f; mind = sqd; }
        }
        return minf;
    }
    bool expand(unsigned pass, int w, int f, unsigned e, Horizon &horizon) {
        static const unsigned i1m3[] = {1, 2, 0};
        static const unsigned i2m3[] = {2, 0, 1};
        if (fc_store[f].pass != pass) {
            const unsigned e1 = i1m3[e];
            if ((dot(fc_store[f].n, sv(w).w) - fc_store[f].d) < -EPA_PLANE_EPS) {
                const int nf = newface(fc_store[f].c[e1], fc_store[f].c[e], w, false);
                if (nf >= 0) {
                    bind(nf, 0, f, e);
                    if (horizon.cf >= 0) bind(horizon.cf, 1, nf, 2);
                    else horizon.ff = nf;
                    horizon.cf = nf;
                    ++horizon.nf;
                    return true;
                }
            } else {
                const unsigned e2 = i2m3[e];
                fc_store[f].pass = (unsigned char) pass;
                if (expand(pass, w, fc_store[f].f[e1], fc_store[f].e[e1], horizon) &&
                    expand(pass, w, fc_store[f].f[e2], fc_store[f].e[e2], horizon)) {
                    remove(hull, f);
                    append(stock, f);
                    return true;
                }
            }
        }
        return false;
    }

    Status evaluate(const V3 &guess) {
        GJK2::Simplex &simplex = *gjk.simplex;
        if (simplex.rank > 1 && gjk.enclose_origin()) {
            while (hull.root >= 0) {
                const int f = hull.root;
                remove(hull, f);
                append(stock, f);
            }
            status = Valid;
            nextsv = 0;
            // orient the simplex
            if (det3(gjk.W(simplex, 0) - gjk.W(simplex, 3), gjk.W(simplex, 1) - gjk.W(simplex, 3), gjk.W(simplex, 2) - gjk.W(simplex, 3)) < 0) {
                const int tc = simplex.c[0]; simplex.c[0] = simplex.c[1]; simplex.c[1] = tc;
                const float tp = simplex.p[0]; simplex.p[0] = simplex.p[1]; simplex.p[1] = tp;
            }
            const int g0 = -1 - simplex.c[0], g1 = -1 - simplex.c[1], g2 = -1 - simplex.c[2], g3 = -1 - simplex.c[3];
            const int tetra[] = {newface(g0, g1, g2, true), newface(g1, g0, g3, true), newface(g2, g1, g3, true), newface(g0, g2, g3, true)};
            if (hull.count == 4) {
                int best = findbest();
                Face outer = fc_store[best];
                unsigned pass = 0;
                unsigned iterations = 0;
                bind(tetra[0], 0, tetra[1], 0);
                bind(tetra[0], 1, tetra[2], 0);
                bind(tetra[0], 2, tetra[3], 0);
                bind(tetra[1], 1, tetra[3], 2);
                bind(tetra[1], 2, tetra[2], 1);
                bind(tetra[2], 2, tetra[3], 1);
                status = Valid;
                for (; iterations < (unsigned) EPA_MAX_ITERATIONS; ++iterations) {
                    if (nextsv < (unsigned) EPA_MAX_VERTICES) {
                        Horizon horizon;
                        const int w = (int) nextsv++;
                        bool valid = true;
                        fc_store[best].pass = (unsigned char) (++pass);
                        gjk.getsupport(fc_store[best].n, sv_store[w]);
                        const float wdist = dot(fc_store[best].n, sv_store[w].w) - fc_store[best].d;
                        if (wdist > EPA_ACCURACY) {
                            for (unsigned j = 0; j < 3 && valid; ++j) {
                                valid &= expand(pass, w, fc_store[best].f[j], fc_store[best].e[j], horizon);
                            }
                            if (valid && horizon.nf >= 3) {
                                bind(horizon.cf, 1, horizon.ff, 2);
                                remove(hull, best);
                                append(stock, best);
                                best = findbest();
                                outer = fc_store[best];
                            } else {
                                status = InvalidHull;
                                break;
                            }
                        } else {
                            status = AccuraryReached;
                            break;
                        }
                    } else {
                        status = OutOfVertices;
                        break;
                    }
                }
                iterations_done = (int) iterations;
                const V3 projection = outer.n * outer.d;
                normal = outer.n;
                depth = outer.d;
                result.rank = 3;
                result.c[0] = outer.c[0];
                result.c[1] = outer.c[1];
                result.c[2] = outer.c[2];
                result.p[0] = length(cross(sv(outer.c[1]).w - projection, sv(outer.c[2]).w - projection));
                result.p[1] = length(cross(sv(outer.c[2]).w - projection, sv(outer.c[0]).w - projection));
                result.p[2] = length(cross(sv(outer.c[0]).w - projection, sv(outer.c[1]).w - projection));
                const float sum = result.p[0] + result.p[1] + result.p[2];
                result.p[0] /= sum;
                result.p[1] /= sum;
                result.p[2] /= sum;
                return status;
            }
        }
        // fallback
        status = FallBack;
        normal = -guess;
        const float nl = length(normal);
        if (nl > 0) normal = normal / nl;
        else normal = V3(1, 0, 0);
        depth = 0;
        result.rank = 1;
        result.c[0] = -1 - simplex.c[0];
        result.p[0] = 1;
        return status;
    }
};

}  // namespace

// btGjkEpaSolver2::Penetration (usemargins = true)
bool epa_penetration(const ConvexView &A, const Xf &wtrs0, const ConvexView &B, const Xf &wtrs1, const V3 &guess, EpaResults &results) {
    MinkowskiDiff shape;
    initialize(A, wtrs0, B, wtrs1, results, shape, true);
    GJK2 gjk;
    const GJK2::Status gjk_status = gjk.evaluate(shape, -guess);
    results.gjk_iterations = gjk.iterations_done;
    switch (gjk_status) {
        case GJK2::Inside: {
            EPA epa(gjk);
            const EPA::Status epa_status = epa.evaluate(-guess);
            results.epa_status = (int) epa_status;
            results.epa_iterations = epa.iterations_done;
            results.epa_vertices = (int) epa.nextsv;
            if (epa_status != EPA::Failed) {
                V3 w0(0, 0, 0);
                for (int i = 0; i < epa.result.rank; ++i) w0 += shape.support(epa.sv(epa.result.c[i]).d, 0) * epa.result.p[i];
                results.status = EpaResults::Penetrating;
                results.witnesses[0] = wtrs0(w0);
                results.witnesses[1] = wtrs0(w0 - epa.normal * epa.depth);
                results.normal = -epa.normal;
                results.distance = -epa.depth;
                return true;
            } else {
                results.status = EpaResults::EPA_Failed;
            }
            break;
        }
        case GJK2::Failed: results.status = EpaResults::GJK_Failed; break;
        default: break;
    }
    return false;
}

// btGjkEpaSolver2::Distance (no margins)
bool epa_distance(const ConvexView &A, const Xf &wtrs0, const ConvexView &B, const Xf &wtrs1, const V3 &guess, EpaResults &results) {
    MinkowskiDiff shape;
    initialize(A, wtrs0, B, wtrs1, results, shape, false);
    GJK2 gjk;
    const GJK2::Status gjk_status = gjk.evaluate(shape, guess);
    if (gjk_status == GJK2::Valid) {
        V3 w0(0, 0, 0), w1(0, 0, 0);
        for (int i = 0; i < gjk.simplex->rank; ++i) {
            const float p = gjk.simplex->p[i];
            w0 += shape.support(gjk.store[gjk.simplex->c[i]].d, 0) * p;
            w1 += shape.support(-gjk.store[gjk.simplex->c[i]].d, 1) * p;
        }
        results.witnesses[0] = wtrs0(w0);
        results.witnesses[1] = wtrs0(w1);
        results.normal = w0 - w1;
        results.distance = length(results.normal);
        results.normal = results.normal / (results.distance > GJK2_MIN_DISTANCE ? results.distance : 1.f);
        return true;
    }
    results.status = gjk_status == GJK2::Inside ? EpaResults::Penetrating : EpaResults::GJK_Failed;
    return false;
}

// btVector3::safeNormalize
static V3 safe_normalize(V3 v) {
    const float l2 = length2(v);
    if (l2 >= SIMD_EPSILON * SIMD_EPSILON) return v / std::sqrt(l2);
    return V3(1, 0, 0);
}

bool epa_calc_pen_depth(const ConvexView &A, const ConvexView &B, const Xf &transA, const Xf &transB, V3 &v, V3 &witnessA, V3 &witnessB,
                        EpaResults *diag) {
    const V3 guessVectors[] = {
        safe_normalize(transB.o - transA.o), safe_normalize(transA.o - transB.o),
        V3(0, 0, 1), V3(0, 1, 0), V3(1, 0, 0), V3(1, 1, 0), V3(1, 1, 1), V3(0, 1, 1), V3(1, 0, 1),
    };
    for (const V3 &guessVector : guessVectors) {
        EpaResults results;
        if (epa_penetration(A, transA, B, transB, guessVector, results)) {
            witnessA = results.witnesses[0];
            witnessB = results.witnesses[1];
            v = results.normal;
            if (diag) *diag = results;
            return true;
        } else if (epa_distance(A, transA, B, transB, guessVector, results)) {
            witnessA = results.witnesses[0];
            witnessB = results.witnesses[1];
            v = results.normal;
            if (diag) *diag = results;
            return false;
        }
    }
    return false;
}

}  // namespace orc
