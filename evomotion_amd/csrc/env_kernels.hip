// robot_walk dynamics for MI355X (gfx950): one environment per lane, one wavefront per workgroup.
//
// What this file replaces (reference, paths relative to its repo root):
//   Environment::do_step / step_world / reset      evo_motion_model/src/environment.cpp:33-48
//   RobotWalk::compute_step / reset_engine         evo_motion_model/src/env/robot_walk.cpp:56-104
//   MuscleController::on_input, Muscle::contract   src/controller/muscle_controller.cpp:10-12, src/robot/muscle.cpp:82-85
//   proprioception states                          src/robot/proprioception_state.cpp:23-129
//   and Bullet3's stepSimulation (sequential-impulse solver) that those call into.
//
// Layout:
//   HBM   struct-of-arrays state, field-major [field][env]: lane i of a wave touches env 64*w+i, so every
//         load/store instruction is one coalesced 256-byte line per field.
//   LDS   per-body solver velocity deltas (6 floats) and world inverse-inertia tiles (6 floats),
//         [slot][lane] (bank = lane, conflict-free).  41 bodies -> 492 slots x 256 B = 123 KiB per wave.
//   SGPR  the skeleton (frames, pivots, masses, hull points) is wave-uniform and sits in __constant__ memory.
//   PGS   the Gauss-Seidel sweep keeps Bullet's row order; rows are rebuilt from compact per-constraint
//         geometry (axes + lever arms) each sweep instead of streaming ~20 floats/row, so one sweep reads
//         ~1.4 K floats/env of scratch instead of ~7.5 K.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dev_math.h"
#include "env_dev.h"
#include "skel_const.h"

namespace evm {

__constant__ EvmSkelC c_skel;

#define DT_F (1.f / 60.f)
#define FPS_F (1.f / DT_F)
#define MARGIN_F 0.04f
#define ERP_F 0.2f
#define WARM_F 0.85f
#define SPLIT_THR_F (-0.04f)
#define SPLIT_TURN_ERP_F 0.1f
#ifndef NUM_ITER
#define NUM_ITER 10
#endif

struct Ctx {
    EnvDev d;  // array bases
    EnvDev t;  // the same arrays advanced to this wave's 64-env tile: element (slot k, lane) = t.arr[k * 64 + lane]
    int env, lane;
    int wave;  // 0..EVM_NW-1 inside the step kernel (wave-uniform), 0 elsewhere
    float *lds;
};

// HBM layout: every array is [tile][slot][64 lanes].  Inside a kernel the tile base is wave-uniform (SGPR pair),
// the lane offset is one VGPR shared by every access, and the slot offset is an immediate or one scalar add:
// no per-access 64-bit address arithmetic.
// the member-vs-member arrays of a tile (null pointers stay null: floor-contacts-only mode)
DEV void ctx_pair_arrays(Ctx &c, const EnvDev &d, size_t tile) {
    const int np = c_skel.npair, nwords = ((np + 31) >> 5) + 1;
    c.t.pmn = d.pmn ? d.pmn + tile * np : nullptr;
    c.t.pmp = d.pmp ? d.pmp + tile * (EVM_PM_STRIDE * np) : nullptr;
    c.t.pact = d.pact ? d.pact + tile * nwords : nullptr;
    c.t.crec = d.crec ? d.crec + tile * (EVM_CR_STRIDE * (c_skel.nm + np)) : nullptr;
}
DEV Ctx make_ctx(const EnvDev &d, float *lds) {
    Ctx c;
    c.d = d;
    c.t = d;
    c.lane = threadIdx.x & 63;
    c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.env = blockIdx.x * 64 + c.lane;
    c.lds = lds;
    const size_t tile = (size_t) blockIdx.x * 64;
    const int nb = c_skel.nb, nm = c_skel.nm, nmus = c_skel.nmus > 0 ? c_skel.nmus : 1;
    c.t.pos = d.pos + tile * (3 * nb); c.t.quat = d.quat + tile * (4 * nb);
    c.t.lin = d.lin + tile * (3 * nb); c.t.ang = d.ang + tile * (3 * nb);
    c.t.hist = d.hist + tile * (6 * nm); c.t.mfn = d.mfn + tile * nm; c.t.mfp = d.mfp + tile * (36 * nm);
    c.t.target = d.target + tile * nmus; c.t.E = d.E + tile * 9; c.t.iinv_stale = d.iinv_stale + tile * (6 * nb);
    c.t.mt = d.mt + tile * 624; c.t.scratch = d.scratch + tile * c_skel.sc_total;
    c.t.diag = d.diag + tile * 2; c.t.stat = d.stat + tile * 2;
    ctx_pair_arrays(c, d, tile);
    return c;
}

// the same view for an ARBITRARY env per lane (the pair kernel's compacted work lists): the tile bases become per-lane
// pointers, everything built on GS / SC / LII keeps working
DEV Ctx make_ctx_env(const EnvDev &d, int env) {
    Ctx c;
    c.d = d;
    c.t = d;
    c.lane = env & 63;
    c.wave = 0;
    c.env = env;
    const size_t tile = (size_t) (env >> 6) * 64;
    c.lds = d.gtile + (size_t) (env >> 6) * d.tile_floats;
    const int nb = c_skel.nb, nm = c_skel.nm, nmus = c_skel.nmus > 0 ? c_skel.nmus : 1;
    c.t.pos = d.pos + tile * (3 * nb); c.t.quat = d.quat + tile * (4 * nb);
    c.t.lin = d.lin + tile * (3 * nb); c.t.ang = d.ang + tile * (3 * nb);
    c.t.hist = d.hist + tile * (6 * nm); c.t.mfn = d.mfn + tile * nm; c.t.mfp = d.mfp + tile * (36 * nm);
    c.t.target = d.target + tile * nmus; c.t.E = d.E + tile * 9; c.t.iinv_stale = d.iinv_stale + tile * (6 * nb);
    c.t.mt = d.mt + tile * 624; c.t.scratch = d.scratch + tile * c_skel.sc_total;
    c.t.diag = d.diag + tile * 2; c.t.stat = d.stat + tile * 2;
    ctx_pair_arrays(c, d, tile);
    return c;
}

#define GS(arr, k) (c.t.arr[((k) << 6) + c.lane])
#define SC(k) (c.t.scratch[((k) << 6) + c.lane])
// field f of the quad-packed record that starts at slot `base` (a multiple of 4), see skel_const.h
#define RC(base, f) (c.t.scratch[(((base) + ((f) & ~3)) << 6) + (c.lane << 2) + ((f) & 3)])
#define LDV(b, k) (c.lds[(((b) * 6 + (k)) << 6) + c.lane])
#define LII(b, k) (c.lds[(((c_skel.nb + (b)) * 6 + (k)) << 6) + c.lane])
// per-body version counters (one int per body) behind the float tiles: dataflow synchronisation of the sweep
#define LVER(c_) (reinterpret_cast<int *>((c_).lds + (size_t) c_skel.nb * 12 * 64))
// partial results of the hull scans: slice i -> (minimum world y, vertex index as int bits)
#define LPART(i, k) (c.lds[(size_t) c_skel.nb * 12 * 64 + (size_t) ((c_skel.nb + 63) / 64) * 64 + ((((i) << 1) + (k)) << 6) + c.lane])

// Wait until body b has been written `expect` times (monotonic counter, workgroup-scope acquire).  Bounded:
// a schedule bug must not hang the GPU; on timeout the diagnostic slot is poisoned and the wave goes on.
#ifndef EVM_SPIN_HOT
#define EVM_SPIN_HOT 4
#endif
DEV void wait_version(const Ctx &c, int b, int expect) {
    int *ver = LVER(c);
    int spins = 0;
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&ver[b], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < expect) {
        if (++spins > EVM_SPIN_HOT) __builtin_amdgcn_s_sleep(1);
        if (spins > (1 << 22)) { c.t.diag[c.lane] = -1.f; break; }
    }
}
// both bodies of a joint visit with ONE LDS round trip: two relaxed polls in flight together, one acquire fence
// once both versions have been reached (lane 0 is always a live env; other lanes may have exited on a ragged tile)
DEV void wait_versions2(const Ctx &c, int a, int expA, int b, int expB) {
    int *ver = LVER(c);
    int spins = 0;
    for (;;) {
        const int va = __hip_atomic_load(&ver[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int vb = __hip_atomic_load(&ver[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (__builtin_amdgcn_readfirstlane(va) >= expA && __builtin_amdgcn_readfirstlane(vb) >= expB) break;
        if (++spins > EVM_SPIN_HOT) __builtin_amdgcn_s_sleep(1);  // poll back to back first: a hop is usually imminent
        if (spins > (1 << 22)) { c.t.diag[c.lane] = -1.f; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
DEV void publish_version(const Ctx &c, int b, int value) {
    __hip_atomic_store(&LVER(c)[b], value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

DEV F3 gs3(const float *tile, int lane, int k) {
    const float *p = tile + (k << 6) + lane;
    return f3(p[0], p[64], p[128]);
}
DEV void ss3(float *tile, int lane, int k, F3 v) {
    float *p = tile + (k << 6) + lane;
    p[0] = v.x; p[64] = v.y; p[128] = v.z;
}
#define G3(arr, k) gs3(c.t.arr, c.lane, (k))
#define S3(arr, k, v) ss3(c.t.arr, c.lane, (k), (v))
#define SC3(k) gs3(c.t.scratch, c.lane, (k))
#define SSC3(k, v) ss3(c.t.scratch, c.lane, (k), (v))

// ---------------------------------------------------------------------------------------------
// body views
// ---------------------------------------------------------------------------------------------
struct BodyK {  // kinematic snapshot used while building rows
    F3 o;
    M33 R;
    F3 v, w;     // lin + externalForceImpulse, ang + externalTorqueImpulse
    F3 w_raw;    // ang
    S33 I;
    float im;
};
struct BodyD {  // what a Gauss-Seidel row needs
    F3 dl, da;
    S33 I;
    float im;
};

DEV S33 lds_inertia(const Ctx &c, int b) {
    S33 s;
    s.xx = LII(b, 0); s.xy = LII(b, 1); s.xz = LII(b, 2); s.yy = LII(b, 3); s.yz = LII(b, 4); s.zz = LII(b, 5);
    return s;
}
DEV BodyK load_bodyk(const Ctx &c, int b) {
    BodyK k;
    k.o = G3(pos, 3 * b);
    k.R = m33(SC3(c_skel.sc_r + 9 * b), SC3(c_skel.sc_r + 9 * b + 3), SC3(c_skel.sc_r + 9 * b + 6));
    const F3 lin = G3(lin, 3 * b);
    k.w_raw = G3(ang, 3 * b);
    const F3 ext = SC3(c_skel.sc_ext + 3 * b);
    k.v = lin + f3(0.f, c_skel.body[b].ext_force_y, 0.f);
    k.w = k.w_raw + ext;
    k.I = lds_inertia(c, b);
    k.im = c_skel.body[b].inv_mass;
    return k;
}
DEV BodyD load_bodyd(const Ctx &c, int b, float im) {
    BodyD k;
    k.dl = f3(LDV(b, 0), LDV(b, 1), LDV(b, 2));
    k.da = f3(LDV(b, 3), LDV(b, 4), LDV(b, 5));
    k.I = lds_inertia(c, b);
    k.im = im;
    return k;
}
DEV BodyD load_bodyd(const Ctx &c, int b) { return load_bodyd(c, b, c_skel.body[b].inv_mass); }
DEV void store_bodyd(const Ctx &c, int b, const BodyD &k) {
    LDV(b, 0) = k.dl.x; LDV(b, 1) = k.dl.y; LDV(b, 2) = k.dl.z;
    LDV(b, 3) = k.da.x; LDV(b, 4) = k.da.y; LDV(b, 5) = k.da.z;
}

// ---------------------------------------------------------------------------------------------
// generic two-body rows: linear (n, relA x n, -(relB x n)) or angular (0, a, -a)
// ---------------------------------------------------------------------------------------------
template <bool LIN>
DEV void row_setup(F3 ax, F3 relA, F3 relB, const BodyK &A, const BodyK &B, float err, float &jd, float &rhs) {
    F3 c1, c2;
    float sum, rel_vel;
    if (LIN) {
        c1 = cross(relA, ax);
        c2 = -cross(relB, ax);
        sum = dot(ax * A.im, ax);
        sum += dot(mul(A.I, c1), c1);
        sum += dot(ax * B.im, ax);
        sum += dot(mul(B.I, c2), c2);
        rel_vel = (dot(ax, A.v) + dot(c1, A.w)) + (-dot(ax, B.v) + dot(c2, B.w));
    } else {
        c1 = ax;
        c2 = -ax;
        sum = dot(mul(A.I, c1), c1);
        sum += dot(mul(B.I, c2), c2);
        rel_vel = dot(c1, A.w) + dot(c2, B.w);
    }
    jd = fabsf(sum) > EVM_EPS ? 1.0f / sum : 0.f;
    rhs = err * jd + (0.f - rel_vel) * jd;
}

template <bool LIN, bool BOUNDED>
DEV float row_iter(F3 ax, F3 relA, F3 relB, BodyD &A, BodyD &B, float jd, float rhs, float lo, float hi,
                   float &applied) {
    F3 c1, c2;
    float d1, d2;
    if (LIN) {
        c1 = cross(relA, ax);
        c2 = -cross(relB, ax);
        d1 = dot(ax, A.dl) + dot(c1, A.da);
        d2 = -dot(ax, B.dl) + dot(c2, B.da);
    } else {
        c1 = ax;
        c2 = -ax;
        d1 = dot(c1, A.da);
        d2 = dot(c2, B.da);
    }
    const F3 angA = mul(A.I, c1), angB = mul(B.I, c2);
    float dI = rhs;
    dI -= d1 * jd;
    dI -= d2 * jd;
    const float sum = applied + dI;
    if (BOUNDED) {  // (selects, not branches: sum < lo -> lo, else sum > hi -> hi, else sum)
        const bool cl = sum < lo, ch = !cl && sum > hi;
        const float nap = cl ? lo : (ch ? hi : sum);
        dI = (cl || ch) ? nap - applied : dI;
        applied = nap;
    } else applied = sum;
    if (LIN) {
        A.dl = A.dl + ax * (A.im * dI);
        B.dl = B.dl - ax * (B.im * dI);
    }
    A.da = A.da + angA * dI;
    B.da = B.da + angB * dI;
    return dI;
}

// Both bodies of a joint visit as (A, B) pairs: one packed instruction serves both sides of a row.
struct BodyPD {
    F3P dl, da;
    S33P I;
    P2 im;
};
// world inverse inertia of both bodies: constant during the sweeps, so a visit reads it BEFORE it waits for its turn
DEV S33P load_inertia_pair(const Ctx &c, int a, int b) {
    S33P I;
    I.xx = p2(LII(a, 0), LII(b, 0)); I.xy = p2(LII(a, 1), LII(b, 1)); I.xz = p2(LII(a, 2), LII(b, 2));
    I.yy = p2(LII(a, 3), LII(b, 3)); I.yz = p2(LII(a, 4), LII(b, 4)); I.zz = p2(LII(a, 5), LII(b, 5));
    return I;
}
DEV BodyPD load_bodypd(const Ctx &c, int a, int b, float imA, float imB, const S33P &I) {
    BodyPD k;
    k.dl = f3p(p2(LDV(a, 0), LDV(b, 0)), p2(LDV(a, 1), LDV(b, 1)), p2(LDV(a, 2), LDV(b, 2)));
    k.da = f3p(p2(LDV(a, 3), LDV(b, 3)), p2(LDV(a, 4), LDV(b, 4)), p2(LDV(a, 5), LDV(b, 5)));
    k.I = I;
    k.im = p2(imA, -imB);  // imS of row_iter
    return k;
}
DEV void store_bodypd(const Ctx &c, int a, int b, const BodyPD &k) {
    LDV(a, 0) = k.dl.x.x; LDV(a, 1) = k.dl.y.x; LDV(a, 2) = k.dl.z.x;
    LDV(a, 3) = k.da.x.x; LDV(a, 4) = k.da.y.x; LDV(a, 5) = k.da.z.x;
    LDV(b, 0) = k.dl.x.y; LDV(b, 1) = k.dl.y.y; LDV(b, 2) = k.dl.z.y;
    LDV(b, 3) = k.da.x.y; LDV(b, 4) = k.da.y.y; LDV(b, 5) = k.da.z.y;
}
// One Gauss-Seidel row on the (A, B) pair.  The B side of a row is the A side with the axis negated; the sign is folded
// into per-visit constants instead of per-row negations: relS = (relA, -relB) (stored that way in the record) and
// imS = (1/mA, -1/mB).  With A = (ax, ax): c = relS x A = (relA x ax, -(relB x ax)); t = A . dl gives the two linear
// projections, the B one entering with a minus sign (a free neg_hi modifier); the updates use A * (imS * dI) and, for
// angular rows, w * (dI, -dI) with w = I A.  Every value equals the two scalar sides' bit for bit.
// ISO: both bodies have an isotropic inverse inertia k * identity (attach spheres), kept in Q.I.xx
template <bool LIN, bool BOUNDED, bool ISO = false>
DEV float row_iter(F3 ax, const F3P &relS, BodyPD &Q, float jd, float rhs, float lo, float hi, float &applied) {
    const F3P A = f3p(p2(ax.x, ax.x), p2(ax.y, ax.y), p2(ax.z, ax.z));
    F3P cc;
    P2 d;
    if (LIN) {
        cc = cross(relS, A);
        const P2 t = dot(A, Q.dl), u = dot(cc, Q.da);
        d = p2(t.x, -t.y) + u;
    } else {
        const P2 t = dot(A, Q.da);
        d = p2(t.x, -t.y);
    }
    float dI = rhs;
    dI -= d.x * jd;
    dI -= d.y * jd;
    const float sum = applied + dI;
    if (BOUNDED) {  // (selects, not branches: sum < lo -> lo, else sum > hi -> hi, else sum)
        const bool cl = sum < lo, ch = !cl && sum > hi;
        const float nap = cl ? lo : (ch ? hi : sum);
        dI = (cl || ch) ? nap - applied : dI;
        applied = nap;
    } else applied = sum;
    if (LIN) {
        const F3P ang = ISO ? cc * Q.I.xx : mul(Q.I, cc);
        Q.dl = Q.dl + A * (Q.im * dI);
        Q.da = Q.da + ang * dI;
    } else {
        const F3P w = ISO ? A * Q.I.xx : mul(Q.I, A);
        Q.da = Q.da + w * p2(dI, -dI);
    }
    return dI;
}

// register block holding one constraint's scratch record (loaded one constraint ahead of its use)
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct Blk42 {
    f32x4 q[EVM_CM_STRIDE / 4];  // the largest record (a member's contact rows); kept as quads so that the
                                 // conditional tail loads merge as whole register tuples (no copies, no early wait)
};
#define KV(k, f) ((k).q[(f) >> 2][(f) & 3])
#define KV3(k, f) f3(KV(k, f), KV(k, (f) + 1), KV(k, (f) + 2))
DEV const f32x4 *rec_quads(const Ctx &c, int slot) {
    return reinterpret_cast<const f32x4 *>(c.t.scratch + ((size_t) slot << 6)) + c.lane;
}
template <int Q0, int Q1>
DEV void blk_load_quads(const f32x4 *p, Blk42 &b) {
#pragma unroll
    for (int q = Q0; q < Q1; q++) b.q[q] = p[q << 6];
}
DEV void blk_load(const Ctx &c, const EvmEntryC &v, Blk42 &b, int nslots) {
    const f32x4 *p = rec_quads(c, v.slot);
    blk_load_quads<0, 4>(p, b);
    if (nslots > 16) {  // wave-uniform: p2p records are 4 quads, hinges 9, fixed / slider 11, contacts 12
        blk_load_quads<4, 9>(p, b);
        if (nslots > 36) {
            blk_load_quads<9, 11>(p, b);
            if (nslots > 44) blk_load_quads<11, 12>(p, b);
        }
    }
}
// store quads [Q0, Q1) of a record from a register image
template <int Q0, int Q1>
DEV void rec_store(const Ctx &c, int slot, const float *r) {
    f32x4 *p = reinterpret_cast<f32x4 *>(c.t.scratch + ((size_t) slot << 6)) + c.lane;
#pragma unroll
    for (int q = Q0; q < Q1; q++) {
        f32x4 x;
        x[0] = r[4 * q]; x[1] = r[4 * q + 1]; x[2] = r[4 * q + 2]; x[3] = r[4 * q + 3];
        p[q << 6] = x;
    }
}
DEV void put3(float *r, int i, F3 v) { r[i] = v.x; r[i + 1] = v.y; r[i + 2] = v.z; }
DEV F3 v3(const float *v, int i) { return f3(v[i], v[i + 1], v[i + 2]); }

// ---------------------------------------------------------------------------------------------
// hinge  (btHingeConstraint::getInfo2InternalUsingFrameOffset)
// scratch: relA3 relB3 p3 q3 ax3 | jd6 | rhs6 | lo hi | applied6
// ---------------------------------------------------------------------------------------------
DEV void hinge_setup(const Ctx &c, int hi) {
    const EvmHingeC &H = c_skel.hinge[hi];
    const BodyK A = load_bodyk(c, H.a), B = load_bodyk(c, H.b);
    const M33 fa = load_m33(H.fa), fb = load_m33(H.fb);
    const F3 A0 = mul(A.R, col0(fa)), A1 = mul(A.R, col1(fa)), A2 = mul(A.R, col2(fa));
    const F3 B1 = mul(B.R, col1(fb)), B2 = mul(B.R, col2(fb));
    const F3 trAo = mul(A.R, load_f3(H.fao)) + A.o;
    const F3 trBo = mul(B.R, load_f3(H.fbo)) + B.o;
    // testLimit (btAngularLimit::test)
    const float angle = atan2f(dot(B1, A0), dot(B1, A1));
    float correction = 0.f;
    bool solve_limit = false;
    if (H.half_range >= 0.f) {
        const float dev = norm_angle(angle - H.center);
        if (dev < -H.half_range) { solve_limit = true; correction = -(dev + H.half_range); }
        else if (dev > H.half_range) { solve_limit = true; correction = H.half_range - dev; }
    }
    const F3 ofs = trBo - trAo;
    float factA = H.factA, factB = H.factB;
    F3 ax1 = A2 * factA + B2 * factB;
    if (len2(ax1) < EVM_EPS) { factA = 0.f; factB = 1.f; ax1 = A2 * factA + B2 * factB; }
    ax1 = normalize(ax1);
    F3 relB = trBo - B.o;
    const F3 projB = ax1 * dot(relB, ax1);
    const F3 orthoB = relB - projB;
    F3 relA = trAo - A.o;
    const F3 projA = ax1 * dot(relA, ax1);
    const F3 orthoA = relA - projA;
    const F3 totalDist = projA - projB;
    relA = orthoA + totalDist * factA;
    relB = orthoB - totalDist * factB;
    F3 p = orthoB * factA + orthoA * factB;
    const float l2 = len2(p);
    if (l2 > EVM_EPS) p = p * (1.0f / sqrtf(l2)); else p = A1;
    const F3 q = cross(ax1, p);
    const float k = FPS_F * ERP_F;
    const F3 u = cross(A2, B2);
    float jd[6], rhs[6];
    row_setup<true>(p, relA, relB, A, B, k * dot(p, ofs), jd[0], rhs[0]);
    row_setup<true>(q, relA, relB, A, B, k * dot(q, ofs), jd[1], rhs[1]);
    row_setup<true>(ax1, relA, relB, A, B, k * dot(ax1, ofs), jd[2], rhs[2]);
    row_setup<false>(p, relA, relB, A, B, k * dot(u, p), jd[3], rhs[3]);
    row_setup<false>(q, relA, relB, A, B, k * dot(u, q), jd[4], rhs[4]);
    float lo = 0.f, hi_ = 0.f;
    jd[5] = 0.f; rhs[5] = 0.f;
    if (solve_limit) {
        const float limit_err = correction;
        const bool low = limit_err > 0.f;
        float err = 0.f;
        err += k * limit_err;
        if (low) { lo = 0.f; hi_ = EVM_INF; } else { lo = -EVM_INF; hi_ = 0.f; }
        const float bounce = H.relaxation;
        if (bounce > 0.f) {
            float vel = dot(A.w_raw, ax1);
            vel -= dot(B.w_raw, ax1);
            if (low) { if (vel < 0.f) { const float nc = -bounce * vel; if (nc > err) err = nc; } }
            else { if (vel > 0.f) { const float nc = -bounce * vel; if (nc < err) err = nc; } }
        }
        err *= H.bias;
        row_setup<false>(ax1, relA, relB, A, B, err, jd[5], rhs[5]);
    }
    float rec[EVM_H_STRIDE];
    rec[0] = relA.x; rec[1] = -relB.x; rec[2] = relA.y; rec[3] = -relB.y; rec[4] = relA.z; rec[5] = -relB.z;  // relS pairs
    put3(rec, 6, p); put3(rec, 9, q); put3(rec, 12, ax1);
#pragma unroll
    for (int r = 0; r < 6; r++) { rec[15 + r] = jd[r]; rec[21 + r] = rhs[r]; rec[28 + r] = 0.f; }
    rec[27] = 0.f; rec[34] = lo; rec[35] = hi_;
    rec_store<0, EVM_H_STRIDE / 4>(c, c_skel.sc_h + EVM_H_STRIDE * hi, rec);
}
// the rows of one hinge visit on registers: k = the record, Q = the two bodies, ap = accumulated impulses (in: the record's)
DEV float hinge_rows(const Blk42 &k, BodyPD &Q, float (&ap)[6]) {
    const F3 p = KV3(k, 6), q = KV3(k, 9), ax1 = KV3(k, 12);
#pragma unroll
    for (int r = 0; r < 6; r++) ap[r] = KV(k, 28 + r);
    const float lo = KV(k, 34), hi_ = KV(k, 35);
    const F3P rel = f3p(p2(KV(k, 0), KV(k, 1)), p2(KV(k, 2), KV(k, 3)), p2(KV(k, 4), KV(k, 5)));  // (relA, -relB) pairs
    float res = 0.f;
    res = fmaxf(res, fabsf(row_iter<true, false>(p, rel, Q, KV(k, 15), KV(k, 21), 0.f, 0.f, ap[0])));
    res = fmaxf(res, fabsf(row_iter<true, false>(q, rel, Q, KV(k, 16), KV(k, 22), 0.f, 0.f, ap[1])));
    res = fmaxf(res, fabsf(row_iter<true, false>(ax1, rel, Q, KV(k, 17), KV(k, 23), 0.f, 0.f, ap[2])));
    res = fmaxf(res, fabsf(row_iter<false, false>(p, rel, Q, KV(k, 18), KV(k, 24), 0.f, 0.f, ap[3])));
    res = fmaxf(res, fabsf(row_iter<false, false>(q, rel, Q, KV(k, 19), KV(k, 25), 0.f, 0.f, ap[4])));
    if (KV(k, 20) != 0.f) res = fmaxf(res, fabsf(row_iter<false, true>(ax1, rel, Q, KV(k, 20), KV(k, 26), lo, hi_, ap[5])));
    return res;
}
DEV float hinge_solve(const Ctx &c, const EvmEntryC &V, const Blk42 &k, BodyPD &Q) {
    const int s = V.slot;
    float ap[6];
    const float res = hinge_rows(k, Q, ap);
    store_bodypd(c, V.a, V.b, Q);
    {
        float w[EVM_H_STRIDE];
#pragma unroll
        for (int r = 0; r < 6; r++) w[28 + r] = ap[r];
        w[34] = KV(k, 34); w[35] = KV(k, 35);
        rec_store<7, 9>(c, s, w);
    }
    return res;
}

// ---------------------------------------------------------------------------------------------
// fixed = btGeneric6DofSpring2Constraint with every axis locked (RO_XYZ): 3 angular rows then 3 linear rows
// scratch: relA3 relB3 angax9 linax9 | jd6 | rhs6 | applied6
// ---------------------------------------------------------------------------------------------
DEV M33 inverse33(const M33 &m) {
    const float co0 = m.r1.y * m.r2.z - m.r1.z * m.r2.y;
    const float co1 = m.r1.z * m.r2.x - m.r1.x * m.r2.z;
    const float co2 = m.r1.x * m.r2.y - m.r1.y * m.r2.x;
    const float det = m.r0.x * co0 + m.r0.y * co1 + m.r0.z * co2;
    const float s = 1.0f / det;
    return m33(f3(co0 * s, (m.r0.z * m.r2.y - m.r0.y * m.r2.z) * s, (m.r0.y * m.r1.z - m.r0.z * m.r1.y) * s),
               f3(co1 * s, (m.r0.x * m.r2.z - m.r0.z * m.r2.x) * s, (m.r0.z * m.r1.x - m.r0.x * m.r1.z) * s),
               f3(co2 * s, (m.r0.y * m.r2.x - m.r0.x * m.r2.y) * s, (m.r0.x * m.r1.y - m.r0.y * m.r1.x) * s));
}
DEV void fixed_setup(const Ctx &c, int fi) {
    const EvmFixedC &X = c_skel.fixed[fi];
    const BodyK A = load_bodyk(c, X.a), B = load_bodyk(c, X.b);
    const M33 cA = mul(A.R, load_m33(X.fa)), cB = mul(B.R, load_m33(X.fb));
    const F3 cAo = mul(A.R, load_f3(X.fao)) + A.o;
    const F3 cBo = mul(B.R, load_f3(X.fbo)) + B.o;
    const M33 invA = inverse33(cA);
    const F3 linDiff = mul(invA, cBo - cAo);
    const M33 rel = mul(invA, cB);
    F3 ang;
    const float fi_ = rel.r2.x;
    if (fi_ < 1.0f) {
        if (fi_ > -1.0f) {
            ang.x = atan2f(-rel.r2.y, rel.r2.z);
            ang.y = asinf(rel.r2.x);
            ang.z = atan2f(-rel.r1.x, rel.r0.x);
        } else {
            ang.x = -atan2f(rel.r0.y, rel.r1.y); ang.y = -EVM_PI * 0.5f; ang.z = 0.f;
        }
    } else {
        ang.x = atan2f(rel.r0.y, rel.r1.y); ang.y = EVM_PI * 0.5f; ang.z = 0.f;
    }
    const F3 axis0 = col0(cB), axis2 = col2(cA);
    F3 a1 = cross(axis2, axis0);
    F3 a0 = cross(a1, axis2);
    F3 a2 = cross(axis0, a1);
    a0 = normalize(a0); a1 = normalize(a1); a2 = normalize(a2);
    const F3 relB = cBo - B.o, relA = cAo - A.o;
    const F3 l0 = col0(cA), l1 = col1(cA), l2 = col2(cA);
    const float k = FPS_F * ERP_F;
    float jd[6], rhs[6];
    row_setup<false>(a0, relA, relB, A, B, k * ang.x * -1.f, jd[0], rhs[0]);
    row_setup<false>(a1, relA, relB, A, B, k * ang.y * -1.f, jd[1], rhs[1]);
    row_setup<false>(a2, relA, relB, A, B, k * ang.z * -1.f, jd[2], rhs[2]);
    row_setup<true>(l0, relA, relB, A, B, k * linDiff.x * 1.f, jd[3], rhs[3]);
    row_setup<true>(l1, relA, relB, A, B, k * linDiff.y * 1.f, jd[4], rhs[4]);
    row_setup<true>(l2, relA, relB, A, B, k * linDiff.z * 1.f, jd[5], rhs[5]);
    float rec[EVM_F_STRIDE];
    rec[0] = relA.x; rec[1] = -relB.x; rec[2] = relA.y; rec[3] = -relB.y; rec[4] = relA.z; rec[5] = -relB.z;  // relS pairs
    put3(rec, 6, a0); put3(rec, 9, a1); put3(rec, 12, a2);
    put3(rec, 15, l0); put3(rec, 18, l1); put3(rec, 21, l2);
#pragma unroll
    for (int r = 0; r < 6; r++) { rec[24 + r] = jd[r]; rec[30 + r] = rhs[r]; rec[36 + r] = 0.f; }
    rec[42] = 0.f; rec[43] = 0.f;
    rec_store<0, EVM_F_STRIDE / 4>(c, c_skel.sc_f + EVM_F_STRIDE * fi, rec);
}
DEV float fixed_rows(const Blk42 &k, BodyPD &Q, float (&ap)[6]) {
#pragma unroll
    for (int r = 0; r < 6; r++) ap[r] = KV(k, 36 + r);
    const F3P rel = f3p(p2(KV(k, 0), KV(k, 1)), p2(KV(k, 2), KV(k, 3)), p2(KV(k, 4), KV(k, 5)));  // (relA, -relB) pairs
    float res = 0.f;
#pragma unroll
    for (int r = 0; r < 3; r++)
        res = fmaxf(res, fabsf(row_iter<false, false>(KV3(k, 6 + 3 * r), rel, Q, KV(k, 24 + r), KV(k, 30 + r), 0.f, 0.f, ap[r])));
#pragma unroll
    for (int r = 0; r < 3; r++)
        res = fmaxf(res, fabsf(row_iter<true, false>(KV3(k, 15 + 3 * r), rel, Q, KV(k, 27 + r), KV(k, 33 + r), 0.f, 0.f, ap[3 + r])));
    return res;
}
DEV float fixed_solve(const Ctx &c, const EvmEntryC &V, const Blk42 &k, BodyPD &Q) {
    const int s = V.slot;
    float ap[6];
    const float res = fixed_rows(k, Q, ap);
    store_bodypd(c, V.a, V.b, Q);
    {
        float w[EVM_F_STRIDE];
#pragma unroll
        for (int r = 0; r < 6; r++) w[36 + r] = ap[r];
        w[42] = 0.f; w[43] = 0.f;
        rec_store<9, 11>(c, s, w);
    }
    return res;
}

// ---------------------------------------------------------------------------------------------
// muscle slider (btSliderConstraint::getInfo2NonVirtual, identity frames, useLinearReferenceFrameA)
// scratch: p3 q3 ax3 p2_3 q2_3 relA3 relB3 | jd6 | rhs6 | lo hi | applied6
// rows: 0,1 angular(p,q)  2,3 linear(p2,q2)  4 linear(ax1) motor/limit  5 angular(ax1) limit
// ---------------------------------------------------------------------------------------------
DEV float motor_factor(float pos, float lowLim, float uppLim, float vel, float timeFact) {
    if (lowLim > uppLim) return 1.0f;
    if (lowLim == uppLim) return 0.0f;
    float lim_fact = 1.0f;
    const float delta_max = vel / timeFact;
    if (delta_max < 0.0f) {
        if ((pos >= lowLim) && (pos < (lowLim - delta_max))) lim_fact = (lowLim - pos) / delta_max;
        else if (pos < lowLim) lim_fact = 0.0f;
        else lim_fact = 1.0f;
    } else if (delta_max > 0.0f) {
        if ((pos <= uppLim) && (pos > (uppLim - delta_max))) lim_fact = (uppLim - pos) / delta_max;
        else if (pos > uppLim) lim_fact = 0.0f;
        else lim_fact = 1.0f;
    } else lim_fact = 0.0f;
    return lim_fact;
}
DEV void slider_setup(const Ctx &c, int mi, bool powered_in, float target_vel) {
    const EvmMuscleC &M = c_skel.muscle[mi];
    const BodyK A = load_bodyk(c, M.sa), B = load_bodyk(c, M.sb);
    const F3 ax1A = col0(A.R), ax1B = col0(B.R);
    const F3 delta = B.o - A.o;
    float depth0 = dot(delta, col0(A.R));
    // testAngLimits (lower == upper == 0)
    float ang_depth = 0.f;
    bool solve_ang = false;
    {
        // the limit is [0,0]: this row exists iff rot != 0, and rot hovers at rounding level (see DESIGN.md,
        // "ill-conditioned decisions"), so evaluate it without contraction
        const F3 axisA0 = col1(A.R), axisA1 = col2(A.R), axisB0 = col1(B.R);
        const float rot = atan2f(xdot(axisB0, axisA1), xdot(axisB0, axisA0));
        if (rot < 0.f) { ang_depth = rot; solve_ang = true; }
        else if (rot > 0.f) { ang_depth = rot; solve_ang = true; }
    }
    // testLinLimits (lower 0, upper = 2 * L0)
    bool solve_lin = false;
    const float lin_pos = depth0;
    const float upper = M.upper_lin;
    if (0.f <= upper) {
        if (depth0 > upper) { depth0 -= upper; solve_lin = true; }
        else if (depth0 < 0.f) { depth0 -= 0.f; solve_lin = true; }
        else depth0 = 0.f;
    } else depth0 = 0.f;
    const F3 ofs = B.o - A.o;
    const float factA = M.factA, factB = M.factB;
    F3 ax1 = normalize(ax1A * factA + ax1B * factB);
    F3 p, q;
    plane_space(ax1, p, q);
    float k = FPS_F * (1.0f * ERP_F);
    const F3 u = cross(ax1A, ax1B);
    float jd[6], rhs[6];
    F3 relB = f3(0.f, 0.f, 0.f) + (B.o - B.o);
    F3 relA = A.o - A.o;
    row_setup<false>(p, relA, relB, A, B, k * dot(u, p), jd[0], rhs[0]);
    row_setup<false>(q, relA, relB, A, B, k * dot(u, q), jd[1], rhs[1]);
    const F3 projB = ax1 * dot(relB, ax1);
    const F3 orthoB = relB - projB;
    const F3 projA = ax1 * dot(relA, ax1);
    const F3 orthoA = relA - projA;
    const float sliderOffs = lin_pos - depth0;
    const F3 totalDist = projA + ax1 * sliderOffs - projB;
    relA = orthoA + totalDist * factA;
    relB = orthoB - totalDist * factB;
    F3 p2 = orthoB * factA + orthoA * factB;
    const float l2 = len2(p2);
    if (l2 > EVM_EPS) p2 = p2 * (1.0f / sqrtf(l2)); else p2 = col1(A.R);
    const F3 q2 = cross(ax1, p2);
    k = FPS_F * (1.0f * ERP_F);
    row_setup<true>(p2, relA, relB, A, B, k * dot(p2, ofs), jd[2], rhs[2]);
    row_setup<true>(q2, relA, relB, A, B, k * dot(q2, ofs), jd[3], rhs[3]);
    // linear limit / motor row
    float lo = 0.f, hi_ = 0.f;
    jd[4] = 0.f; rhs[4] = 0.f;
    {
        float limit_err = 0.f;
        int limit = 0;
        if (solve_lin) { limit_err = depth0; limit = limit_err > 0.f ? 2 : 1; }
        bool powered = powered_in;
        if (limit || powered) {
            const float lostop = 0.f, histop = upper;
            if (limit && (lostop == histop)) powered = false;
            float err = 0.f;
            if (powered) {
                const float mot = motor_factor(lin_pos, lostop, histop, target_vel, FPS_F * ERP_F);
                err -= 1.0f * mot * target_vel;
                lo += -M.max_impulse;
                hi_ += M.max_impulse;
            }
            if (limit) {
                k = FPS_F * ERP_F;
                err += k * limit_err;
                if (lostop == histop) { lo = -EVM_INF; hi_ = EVM_INF; }
                else if (limit == 1) { lo = -EVM_INF; hi_ = 0.f; }
                else { lo = 0.f; hi_ = EVM_INF; }
                err *= 1.0f;
            }
            row_setup<true>(ax1, relA, relB, A, B, err, jd[4], rhs[4]);
        }
    }
    // angular limit row
    jd[5] = 0.f; rhs[5] = 0.f;
    if (solve_ang) {
        float err = 0.f;
        err += (FPS_F * ERP_F) * ang_depth;
        err *= 1.0f;
        row_setup<false>(ax1, relA, relB, A, B, err, jd[5], rhs[5]);
    }
    float rec[EVM_S_STRIDE];
    rec[0] = relA.x; rec[1] = -relB.x; rec[2] = relA.y; rec[3] = -relB.y; rec[4] = relA.z; rec[5] = -relB.z;  // relS pairs
    put3(rec, 6, p); put3(rec, 9, q); put3(rec, 12, ax1); put3(rec, 15, p2); put3(rec, 18, q2);
#pragma unroll
    for (int r = 0; r < 6; r++) { rec[21 + r] = jd[r]; rec[27 + r] = rhs[r]; rec[36 + r] = 0.f; }
    rec[33] = lo; rec[34] = hi_; rec[35] = 0.f; rec[42] = 0.f; rec[43] = 0.f;
    rec_store<0, EVM_S_STRIDE / 4>(c, c_skel.sc_s + EVM_S_STRIDE * mi, rec);
    SC(c_skel.sc_mobs + 4 * mi) = lin_pos;  // btSliderConstraint::getLinearPos(), MuscleState
}
template <bool ISO>
DEV float slider_rows(const Blk42 &kk, BodyPD &Q, float (&ap)[6]) {
    const F3 p = KV3(kk, 6), q = KV3(kk, 9), ax1 = KV3(kk, 12), p2_ = KV3(kk, 15), q2 = KV3(kk, 18);
#pragma unroll
    for (int r = 0; r < 6; r++) ap[r] = KV(kk, 36 + r);
    const float lo = KV(kk, 33), hi_ = KV(kk, 34);
    const F3P rel = f3p(p2(KV(kk, 0), KV(kk, 1)), p2(KV(kk, 2), KV(kk, 3)), p2(KV(kk, 4), KV(kk, 5)));  // (relA, -relB) pairs
    float res = 0.f;
    res = fmaxf(res, fabsf(row_iter<false, false, ISO>(p, rel, Q, KV(kk, 21), KV(kk, 27), 0.f, 0.f, ap[0])));
    res = fmaxf(res, fabsf(row_iter<false, false, ISO>(q, rel, Q, KV(kk, 22), KV(kk, 28), 0.f, 0.f, ap[1])));
    res = fmaxf(res, fabsf(row_iter<true, false, ISO>(p2_, rel, Q, KV(kk, 23), KV(kk, 29), 0.f, 0.f, ap[2])));
    res = fmaxf(res, fabsf(row_iter<true, false, ISO>(q2, rel, Q, KV(kk, 24), KV(kk, 30), 0.f, 0.f, ap[3])));
    if (KV(kk, 25) != 0.f) res = fmaxf(res, fabsf(row_iter<true, true, ISO>(ax1, rel, Q, KV(kk, 25), KV(kk, 31), lo, hi_, ap[4])));
    if (KV(kk, 26) != 0.f) res = fmaxf(res, fabsf(row_iter<false, false, ISO>(ax1, rel, Q, KV(kk, 26), KV(kk, 32), 0.f, 0.f, ap[5])));
    return res;
}
template <bool ISO>
DEV float slider_solve(const Ctx &c, const EvmEntryC &V, const Blk42 &kk, BodyPD &Q) {
    const int s = V.slot;
    float ap[6];
    const float res = slider_rows<ISO>(kk, Q, ap);
    // the sphere pair stays in the caller's registers for the two p2p constraints that follow
    {
        float w[EVM_S_STRIDE];
#pragma unroll
        for (int r = 0; r < 6; r++) w[36 + r] = ap[r];
        w[42] = 0.f; w[43] = 0.f;
        rec_store<9, 11>(c, s, w);
    }
    return res;
}

// ---------------------------------------------------------------------------------------------
// point-to-point (btPoint2PointConstraint::getInfo2NonVirtual); which = 0: member A - sphere A, 1: B side
// scratch: a1_3 a2_3 | jd3 | rhs3 | applied3
// ---------------------------------------------------------------------------------------------
DEV void p2p_setup(const Ctx &c, int mi, int which) {
    const EvmMuscleC &M = c_skel.muscle[mi];
    const int ba = which ? M.mb : M.ma, bb = which ? M.sb : M.sa;
    const BodyK A = load_bodyk(c, ba), B = load_bodyk(c, bb);
    const F3 a1 = mul(A.R, load_f3(which ? M.piv_b : M.piv_a));
    const F3 a2 = mul(B.R, f3(0.f, 0.f, 0.f));
    const float k = FPS_F * ERP_F;
    const int s = c_skel.sc_p + EVM_P_STRIDE * (2 * mi + which);
    float rec[EVM_P_STRIDE];
    row_setup<true>(f3(1.f, 0.f, 0.f), a1, a2, A, B, k * (a2.x + B.o.x - a1.x - A.o.x), rec[6], rec[9]);
    row_setup<true>(f3(0.f, 1.f, 0.f), a1, a2, A, B, k * (a2.y + B.o.y - a1.y - A.o.y), rec[7], rec[10]);
    row_setup<true>(f3(0.f, 0.f, 1.f), a1, a2, A, B, k * (a2.z + B.o.z - a1.z - A.o.z), rec[8], rec[11]);
    put3(rec, 0, a1); put3(rec, 3, a2);
    rec[12] = 0.f; rec[13] = 0.f; rec[14] = 0.f; rec[15] = 0.f;
    rec_store<0, EVM_P_STRIDE / 4>(c, s, rec);
}
// p2p rows along the world axes.  The pivot in the attach sphere is the origin (muscle.cpp:52,55), so the
// sphere-side lever arm a2 is exactly zero: body B only takes the linear part.
// kk: the p2p record (4 quads), A: the member, dlB: the attach sphere's linear delta (kept by the caller), s: record slot
struct Blk16 {
    f32x4 q[4];
};
DEV float p2p_rows(const Blk16 &kk, BodyD &A, F3 &dlB, float imB, float &ap0, float &ap1, float &ap2) {
    const F3 a1 = KV3(kk, 0);
    ap0 = KV(kk, 12); ap1 = KV(kk, 13); ap2 = KV(kk, 14);
    float res = 0.f;
    {   // x: c1 = a1 x e_x = (0, a1.z, -a1.y)
        const F3 angA = f3(A.I.xy * a1.z - A.I.xz * a1.y, A.I.yy * a1.z - A.I.yz * a1.y, A.I.yz * a1.z - A.I.zz * a1.y);
        const float d1 = A.dl.x + (a1.z * A.da.y - a1.y * A.da.z);
        float dI = KV(kk, 9);
        dI -= d1 * KV(kk, 6);
        dI -= (-dlB.x) * KV(kk, 6);
        ap0 += dI;
        A.dl.x += A.im * dI; A.da = A.da + angA * dI; dlB.x -= imB * dI;
        res = fmaxf(res, fabsf(dI));
    }
    {   // y: c1 = a1 x e_y = (-a1.z, 0, a1.x)
        const F3 angA = f3(A.I.xz * a1.x - A.I.xx * a1.z, A.I.yz * a1.x - A.I.xy * a1.z, A.I.zz * a1.x - A.I.xz * a1.z);
        const float d1 = A.dl.y + (a1.x * A.da.z - a1.z * A.da.x);
        float dI = KV(kk, 10);
        dI -= d1 * KV(kk, 7);
        dI -= (-dlB.y) * KV(kk, 7);
        ap1 += dI;
        A.dl.y += A.im * dI; A.da = A.da + angA * dI; dlB.y -= imB * dI;
        res = fmaxf(res, fabsf(dI));
    }
    {   // z: c1 = a1 x e_z = (a1.y, -a1.x, 0)
        const F3 angA = f3(A.I.xx * a1.y - A.I.xy * a1.x, A.I.xy * a1.y - A.I.yy * a1.x, A.I.xz * a1.y - A.I.yz * a1.x);
        const float d1 = A.dl.z + (a1.y * A.da.x - a1.x * A.da.y);
        float dI = KV(kk, 11);
        dI -= d1 * KV(kk, 8);
        dI -= (-dlB.z) * KV(kk, 8);
        ap2 += dI;
        A.dl.z += A.im * dI; A.da = A.da + angA * dI; dlB.z -= imB * dI;
        res = fmaxf(res, fabsf(dI));
    }
    return res;
}
DEV float p2p_solve(const Ctx &c, int s, const Blk16 &kk, BodyD &A, F3 &dlB, float imB) {
    float ap0, ap1, ap2;
    const float res = p2p_rows(kk, A, dlB, imB, ap0, ap1, ap2);
    {
        float w[EVM_P_STRIDE];
        w[12] = ap0; w[13] = ap1; w[14] = ap2; w[15] = 0.f;
        rec_store<3, 4>(c, s, w);
    }
    return res;
}

// ---------------------------------------------------------------------------------------------
// contacts: member hull vs floor plane, Bullet-style persistent manifold (A = floor/static, B = member)
// manifold point fields: 0..2 localA, 3..5 localB, 6 dist, 7 applied, 8 applied_lateral
// ---------------------------------------------------------------------------------------------
#define MFP(m, slot, f) GS(mfp, ((m) * 4 + (slot)) * 9 + (f))

struct MPoint {
    F3 la, lb;
    float dist, ap, apl;
};
DEV MPoint sel(bool cnd, const MPoint &a, const MPoint &b) {
    MPoint r;
    r.la = f3(cnd ? a.la.x : b.la.x, cnd ? a.la.y : b.la.y, cnd ? a.la.z : b.la.z);
    r.lb = f3(cnd ? a.lb.x : b.lb.x, cnd ? a.lb.y : b.lb.y, cnd ? a.lb.z : b.lb.z);
    r.dist = cnd ? a.dist : b.dist; r.ap = cnd ? a.ap : b.ap; r.apl = cnd ? a.apl : b.apl;
    return r;
}
DEV MPoint load_mp(const Ctx &c, int m, int slot) {
    MPoint p;
    p.la = f3(MFP(m, slot, 0), MFP(m, slot, 1), MFP(m, slot, 2));
    p.lb = f3(MFP(m, slot, 3), MFP(m, slot, 4), MFP(m, slot, 5));
    p.dist = MFP(m, slot, 6); p.ap = MFP(m, slot, 7); p.apl = MFP(m, slot, 8);
    return p;
}
DEV void store_mp(const Ctx &c, int m, int slot, const MPoint &p) {
    MFP(m, slot, 0) = p.la.x; MFP(m, slot, 1) = p.la.y; MFP(m, slot, 2) = p.la.z;
    MFP(m, slot, 3) = p.lb.x; MFP(m, slot, 4) = p.lb.y; MFP(m, slot, 5) = p.lb.z;
    MFP(m, slot, 6) = p.dist; MFP(m, slot, 7) = p.ap; MFP(m, slot, 8) = p.apl;
}

// Deepest-vertex scan of one hull slice: the first strict minimum of world y over the slice, every product and
// sum individually rounded (no fma contraction) so that the discrete choice follows the same rounding as a plain
// multiply/add sequence.  Two vertices per packed instruction (pair layout of EvmSkelC::hull); the coordinates
// are wave-uniform and arrive as wide scalar loads, four pairs per trip.  Even and odd vertices keep separate
// running minima, merged at the end (lower value first, then lower index).
DEV void hull_scan(const Ctx &c, int si, F3 r1, float oy) {
    const EvmScanC &S = c_skel.scan[si];
    const int m = S.member;
    const EvmMemberC &MB = c_skel.member[m];
    const P2 rx = p2(r1.x, r1.x), ry = p2(r1.y, r1.y), rz = p2(r1.z, r1.z), oyp = p2(oy, oy);
    const float *hp = c_skel.hull + 3 * (MB.hull_off + S.begin);  // 6 floats per vertex pair
    const int np = (S.end - S.begin + 1) >> 1;
    P2 best = p2(EVM_INF, EVM_INF);
    int be = 0, bo = 0;  // pair index of the even / odd minimum
#define EVM_SCAN_PAIR(H, Q, PI)                                                                       \
    {                                                                                                 \
        const P2 x = p2(H[6 * (Q)], H[6 * (Q) + 1]), y = p2(H[6 * (Q) + 2], H[6 * (Q) + 3]),          \
                 z = p2(H[6 * (Q) + 4], H[6 * (Q) + 5]);                                              \
        const P2 wy = xa2(xa2(xa2(xm2(rx, x), xm2(ry, y)), xm2(rz, z)), oyp);                         \
        const bool ce = wy.x < best.x, co = wy.y < best.y;                                            \
        best = p2(ce ? wy.x : best.x, co ? wy.y : best.y);                                            \
        be = ce ? (PI) : be;                                                                          \
        bo = co ? (PI) : bo;                                                                          \
    }
    int p = 0;
    for (; p + 4 <= np; p += 4) {
        float h[24];
#pragma unroll
        for (int k = 0; k < 24; k++) h[k] = hp[6 * p + k];
#pragma unroll
        for (int q = 0; q < 4; q++) EVM_SCAN_PAIR(h, q, p + q)
    }
    for (; p < np; p++) {
        float h[6];
#pragma unroll
        for (int k = 0; k < 6; k++) h[k] = hp[6 * p + k];
        EVM_SCAN_PAIR(h, 0, p)
    }
#undef EVM_SCAN_PAIR
    float b = best.x;
    int bi = 2 * be;
    if (best.y < b || (best.y == b && 2 * bo + 1 < bi)) { b = best.y; bi = 2 * bo + 1; }
    LPART(si, 0) = b;
    LPART(si, 1) = __int_as_float(S.begin + bi);
}

// manifold maintenance for member m from the scanned deepest vertex; returns the number of cached points afterwards
// drop: lanes whose env starts a reset with this step (their cached points are discarded: removeRigidBody/addRigidBody)
DEV int contact_update(const Ctx &c, int m, MPoint *pts_out = nullptr, bool drop = false) {
    const EvmMemberC &MB = c_skel.member[m];
    const F3 o = G3(pos, 3 * m);
    const M33 R = m33(SC3(c_skel.sc_r + 9 * m), SC3(c_skel.sc_r + 9 * m + 3), SC3(c_skel.sc_r + 9 * m + 6));
    float best = LPART(MB.scan_first, 0);
    int bi = __float_as_int(LPART(MB.scan_first, 1));
    for (int k = 1; k < MB.scan_count; k++) {  // slices in vertex order: a strict compare keeps the first minimum
        const float b2 = LPART(MB.scan_first + k, 0);
        const int i2 = __float_as_int(LPART(MB.scan_first + k, 1));
        if (b2 < best) { best = b2; bi = i2; }
    }
    const float *hp = c_skel.hull;
    int n = drop ? 0 : GS(mfn, m);
    const float thr = MB.break_thr;
    const float depth = (best - c_skel.floor_top_y) - (MARGIN_F + MARGIN_F);
    const bool add = !(depth > thr);
    if (!__any(add || n > 0)) {
        if (__any(drop)) GS(mfn, m) = n;  // n == 0 in every lane here
        return 0;
    }

    MPoint p0 = load_mp(c, m, 0), p1 = load_mp(c, m, 1), p2 = load_mp(c, m, 2), p3 = load_mp(c, m, 3);
    const F3 fo = load_f3(c_skel.floor_o);
    if (add) {
        const int g = MB.hull_off + bi, hb = 6 * (g >> 1) + (g & 1);
        const F3 ps = f3(hp[hb], hp[hb + 2], hp[hb + 4]);
        const F3 w = f3(xa(xa(xa(xm(R.r0.x, ps.x), xm(R.r0.y, ps.y)), xm(R.r0.z, ps.z)), o.x),
                        best,
                        xa(xa(xa(xm(R.r2.x, ps.x), xm(R.r2.y, ps.y)), xm(R.r2.z, ps.z)), o.z));
        const F3 pointOnB = f3(w.x, w.y - MARGIN_F, w.z);
        const F3 pointA = pointOnB + f3(0.f, -1.f, 0.f) * depth;
        MPoint np;
        np.la = pointA - fo;                  // floor basis is the identity
        np.lb = tmul(R, pointOnB - o);        // btTransform::invXform
        np.dist = depth; np.ap = 0.f; np.apl = 0.f;
        // getCacheEntry
        float shortest = thr * thr;
        int nearest = -1;
        { const F3 d = p0.la - np.la; const float dd = dot(d, d); if (0 < n && dd < shortest) { shortest = dd; nearest = 0; } }
        { const F3 d = p1.la - np.la; const float dd = dot(d, d); if (1 < n && dd < shortest) { shortest = dd; nearest = 1; } }
        { const F3 d = p2.la - np.la; const float dd = dot(d, d); if (2 < n && dd < shortest) { shortest = dd; nearest = 2; } }
        { const F3 d = p3.la - np.la; const float dd = dot(d, d); if (3 < n && dd < shortest) { shortest = dd; nearest = 3; } }
        int ins;
        if (nearest >= 0) {
            ins = nearest;
            const MPoint old = sel(ins == 0, p0, sel(ins == 1, p1, sel(ins == 2, p2, p3)));
            np.ap = old.ap; np.apl = old.apl;
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
    // refreshContactPoints: distances first, then removal from the last slot down (swap-with-last)
#define REFRESH_DIST(P)                                       \
    {                                                         \
        const F3 posA_ = P.la + fo;                           \
        const F3 posB_ = mul(R, P.lb) + o;                    \
        P.dist = -(posA_.y - posB_.y); /* dot(posA - posB, (0,-1,0)) */ \
    }
    REFRESH_DIST(p3) REFRESH_DIST(p2) REFRESH_DIST(p1) REFRESH_DIST(p0)
#undef REFRESH_DIST
#define REFRESH_SLOT(I, P)                                                                         \
    if (I < n) {                                                                                   \
        const F3 posA = P.la + fo;                                                                 \
        const F3 posB = mul(R, P.lb) + o;                                                          \
        bool rm = !(P.dist <= thr);                                                                \
        if (!rm) {                                                                                 \
            const F3 projected = posA - f3(0.f, -1.f, 0.f) * P.dist;                               \
            const F3 diff = posB - projected;                                                      \
            rm = dot(diff, diff) > thr * thr;                                                      \
        }                                                                                          \
        if (rm) {                                                                                  \
            const int last = n - 1;                                                                \
            const MPoint lp = sel(last == 0, p0, sel(last == 1, p1, sel(last == 2, p2, p3)));      \
            P = lp;                                                                                \
            n--;                                                                                   \
        }                                                                                          \
    }
    REFRESH_SLOT(3, p3)
    REFRESH_SLOT(2, p2)
    REFRESH_SLOT(1, p1)
    REFRESH_SLOT(0, p0)
#undef REFRESH_SLOT
    store_mp(c, m, 0, p0); store_mp(c, m, 1, p1); store_mp(c, m, 2, p2); store_mp(c, m, 3, p3);
    if (pts_out) { pts_out[0] = p0; pts_out[1] = p1; pts_out[2] = p2; pts_out[3] = p3; }
    GS(mfn, m) = n;
    return n;
}

// contact rows of member m: setup, warm start, split-impulse recovery (all of it touches body m only)
// pts: the member's four manifold points when the caller still has them in registers (split pipeline), else re-read
DEV void contact_setup(const Ctx &c, int m, int n, const MPoint *pts = nullptr) {
    const EvmMemberC &MB = c_skel.member[m];
    const BodyK B = load_bodyk(c, m);
    BodyD D = load_bodyd(c, m);
    const F3 nrm = f3(0.f, -1.f, 0.f);
    const float invdt = 1.f / DT_F;
    F3 push = f3(0.f, 0.f, 0.f), turn = f3(0.f, 0.f, 0.f);
    // split-impulse bookkeeping for up to 4 points
    F3 pc2[4], pang[4];
    float pjd[4], prhs[4], ppush[4];
    bool any_pen = false;
    float rec[EVM_CM_STRIDE];  // record image; jd_n == 0 marks "no point" for the sweeps
#pragma unroll
    for (int i = 0; i < EVM_CM_STRIDE; i++) rec[i] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        pjd[j] = 0.f; prhs[j] = 0.f; ppush[j] = 0.f; pc2[j] = f3(0, 0, 0); pang[j] = f3(0, 0, 0);
        if (j < n) {
            const F3 lb = pts ? pts[j].lb : f3(MFP(m, j, 3), MFP(m, j, 4), MFP(m, j, 5));
            const float dist = pts ? pts[j].dist : MFP(m, j, 6);
            const F3 posB = mul(B.R, lb) + B.o;
            const F3 rel = posB - B.o;
            // convertContact: relative velocity of the contact point (static body A contributes 0)
            const F3 vel2 = B.v + cross(B.w, rel);
            const F3 vel = f3(0.f, 0.f, 0.f) - vel2;
            const float rel_vel = dot(nrm, vel);
            // normal row
            const F3 tq = cross(rel, nrm);
            const F3 c2 = -tq;
            const F3 angB = mul(B.I, c2);
            const F3 vec = cross(-angB, rel);
            const float denom1 = B.im + dot(nrm, vec);
            const float jd = 1.0f / (0.f + denom1 + 0.f);
            const F3 n2 = -nrm;
            const float applied = (pts ? pts[j].ap : MFP(m, j, 7)) * WARM_F;
            // warm start: internalApplyImpulse(-n2 * invMass, -angB, -applied)
            D.dl = D.dl + ((-n2) * B.im) * (-applied);
            D.da = D.da + (-angB) * (-applied);
            const float vel2Dotn = dot(n2, B.v) + dot(c2, B.w);
            const float rv = 0.f + vel2Dotn;
            float positionalError = 0.f;
            float velocityError = 0.f - rv;
            const float penetration = dist + 0.f;
            if (penetration > 0.f) velocityError -= penetration * invdt;
            else positionalError = -penetration * ERP_F * invdt;
            const float penImp = positionalError * jd, velImp = velocityError * jd;
            float rhs, rhs_pen;
            if (penetration > SPLIT_THR_F) { rhs = penImp + velImp; rhs_pen = 0.f; }
            else { rhs = velImp; rhs_pen = penImp; }
            // friction direction
            F3 lat = vel - nrm * rel_vel;
            const float lat2 = len2(lat);
            if (lat2 > EVM_EPS) lat = lat * (1.f / sqrtf(lat2));
            else { F3 d2; plane_space(nrm, lat, d2); }
            const F3 fn2 = -lat;
            const F3 fc2 = cross(rel, fn2);
            const F3 fangB = mul(B.I, fc2);
            const F3 fvec = cross(-fangB, rel);
            const float fjd = 1.0f / (0.f + (B.im + dot(lat, fvec)));
            const float fv2 = dot(fn2, B.v) + dot(fc2, B.w_raw);  // no external torque impulse here
            const float frhs = (0.f - (0.f + fv2)) * fjd;
            const float fapplied = (pts ? pts[j].apl : MFP(m, j, 8)) * WARM_F;
            D.dl = D.dl + ((-fn2) * B.im) * (-fapplied);
            D.da = D.da + (-fangB) * (-fapplied);
            put3(rec, EVM_C_STRIDE * j, rel); put3(rec, EVM_C_STRIDE * j + 3, lat);
            rec[EVM_C_STRIDE * j + 6] = jd; rec[EVM_C_STRIDE * j + 7] = rhs;
            rec[EVM_C_STRIDE * j + 8] = fjd; rec[EVM_C_STRIDE * j + 9] = frhs;
            // the accumulated impulses live in the record during the sweeps (contact_writeback returns them)
            rec[40 + 2 * j] = applied; rec[41 + 2 * j] = fapplied;
            pjd[j] = jd; prhs[j] = rhs_pen; pc2[j] = c2; pang[j] = angB;
            any_pen = any_pen || (rhs_pen != 0.f);
        }
    }
    store_bodyd(c, m, D);
    rec_store<0, EVM_CM_STRIDE / 4>(c, c_skel.sc_c + EVM_CM_STRIDE * m, rec);
    if (__any(any_pen)) {
        for (int it = 0; it < NUM_ITER; it++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (prhs[j] != 0.f) {
                    float dI = prhs[j];
                    const float d2 = dot(f3(0.f, 1.f, 0.f), push) + dot(pc2[j], turn);
                    dI -= d2 * pjd[j];
                    const float sum = ppush[j] + dI;
                    if (sum < 0.f) { dI = 0.f - ppush[j]; ppush[j] = 0.f; } else ppush[j] = sum;
                    push = push + (f3(0.f, 1.f, 0.f) * B.im) * dI;
                    turn = turn + pang[j] * dI;
                }
            }
        }
    }
    SSC3(c_skel.sc_pt + 6 * m, push); SSC3(c_skel.sc_pt + 6 * m + 3, turn);
}

// Gauss-Seidel rows of member m's contact points; k = the member's record, requested one schedule entry ahead
// rows only: w[40..47] = the accumulated impulses after the visit
DEV float contact_rows(const Blk42 &k, BodyD &D, float mu, float (&w)[EVM_CM_STRIDE]) {
    float res = 0.f;
    float apn[4];
#pragma unroll
    for (int i = 40; i < EVM_CM_STRIDE; i++) w[i] = KV(k, i);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        apn[j] = 0.f;
        const bool has = KV(k, 10 * j + 6) != 0.f;  // jd_n = 1 / denominator > 0 for a live point
        if (!__any(has)) continue;
        if (has) {  // resolveSingleConstraintRowLowerLimit
            const F3 rel = KV3(k, 10 * j);
            const float jd = KV(k, 10 * j + 6), rhs = KV(k, 10 * j + 7);
            const F3 c2 = -cross(rel, f3(0.f, -1.f, 0.f));
            const F3 angB = mul(D.I, c2);
            float ap = KV(k, 40 + 2 * j);
            float dI = rhs;
            const float d2 = D.dl.y + dot(c2, D.da);
            dI -= d2 * jd;
            const float sum = ap + dI;
            if (sum < 0.f) { dI = 0.f - ap; ap = 0.f; } else ap = sum;
            D.dl = D.dl + (f3(0.f, 1.f, 0.f) * D.im) * dI;
            D.da = D.da + angB * dI;
            w[40 + 2 * j] = ap;
            apn[j] = ap;
            res = fmaxf(res, fabsf(dI));
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (!__any(apn[j] > 0.f)) continue;
        if (apn[j] > 0.f) {  // friction row with limits +-mu * normal impulse
            const F3 rel = KV3(k, 10 * j), lat = KV3(k, 10 * j + 3);
            const float jd = KV(k, 10 * j + 8), rhs = KV(k, 10 * j + 9);
            const F3 n2 = -lat;
            const F3 c2 = cross(rel, n2);
            const F3 angB = mul(D.I, c2);
            const float lim = mu * apn[j];
            float ap = KV(k, 41 + 2 * j);
            float dI = rhs;
            const float d2 = dot(n2, D.dl) + dot(c2, D.da);
            dI -= d2 * jd;
            const float sum = ap + dI;
            if (sum < -lim) { dI = -lim - ap; ap = -lim; }
            else if (sum > lim) { dI = lim - ap; ap = lim; }
            else ap = sum;
            D.dl = D.dl + (n2 * D.im) * dI;
            D.da = D.da + angB * dI;
            w[41 + 2 * j] = ap;
            res = fmaxf(res, fabsf(dI));
        }
    }
    return res;
}
DEV float contact_iter(const Ctx &c, int m, const Blk42 &k, BodyD &D) {
    float w[EVM_CM_STRIDE];
    const float res = contact_rows(k, D, c_skel.member[m].mu, w);
    store_bodyd(c, m, D);
    rec_store<10, 12>(c, c_skel.sc_c + EVM_CM_STRIDE * m, w);
    return res;
}
// after the sweeps: the accumulated impulses go back into the persistent manifold (warm start of the next step)
DEV void contact_writeback(const Ctx &c, int m, int n) {
    const int rec = c_skel.sc_c + EVM_CM_STRIDE * m;
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (j < n) { MFP(m, j, 7) = RC(rec, 40 + 2 * j); MFP(m, j, 8) = RC(rec, 41 + 2 * j); }
}

// ---------------------------------------------------------------------------------------------
// MT19937 stream per env (std::mt19937 + libstdc++ uniform_real_distribution<float>, robot_walk.h:34-35)
// ---------------------------------------------------------------------------------------------
DEV float mt_uniform01(const Ctx &c) {
    int idx = c.d.mt_idx[c.env];
    uint32_t *mt = c.t.mt + c.lane;
    if (idx >= 624) {
        for (int i = 0; i < 624; i++) {
            const uint32_t a = mt[i << 6], b = mt[((i + 1) % 624) << 6];
            const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
            mt[i << 6] = mt[((i + 397) % 624) << 6] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        idx = 0;
    }
    uint32_t y = mt[idx << 6];
    c.d.mt_idx[c.env] = idx + 1;
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    float r = xm((float) y, 2.3283064365386963e-10f);
    if (r >= 1.0f) r = 0.99999994f;
    return r;
}

// the k-th (k = 0..2) upcoming draw of the stream WITHOUT consuming it (the split pipeline learns the rotation of a reset
// one kernel before the wave that owns the stream advances it).  Past the end of the block the value is what the
// in-place regeneration will put there: new[j] = old[j + 397] ^ twist(old[j], old[j + 1]), j <= 2.
DEV float mt_peek01(const Ctx &c, int k) {
    const int p = c.d.mt_idx[c.env] + k;
    const uint32_t *mt = c.t.mt + c.lane;
    uint32_t y;
    if (p < 624) y = mt[p << 6];
    else {
        const int j = p - 624;
        const uint32_t a = mt[j << 6], b = mt[(j + 1) << 6];
        const uint32_t t = (a & 0x80000000u) | (b & 0x7fffffffu);
        y = mt[(j + 397) << 6] ^ (t >> 1) ^ ((t & 1u) ? 0x9908b0dfu : 0u);
    }
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    float r = xm((float) y, 2.3283064365386963e-10f);
    if (r >= 1.0f) r = 0.99999994f;
    return r;
}
// (float) sin((double) x), (float) cos((double) x) for |x| <= 1.2 (reset angles: <= pi / 3)
DEV void sincos_small(float xf, float &s, float &c) {
    const double x = (double) xf, z = x * x;
    if (fabs(x) > 1.2) { s = (float) sin(x); c = (float) cos(x); return; }  // not a reset angle of the reference: library path
    double ps = -1.0 / 51090942171709440000.0;                    // -1/21!
    ps = fma(ps, z, 1.0 / 121645100408832000.0);                   //  1/19!
    ps = fma(ps, z, -1.0 / 355687428096000.0);                     // -1/17!
    ps = fma(ps, z, 1.0 / 1307674368000.0);                        //  1/15!
    ps = fma(ps, z, -1.0 / 6227020800.0);                          // -1/13!
    ps = fma(ps, z, 1.0 / 39916800.0);                             //  1/11!
    ps = fma(ps, z, -1.0 / 362880.0);                              // -1/9!
    ps = fma(ps, z, 1.0 / 5040.0);                                 //  1/7!
    ps = fma(ps, z, -1.0 / 120.0);                                 // -1/5!
    ps = fma(ps, z, 1.0 / 6.0);                                    //  1/3!
    s = (float) fma(-ps * z, x, x);                                // x - x^3 (1/6 - ...)
    double pc = 1.0 / 2432902008176640000.0;                       //  1/20!
    pc = fma(pc, z, -1.0 / 6402373705728000.0);                    // -1/18!
    pc = fma(pc, z, 1.0 / 20922789888000.0);                       //  1/16!
    pc = fma(pc, z, -1.0 / 87178291200.0);                         // -1/14!
    pc = fma(pc, z, 1.0 / 479001600.0);                            //  1/12!
    pc = fma(pc, z, -1.0 / 3628800.0);                             // -1/10!
    pc = fma(pc, z, 1.0 / 40320.0);                                //  1/8!
    pc = fma(pc, z, -1.0 / 720.0);                                 // -1/6!
    pc = fma(pc, z, 1.0 / 24.0);                                   //  1/4!
    pc = fma(pc, z, -0.5);                                         // -1/2!
    c = (float) fma(pc, z, 1.0);
}
// reset rotation from its three uniform draws -> glm::eulerAngleYXZ (robot_walk.cpp:80-86)
DEV M33 repose_rotation(float u0, float u1, float u2) {
    const float angle_limit = c_skel.reset_angle_limit;
    const float half = angle_limit / 2.f;
    const float yaw = xs_(xm(u0, angle_limit), half);
    const float roll = xs_(xm(u1, angle_limit), half);
    const float pitch = xs_(xm(u2, angle_limit), half);
    // sin/cos through fp64 so that the fp32 results are the correctly rounded ones (what glibc returns for
    // all but a handful of arguments); 6 evaluations per reset.  The angles are at most pi / 3 in magnitude: no range
    // reduction is needed and the Taylor series to x^21 / x^20 is exact to fp64 rounding (next term < 2e-22), 20 fmas
    // instead of the ~2000 instructions of six library calls — which sat on the root body's item of the integration kernel and
    // in the sweeps kernel's epilogue in nearly every step (some env of a wave finishes an episode almost every call).
    float ch, sh, cp, sp, cb, sb;
    sincos_small(yaw, sh, ch);
    sincos_small(pitch, sp, cp);
    sincos_small(roll, sb, cb);
    // glm::eulerAngleYXZ(yaw, pitch, roll), stored as rows, every operation individually rounded
    return m33(f3(xa(xm(ch, cb), xm(xm(sh, sp), sb)), xa(xm(-ch, sb), xm(xm(sh, sp), cb)), xm(sh, cp)),
               f3(xm(sb, cp), xm(cb, cp), -sp),
               f3(xa(xm(-sh, cb), xm(xm(ch, sp), sb)), xa(xm(sb, sh), xm(xm(ch, sp), cb)), xm(ch, cp)));
}
DEV M33 repose_draw(const Ctx &c) {  // consumes three draws
    const float u0 = mt_uniform01(c), u1 = mt_uniform01(c), u2 = mt_uniform01(c);
    return repose_rotation(u0, u1, u2);
}
// one body of RigidBodyItem::reset (item.cpp:77-86): new origin E t0 + root_pos, zero velocities; the world inverse inertia
// tensor stays that of the last integrated transform
DEV void repose_body(const Ctx &c, int b, const M33 &E, bool was_pending) {
    if (!was_pending) {
        const Q4 q = q4(GS(quat, 4 * b), GS(quat, 4 * b + 1), GS(quat, 4 * b + 2), GS(quat, 4 * b + 3));
        const S33 I = inertia_world(mat_from_quat(q), load_f3(c_skel.body[b].inv_inertia));
        GS(iinv_stale, 6 * b) = I.xx; GS(iinv_stale, 6 * b + 1) = I.xy; GS(iinv_stale, 6 * b + 2) = I.xz;
        GS(iinv_stale, 6 * b + 3) = I.yy; GS(iinv_stale, 6 * b + 4) = I.yz; GS(iinv_stale, 6 * b + 5) = I.zz;
    }
    const F3 e0 = col0(E), e1 = col1(E), e2 = col2(E);
    const F3 t0 = load_f3(c_skel.body[b].t0);
    const F3 rp = load_f3(c_skel.root_pos);
    const F3 o = f3(xa(xa(xa(xm(e0.x, t0.x), xm(e1.x, t0.y)), xm(e2.x, t0.z)), rp.x),
                    xa(xa(xa(xm(e0.y, t0.x), xm(e1.y, t0.y)), xm(e2.y, t0.z)), rp.y),
                    xa(xa(xa(xm(e0.z, t0.x), xm(e1.z, t0.y)), xm(e2.z, t0.z)), rp.z));
    S3(pos, 3 * b, o);
    S3(lin, 3 * b, f3(0.f, 0.f, 0.f));
    S3(ang, 3 * b, f3(0.f, 0.f, 0.f));
    if (b < c_skel.nm) SSC3(c_skel.sc_ms + 3 * b, o);  // motion state := new transform (item.cpp:81)
}
// per-env bookkeeping of a reset: manifolds dropped, rotation kept for the first step, counters
DEV void repose_finish(const Ctx &c, const M33 &E, int flags, bool drop_manifolds = true) {
    if (drop_manifolds) {
        for (int m = 0; m < c_skel.nm; m++) GS(mfn, m) = 0;
        for (int p = 0; p < c_skel.npair; p++) c.t.pmn[(p << 6) + c.lane] = 0;
    }
    GS(E, 0) = E.r0.x; GS(E, 1) = E.r0.y; GS(E, 2) = E.r0.z;
    GS(E, 3) = E.r1.x; GS(E, 4) = E.r1.y; GS(E, 5) = E.r1.z;
    GS(E, 6) = E.r2.x; GS(E, 7) = E.r2.y; GS(E, 8) = E.r2.z;
    c.d.flags[c.env] = flags | EVM_FLAG_PENDING;
    c.d.curr_step[c.env] = 0;                       // robot_walk.cpp:100-101 (nothing reads them in between)
    c.d.remaining[c.env] = c_skel.init_remaining;
}
// RobotWalk::reset_engine up to the settle steps (robot_walk.cpp:76-96, item.cpp:77-86), one lane
DEV void repose(const Ctx &c) {
    const M33 E = repose_draw(c);
    const int flags = c.d.flags[c.env];
    const bool was_pending = (flags & EVM_FLAG_PENDING) != 0;
    for (int b = 0; b < c_skel.nb; b++) repose_body(c, b, E, was_pending);
    repose_finish(c, E, flags);
}

// ---------------------------------------------------------------------------------------------
// observation + reward + termination (robot_walk.cpp:56-74; proprioception_state.cpp)
// ---------------------------------------------------------------------------------------------
DEV void euler_zyx(Q4 q, float &yaw, float &pitch, float &roll) {
    const float sqx = q.x * q.x, sqy = q.y * q.y, sqz = q.z * q.z, squ = q.w * q.w;
    const float sarg = -2.0f * (q.x * q.z - q.w * q.y);
    if (sarg <= -0.99999f) { pitch = -0.5f * EVM_PI; roll = 0.f; yaw = 2.0f * atan2f(q.x, -q.y); }
    else if (sarg >= 0.99999f) { pitch = 0.5f * EVM_PI; roll = 0.f; yaw = 2.0f * atan2f(-q.x, q.y); }
    else {
        pitch = asinf(sarg);
        roll = atan2f(2.0f * (q.y * q.z + q.w * q.x), squ - sqx - sqy + sqz);
        yaw = atan2f(2.0f * (q.x * q.y + q.w * q.z), squ + sqx - sqy - sqz);
    }
}

struct BodyState {  // a body right after integration
    F3 o, lin, ang, ms;
    Q4 q;
};
// the 19-value block of member m (proprioception_state.cpp:23-58,86-112); root_ms = the root's motion-state origin
DEV void member_values(const Ctx &c, int m, F3 root_ms, float (&v)[19], const BodyState *st = nullptr) {
    const float PI_F = (float) 3.14159265358979323846;
    // st: the member's freshly integrated state when the caller still has it in registers, else read back
    const Q4 qs = st ? st->q : q4(GS(quat, 4 * m), GS(quat, 4 * m + 1), GS(quat, 4 * m + 2), GS(quat, 4 * m + 3));
    const Q4 q = quat_from_mat(mat_from_quat(qs));  // getWorldTransform().getRotation()
    float yaw, pitch, roll;
    euler_zyx(q, yaw, pitch, roll);
    const F3 lv = st ? st->lin : G3(lin, 3 * m), av = st ? st->ang : G3(ang, 3 * m);
    const F3 ll = G3(hist, 6 * m), la = G3(hist, 6 * m + 3);
    const F3 dl = ll - lv, da = la - av;
    S3(hist, 6 * m, lv); S3(hist, 6 * m + 3, av);
    v[0] = yaw / PI_F; v[1] = pitch / PI_F; v[2] = roll / PI_F;
    v[3] = lv.x; v[4] = lv.y; v[5] = lv.z;
    v[6] = av.x / PI_F; v[7] = av.y / PI_F; v[8] = av.z / PI_F;
    v[9] = dl.x; v[10] = dl.y; v[11] = dl.z;
    v[12] = da.x / PI_F; v[13] = da.y / PI_F; v[14] = da.z / PI_F;
    v[15] = 0.f;  // floor_touched is never raised after construction (proprioception_state.cpp:18,39-40)
    if (m == c_skel.root) {
        const F3 p = st ? st->o : G3(pos, 3 * m);
        v[16] = logf(sqrtf(dot(p, p)) + 1.f);
        v[17] = p.y;
        v[18] = atan2f(p.z, p.x);
    } else {
        const F3 d = (st ? st->ms : SC3(c_skel.sc_ms + 3 * m)) - root_ms;
        v[16] = d.x; v[17] = d.y; v[18] = d.z;
    }
}
DEV void observe_member(const Ctx &c, int m, float *obs, F3 root_ms) {
    float v[19];
    member_values(c, m, root_ms, v);
    float *o = obs + (size_t) c.env * c_skel.obs_dim + 19 * c_skel.state_index[m];
#pragma unroll
    for (int k = 0; k < 19; k++) o[k] = v[k];
}
// A K-value block of all 64 envs of a full tile, written row-contiguously: the values cross lanes through a wave-private
// LDS buffer [K][64] so that consecutive lanes write consecutive floats of an env's row (K-float runs) instead of one
// float per 1 484-byte row each.  `who` = ballot of the envs that emit an observation in this call.
template <int K>
DEV void store_block_rows(const Ctx &c, float *obs, int col0, const float (&v)[K], float *buf, unsigned long long who) {
#pragma unroll
    for (int k = 0; k < K; k++) buf[(k << 6) + c.lane] = v[k];
    __builtin_amdgcn_wave_barrier();
    float *tile_obs = obs + (size_t) (c.env - c.lane) * c_skel.obs_dim + col0;
    for (int idx = c.lane; idx < K * 64; idx += 64) {
        const int e = idx / K, k = idx - e * K;
        if ((who >> e) & 1ull) tile_obs[(size_t) e * c_skel.obs_dim + k] = buf[(k << 6) + e];
    }
    __builtin_amdgcn_wave_barrier();
}
DEV void observe_muscle(const Ctx &c, int mi, float *obs) {  // MuscleState (proprioception_state.cpp:124-129)
    float *o = obs + (size_t) c.env * c_skel.obs_dim + 19 * c_skel.nm + 4 * mi;
    o[0] = SC(c_skel.sc_mobs + 4 * mi + 0);
    o[1] = SC(c_skel.sc_mobs + 4 * mi + 1);
    o[2] = SC(c_skel.sc_mobs + 4 * mi + 2);
    o[3] = SC(c_skel.sc_mobs + 4 * mi + 3);
}
// reward, counters, termination (robot_walk.cpp:61-72 / robot_jump.cpp:71-84): one writer per env
DEV void observe_tail(const Ctx &c, float *reward, uint8_t *done) {
    const int root = c_skel.root;
    // robot_walk.cpp:61-68: the root's z velocity; robot_jump.cpp:71-80: max(vy, 0) + vz and a strict fail test
    const float vz = c_skel.env_kind == 1 ? fmaxf(GS(lin, 3 * root + 1), 0.f) + GS(lin, 3 * root + 2) : GS(lin, 3 * root + 2);
    int remaining = c.d.remaining[c.env], cs = c.d.curr_step[c.env];
    if (vz < c_skel.min_vel) remaining -= 1;
    else if (vz >= c_skel.target_vel) remaining += 1;
    const bool win = cs >= c_skel.max_steps;
    const bool fail = c_skel.env_kind == 1 ? remaining < 0 : remaining <= 0;
    c.d.remaining[c.env] = remaining;
    c.d.curr_step[c.env] = cs + 1;
    reward[c.env] = vz;
    done[c.env] = (win | fail) ? 1 : 0;
}
DEV void observe(const Ctx &c, float *obs, float *reward, uint8_t *done) {  // monolithic kernel: dealt to its waves
    const F3 root_ms = SC3(c_skel.sc_ms + 3 * c_skel.root);
    for (int m = c.wave; m < c_skel.nm; m += EVM_NW) observe_member(c, m, obs, root_ms);
    for (int mi = c.wave; mi < c_skel.nmus; mi += EVM_NW) observe_muscle(c, mi, obs);
    if (c.wave == 0) observe_tail(c, reward, done);
}

// ---------------------------------------------------------------------------------------------
// one stepSimulation(1/60) for this lane's env
// ---------------------------------------------------------------------------------------------
#ifdef EVM_STAMPS
#define STAMPX(i) do { if (c.lane == 0) c.d.stamps[(size_t) blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMP(i) do { if (c.lane == 0 && c.wave == 0) c.d.stamps[(size_t) blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// one body at the start of a step: world basis, world inverse inertia and zeroed solver deltas into the tile
// (LDS, or its global staging copy in the split pipeline), implicit gyroscopic impulse
DEV void body_prepare(const Ctx &c, int b, bool pending, bool any_pending, const M33 &E) {
    const EvmBodyC &BC = c_skel.body[b];
    const Q4 q0 = q4(GS(quat, 4 * b), GS(quat, 4 * b + 1), GS(quat, 4 * b + 2), GS(quat, 4 * b + 3));
    M33 R = mat_from_quat(q0);
    const F3 invI = load_f3(BC.inv_inertia);
    // attach spheres: R diag(k, k, k) R^T = k R R^T = k * identity for any rotation.  (Their gyroscopic term
    // w x (I w) is zero in exact arithmetic, but Bullet's implicit formula divides its rounding noise by I —
    // about 1e-7 |w|^2 rad/s per step — and that noise is part of the reference's trajectory: it stays.)
    S33 I;
    if (BC.isotropic) { I.xx = invI.x; I.xy = 0.f; I.xz = 0.f; I.yy = invI.x; I.yz = 0.f; I.zz = invI.x; }
    else I = inertia_world(R, invI);
    if (any_pending) {
        // first step after reset(): transform = E * M0 (non-orthonormal) and the inverse
        // inertia tensor is still the one of the last integrated transform
        const M33 Rp = glm_mul_basis(E, load_m33(BC.m0));
        S33 Ip;
        Ip.xx = GS(iinv_stale, 6 * b); Ip.xy = GS(iinv_stale, 6 * b + 1); Ip.xz = GS(iinv_stale, 6 * b + 2);
        Ip.yy = GS(iinv_stale, 6 * b + 3); Ip.yz = GS(iinv_stale, 6 * b + 4); Ip.zz = GS(iinv_stale, 6 * b + 5);
        if (pending) { R = Rp; I = Ip; }
    }
    SSC3(c_skel.sc_r + 9 * b, R.r0); SSC3(c_skel.sc_r + 9 * b + 3, R.r1); SSC3(c_skel.sc_r + 9 * b + 6, R.r2);
    LII(b, 0) = I.xx; LII(b, 1) = I.xy; LII(b, 2) = I.xz; LII(b, 3) = I.yy; LII(b, 4) = I.yz; LII(b, 5) = I.zz;
#pragma unroll
    for (int k = 0; k < 6; k++) LDV(b, k) = 0.f;
    LVER(c)[b] = 0;
    // btRigidBody::computeGyroscopicImpulseImplicit_Body (contraction-free: see dev_math.h, namespace nc)
    const F3 idl = f3(1.f / invI.x, 1.f / invI.y, 1.f / invI.z);
    SSC3(c_skel.sc_ext + 3 * b, nc::gyro_impulse(quat_from_mat(R), idl, G3(ang, 3 * b), DT_F));
}

// one body at the end of a step: velocities += solver deltas, split-impulse pose correction, transform integration
DEV BodyState body_integrate(const Ctx &c, int b) {
    F3 o = G3(pos, 3 * b);
    M33 R = m33(SC3(c_skel.sc_r + 9 * b), SC3(c_skel.sc_r + 9 * b + 3), SC3(c_skel.sc_r + 9 * b + 6));
    const F3 dl = f3(LDV(b, 0), LDV(b, 1), LDV(b, 2)), da = f3(LDV(b, 3), LDV(b, 4), LDV(b, 5));
    F3 lin = G3(lin, 3 * b) + dl;
    F3 ang = G3(ang, 3 * b) + da;
    if (b < c_skel.nm) {
        const F3 push = SC3(c_skel.sc_pt + 6 * b), turn = SC3(c_skel.sc_pt + 6 * b + 3);
        const bool nz = push.x != 0.f || push.y != 0.f || push.z != 0.f || turn.x != 0.f || turn.y != 0.f || turn.z != 0.f;
        if (__any(nz)) {
            F3 o2; Q4 q2;
            integrate_transform(o, R, push, turn * SPLIT_TURN_ERP_F, DT_F, o2, q2);
            if (nz) { o = o2; R = mat_from_quat(q2); }
        }
    }
    lin = lin + f3(0.f, c_skel.body[b].ext_force_y, 0.f);
    ang = ang + SC3(c_skel.sc_ext + 3 * b);
    F3 o2; Q4 q2;
    integrate_transform(o, R, lin, ang, DT_F, o2, q2);
    S3(pos, 3 * b, o2);
    GS(quat, 4 * b) = q2.x; GS(quat, 4 * b + 1) = q2.y; GS(quat, 4 * b + 2) = q2.z; GS(quat, 4 * b + 3) = q2.w;
    S3(lin, 3 * b, lin);
    S3(ang, 3 * b, ang);
    BodyState st;
    st.o = o2; st.q = q2; st.lin = lin; st.ang = ang;
    st.ms = integ_pos(o2, lin, 0.f - DT_F);  // btDefaultMotionState, one step behind
    if (b < c_skel.nm) SSC3(c_skel.sc_ms + 3 * b, st.ms);
    return st;
}

// the NUM_ITER projected Gauss-Seidel sweeps of one tile (whole workgroup; tile state in LDS), then the per-constraint
// readbacks.  cmask: members with a cached contact point in any lane of the tile.
DEV void sweeps_run(const Ctx &c, bool any_pending, unsigned cmask, int ncontact) {
    const int W = c.wave;
#ifdef EVM_STAMPS3
    if (W == 0 && c.lane < 16) c.d.stamps[(size_t) blockIdx.x * 16 + c.lane] = 0;
    unsigned long long t_type[6] = {0, 0, 0, 0, 0, 0};
    unsigned n_type[6] = {0, 0, 0, 0, 0, 0};
#endif
    // ---- projected Gauss-Seidel sweeps ----
    // A wave walks its slice of the level schedule (EvmSkelC::sched); a workgroup barrier closes each level.
    const int ns = c_skel.nwsched[W];
    float res = 0.f;
    // Dataflow sweep.  Visits that share no body commute exactly, so the only ordering that matters is, per body,
    // the order of the visits that touch it.  Every body carries a version counter in LDS (= how many visits have
    // written it); a visit waits until its two bodies have reached the versions it expects, solves, and publishes the
    // new versions.  No workgroup barrier inside or between sweeps; every wave's list follows one global topological
    // order, so the globally lowest unfinished entry is always at the head of some wave's list: no deadlock.
    // `after_loads` runs once the entry's LDS reads have landed and before its rows (a stretch of pure VALU work): the
    // place to issue the scalar load of a later descriptor — SMEM and LDS share lgkmcnt and every LDS wait is a full
    // lgkmcnt(0), so a scalar load issued anywhere else is waited for almost immediately.
#ifdef EVM_STAMPS2
    unsigned long long t_wait = 0, t_solve = 0;
    const unsigned long long t_loop0 = __builtin_amdgcn_s_memtime();
#endif
    auto run_visit = [&](const EvmEntryC &V, const Blk42 &k, int it, auto &&after_loads) -> float {
        const int expA = it * V.psA + (V.need & 0xffff);
        const int expB = it * V.psB + (V.need >> 16);
#ifdef EVM_STAMPS2
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
        const S33P I = load_inertia_pair(c, V.a, V.b);
        wait_versions2(c, V.a, expA, V.b, expB);
#ifdef EVM_STAMPS2
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
#ifdef EVM_STAMPS3
        const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
        float r;
        {
            BodyPD Q = load_bodypd(c, V.a, V.b, V.imA, V.imB, I);
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
            after_loads();
            r = V.type == 0 ? hinge_solve(c, V, k, Q) : fixed_solve(c, V, k, Q);
        }
        publish_version(c, V.a, expA + 1);
        publish_version(c, V.b, expB + 1);
#ifdef EVM_STAMPS3
        {
            const unsigned long long dt = __builtin_amdgcn_s_memtime() - ts0;
#pragma unroll
            for (int q = 0; q < 2; q++) if (V.type == q) { t_type[q] += dt; n_type[q]++; }
        }
#endif
#ifdef EVM_STAMPS2
        t_wait += t1 - t0; t_solve += __builtin_amdgcn_s_memtime() - t1;
#endif
        return r;
    };
    // A whole muscle (type 5): slider between the two attach spheres, then the p2p of each sphere to its member, in Bullet's
    // order.  The spheres are touched by nothing else: no version counters, one LDS load at the start and one store at the
    // end.  The slider rows need nothing from outside, so they run before the entry waits for member a; member b is waited
    // for only after a's rows are done and published.
    auto run_muscle = [&](const EvmEntryC &V, const Blk42 &k, int it, auto &&after_loads) -> float {
        const int sa = V.spheres & 0xffff, sb = V.spheres >> 16;
        const int expA = it * V.psA + (V.need & 0xffff);
        const int expB = it * V.psB + (V.need >> 16);
#if defined(EVM_STAMPS2) || defined(EVM_STAMPS3)
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
        // the two p2p records: requested now, needed after the slider rows
        Blk16 ra, rb;
        {
            const int mi = (V.slot - c_skel.sc_s) / EVM_S_STRIDE;
            const f32x4 *pa = rec_quads(c, c_skel.sc_p + EVM_P_STRIDE * (2 * mi)), *pb = rec_quads(c, c_skel.sc_p + EVM_P_STRIDE * (2 * mi + 1));
#pragma unroll
            for (int q = 0; q < 4; q++) { ra.q[q] = pa[q << 6]; rb.q[q] = pb[q << 6]; }
        }
        const bool iso = V.iso && !any_pending;
        S33P I;
        if (iso) {  // inverse inertia k * identity, straight from the descriptor
            I.xx = p2(V.kA, V.kB);
            I.xy = I.xz = I.yz = p2(0.f, 0.f);
            I.yy = I.zz = I.xx;
        } else {
            I = load_inertia_pair(c, sa, sb);
        }
        BodyPD Q = load_bodypd(c, sa, sb, V.imSa, V.imSb, I);
        const S33 Ia = lds_inertia(c, V.a), Ib = lds_inertia(c, V.b);  // the members' tensors: constant during the sweeps
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        after_loads();
        float r = iso ? slider_solve<true>(c, V, k, Q) : slider_solve<false>(c, V, k, Q);
        const int mi = (V.slot - c_skel.sc_s) / EVM_S_STRIDE;
        unsigned long long twait = 0;
        {   // p2p(a, sa)
#if defined(EVM_STAMPS2) || defined(EVM_STAMPS3)
            const unsigned long long w0 = __builtin_amdgcn_s_memtime();
#endif
            wait_version(c, V.a, expA);
#if defined(EVM_STAMPS2) || defined(EVM_STAMPS3)
            twait += __builtin_amdgcn_s_memtime() - w0;
#endif
            BodyD A;
            A.dl = f3(LDV(V.a, 0), LDV(V.a, 1), LDV(V.a, 2));
            A.da = f3(LDV(V.a, 3), LDV(V.a, 4), LDV(V.a, 5));
            A.I = Ia; A.im = V.imA;
            F3 dlS = lo(Q.dl);
            r = fmaxf(r, p2p_solve(c, c_skel.sc_p + EVM_P_STRIDE * (2 * mi), ra, A, dlS, V.imSa));
            store_bodyd(c, V.a, A);
            publish_version(c, V.a, expA + 1);
            Q.dl.x.x = dlS.x; Q.dl.y.x = dlS.y; Q.dl.z.x = dlS.z;
        }
        {   // p2p(b, sb)
#if defined(EVM_STAMPS2) || defined(EVM_STAMPS3)
            const unsigned long long w0 = __builtin_amdgcn_s_memtime();
#endif
            wait_version(c, V.b, expB);
#if defined(EVM_STAMPS2) || defined(EVM_STAMPS3)
            twait += __builtin_amdgcn_s_memtime() - w0;
#endif
            BodyD B;
            B.dl = f3(LDV(V.b, 0), LDV(V.b, 1), LDV(V.b, 2));
            B.da = f3(LDV(V.b, 3), LDV(V.b, 4), LDV(V.b, 5));
            B.I = Ib; B.im = V.imB;
            F3 dlS = hi(Q.dl);
            r = fmaxf(r, p2p_solve(c, c_skel.sc_p + EVM_P_STRIDE * (2 * mi + 1), rb, B, dlS, V.imSb));
            store_bodyd(c, V.b, B);
            publish_version(c, V.b, expB + 1);
            Q.dl.x.y = dlS.x; Q.dl.y.y = dlS.y; Q.dl.z.y = dlS.z;
        }
        // the spheres' deltas: imS sign convention of the pair (im = (1/m, -1/m)) does not touch dl / da themselves
        store_bodypd(c, sa, sb, Q);
#ifdef EVM_STAMPS3
        { t_type[2] += __builtin_amdgcn_s_memtime() - t0 - twait; n_type[2]++; t_type[3] += twait; n_type[3]++; }
#endif
#ifdef EVM_STAMPS2
        t_wait += twait; t_solve += __builtin_amdgcn_s_memtime() - t0 - twait;
#endif
        return r;
    };
    // contact rows of a member: after all of its joint visits of this sweep, before the next sweep's
    auto run_contact = [&](const EvmEntryC &V, int it, const Blk42 &k, auto &&after_loads) -> float {
        const int m = V.a, ps = V.psA;
#ifdef EVM_STAMPS2
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
        BodyD D;
        D.I = lds_inertia(c, m);
        wait_version(c, m, it * ps + ps - 1);
#ifdef EVM_STAMPS2
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
        float r = 0.f;
#ifdef EVM_STAMPS3
        const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
        const bool active = (cmask & (1u << m)) != 0;
        if (active) {
            D.dl = f3(LDV(m, 0), LDV(m, 1), LDV(m, 2));
            D.da = f3(LDV(m, 3), LDV(m, 4), LDV(m, 5));
            D.im = c_skel.body[m].inv_mass;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        after_loads();
        if (active) r = contact_iter(c, m, k, D);
        publish_version(c, m, (it + 1) * ps);
#ifdef EVM_STAMPS3
        {
            const unsigned long long dt = __builtin_amdgcn_s_memtime() - ts0;
#ifdef EVM_STAMPS3_HULL  // split the active contact entries by hull size (cubes vs the 509-vertex feet)
            if (active && c_skel.member[m].hull_n <= 8) { t_type[3] += dt; n_type[3]++; } else
#endif
            if (active) { t_type[4] += dt; n_type[4]++; } else { t_type[5] += dt; n_type[5]++; }
        }
#endif
#ifdef EVM_STAMPS2
        t_wait += t1 - t0; t_solve += __builtin_amdgcn_s_memtime() - t1;
#endif
        return r;
    };
    auto run_entry = [&](const EvmEntryC &V, const Blk42 &k, int it, auto &&after_loads) -> float {
        return V.type == 4 ? run_contact(V, it, k, after_loads) : (V.type == 5 ? run_muscle(V, k, it, after_loads) : run_visit(V, k, it, after_loads));
    };
    // a member without a cached point in any lane of the tile has nothing to prefetch
    auto entry_slots = [&](const EvmEntryC &V) { return (V.type == 4 && !(cmask & (1u << V.a))) ? 0 : V.nslots; };
    const EvmEntryC *stream = c_skel.wsched[W];
    // The wave's entries of all NUM_ITER sweeps form one stream (no barrier between sweeps: the version counters
    // carry the sweep number).  Three-stage software pipeline over the stream: the descriptor (scalar loads) of
    // entry j + 2 and the record (vector loads, two register blocks in ping-pong) of entry j + 1 are requested
    // before entry j runs, so neither latency sits between a wave's consecutive visits.
    if (ns == 1) {
        // a record may only be requested after the wave's previous run of the same entry has stored it; with a
        // single entry that is the immediately preceding one, so there is nothing to overlap
        const EvmEntryC v = stream[0];
        for (int it = 0; it < NUM_ITER; it++) {
            Blk42 k;
            blk_load(c, v, k, entry_slots(v));
            res = run_entry(v, k, it, [] {});
        }
    } else if (ns > 1) {
        const int T = NUM_ITER * ns;
        int f_i = 0, f_it = 0;  // stream position of the next descriptor to fetch
        auto fetch = [&](EvmEntryC &v, int &it_) {
            v = stream[f_i];  // one s_load_dwordx16; its address depends on the stream position only
            it_ = f_it;
            if (++f_i == ns) { f_i = 0; f_it++; }  // runs past the end by up to three entries: fetched, never run
        };
        Blk42 ka, kb;
        int it0, it1, it2, it3;
        EvmEntryC v0, v1, v2, v3;
        fetch(v0, it0);
        fetch(v1, it1);
        blk_load(c, v0, ka, entry_slots(v0));
        for (int j = 0; j < T; j += 2) {
            blk_load(c, v1, kb, entry_slots(v1));
            {
                const float r = run_entry(v0, ka, it0, [&] { fetch(v2, it2); });
                if (it0 == NUM_ITER - 1) res = fmaxf(res, r);
            }
            blk_load(c, v2, ka, entry_slots(v2));
            if (j + 1 < T) {
                const float r = run_entry(v1, kb, it1, [&] { fetch(v3, it3); });
                if (it1 == NUM_ITER - 1) res = fmaxf(res, r);
            } else {
                fetch(v3, it3);
            }
            v0 = v2; it0 = it2;
            v1 = v3; it1 = it3;
        }
    }
#ifdef EVM_STAMPS2
#ifdef EVM_STAMPS4  // whole entry stream of the wave vs the time inside its entries (difference = per-entry overhead)
    if (c.lane == 0) { c.d.stamps[(size_t) blockIdx.x * 16 + 2 * W] = __builtin_amdgcn_s_memtime() - t_loop0; c.d.stamps[(size_t) blockIdx.x * 16 + 2 * W + 1] = t_wait + t_solve; }
#else
    if (c.lane == 0) { c.d.stamps[(size_t) blockIdx.x * 16 + 2 * W] = t_wait; c.d.stamps[(size_t) blockIdx.x * 16 + 2 * W + 1] = t_solve; }
#endif
#endif
#ifdef EVM_STAMPS3
    __syncthreads();
    if (c.lane == 0) {
#pragma unroll
        for (int q = 0; q < 6; q++) {
            atomicAdd(&c.d.stamps[(size_t) blockIdx.x * 16 + 2 * q], t_type[q]);
            atomicAdd(&c.d.stamps[(size_t) blockIdx.x * 16 + 2 * q + 1], (unsigned long long) n_type[q]);
        }
    }
#endif
    {   // batch-level residual: this wave's rows, max over its lanes, one atomic per wave
        int r = __float_as_int(res);
        const unsigned long long live = __ballot(true);  // lanes outside the batch / the mask have left the kernel
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const int other = __shfl_xor(r, o);
            if ((live >> ((c.lane & 63) ^ o)) & 1ull) r = max(r, other);
        }
        if ((c.lane & 63) == (int) __builtin_ctzll(live)) atomicMax(c.d.resid, r);
    }
    __syncthreads();
    STAMP(5);
    if (W == 0) {
        GS(diag, 0) = res;
        GS(diag, 1) = (float) ncontact;
    }

    // ---- contact impulses back into the manifolds (by the wave that ran the member's contact rows) ----
    for (int i = 0; i < ns; i++) {
        const EvmEntryC &V = c_skel.wsched[W][i];
        if (V.type == 4 && (cmask & (1u << V.a))) contact_writeback(c, V.a, GS(mfn, V.a));
    }
    // ---- muscle readbacks: getAppliedImpulse() = impulse of the last row written back ----
    for (int mi = W; mi < c_skel.nmus; mi += EVM_NW) {
        const int s = c_skel.sc_s + EVM_S_STRIDE * mi;
        const float jd4 = RC(s, 25), jd5 = RC(s, 26);
        const float a3 = RC(s, 39), a4 = RC(s, 40), a5 = RC(s, 41);
        SC(c_skel.sc_mobs + 4 * mi + 1) = jd5 != 0.f ? a5 : (jd4 != 0.f ? a4 : a3);
        SC(c_skel.sc_mobs + 4 * mi + 2) = RC(c_skel.sc_p + EVM_P_STRIDE * (2 * mi), 14);
        SC(c_skel.sc_mobs + 4 * mi + 3) = RC(c_skel.sc_p + EVM_P_STRIDE * (2 * mi + 1), 14);
    }
    STAMP(6);
}

DEV void physics_step(const Ctx &c, int flags) {
    // Four waves (one per SIMD of the CU) work on the SAME 64 environments and share the LDS tile.  Work items
    // of a phase that touch disjoint bodies are dealt to the waves; __syncthreads() closes every phase.  The
    // Gauss-Seidel sweep keeps Bullet's order up to exact commutation (level schedule in EvmSkelC::sched).
    const int W = c.wave;
    STAMP(0);
    const bool pending = (flags & EVM_FLAG_PENDING) != 0;
    const bool powered = (flags & EVM_FLAG_POWERED) != 0;
    const bool any_pending = __any(pending);
    M33 E;
    if (any_pending)
        E = m33(f3(GS(E, 0), GS(E, 1), GS(E, 2)), f3(GS(E, 3), GS(E, 4), GS(E, 5)), f3(GS(E, 6), GS(E, 7), GS(E, 8)));

    // ---- bodies: world basis, inverse inertia tile -> LDS, implicit gyroscopic impulse, zero deltas ----
    for (int b = W; b < c_skel.nb; b += EVM_NW) body_prepare(c, b, pending, any_pending, E);
    __syncthreads();
    STAMP(1);

    // ---- collision: hull vs floor plane, persistent manifolds (members dealt to waves by hull size) ----
    for (int i = 0; i < c_skel.nscan; i++)
        if (c_skel.scan[i].wave == W) hull_scan(c, i, SC3(c_skel.sc_r + 9 * c_skel.scan[i].member + 3), GS(pos, 3 * c_skel.scan[i].member + 1));
    STAMP(9);   // wave 0: its own scans done
    __syncthreads();
    STAMP(10);  // all scans done
    for (int m = 0; m < c_skel.nm; m++)
        if (c_skel.member_wave[m] == W && c_skel.member[m].contact_response) contact_update(c, m);
    STAMP(11);  // wave 0: its manifolds done
    __syncthreads();
    int ncontact = 0;
    unsigned cmask = 0;  // wave-uniform: members with a cached point in any lane
    for (int m = 0; m < c_skel.nm; m++) {
        const int n = GS(mfn, m);
        ncontact += n;
        if (__any(n > 0)) cmask |= 1u << m;
    }
    STAMP(2);

    // ---- joint rows: every constraint's record is independent of the others ----
    for (int v = W; v < c_skel.nvisit; v += EVM_NW) {
        const EvmVisitC &V = c_skel.visit[v];
        switch (V.type) {
            case 0: hinge_setup(c, (V.slot - c_skel.sc_h) / EVM_H_STRIDE); break;
            case 1: fixed_setup(c, (V.slot - c_skel.sc_f) / EVM_F_STRIDE); break;
            case 2: { const int mi = (V.slot - c_skel.sc_s) / EVM_S_STRIDE; slider_setup(c, mi, powered, GS(target, mi)); break; }
            default: { const int k = (V.slot - c_skel.sc_p) / EVM_P_STRIDE; p2p_setup(c, k >> 1, k & 1); break; }
        }
    }
    STAMP(3);
    // ---- contact rows: setup + warm start + split impulse (touches the member's own deltas only) ----
    for (int m = 0; m < c_skel.nm; m++) {
        if (c_skel.member_wave[m] != W) continue;
        if (cmask & (1u << m)) contact_setup(c, m, GS(mfn, m));
        else { SSC3(c_skel.sc_pt + 6 * m, f3(0.f, 0.f, 0.f)); SSC3(c_skel.sc_pt + 6 * m + 3, f3(0.f, 0.f, 0.f)); }
    }
    __syncthreads();
    STAMP(4);

    sweeps_run(c, any_pending, cmask, ncontact);

    // ---- write back velocities, split-impulse pose correction, integrate transforms ----
    for (int b = W; b < c_skel.nb; b += EVM_NW) body_integrate(c, b);
    __syncthreads();
    STAMP(7);
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// MODE bits: 1 apply action, 2 observe, 4 auto-reset rollout form
template <int MODE>
__global__ __launch_bounds__(64 * EVM_NW) void k_env_step(EnvDev d, const float *__restrict__ action, float *obs,
                                                          float *reward, uint8_t *done, uint8_t *valid,
                                                          const uint8_t *__restrict__ mask) {
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    Ctx c = make_ctx(d, lds_dyn);
    // lanes outside the batch / the mask drop out; every wave of the workgroup owns the same lanes, so either
    // all four waves keep running (and meet at every barrier) or all four leave here
    if (c.env >= d.n_real) return;
    if (mask && !mask[c.env]) return;
    const bool lead = c.wave == 0;  // per-env bookkeeping is written by wave 0 only
    if (MODE & 4) {
        if (lead && (d.flags[c.env] & EVM_FLAG_DONE)) {
            repose(c);
            d.flags[c.env] &= ~EVM_FLAG_DONE;
            d.settle_left[c.env] = c_skel.settle_steps;
            GS(stat, 1) += 1;
        }
        __syncthreads();
    }
    int flags = d.flags[c.env];
    const int settle0 = (MODE & 4) ? d.settle_left[c.env] : 0;
    const bool settling = settle0 > 0;
    if (MODE & 1) {
        if (lead && !settling) {  // MuscleController::on_input -> Muscle::contract
            for (int mi = 0; mi < c_skel.nmus; mi++)
                GS(target, mi) = action[(size_t) c.env * c_skel.nmus + mi] * c_skel.muscle[mi].speed;
            if (MODE & 4) GS(stat, 0) += 1;
        }
        if (!settling) flags |= EVM_FLAG_POWERED;
        __syncthreads();  // targets are read by whichever wave sets up the slider
    }
    physics_step(c, flags);
    flags &= ~EVM_FLAG_PENDING;
    bool do_observe = (MODE & 2) != 0;
    if (MODE & 4) {
        if (settling) do_observe = settle0 == 1;
        if (lead) {
            if (settling) d.settle_left[c.env] = settle0 - 1;
            valid[c.env] = do_observe ? (settling ? 2 : 1) : 0;  // 1 = do_step transition, 2 = reset()'s own step
        }
    }
    if (do_observe) {
        observe(c, obs, reward, done);
        STAMP(8);
        if (lead && (MODE & 4) && done[c.env]) flags |= EVM_FLAG_DONE;
    }
    if (lead) d.flags[c.env] = flags;
}

}  // namespace evm
#include "pairs_dev.h"
namespace evm {

// =================================================================================================================
// Split pipeline.  At the mandated 4096 envs/GPU the monolithic kernel occupies 64 of the 256 CUs, and 40 % of its time
// goes to phases that are embarrassingly parallel over bodies / constraints / members.  Here those phases run as
// separate launches over (tile, part) grids that fill the chip; only the Gauss-Seidel sweeps stay one workgroup per
// tile.  The tile that the sweeps keep in LDS (solver deltas, world inverse inertia, scan minima) has a global staging
// copy of identical layout (EnvDev::gtile): the pre kernels fill it through the same LII / LDV / LPART accessors
// (Ctx::lds points at it), the sweeps kernel loads it into LDS and stores the final deltas back, the post kernel
// integrates from it.
//   k_split_pre_a     bodies (basis, inertia, gyroscopic impulse) and hull-scan slices          items independent
//   k_split_pre_b     joint records (+ muscle targets from the action), manifolds + contact rows items independent
//   k_split_sweeps    10 sweeps + readbacks + the root's next motion state
//   k_split_post      integration; each member's observation block by the wave that integrated it; the root's wave
//                     also does reward / termination / rollout bookkeeping
// =================================================================================================================
#define EVM_SPLIT_WAVES 4  // waves per workgroup of the pre / post kernels

DEV float *tile_stage(const EnvDev &d) { return d.gtile + (size_t) blockIdx.x * d.tile_floats; }
// Per-lane view of the step that is starting.  In the rollout form an env whose last transition was terminal (DONE)
// begins its reset with this call: nothing is written for that before the sweeps kernel (a single workgroup per tile, so
// there is one writer and no reader left); the setup kernels derive the state the reset will produce ("effective"):
// pending with the rotation the post kernel drew ahead (sc_nexte), settle counter at its start value, manifolds dropped.
struct LaneState {
    int flags;            // live flags (not yet updated for a starting reset)
    bool fin;             // a reset starts with this call
    bool was_pending;     // PENDING before that
    bool pending, any_pending, settling;   // effective
    int settle0;                           // effective
    M33 E;                                 // effective reset rotation (valid where pending)
};
template <int MODE>
DEV LaneState lane_state(const Ctx &c) {
    LaneState L;
    L.flags = c.d.flags[c.env];
    L.fin = (MODE & 4) && (L.flags & EVM_FLAG_DONE) != 0;
    L.was_pending = (L.flags & EVM_FLAG_PENDING) != 0;
    L.pending = L.was_pending || L.fin;
    L.any_pending = __any(L.pending);
    L.settle0 = (MODE & 4) ? (L.fin ? c_skel.settle_steps : c.d.settle_left[c.env]) : 0;
    L.settling = L.settle0 > 0;
    if (L.any_pending) {
        L.E = m33(f3(GS(E, 0), GS(E, 1), GS(E, 2)), f3(GS(E, 3), GS(E, 4), GS(E, 5)), f3(GS(E, 6), GS(E, 7), GS(E, 8)));
        if (__any(L.fin)) {
            const M33 N = m33(SC3(c_skel.sc_nexte), SC3(c_skel.sc_nexte + 3), SC3(c_skel.sc_nexte + 6));
            if (L.fin) L.E = N;
        }
    }
    return L;
}
// lanes outside the batch / the mask drop out (whole waves of a tile agree: every wave owns the same lanes)
#define EVM_SPLIT_GUARD()                      \
    if (c.env >= d.n_real) return;             \
    if (mask && !mask[c.env]) return;

template <int MODE>
__global__ __launch_bounds__(64 * EVM_SPLIT_WAVES) void k_split_pre_a(EnvDev d, const uint8_t *__restrict__ mask) {
    Ctx c = make_ctx(d, tile_stage(d));
    // The NEXT step's narrowphase work lists start empty (this step's were zeroed by the previous step: the broadphase items below
    // append to them).  Before the guard: one wavefront of the grid does it whatever the mask leaves of its tile.
    if (c_skel.self_collision && blockIdx.x == 0 && blockIdx.y == gridDim.y - 1 && c.wave == EVM_SPLIT_WAVES - 1) {
        int *nxt = pc_next(d);
        for (int k = c.lane; k <= c_skel.npair + 1; k += 64) nxt[k] = 0;
    }
    EVM_SPLIT_GUARD()
    const LaneState L = lane_state<MODE>(c);
    const int vw = blockIdx.y * EVM_SPLIT_WAVES + c.wave, nvw = gridDim.y * EVM_SPLIT_WAVES;
    if (c_skel.self_collision && vw == nvw - 1) {  // the pair kernel ORs its live pairs in, the setup kernels the split-impulse flag
        const int nwords = ((c_skel.npair + 31) >> 5) + 1;
        for (int k = 0; k < nwords; k++) c.t.pact[(k << 6) + c.lane] = 0u;
    }
    // one item list (bodies, then scan slices, then — member-vs-member mode — the broadphase of the member pairs) dealt round
    // robin, so that no wave gets the head of two kinds
    const int nbroad = c_skel.self_collision ? c_skel.npair : 0;
    for (int j = vw; j < c_skel.nb + c_skel.nscan + nbroad; j += nvw) {
        if (j < c_skel.nb) {
            if (__any(L.fin)) { if (L.fin) repose_body(c, j, L.E, L.was_pending); }  // RigidBodyItem::reset of a starting reset
            body_prepare(c, j, L.pending, L.any_pending, L.E);
            continue;
        }
        if (j >= c_skel.nb + c_skel.nscan) {
            // broadphase of one member pair: the envs whose boxes overlap, or that hold a cached point, go to the pair's work
            // list for the narrowphase.  The members' transforms are derived here, as body_prepare / repose_body derive them
            // (those run in other waves of this kernel): basis from the quaternion, or E * M0 in the step that follows a reset;
            // origin from the state, or the re-posed one when the reset starts with this step.
            const int p = j - c_skel.nb - c_skel.nscan;
            M33 Rm[2];
            F3 om[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                const int m = s2 == 0 ? (int) c_skel.pair[p].a : (int) c_skel.pair[p].b;
                M33 R = mat_from_quat(q4(GS(quat, 4 * m), GS(quat, 4 * m + 1), GS(quat, 4 * m + 2), GS(quat, 4 * m + 3)));
                if (L.any_pending) {
                    const M33 Rp = glm_mul_basis(L.E, load_m33(c_skel.body[m].m0));
                    if (L.pending) R = Rp;
                }
                F3 o = G3(pos, 3 * m);
                if (__any(L.fin)) {  // (repose_body's arithmetic)
                    const F3 e0 = col0(L.E), e1 = col1(L.E), e2 = col2(L.E);
                    const F3 t0 = load_f3(c_skel.body[m].t0), rp = load_f3(c_skel.root_pos);
                    const F3 orp = f3(xa(xa(xa(xm(e0.x, t0.x), xm(e1.x, t0.y)), xm(e2.x, t0.z)), rp.x),
                                      xa(xa(xa(xm(e0.y, t0.x), xm(e1.y, t0.y)), xm(e2.y, t0.z)), rp.y),
                                      xa(xa(xa(xm(e0.z, t0.x), xm(e1.z, t0.y)), xm(e2.z, t0.z)), rp.z));
                    if (L.fin) o = orp;
                }
                Rm[s2] = R; om[s2] = o;
            }
            pair_broadphase(c, p, L.fin, Rm[0], om[0], Rm[1], om[1]);
            continue;
        }
        const int i = j - c_skel.nb;
        // the scan needs row 1 of the member's basis; the bodies are being prepared by other waves, so it is rebuilt here
        const int m = c_skel.scan[i].member;
        M33 R = mat_from_quat(q4(GS(quat, 4 * m), GS(quat, 4 * m + 1), GS(quat, 4 * m + 2), GS(quat, 4 * m + 3)));
        if (L.any_pending) {
            const M33 Rp = glm_mul_basis(L.E, load_m33(c_skel.body[m].m0));
            if (L.pending) R = Rp;
        }
        float oy = GS(pos, 3 * m + 1);
        if (__any(L.fin)) {  // the member's origin after the re-pose (another wave is writing it right now)
            const F3 e0 = col0(L.E), e1 = col1(L.E), e2 = col2(L.E);
            const F3 t0 = load_f3(c_skel.body[m].t0);
            const float ry = xa(xa(xa(xm(e0.y, t0.x), xm(e1.y, t0.y)), xm(e2.y, t0.z)), c_skel.root_pos[1]);
            if (L.fin) oy = ry;
        }
        hull_scan(c, i, R.r1, oy);
    }
}

#ifndef EVM_PRE_B_WAVES
#define EVM_PRE_B_WAVES 3
#endif
// The items of k_split_pre_b for the virtual wave vw of nvw (a function of its own: the narrowphase kernel's record blocks run
// the same items, k_split_pairs_rec below)
template <int MODE>
DEV void pre_b_items(const Ctx &c, const LaneState &L, int vw, int nvw, const float *__restrict__ action, int broad, int tile_ix) {
    (void) tile_ix;
    const bool powered = (L.flags & EVM_FLAG_POWERED) != 0 || ((MODE & 1) && !L.settling);
#ifdef EVM_STAMPS5
    int s5_kind = -1;
    unsigned long long s5_t0 = 0;
#endif
    // one item list: members first (manifold + contact rows, the longest items), then the joint visits
    (void) broad;
    for (int j = vw; j < c_skel.nm + c_skel.nvisit; j += nvw) {
#ifdef EVM_STAMPS5  // diagnostic: longest item of each kind, per tile (cycles); tools/stamps5.py
        {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (s5_kind >= 0 && c.lane == 0) atomicMax(&c.d.stamps[(size_t) tile_ix * 16 + s5_kind], now - s5_t0);
            s5_kind = j < c_skel.nm ? (c_skel.member[j].hull_n > 64 ? 5 : 4) : c_skel.visit[j - c_skel.nm].type;
            s5_t0 = now;
        }
#endif

        if (j < c_skel.nm) {
            // manifold maintenance, then the contact rows (warm start lands in the member's own deltas)
            const int m = j;
            int n = 0;
            MPoint pts[4];
#ifdef EVM_STAMPS5
            const unsigned long long s5_a = __builtin_amdgcn_s_memtime();
#endif
            if (c_skel.member[m].contact_response) n = contact_update(c, m, pts, L.fin);
            else if (__any(L.fin)) { if (L.fin) GS(mfn, m) = 0; }
#ifdef EVM_STAMPS5  // slots 6 / 7: longest manifold update, longest contact-row setup of a member
            const unsigned long long s5_b = __builtin_amdgcn_s_memtime();
            if (c.lane == 0) atomicMax(&c.d.stamps[(size_t) tile_ix * 16 + 6], s5_b - s5_a);
#endif
            const bool touching = __any(n > 0);
            if (c_skel.self_collision) {
                // member-vs-member mode: the rows go to the two-body record the sweeps kernel's contact rounds read; warm
                // start and split-impulse recovery happen there, in manifold order across floor and pair contacts
                if (touching) floor_record(c, m, n, pts);
                SSC3(c_skel.sc_pt + 6 * m, f3(0.f, 0.f, 0.f)); SSC3(c_skel.sc_pt + 6 * m + 3, f3(0.f, 0.f, 0.f));
                continue;
            }
            if (touching) contact_setup(c, m, n, pts);
#ifdef EVM_STAMPS5
            if (c.lane == 0) atomicMax(&c.d.stamps[(size_t) tile_ix * 16 + 7], __builtin_amdgcn_s_memtime() - s5_b);
#endif
            if (!touching) { SSC3(c_skel.sc_pt + 6 * m, f3(0.f, 0.f, 0.f)); SSC3(c_skel.sc_pt + 6 * m + 3, f3(0.f, 0.f, 0.f)); }
            continue;
        }
        const EvmVisitC &V = c_skel.visit[j - c_skel.nm];
        switch (V.type) {
            case 0: hinge_setup(c, (V.slot - c_skel.sc_h) / EVM_H_STRIDE); break;
            case 1: fixed_setup(c, (V.slot - c_skel.sc_f) / EVM_F_STRIDE); break;
            case 2: {
                const int mi = (V.slot - c_skel.sc_s) / EVM_S_STRIDE;
                if ((MODE & 1) && !L.settling)  // MuscleController::on_input -> Muscle::contract
                    GS(target, mi) = action[(size_t) c.env * c_skel.nmus + mi] * c_skel.muscle[mi].speed;
                slider_setup(c, mi, powered, GS(target, mi));
                break;
            }
            default: { const int k = (V.slot - c_skel.sc_p) / EVM_P_STRIDE; p2p_setup(c, k >> 1, k & 1); break; }
        }
    }
#ifdef EVM_STAMPS5
    if (s5_kind >= 0 && c.lane == 0) atomicMax(&c.d.stamps[(size_t) tile_ix * 16 + s5_kind], __builtin_amdgcn_s_memtime() - s5_t0);
#endif
}

template <int MODE>
__global__ __launch_bounds__(64 * EVM_SPLIT_WAVES) __attribute__((amdgpu_waves_per_eu(EVM_PRE_B_WAVES, EVM_PRE_B_WAVES))) void k_split_pre_b(EnvDev d, const float *__restrict__ action,
                                                                        const uint8_t *__restrict__ mask, int broad) {
    Ctx c = make_ctx(d, tile_stage(d));
    EVM_SPLIT_GUARD()
    const LaneState L = lane_state<MODE>(c);
    const int vw = blockIdx.y * EVM_SPLIT_WAVES + c.wave, nvw = gridDim.y * EVM_SPLIT_WAVES;
    pre_b_items<MODE>(c, L, vw, nvw, action, broad, (int) blockIdx.x);
}

// Member-vs-member mode, narrowphase (pairs_dev.h).  The envs that need a pair are compacted over the whole batch by the
// broadphase items of k_split_pre_a, so a wavefront is full of real work whatever fraction of the envs has that pair close.
//   blocks [0, EVM_BIG_BLOCKS)   the pairs with a big hull (the 451-vertex feet): one query per QUARTER wavefront, the 16 lanes
//                                of a row sharing the hull scans (narrow_dev.h, support_group; the hull sits in LDS); the
//                                blocks walk the flat (pair, env) list
//   the rest, (tile, pair)       the other pairs: one env per lane, the pair wave-uniform, 64 entries of the pair's own list per
//                                block, pairs in decreasing cost order
// One launch for both, so that the few long big-hull wavefronts and the many short ones share the chip.
#ifdef EVM_BIG_SOLO   // measurement build: every big-hull query on a wavefront of its own (the urgent list's form) instead of four per
                      // wavefront.  Measured (512 steps, 4096 envs): 0.3677 against 0.3219 ms per step — four times the wavefronts,
                      // each well under four times shorter (the simplex update is the same serial work on 64 lanes as on 16)
#define EVM_BIG_BLOCKS 8192
#else
#define EVM_BIG_BLOCKS 2048
#endif
#define EVM_URGENT_BLOCKS 256   // the launch's first blocks: the urgent list (pairs_dev.h), one query per wavefront
#ifndef EVM_PAIRS_WAVES
#define EVM_PAIRS_WAVES 2   // wavefronts per SIMD the narrowphase kernel is compiled for (128 arch VGPRs + AGPR spill space at 2)
#endif
template <int MODE>
DEV void narrow_block(const EnvDev &d, int blk, int tiles) {   // blk: block index in the launch
#ifdef EVM_KSTAMPS  // diagnostic build (tools/kstamps.py): working wavefronts of the narrowphase kernel, cycles and extent
    struct KStamp {
        unsigned long long *st, t0, r0; int kind;
        __device__ void begin(unsigned long long *s, int k) { st = s; kind = k; t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
        __device__ ~KStamp() {
            if (!st || threadIdx.x != 0) return;
            const unsigned long long dt = __builtin_amdgcn_s_memtime() - t0, r1 = __builtin_amdgcn_s_memrealtime();
            atomicAdd(&st[3 * kind], dt); atomicAdd(&st[3 * kind + 1], 1ull); atomicMax(&st[3 * kind + 2], dt);
            atomicMin(&st[6], r0); atomicMax(&st[7], r1);
        }
    } ks;
    ks.st = nullptr;
#endif
    if (blk < EVM_SPEC_SLOTS) {
        // Speculation blocks, the launch's very first: entry i (< EVM_SPEC_SLOTS) of the urgent list is ALSO taken by block i here,
        // which runs the pair's penetration query at once and leaves the answer in slot i (narrow_dev.h: speculate_pen_depth); the
        // urgent block that owns the entry (below) asks for it, or calls the run off, when its GJK is through.
        const int cnt = pc_cur(d)[c_skel.npair + 1];
        if (blk >= cnt || d.spec == nullptr) return;
#ifdef EVM_KSTAMPS
        if (threadIdx.x == 0) gj::epa::g_ust_on = 0;
#endif
        const int e = d.blist[(size_t) c_skel.npair * d.n - 1 - blk], p = e >> 20, env = e & 0xfffff;
        // hulls of at most 64 vertices live in the solver's registers: only a bigger one needs the table in LDS
        const bool small = c_skel.member[c_skel.pair[p].a].hull_n <= 64 && c_skel.member[c_skel.pair[p].b].hull_n <= 64;
        int hoff = -1;
        if (!small) {
            const bool all = c_skel.hull_pts <= EVM_LDS_HULL_PTS;
            const int h0 = all ? 0 : c_skel.big_hull_off, hn = all ? c_skel.hull_pts : c_skel.big_hull_n;
            hoff = all ? -2 : c_skel.big_hull_off;
            for (int v = threadIdx.x; v < hn; v += 64) {
                const int g = h0 + v, hb = 6 * (g >> 1) + (g & 1);
                gj::g_lds_hull[v] = gj::gj_f4{c_skel.hull[hb], c_skel.hull[hb + 2], c_skel.hull[hb + 4], 0.f};
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) atomicAdd(&d.errs[5], 1);
        const Ctx c = make_ctx_env(d, env);
        pair_speculate(c, p, hoff, d.spec + (size_t) EVM_SPEC_WORDS * blk);
        return;
    }
    blk -= EVM_SPEC_SLOTS;
    const bool urgent_blk = blk < EVM_URGENT_BLOCKS;
    blk -= EVM_URGENT_BLOCKS;
#ifdef EVM_KSTAMPS
    if (threadIdx.x == 0) { gj::epa::g_ust_on = urgent_blk ? 1 : 0; gj::epa::g_ust[0] = __builtin_amdgcn_s_memtime(); }
#endif
    if (urgent_blk || blk < EVM_BIG_BLOCKS) {
        const int cnt = pc_cur(d)[c_skel.npair + (urgent_blk ? 1 : 0)];
#ifdef EVM_BIG_SOLO
        if (urgent_blk ? blk + EVM_URGENT_BLOCKS >= cnt : blk >= cnt) return;
#else
        if (urgent_blk ? blk + EVM_URGENT_BLOCKS >= cnt : blk * 4 >= cnt) return;
#endif
#ifdef EVM_KSTAMPS
        ks.begin(d.stamps, 0);
#endif
        // the hull table into LDS as [vertex][xyz]: all of it when it fits (the boxes' supports then stay off the memory system
        // too: a GJK iteration of a foot-vs-box query has no global access left), else the big hull alone
        const bool all = c_skel.hull_pts <= EVM_LDS_HULL_PTS;
        const int h0 = all ? 0 : c_skel.big_hull_off, hn = all ? c_skel.hull_pts : c_skel.big_hull_n;
        const int hoff = all ? -2 : c_skel.big_hull_off;
        for (int v = threadIdx.x; v < hn; v += 64) {  // from the pair-packed table
            const int g = h0 + v, hb = 6 * (g >> 1) + (g & 1);
            gj::g_lds_hull[v] = gj::gj_f4{c_skel.hull[hb], c_skel.hull[hb + 2], c_skel.hull[hb + 4], 0.f};
        }
        __syncthreads();
#ifdef EVM_KSTAMPS
        if (threadIdx.x == 0) atomicAdd(&d.stamps[16], __builtin_amdgcn_s_memtime() - ks.t0);  // hull staging
#endif
        if (urgent_blk) {   // ONE query per wavefront, carried by all 64 lanes: the penetration solver's parallel parts get the whole wave
            __builtin_amdgcn_s_setprio(2);
            for (int i = blk + EVM_URGENT_BLOCKS; i < cnt; i += EVM_URGENT_BLOCKS) {
#ifdef EVM_KSTAMPS
                if (threadIdx.x == 0) { gj::epa::g_ust[1] = __builtin_amdgcn_s_memtime(); for (int k = 0; k < 8; k++) gj::epa::g_uph[k] = 0ull; }
#endif
                const int e = d.blist[(size_t) c_skel.npair * d.n - 1 - i], p = e >> 20, env = e & 0xfffff;
                const Ctx c = make_ctx_env(d, env);
                const bool fin = (MODE & 4) && (d.flags[env] & EVM_FLAG_DONE) != 0;
                pair_item<true, true>(c, p, fin, hoff, (i < EVM_SPEC_SLOTS && d.spec != nullptr) ? d.spec + (size_t) EVM_SPEC_WORDS * i : nullptr);
            }
            return;
        }
#ifdef EVM_BIG_SOLO
        for (int i = blk; i < cnt; i += EVM_BIG_BLOCKS) {
            const int e = d.blist[i], p = e >> 20, env = e & 0xfffff;
            const Ctx c = make_ctx_env(d, env);
            const bool fin = (MODE & 4) && (d.flags[env] & EVM_FLAG_DONE) != 0;
            pair_item<true, true>(c, p, fin, hoff, nullptr);
        }
        return;
#endif
        for (int i0 = blk * 4; i0 < cnt; i0 += EVM_BIG_BLOCKS * 4) {
            const int i = i0 + (int) (threadIdx.x >> 4);
            if (i < cnt) {  // (a row without an entry sits the iteration out; rows are independent of each other)
                const int e = d.blist[i], p = e >> 20, env = e & 0xfffff;
                const Ctx c = make_ctx_env(d, env);
                const bool fin = (MODE & 4) && (d.flags[env] & EVM_FLAG_DONE) != 0;
                pair_item<true>(c, p, fin, hoff);
            }
        }
        return;
    }
    const int bx = blk - EVM_BIG_BLOCKS;
    const int p = c_skel.pair_order[bx / tiles];
    const int cnt = pc_cur(d)[p], base = (bx % tiles) * 64;
    if (base >= cnt) return;
    const int i = base + (int) threadIdx.x;
    if (i >= cnt) return;
#ifdef EVM_KSTAMPS
    ks.begin(d.stamps, 1);
#endif
    const int env = d.plist[(size_t) p * d.n + i];
    const Ctx c = make_ctx_env(d, env);
    const bool fin = (MODE & 4) && (d.flags[env] & EVM_FLAG_DONE) != 0;
    pair_item<false>(c, p, fin);
}

template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(EVM_PAIRS_WAVES, EVM_PAIRS_WAVES))) void k_split_pairs(EnvDev d, const uint8_t *__restrict__ mask, int tiles) {
    (void) mask;  // (masked-out envs never enter a list)
    narrow_block<MODE>(d, (int) blockIdx.x, tiles);
}

// GT: the tile does not fit the LDS (more than ~53 bodies): the kernel works in place on the tile's global staging copy — the
// same layout, so the copy-in below degenerates to the attach spheres' zero fill and the copy-out to a self-copy; the waves
// still meet through the version counters (workgroup scope: one CU, one L1).  Slower, but a skeleton is not refused for its size.
template <bool GT>
__global__ __launch_bounds__(64 * EVM_NW) void k_split_sweeps(EnvDev d, const uint8_t *__restrict__ mask, int autoreset) {
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    Ctx c = make_ctx(d, GT ? tile_stage(d) : lds_dyn);
    EVM_SPLIT_GUARD()
    const int flags_in = d.flags[c.env];
    const bool fin = autoreset && (flags_in & EVM_FLAG_DONE) != 0;  // a reset starts with this step (see LaneState)
    const bool any_pending = __any((flags_in & EVM_FLAG_PENDING) != 0 || fin);
    // tile: staging copy -> LDS (solver deltas with the contact warm start, world inverse inertia); versions start at 0
    {
        // An attach sphere (isotropic, no contacts) starts every step with zero deltas, and its inertia tile is read only by
        // the general muscle path (a step that follows a reset): neither is fetched otherwise.
        const float *g = tile_stage(d);
        const int nb6 = c_skel.nb * 6;
        for (int b = c.wave; b < c_skel.nb; b += EVM_NW) {
            const bool light = b >= c_skel.nm && c_skel.body[b].isotropic;
            float v[12];
            if (!light || any_pending) {
#pragma unroll
                for (int k = 0; k < 6; k++) v[6 + k] = g[((nb6 + b * 6 + k) << 6) + c.lane];
            }
            if (!light) {
#pragma unroll
                for (int k = 0; k < 6; k++) v[k] = g[((b * 6 + k) << 6) + c.lane];
            } else {
#pragma unroll
                for (int k = 0; k < 6; k++) v[k] = 0.f;
            }
#pragma unroll
            for (int k = 0; k < 6; k++) c.lds[((b * 6 + k) << 6) + c.lane] = v[k];
            if (!light || any_pending) {
#pragma unroll
                for (int k = 0; k < 6; k++) c.lds[((nb6 + b * 6 + k) << 6) + c.lane] = v[6 + k];
            }
        }
        if (c.wave == 0) for (int b = 0; b < c_skel.nb; b++) LVER(c)[b] = 0;  // every live lane writes: a ragged tile may have one
    }
    int ncontact = 0;
    unsigned cmask = 0;  // wave-uniform: members with a cached point in any lane
    for (int m = 0; m < c_skel.nm; m++) {
        const int n = GS(mfn, m);
        ncontact += n;
        if (__any(n > 0)) cmask |= 1u << m;
    }
    __syncthreads();
    sweeps_run(c, any_pending, cmask, ncontact);
    __syncthreads();
    {   // final deltas -> staging copy
        float *g = tile_stage(d);
        const int nfl = c_skel.nb * 6;
        for (int k = c.wave; k < nfl; k += EVM_NW) g[(k << 6) + c.lane] = c.lds[(k << 6) + c.lane];
    }
    if (c.wave == 0) {
        if (__any(fin)) {
            // the one writer of a starting reset's bookkeeping (RobotWalk::reset_engine): the stream advances by the
            // three draws whose rotation the setup kernels already used, flags / counters / settle count are set
            if (fin) {
                const M33 E = repose_draw(c);
                repose_finish(c, E, flags_in & ~EVM_FLAG_DONE, false);  // the manifolds were dropped by the setup kernel
                d.settle_left[c.env] = c_skel.settle_steps;
                GS(stat, 1) += 1;
            }
        }
        SC(c_skel.sc_snap) = __int_as_float(d.flags[c.env]);
        SC(c_skel.sc_snap + 1) = __int_as_float(d.settle_left[c.env]);
        // the root's motion-state origin after this step, for every member's observation block (the post kernel's
        // waves integrate different bodies concurrently and cannot read each other's results)
        const int b = c_skel.root;
        F3 o = G3(pos, 3 * b);
        const F3 dl = f3(LDV(b, 0), LDV(b, 1), LDV(b, 2));
        F3 lin = G3(lin, 3 * b) + dl;
        const F3 push = SC3(c_skel.sc_pt + 6 * b), turn = SC3(c_skel.sc_pt + 6 * b + 3);
        const bool nz = push.x != 0.f || push.y != 0.f || push.z != 0.f || turn.x != 0.f || turn.y != 0.f || turn.z != 0.f;
        if (nz) o = integ_pos(o, push, DT_F);
        lin = lin + f3(0.f, c_skel.body[b].ext_force_y, 0.f);
        const F3 o2 = integ_pos(o, lin, DT_F);
        SSC3(c_skel.sc_rootms, integ_pos(o2, lin, 0.f - DT_F));
    }
}

// The integration / observation items of k_split_post for processor vw of nvw, one env per lane whatever the lanes' envs are
// (row-wise stores): the lane-group sweeps kernel runs them itself at its end (k_sweeps_g, fused form), on the 16 envs and 16
// (wave, lane group) processors of its workgroup — a launch of its own cost 15 us per step for 3 us of work.  `mode` as
// k_split_post's MODE (bit 0 motors powered, bit 1 observe, bit 2 rollout form).
struct PostArgs {
    float *obs, *reward;
    uint8_t *done, *valid;
    int mode;   // < 0: not fused (k_split_post follows)
};
DEV void post_items(const Ctx &c, const EnvDev &d, const PostArgs &pa, int vw, int nvw) {
    const int MODE = pa.mode;
    LaneState L;  // from the snapshot taken by the sweeps kernel: the live values are rewritten below by the root's processor
    L.flags = __float_as_int(SC(c_skel.sc_snap));
    L.settle0 = (MODE & 4) ? __float_as_int(SC(c_skel.sc_snap + 1)) : 0;
    L.settling = L.settle0 > 0;
    bool do_observe = (MODE & 2) != 0;
    if ((MODE & 4) && L.settling) do_observe = L.settle0 == 1;
    const bool any_observe = __any(do_observe);
    F3 root_ms = f3(0.f, 0.f, 0.f);
    if (any_observe) root_ms = SC3(c_skel.sc_rootms);
    for (int b = vw; b < c_skel.nb; b += nvw) {
        const BodyState st = body_integrate(c, b);
        if (b < c_skel.nm && any_observe) {
            float v[19];
            if (do_observe) {
                member_values(c, b, root_ms, v, &st);
                float *o = pa.obs + (size_t) c.env * c_skel.obs_dim + 19 * c_skel.state_index[b];
#pragma unroll
                for (int k = 0; k < 19; k++) o[k] = v[k];
            }
        }
        if (b == c_skel.root) {
            // one writer per env: reward / termination / counters / rollout bookkeeping
            int flags = L.flags & ~EVM_FLAG_PENDING;
            if ((MODE & 1) && !L.settling) {
                flags |= EVM_FLAG_POWERED;
                if (MODE & 4) GS(stat, 0) += 1;
            }
            if (MODE & 4) {
                if (L.settling) d.settle_left[c.env] = L.settle0 - 1;
                pa.valid[c.env] = do_observe ? (L.settling ? 2 : 1) : 0;  // 1 = do_step transition, 2 = reset()'s own step
            }
            bool fin_next = false;
            if (do_observe) {
                observe_tail(c, pa.reward, pa.done);
                if ((MODE & 4) && pa.done[c.env]) { flags |= EVM_FLAG_DONE; fin_next = true; }
            }
            d.flags[c.env] = flags;
            if ((MODE & 4) && __any(fin_next)) {
                // the next call starts this env's reset: draw its rotation ahead (read-only) for that call's setup kernels
                const M33 N = repose_rotation(mt_peek01(c, 0), mt_peek01(c, 1), mt_peek01(c, 2));
                if (fin_next) { SSC3(c_skel.sc_nexte, N.r0); SSC3(c_skel.sc_nexte + 3, N.r1); SSC3(c_skel.sc_nexte + 6, N.r2); }
            }
        }
    }
    if (any_observe)
        for (int mi = vw; mi < c_skel.nmus; mi += nvw)
            if (do_observe) observe_muscle(c, mi, pa.obs);
}

template <int MODE>
__global__ __launch_bounds__(64 * EVM_SPLIT_WAVES) void k_split_post(EnvDev d, float *obs, float *reward, uint8_t *done,
                                                                       uint8_t *valid, const uint8_t *__restrict__ mask) {
    Ctx c = make_ctx(d, tile_stage(d));
    EVM_SPLIT_GUARD()
    LaneState L;  // from the snapshot taken by the sweeps kernel: the live values are rewritten below by the root's wave
    L.flags = __float_as_int(SC(c_skel.sc_snap));
    L.settle0 = (MODE & 4) ? __float_as_int(SC(c_skel.sc_snap + 1)) : 0;
    L.settling = L.settle0 > 0;
    const int vw = blockIdx.y * EVM_SPLIT_WAVES + c.wave, nvw = gridDim.y * EVM_SPLIT_WAVES;
    bool do_observe = (MODE & 2) != 0;
    if ((MODE & 4) && L.settling) do_observe = L.settle0 == 1;
    const bool any_observe = __any(do_observe);
    __shared__ float sbuf[EVM_SPLIT_WAVES][19 * 64];
    float *buf = sbuf[c.wave];
    const unsigned long long who = __ballot(do_observe);
    const bool full_tile = __popcll(__ballot(true)) == 64;  // every lane is a live env: lanes can write each other's rows
    F3 root_ms = f3(0.f, 0.f, 0.f);
    if (any_observe) root_ms = SC3(c_skel.sc_rootms);
    for (int b = vw; b < c_skel.nb; b += nvw) {
#ifdef EVM_STAMPS5  // slots 8 / 9 / 10: longest attach-sphere item, member item, the root's item of k_split_post
        const unsigned long long s5_p0 = __builtin_amdgcn_s_memtime();
#endif
        const BodyState st = body_integrate(c, b);
        if (b < c_skel.nm && any_observe) {
            float v[19];
            if (do_observe) member_values(c, b, root_ms, v, &st);
            if (full_tile) {
                store_block_rows<19>(c, obs, 19 * c_skel.state_index[b], v, buf, who);
            } else if (do_observe) {
                float *o = obs + (size_t) c.env * c_skel.obs_dim + 19 * c_skel.state_index[b];
#pragma unroll
                for (int k = 0; k < 19; k++) o[k] = v[k];
            }
        }
        if (b == c_skel.root) {
            // one writer per env: reward / termination / counters / rollout bookkeeping
            int flags = L.flags & ~EVM_FLAG_PENDING;
            if ((MODE & 1) && !L.settling) {
                flags |= EVM_FLAG_POWERED;
                if (MODE & 4) GS(stat, 0) += 1;
            }
            if (MODE & 4) {
                if (L.settling) d.settle_left[c.env] = L.settle0 - 1;
                valid[c.env] = do_observe ? (L.settling ? 2 : 1) : 0;  // 1 = do_step transition, 2 = reset()'s own step
            }
            bool fin_next = false;
            if (do_observe) {
                observe_tail(c, reward, done);
                if ((MODE & 4) && done[c.env]) { flags |= EVM_FLAG_DONE; fin_next = true; }
            }
            d.flags[c.env] = flags;
            if ((MODE & 4) && __any(fin_next)) {
                // the next call starts this env's reset: draw its rotation ahead (read-only) for that call's setup kernels
                const M33 N = repose_rotation(mt_peek01(c, 0), mt_peek01(c, 1), mt_peek01(c, 2));
                if (fin_next) { SSC3(c_skel.sc_nexte, N.r0); SSC3(c_skel.sc_nexte + 3, N.r1); SSC3(c_skel.sc_nexte + 6, N.r2); }
            }
        }
#ifdef EVM_STAMPS5
        if (c.lane == 0) atomicMax(&c.d.stamps[(size_t) blockIdx.x * 16 + (b == c_skel.root ? 10 : b < c_skel.nm ? 9 : 8)], __builtin_amdgcn_s_memtime() - s5_p0);
#endif
    }
    if (any_observe)
        for (int mi = vw; mi < c_skel.nmus; mi += nvw) {
            if (full_tile) {
                float v[4] = {SC(c_skel.sc_mobs + 4 * mi + 0), SC(c_skel.sc_mobs + 4 * mi + 1), SC(c_skel.sc_mobs + 4 * mi + 2),
                              SC(c_skel.sc_mobs + 4 * mi + 3)};
                store_block_rows<4>(c, obs, 19 * c_skel.nm + 4 * mi, v, buf, who);
            } else if (do_observe) {
                observe_muscle(c, mi, obs);
            }
        }
}

__global__ __launch_bounds__(64) void k_env_repose(EnvDev d, const uint8_t *__restrict__ mask) {
    Ctx c = make_ctx(d, nullptr);
    if (c.env >= d.n_real) return;
    if (mask && !mask[c.env]) return;
    repose(c);
    d.flags[c.env] &= ~EVM_FLAG_DONE;
    d.settle_left[c.env] = 0;
}

// creation state: world transform = first_model_matrix (E = identity, reset-pending), MT19937 seeded
__global__ __launch_bounds__(64) void k_env_init(EnvDev d, uint64_t seed) {
    Ctx c = make_ctx(d, nullptr);
    if (c.env >= d.n) return;
    const size_t e = c.env;
    uint32_t x = (uint32_t) (seed + (uint64_t) c.env);
    GS(mt, 0) = x;
    for (int i = 1; i < 624; i++) {
        x = 1812433253u * (x ^ (x >> 30)) + (uint32_t) i;
        GS(mt, i) = x;
    }
    d.mt_idx[e] = 624;
    for (int b = 0; b < c_skel.nb; b++) {
        const EvmBodyC &BC = c_skel.body[b];
        const M33 M0 = load_m33(BC.m0);
        const S33 I = inertia_world(M0, load_f3(BC.inv_inertia));
        GS(iinv_stale, 6 * b) = I.xx; GS(iinv_stale, 6 * b + 1) = I.xy; GS(iinv_stale, 6 * b + 2) = I.xz;
        GS(iinv_stale, 6 * b + 3) = I.yy; GS(iinv_stale, 6 * b + 4) = I.yz; GS(iinv_stale, 6 * b + 5) = I.zz;
        S3(pos, 3 * b, load_f3(BC.t0));
        const Q4 q = quat_from_mat(M0);
        GS(quat, 4 * b) = q.x; GS(quat, 4 * b + 1) = q.y; GS(quat, 4 * b + 2) = q.z; GS(quat, 4 * b + 3) = q.w;
        S3(lin, 3 * b, f3(0.f, 0.f, 0.f));
        S3(ang, 3 * b, f3(0.f, 0.f, 0.f));
    }
    for (int m = 0; m < c_skel.nm; m++) {
        GS(mfn, m) = 0;
        S3(hist, 6 * m, f3(0.f, 0.f, 0.f)); S3(hist, 6 * m + 3, f3(0.f, 0.f, 0.f));
        SSC3(c_skel.sc_ms + 3 * m, load_f3(c_skel.body[m].t0));
        for (int k = 0; k < 36; k++) GS(mfp, m * 36 + k) = 0.f;
    }
    for (int mi = 0; mi < c_skel.nmus; mi++) {
        GS(target, mi) = 0.f;
        for (int k = 0; k < 4; k++) SC(c_skel.sc_mobs + 4 * mi + k) = 0.f;
    }
    for (int p = 0; p < c_skel.npair; p++) {
        c.t.pmn[(p << 6) + c.lane] = 0;
        for (int k = 0; k < EVM_PM_STRIDE; k++) c.t.pmp[((p * EVM_PM_STRIDE + k) << 6) + c.lane] = 0.f;
    }
    for (int k = 0; k < 9; k++) GS(E, k) = (k % 4 == 0) ? 1.f : 0.f;
    d.flags[e] = EVM_FLAG_PENDING;
    d.curr_step[e] = 0;
    d.remaining[e] = c_skel.init_remaining;
    d.settle_left[e] = 0;
    GS(diag, 0) = 0.f; GS(diag, 1) = 0.f;
}

__global__ __launch_bounds__(64) void k_env_poses(EnvDev d, float *out) {
    Ctx c = make_ctx(d, nullptr);
    if (c.env >= d.n_real) return;
    const bool pending = (d.flags[c.env] & EVM_FLAG_PENDING) != 0;
    M33 E = m33(f3(GS(E, 0), GS(E, 1), GS(E, 2)), f3(GS(E, 3), GS(E, 4), GS(E, 5)), f3(GS(E, 6), GS(E, 7), GS(E, 8)));
    for (int b = 0; b < c_skel.nb; b++) {
        float *o = out + ((size_t) c.env * c_skel.nb + b) * 7;
        const F3 p = G3(pos, 3 * b);
        Q4 q = q4(GS(quat, 4 * b), GS(quat, 4 * b + 1), GS(quat, 4 * b + 2), GS(quat, 4 * b + 3));
        if (pending) {
            q = quat_from_mat(glm_mul_basis(E, load_m33(c_skel.body[b].m0)));
        }
        o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = q.x; o[4] = q.y; o[5] = q.z; o[6] = q.w;
    }
}

}  // namespace evm
#include "sweep_groups.h"
namespace evm {
// Narrowphase + the setup kernel's records in ONE launch (member-vs-member mode, the default pipeline): the narrowphase is a few
// hundred latency-bound wavefronts that leave most of the chip idle for 80 us, and the joint / floor records of k_split_pre_b
// depend on nothing it produces.  Blocks [0, narrow) are narrow_block's, the rest are k_split_pre_b's virtual waves (one 64-lane
// block each, the long narrowphase blocks first in dispatch order).  The broadphase items run before, in k_split_pre_a.
template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(EVM_PAIRS_WAVES, EVM_PAIRS_WAVES))) void k_split_pairs_rec(EnvDev d, const float *__restrict__ action,
                                                                        const uint8_t *__restrict__ mask, int tiles, int nvw) {
    const int narrow = EVM_SPEC_SLOTS + EVM_URGENT_BLOCKS + EVM_BIG_BLOCKS + tiles * d.npair_host;
    if ((int) blockIdx.x < narrow) { narrow_block<MODE>(d, (int) blockIdx.x, tiles); return; }
    const int rb = (int) blockIdx.x - narrow, tile = rb / nvw, vw = rb - tile * nvw;
    const Ctx c = make_ctx_at(d, d.gtile + (size_t) tile * d.tile_floats, tile, (int) threadIdx.x, 0);
    if (c.env >= d.n_real) return;
    if (mask && !mask[c.env]) return;
    const LaneState L = lane_state<MODE>(c);
    pre_b_items<MODE>(c, L, vw, nvw, action, 0, tile);
}
}  // namespace evm
#ifdef EVM_ISA_PROBE
namespace evm {
// Diagnostic (tools/chain_isa.py): the row arithmetic of ONE hinge visit and of one link of the root's hinge chain as kernels of
// their own, so that their ISA can be read in isolation (instruction count and longest dependent path per hinge).
__global__ __launch_bounds__(64) void k_probe_hinge_rows(const f32x4 *__restrict__ rec, const float *__restrict__ body, float *__restrict__ out) {
    Blk42 k;
#pragma unroll
    for (int i = 0; i < EVM_H_STRIDE / 4; i++) k.q[i] = rec[(i << 6) + threadIdx.x];
    BodyPD Q;
    const float *b = body + threadIdx.x;
    Q.dl = f3p(p2(b[0], b[64]), p2(b[128], b[192]), p2(b[256], b[320]));
    Q.da = f3p(p2(b[384], b[448]), p2(b[512], b[576]), p2(b[640], b[704]));
    Q.I.xx = p2(b[768], b[832]); Q.I.xy = p2(b[896], b[960]); Q.I.xz = p2(b[1024], b[1088]);
    Q.I.yy = p2(b[1152], b[1216]); Q.I.yz = p2(b[1280], b[1344]); Q.I.zz = p2(b[1408], b[1472]);
    Q.im = p2(b[1536], b[1600]);
    float ap[6];
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("; PROBE_BEGIN hinge_rows" ::: "memory");
    const float res = hinge_rows(k, Q, ap);
    asm volatile("; PROBE_END hinge_rows" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    float *o = out + threadIdx.x;
    o[0] = Q.dl.x.x; o[64] = Q.dl.y.x; o[128] = Q.dl.z.x; o[192] = Q.da.x.x; o[256] = Q.da.y.x; o[320] = Q.da.z.x;
    o[384] = Q.dl.x.y; o[448] = Q.dl.y.y; o[512] = Q.dl.z.y; o[576] = Q.da.x.y; o[640] = Q.da.y.y; o[704] = Q.da.z.y;
#pragma unroll
    for (int r = 0; r < 6; r++) o[768 + 64 * r] = ap[r];
    o[1152] = res;
}
}  // namespace evm
#endif
namespace evm {

// ---------------------------------------------------------------------------------------------
// host-callable launchers (used by env_host.cpp)
// ---------------------------------------------------------------------------------------------
hipError_t upload_skeleton(const EvmSkelC *h, hipStream_t s) {
    return hipMemcpyToSymbolAsync(HIP_SYMBOL(c_skel), h, sizeof(EvmSkelC), 0, hipMemcpyHostToDevice, s);
}
size_t step_lds_bytes(int nb, int nscan) {  // tiles + version counters + hull-scan partials
    return (size_t) nb * 12 * 64 * sizeof(float) + (size_t) ((nb + 63) / 64) * 256 + (size_t) nscan * 2 * 64 * sizeof(float);
}

#define EVM_MAX_DEVICES 64
static int current_device() {
    int dev = 0;
    (void) hipGetDevice(&dev);
    return dev >= 0 && dev < EVM_MAX_DEVICES ? dev : 0;
}
template <int MODE>
static hipError_t launch_mode(const EnvDev &d, size_t lds, const float *action, float *obs, float *reward, uint8_t *done,
                              uint8_t *valid, const uint8_t *mask, hipStream_t s) {
    static bool attr_set[EVM_MAX_DEVICES] = {};  // the attribute is per device
    const int dev = current_device();
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_env_step<MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int) (160 * 1024));
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((k_env_step<MODE>), dim3(d.n / 64), dim3(64 * EVM_NW), lds, s, d, action, obs, reward, done, valid, mask);
    return hipGetLastError();
}
template <int MODE>
static hipError_t launch_split(const EnvDev &d, size_t lds, const float *action, float *obs, float *reward, uint8_t *done,
                               uint8_t *valid, const uint8_t *mask, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    static bool attr_set[EVM_MAX_DEVICES] = {};
    const int dev = current_device();
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_split_sweeps<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int) (160 * 1024));
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    const int tiles = d.n / 64;
    // enough (tile, part) workgroups to cover the chip a few times over at small batches, one part at large ones
    static int target_waves = 0;
    if (!target_waves) { const char *e = getenv("EVM_SPLIT_TARGET_WAVES"); target_waves = e ? atoi(e) : 4608; if (target_waves < 64) target_waves = 4608; }
    int parts = (target_waves + tiles * EVM_SPLIT_WAVES - 1) / (tiles * EVM_SPLIT_WAVES);
    if (parts < 1) parts = 1;
    if (parts > 32) parts = 32;
    const dim3 gp(tiles, parts), bp(64 * EVM_SPLIT_WAVES);
    hipLaunchKernelGGL((k_split_pre_a<MODE>), gp, bp, 0, s, d, mask);
    static int merge = -1;   // EVM_PAIRS_MERGE=0: the narrowphase and the records as two launches (A/B)
    if (merge < 0) { const char *e = getenv("EVM_PAIRS_MERGE"); merge = (e && e[0] == '0') ? 0 : 1; }
    if (d.pmn && merge) {
        const int nvw = parts * EVM_SPLIT_WAVES;
        hipLaunchKernelGGL((k_split_pairs_rec<MODE>), dim3(EVM_SPEC_SLOTS + EVM_URGENT_BLOCKS + EVM_BIG_BLOCKS + tiles * d.npair_host + tiles * nvw), dim3(64), 0, s, d, action, mask, tiles, nvw);
    } else {
        hipLaunchKernelGGL((k_split_pre_b<MODE>), gp, bp, 0, s, d, action, mask, 0);
        if (d.pmn) hipLaunchKernelGGL((k_split_pairs<MODE>), dim3(EVM_SPEC_SLOTS + EVM_URGENT_BLOCKS + EVM_BIG_BLOCKS + tiles * d.npair_host), dim3(64), 0, s, d, mask, tiles);
    }
    if (e0) (void) hipEventRecord(e0, s);
    bool fused_post = false;
    if (d.gs) {
        static bool attr_g[EVM_MAX_DEVICES] = {};
        if (!attr_g[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sweeps_g),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int) (160 * 1024));
            if (e != hipSuccess) return e;
            attr_g[dev] = true;
        }
        static int fuse = -1;   // EVM_FUSE_POST=0: integration / observation as a launch of its own (A/B)
        if (fuse < 0) { const char *e = getenv("EVM_FUSE_POST"); fuse = (e && e[0] == '0') ? 0 : 1; }
        fused_post = fuse != 0;
        PostArgs pa;
        pa.obs = obs; pa.reward = reward; pa.done = done; pa.valid = valid; pa.mode = fused_post ? MODE : -1;
        hipLaunchKernelGGL(k_sweeps_g, dim3(tiles * (64 / EVM_G_ENVS)), dim3(64 * d.g_waves), (size_t) d.g_lds, s, d, mask, (MODE & 4) ? 1 : 0, pa);
    } else {
        if (d.gtile_only) hipLaunchKernelGGL(k_split_sweeps<true>, dim3(tiles), dim3(64 * EVM_NW), 0, s, d, mask, (MODE & 4) ? 1 : 0);
        else hipLaunchKernelGGL(k_split_sweeps<false>, dim3(tiles), dim3(64 * EVM_NW), lds, s, d, mask, (MODE & 4) ? 1 : 0);
    }
    if (e1) (void) hipEventRecord(e1, s);
    if (!fused_post) hipLaunchKernelGGL((k_split_post<MODE>), gp, bp, 0, s, d, obs, reward, done, valid, mask);
    return hipGetLastError();
}
hipError_t launch_step(const EnvDev &d, size_t lds_bytes, int split, int mode, const float *action, float *obs, float *reward,
                       uint8_t *done, uint8_t *valid, const uint8_t *mask, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    // split < 0: choose by batch.  Up to 128 tiles (8192 envs) the monolithic kernel leaves most of the chip idle and the
    // split pipeline wins (0.179 vs 0.227 ms at 64 tiles); with every CU holding a tile it only adds launches and staging
    // traffic (0.472 vs 0.419 ms at 256 tiles).
    if (split < 0) split = d.n / 64 <= 128 ? 1 : 0;
    if (d.pmn) split = 1;  // member-vs-member contacts live in the split pipeline with the lane-group sweeps kernel only
    if (d.gtile_only) split = 1;  // a tile beyond the LDS: the pipeline's tile kernel on the global staging copy
    if (split) {
        // the sweeps kernel keeps only the body tiles and the version counters in LDS (no scan minima)
        const size_t lds = lds_bytes;  // same layout as the staging copy
        switch (mode) {
            case 0: return launch_split<0>(d, lds, action, obs, reward, done, valid, mask, s, e0, e1);
            case 2: return launch_split<2>(d, lds, action, obs, reward, done, valid, mask, s, e0, e1);
            case 3: return launch_split<3>(d, lds, action, obs, reward, done, valid, mask, s, e0, e1);
            case 7: return launch_split<7>(d, lds, action, obs, reward, done, valid, mask, s, e0, e1);
            default: return hipErrorInvalidValue;
        }
    }
    switch (mode) {
        case 0: return launch_mode<0>(d, lds_bytes, action, obs, reward, done, valid, mask, s);
        case 2: return launch_mode<2>(d, lds_bytes, action, obs, reward, done, valid, mask, s);
        case 3: return launch_mode<3>(d, lds_bytes, action, obs, reward, done, valid, mask, s);
        case 7: return launch_mode<7>(d, lds_bytes, action, obs, reward, done, valid, mask, s);
        default: return hipErrorInvalidValue;
    }
}
hipError_t launch_repose(const EnvDev &d, const uint8_t *mask, hipStream_t s) {
    hipLaunchKernelGGL(k_env_repose, dim3(d.n / 64), dim3(64), 0, s, d, mask);
    return hipGetLastError();
}
hipError_t launch_init(const EnvDev &d, uint64_t seed, hipStream_t s) {
    hipLaunchKernelGGL(k_env_init, dim3(d.n / 64), dim3(64), 0, s, d, seed);
    return hipGetLastError();
}
hipError_t launch_poses(const EnvDev &d, float *out, hipStream_t s) {
    hipLaunchKernelGGL(k_env_poses, dim3(d.n / 64), dim3(64), 0, s, d, out);
    return hipGetLastError();
}

}  // namespace evm
