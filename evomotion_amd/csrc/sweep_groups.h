// Lane-group Gauss-Seidel sweeps for the split pipeline (included by env_kernels.hip: same translation unit, so it shares
// c_skel and the row arithmetic of the 64-env tile kernels).
//
// The 64-env tile kernel (k_split_sweeps) gives one environment to every lane and one constraint to every wave: at the
// mandated 4096 envs per GPU that is 64 workgroups on 256 CUs, and every sweep re-reads each constraint record from global
// memory.  Here a wavefront is 4 lane groups x 16 environments:
//   * a workgroup owns a QUARTER tile (16 envs): 4096 envs = 256 workgroups, one per CU;
//   * the lane groups of a wave work on four DIFFERENT constraints of one type (the spider's four legs), so one instruction
//     stream solves four constraints; the host packs Bullet's visit order into such "group entries" (skel_const.h,
//     EvmGSchedC; skeleton_host.cpp, build_group_schedule) without changing any body's visit order;
//   * with 16 envs per workgroup everything the sweeps touch fits the CU's 160 KiB LDS: per body 3 quads (solver deltas +
//     world inverse inertia), and the whole joint-record image (hinge 9, fixed 11, slider 11, p2p 4 quads) — read from
//     global memory ONCE per step instead of once per sweep; only the contact records of touching members stay in global
//     memory (prefetched one entry ahead, as before);
//   * an LDS quad is [quad][16 lanes] x 16 B: the 16 lanes of a group read 256 contiguous bytes, whatever record or body
//     the group works on, which is conflict-free for ds_read_b128's 16-lane phases.
// Waves of a workgroup (1..4, EvmGSchedC::nwaves) synchronise through the per-body version counters of the tile kernel.
#pragma once

namespace evm {

struct GCtx {
    f32x4 *q;    // LDS image, joint records: quad row qi of this lane's env = q[(qi << 4) + e]
    f32x4 *qb;   // LDS image, bodies (3 quad rows per body): row qi = qb[qi * 17 + e].  The one-quad pad per row shifts the banks
                 // by four words per row: the joint rows (16 lanes = 16 envs, one row) and the contact rounds (16 lanes = a few envs,
                 // different bodies) are both conflict-free on ds_read_b128
    int e, g;    // env within the quarter tile, lane group
    int QR;      // first quad of the joint-record image
    int *ver;    // per-body version counters
    bool multi;  // more than one wave per workgroup: versions are live
#ifdef EVM_GSTAMPS2
    unsigned long long wait_cycles;
#endif
#ifdef EVM_GSTAMPS3
    unsigned long long *st3;  // d.stamps: [0] loads, [1] rows, [2] hand-over, [3] stores (cycle sums over all hinge chain entries), [4] entries, [5] phases
#endif
};
#define GQ(G, qi) ((G).q[((qi) << 4) + (G).e])
#define GB(G, qi) ((G).qb[(qi) * 17 + (G).e])

#ifdef EVM_GSTAMPS2
__device__ unsigned long long g_wait_cycles_dummy;
#define GWAIT_T0 const unsigned long long gw_t0 = __builtin_amdgcn_s_memtime();
#define GWAIT_T1 G.wait_cycles += __builtin_amdgcn_s_memtime() - gw_t0;
#else
#define GWAIT_T0
#define GWAIT_T1
#endif
DEV void g_wait2(GCtx &G, const Ctx &c, int a, int expA, int b, int expB) {
    if (!G.multi) return;
    GWAIT_T0
    int spins = 0;
    for (;;) {
        const int va = __hip_atomic_load(&G.ver[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int vb = __hip_atomic_load(&G.ver[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (va >= expA && vb >= expB) break;
        if (++spins > EVM_SPIN_HOT) __builtin_amdgcn_s_sleep(1);
        // a schedule bug must not hang the GPU: give up and stop waiting for the rest of the launch; the batch residual is
        // poisoned with +inf (sticky: evm_env_get_residual returns it until cleared; the tests and the soak assert on it)
        if (spins > (1 << 20)) { atomicMax(c.d.resid, 0x7f800000); atomicAdd(&c.d.errs[0], 1); G.multi = false; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    GWAIT_T1
}
DEV void g_publish(const GCtx &G, int a, int va, int b, int vb) {
    if (G.ver == nullptr) return;
    __hip_atomic_store(&G.ver[a], va, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (b != a) __hip_atomic_store(&G.ver[b], vb, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

DEV BodyPD g_load_pair(const GCtx &G, int a, int b, float imA, float imB) {
    const f32x4 a0 = GB(G, 3 * a), a1 = GB(G, 3 * a + 1), a2 = GB(G, 3 * a + 2);
    const f32x4 b0 = GB(G, 3 * b), b1 = GB(G, 3 * b + 1), b2 = GB(G, 3 * b + 2);
    BodyPD k;
    k.dl = f3p(p2(a0[0], b0[0]), p2(a0[1], b0[1]), p2(a0[2], b0[2]));
    k.da = f3p(p2(a0[3], b0[3]), p2(a1[0], b1[0]), p2(a1[1], b1[1]));
    k.I.xx = p2(a1[2], b1[2]); k.I.xy = p2(a1[3], b1[3]); k.I.xz = p2(a2[0], b2[0]);
    k.I.yy = p2(a2[1], b2[1]); k.I.yz = p2(a2[2], b2[2]); k.I.zz = p2(a2[3], b2[3]);
    k.im = p2(imA, -imB);
    return k;
}
DEV void g_store_pair(const GCtx &G, int a, int b, const BodyPD &Q) {
    f32x4 x;
    x[0] = Q.dl.x.x; x[1] = Q.dl.y.x; x[2] = Q.dl.z.x; x[3] = Q.da.x.x; GB(G, 3 * a) = x;
    x[0] = Q.da.y.x; x[1] = Q.da.z.x; x[2] = Q.I.xx.x; x[3] = Q.I.xy.x; GB(G, 3 * a + 1) = x;
    x[0] = Q.dl.x.y; x[1] = Q.dl.y.y; x[2] = Q.dl.z.y; x[3] = Q.da.x.y; GB(G, 3 * b) = x;
    x[0] = Q.da.y.y; x[1] = Q.da.z.y; x[2] = Q.I.xx.y; x[3] = Q.I.xy.y; GB(G, 3 * b + 1) = x;
}
DEV BodyD g_load_body(const GCtx &G, int b, float im) {
    const f32x4 b0 = GB(G, 3 * b), b1 = GB(G, 3 * b + 1), b2 = GB(G, 3 * b + 2);
    BodyD k;
    k.dl = f3(b0[0], b0[1], b0[2]);
    k.da = f3(b0[3], b1[0], b1[1]);
    k.I.xx = b1[2]; k.I.xy = b1[3]; k.I.xz = b2[0]; k.I.yy = b2[1]; k.I.yz = b2[2]; k.I.zz = b2[3];
    k.im = im;
    return k;
}
DEV void g_store_body(const GCtx &G, int b, const BodyD &k) {
    f32x4 x;
    x[0] = k.dl.x; x[1] = k.dl.y; x[2] = k.dl.z; x[3] = k.da.x; GB(G, 3 * b) = x;
    x[0] = k.da.y; x[1] = k.da.z; x[2] = k.I.xx; x[3] = k.I.xy; GB(G, 3 * b + 1) = x;
}

DEV void g_store_applied6(const GCtx &G, int quad, const float (&ap)[6], float t2, float t3) {
    f32x4 x;
    x[0] = ap[0]; x[1] = ap[1]; x[2] = ap[2]; x[3] = ap[3]; GQ(G, quad) = x;
    x[0] = ap[4]; x[1] = ap[5]; x[2] = t2; x[3] = t3; GQ(G, quad + 1) = x;
}

// One group entry of each type: the record and the two bodies are requested together (one LDS round trip), then the rows
// run on registers, then the bodies' deltas and the accumulated impulses go back.
DEV float g_hinge(const GCtx &G, int rec, int a, int b, float imA, float imB) {
    Blk42 k;
    const f32x4 *p = &GQ(G, G.QR + rec);
#pragma unroll
    for (int i = 0; i < EVM_H_STRIDE / 4; i++) k.q[i] = p[i << 4];
    BodyPD Q = g_load_pair(G, a, b, imA, imB);
    float ap[6];
    const float res = hinge_rows(k, Q, ap);
    g_store_pair(G, a, b, Q);
    g_store_applied6(G, G.QR + rec + 7, ap, KV(k, 34), KV(k, 35));  // fields 28..33 applied, 34 lo, 35 hi
    return res;
}
DEV float g_fixed(const GCtx &G, int rec, int a, int b, float imA, float imB) {
    Blk42 k;
    const f32x4 *p = &GQ(G, G.QR + rec);
#pragma unroll
    for (int i = 0; i < EVM_F_STRIDE / 4; i++) k.q[i] = p[i << 4];
    BodyPD Q = g_load_pair(G, a, b, imA, imB);
    float ap[6];
    const float res = fixed_rows(k, Q, ap);
    g_store_pair(G, a, b, Q);
    g_store_applied6(G, G.QR + rec + 9, ap, 0.f, 0.f);  // fields 36..41
    return res;
}
template <bool ISO>
DEV float g_slider(const GCtx &G, int rec, int a, int b, float imA, float imB) {
    Blk42 k;
    const f32x4 *p = &GQ(G, G.QR + rec);
#pragma unroll
    for (int i = 0; i < EVM_S_STRIDE / 4; i++) k.q[i] = p[i << 4];
    BodyPD Q = g_load_pair(G, a, b, imA, imB);
    float ap[6];
    const float res = slider_rows<ISO>(k, Q, ap);
    g_store_pair(G, a, b, Q);
    g_store_applied6(G, G.QR + rec + 9, ap, 0.f, 0.f);  // fields 36..41
    return res;
}
// a = the member, b = the attach sphere (pivot at its origin: only its linear delta takes part)
DEV float g_p2p(const GCtx &G, int rec, int a, int b, float imA, float imB) {
    Blk16 kk;
    const f32x4 *p = &GQ(G, G.QR + rec);
#pragma unroll
    for (int i = 0; i < 4; i++) kk.q[i] = p[i << 4];
    BodyD A = g_load_body(G, a, imA);
    const f32x4 s0 = GB(G, 3 * b);
    F3 dlS = f3(s0[0], s0[1], s0[2]);
    float ap0, ap1, ap2;
    const float res = p2p_rows(kk, A, dlS, imB, ap0, ap1, ap2);
    g_store_body(G, a, A);
    f32x4 x;
    x[0] = dlS.x; x[1] = dlS.y; x[2] = dlS.z; x[3] = s0[3]; GB(G, 3 * b) = x;
    x[0] = ap0; x[1] = ap1; x[2] = ap2; x[3] = 0.f; GQ(G, G.QR + rec + 3) = x;
    return res;
}
// Chain entries.  The visits of a chain entry are CONSECUTIVE visits on one shared body a (the spider's root: its four
// hinges, then the four p2p constraints of the muscles attached to it), each with its own second body: they cannot run side
// by side, but they can share one entry — one descriptor, one LDS round trip — and run as phases, lane group g in phase g
// (the other groups masked off), with the shared body's deltas handed from group to group through the cross-lane network
// instead of an LDS store / version hop / reload per visit.  nact = filled slots (slots 0..nact-1, in Bullet's order).
DEV float g_bcast(float v, int src_lane) { return __shfl(v, src_lane); }
DEV float g_hinge_chain(const GCtx &G, int rec, int a, int b, float imA, float imB, int nact) {
    Blk42 k;
    BodyPD Q;
#ifdef EVM_GSTAMPS3  // (tools/gstamps3.py) where the cycles of a chain entry go; every stamp waits for the outstanding LDS traffic first
#define GS3_STAMP(var) __builtin_amdgcn_s_waitcnt(0xC07F); const unsigned long long var = __builtin_amdgcn_s_memtime();
    GS3_STAMP(s3_a)
    unsigned long long s3_rows = 0, s3_hand = 0;
#endif
    if (rec >= 0) {
        const f32x4 *p = &GQ(G, G.QR + rec);
#pragma unroll
        for (int i = 0; i < EVM_H_STRIDE / 4; i++) k.q[i] = p[i << 4];
        Q = g_load_pair(G, a, b, imA, imB);  // every group reads the shared body; only group 0's copy is current in phase 0
    }
    float ap[6];
    float res = 0.f;
#ifdef EVM_GSTAMPS3
    GS3_STAMP(s3_b)
#endif
    for (int ph = 0; ph < nact; ph++) {
#ifdef EVM_GSTAMPS3
        GS3_STAMP(s3_p0)
#endif
        if (G.g == ph) res = hinge_rows(k, Q, ap);
#ifdef EVM_GSTAMPS3
        GS3_STAMP(s3_p1)
#endif
        const int src = (ph << 4) + G.e;  // the shared body's deltas after this phase, to every group
        Q.dl.x.x = g_bcast(Q.dl.x.x, src); Q.dl.y.x = g_bcast(Q.dl.y.x, src); Q.dl.z.x = g_bcast(Q.dl.z.x, src);
        Q.da.x.x = g_bcast(Q.da.x.x, src); Q.da.y.x = g_bcast(Q.da.y.x, src); Q.da.z.x = g_bcast(Q.da.z.x, src);
#ifdef EVM_GSTAMPS3
        GS3_STAMP(s3_p2)
        s3_rows += s3_p1 - s3_p0; s3_hand += s3_p2 - s3_p1;
#endif
    }
#ifdef EVM_GSTAMPS3
    GS3_STAMP(s3_c)
#endif
    if (rec >= 0) {
        f32x4 x;  // the second body and the impulses of this group's visit; the shared body once (group 0 holds the final values too)
        x[0] = Q.dl.x.y; x[1] = Q.dl.y.y; x[2] = Q.dl.z.y; x[3] = Q.da.x.y; GB(G, 3 * b) = x;
        x[0] = Q.da.y.y; x[1] = Q.da.z.y; x[2] = Q.I.xx.y; x[3] = Q.I.xy.y; GB(G, 3 * b + 1) = x;
        g_store_applied6(G, G.QR + rec + 7, ap, KV(k, 34), KV(k, 35));
        if (G.g == 0) {
            x[0] = Q.dl.x.x; x[1] = Q.dl.y.x; x[2] = Q.dl.z.x; x[3] = Q.da.x.x; GB(G, 3 * a) = x;
            x[0] = Q.da.y.x; x[1] = Q.da.z.x; x[2] = Q.I.xx.x; x[3] = Q.I.xy.x; GB(G, 3 * a + 1) = x;
        }
    }
#ifdef EVM_GSTAMPS3
    GS3_STAMP(s3_d)
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&G.st3[0], s3_b - s3_a); atomicAdd(&G.st3[1], s3_rows); atomicAdd(&G.st3[2], s3_hand); atomicAdd(&G.st3[3], s3_d - s3_c);
        atomicAdd(&G.st3[4], 1ull); atomicAdd(&G.st3[5], (unsigned long long) nact);
    }
#endif
    return res;
}
// a = the shared member, b = this group's attach sphere
DEV float g_p2p_chain(const GCtx &G, int rec, int a, int b, float imA, float imB, int nact) {
    Blk16 kk;
    BodyD A;
    f32x4 s0;
    if (rec >= 0) {
        const f32x4 *p = &GQ(G, G.QR + rec);
#pragma unroll
        for (int i = 0; i < 4; i++) kk.q[i] = p[i << 4];
        A = g_load_body(G, a, imA);
        s0 = GB(G, 3 * b);
    }
    F3 dlS = f3(s0[0], s0[1], s0[2]);
    float ap0 = 0.f, ap1 = 0.f, ap2 = 0.f;
    float res = 0.f;
    for (int ph = 0; ph < nact; ph++) {
        if (G.g == ph) res = p2p_rows(kk, A, dlS, imB, ap0, ap1, ap2);
        const int src = (ph << 4) + G.e;
        A.dl.x = g_bcast(A.dl.x, src); A.dl.y = g_bcast(A.dl.y, src); A.dl.z = g_bcast(A.dl.z, src);
        A.da.x = g_bcast(A.da.x, src); A.da.y = g_bcast(A.da.y, src); A.da.z = g_bcast(A.da.z, src);
    }
    if (rec >= 0) {
        f32x4 x;
        x[0] = dlS.x; x[1] = dlS.y; x[2] = dlS.z; x[3] = s0[3]; GB(G, 3 * b) = x;
        x[0] = ap0; x[1] = ap1; x[2] = ap2; x[3] = 0.f; GQ(G, G.QR + rec + 3) = x;
        if (G.g == 0) g_store_body(G, a, A);
    }
    return res;
}
// k = the member's contact record (global memory, requested an entry ahead); rec = its first scratch slot
DEV float g_contact(const GCtx &G, const Ctx &c, const Blk42 &k, int rec, int m, float im, float mu) {
    BodyD D = g_load_body(G, m, im);
    float w[EVM_CM_STRIDE];
    const float res = contact_rows(k, D, mu, w);
    g_store_body(G, m, D);
    rec_store<10, 12>(c, rec, w);
    return res;
}

}  // namespace evm
#include "contact_rounds.h"
namespace evm {

DEV Ctx make_ctx_at(const EnvDev &d, float *lds, int tile64, int lane64, int wave) {
    Ctx c;
    c.d = d;
    c.t = d;
    c.lane = lane64;
    c.wave = wave;
    c.env = tile64 * 64 + lane64;
    c.lds = lds;
    const size_t tile = (size_t) tile64 * 64;
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

__global__ __launch_bounds__(64 * EVM_G_MAX_WAVES) void k_sweeps_g(EnvDev d, const uint8_t *__restrict__ mask, int autoreset, PostArgs pa) {
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    const EvmGSchedC *__restrict__ gs = d.gs;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    // block -> (64-env tile, quarter): the four quarters of a tile read the same 256-byte lines of the [slot][64 lanes]
    // arrays, so they go to the same XCD (blocks are dealt to the 8 XCDs round robin) when the tile count allows
    const int tiles = d.n >> 6;
    int tile64, sub;
    if ((tiles & 7) == 0) { const int x = blockIdx.x & 7, y = blockIdx.x >> 3; tile64 = x + 8 * (y >> 2); sub = y & 3; }
    else { tile64 = blockIdx.x >> 2; sub = blockIdx.x & 3; }
    GCtx G;
    G.qb = reinterpret_cast<f32x4 *>(lds_dyn);
    G.q = G.qb + 3 * c_skel.nb;  // record row r (r >= QR = 3 nb) at qb[3 nb * 17 + (r - QR) * 16 + e] = q[(r << 4) + e]
    G.e = lane & (EVM_G_ENVS - 1);
    G.g = lane >> 4;
    G.multi = nw > 1;  // (cleared by a wait that times out)
#ifdef EVM_GSTAMPS2
    G.wait_cycles = 0;
#endif
#ifdef EVM_GSTAMPS3
    G.st3 = d.stamps;
#endif
    const int nb = c_skel.nb, nm = c_skel.nm;
    const int nrq = gs->nrq, total = gs->total;
    G.QR = 3 * nb;
    f32x4 *tbl = G.q + (size_t) (G.QR + nrq) * EVM_G_ENVS;  // [total][4 slots][2 quads]
    f32x4 *hdr = tbl + (size_t) total * (2 * EVM_G_SLOTS);   // [total]
    int *ver_base = reinterpret_cast<int *>(hdr + total);
    G.ver = nw > 1 ? ver_base : nullptr;
    int *resmax = ver_base + ((nb + 3) & ~3);                    // [16] per-env residual of the last sweep (float bits, >= 0)
    // ---- schedule table -> LDS (every thread, before any lane leaves) ----
    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(&gs->slot[0][0]);
        for (int i = threadIdx.x; i < total * 2 * EVM_G_SLOTS; i += blockDim.x) tbl[i] = src[i];
        const f32x4 *hs = reinterpret_cast<const f32x4 *>(&gs->entry[0]);
        for (int i = threadIdx.x; i < total; i += blockDim.x) hdr[i] = hs[i];
        for (int i = threadIdx.x; i < ((nb + 3) & ~3) + EVM_G_ENVS; i += blockDim.x) ver_base[i] = 0;
        if (gs->with_contacts == 0) {  // member-vs-member mode: empty contact program, workgroup-wide words, inverse masses
            unsigned *prog0 = reinterpret_cast<unsigned *>(resmax + EVM_G_ENVS);
            for (int i = threadIdx.x; i < 2 * 16 * EVM_G_ENVS; i += blockDim.x) prog0[i] = 0xffffffffu;
            int *meta0 = reinterpret_cast<int *>(prog0 + 2 * 16 * EVM_G_ENVS);
            if (threadIdx.x < 16) meta0[threadIdx.x] = 0;
            float *imt0 = reinterpret_cast<float *>(meta0 + 16);
            for (int i = threadIdx.x; i < nb; i += blockDim.x) imt0[i] = c_skel.body[i].inv_mass;
        }
    }
#if defined(EVM_GSTAMPS) || defined(EVM_GSTAMPS2)  // diagnostic builds (tools/gstamps.py)
    const unsigned long long gs_t0 = __builtin_amdgcn_s_memtime();
#endif
#ifdef EVM_GSTAMPS  // cycles per phase and per entry type, wave 0 of quarter 0
    unsigned long long gs_type[5] = {0, 0, 0, 0, 0};
    unsigned gs_n[5] = {0, 0, 0, 0, 0};
#endif
    Ctx c = make_ctx_at(d, nullptr, tile64, sub * EVM_G_ENVS + G.e, wave);
    // Lanes outside the batch / the mask.  Floor-contacts mode: they drop out; every wave of the workgroup owns the same 16
    // envs, so either every wave keeps a live lane (and meets the others at the barriers) or the whole workgroup leaves here.
    // Member-vs-member mode regroups the lanes by env for the contact phases (below), so there every lane stays: a dead
    // lane computes on whatever its (allocated, initialised) env slot holds and writes nothing but LDS columns of dead envs;
    // all global stores are guarded by `live` / `clive`.
    const bool scm = gs->with_contacts == 0;  // (wave-uniform)
    const bool live = c.env < d.n_real && (!mask || mask[c.env]);
    if (!__any(live)) return;   // (the same decision in every wave of the workgroup)
    if (!scm && !live) return;
#ifdef EVM_GSTAMPS
    const unsigned long long gs_ta = __builtin_amdgcn_s_memtime();
#endif
    const int flags_in = d.flags[c.env];
    const bool fin = live && autoreset && (flags_in & EVM_FLAG_DONE) != 0;  // a reset starts with this step (see LaneState)
    const bool any_pending = __any(live && ((flags_in & EVM_FLAG_PENDING) != 0 || fin));
    // (member-vs-member mode) the contact phases' lane -> (env, slot) map and the env's manifold bookkeeping, requested now so
    // that the loads fly while the record image is copied
    CBank K0;
    K0.id = -1; K0.round = -1;
    int nrounds = 0;
    unsigned *prog = reinterpret_cast<unsigned *>(resmax + EVM_G_ENVS);  // [2 banks][16 slots][16 envs]
    int *meta = reinterpret_cast<int *>(prog + 2 * 16 * EVM_G_ENVS);      // per wave: rounds, second bank in use, split impulse needed, -
    float *imt = reinterpret_cast<float *>(meta + 16);                    // inverse mass per body
    unsigned w1 = 0xffffffffu;  // this slot's manifold of the overflow bank (served from global memory)
    bool use_b1 = false;
    const int epw = EVM_G_ENVS / nw, nslots = 64 / epw;
    const int ce = wave * epw + lane % epw, cs = lane / epw;
    GCtx GC = G;
    GC.e = ce;
    const Ctx cc = make_ctx_at(d, nullptr, tile64, sub * EVM_G_ENVS + ce, wave);
    const bool clive = cc.env < d.n_real && (!mask || mask[cc.env]);
    // (the same three for the env this lane's MANIFOLD belongs to: its own env, or — a guest slot — another env of the wave)
    Ctx ck = cc;
    int ce_k = ce;
    bool clive_k = clive;
    int nn_c[EVM_MAX_MEMBERS];
    unsigned pw_c[EVM_PACT_WORDS];
    unsigned pflags_c = 0u;
    if (scm) {
        const int nwords_c = (c_skel.npair + 31) >> 5;
#pragma unroll
        for (int m = 0; m < EVM_MAX_MEMBERS; m++) nn_c[m] = m < nm ? cc.t.mfn[(m << 6) + cc.lane] : 0;
#pragma unroll
        for (int k = 0; k < EVM_PACT_WORDS; k++) pw_c[k] = k < nwords_c ? cc.t.pact[(k << 6) + cc.lane] : 0u;
        pflags_c = cc.t.pact[(nwords_c << 6) + cc.lane];
    }
    // ---- joint-record image: global scratch [quad][64 lanes] -> LDS [quad][16 lanes], by LDS-DMA (global_load_lds_dwordx4:
    // no register round trip, so every request of the wave is in flight at once).  One instruction moves four consecutive
    // quads: lane (g, e) fetches quad ib + g of its env, and the hardware writes lane l's 16 bytes at base + 16 l — which is
    // exactly quad ib + g, env e of the image.  (The copy is a 25 MB burst over the whole chip: bandwidth, not latency.)
    {
        const f32x4 *grec = reinterpret_cast<const f32x4 *>(c.t.scratch + ((size_t) c_skel.sc_h << 6)) + c.lane;
        for (int ib = wave * EVM_G_SLOTS; ib < nrq; ib += EVM_G_SLOTS * nw) {
            const int i = ib + G.g;
            if (i < nrq)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (grec + ((size_t) i << 6)),
                                                 (__attribute__((address_space(3))) void *) (G.q + ((size_t) (G.QR + ib) << 4)), 16, 0, 0);
        }
    }
#ifdef EVM_GSTAMPS
    const unsigned long long gs_tb = __builtin_amdgcn_s_memtime();
#endif
    // ---- bodies: staging copy of the tile ([slot][64 lanes]: deltas with the contact warm start, world inverse inertia) ----
    {
        // An attach sphere (isotropic, no contacts) starts every step with zero deltas and a world inverse inertia of
        // k * identity; only a step that follows a reset runs on the (stale) tensor the setup kernel staged.
        const float *gt = d.gtile + (size_t) tile64 * d.tile_floats + c.lane;
        const int nb6 = nb * 6;
        const int step = EVM_G_SLOTS * nw;
        auto body_read = [&](int b, float (&v)[12]) {
            const bool light = b >= nm && c_skel.body[b].isotropic;
            if (!light) {
#pragma unroll
                for (int k = 0; k < 6; k++) v[k] = gt[(b * 6 + k) << 6];
            } else {
#pragma unroll
                for (int k = 0; k < 6; k++) v[k] = 0.f;
            }
            if (!light || any_pending) {
#pragma unroll
                for (int k = 0; k < 6; k++) v[6 + k] = gt[(nb6 + b * 6 + k) << 6];
            } else {
                const float kI = c_skel.body[b].inv_inertia[0];
                v[6] = kI; v[7] = 0.f; v[8] = 0.f; v[9] = kI; v[10] = 0.f; v[11] = kI;
            }
        };
        auto body_put = [&](int b, const float (&v)[12]) {
            f32x4 x;
            x[0] = v[0]; x[1] = v[1]; x[2] = v[2]; x[3] = v[3]; GB(G, 3 * b) = x;
            x[0] = v[4]; x[1] = v[5]; x[2] = v[6]; x[3] = v[7]; GB(G, 3 * b + 1) = x;
            x[0] = v[8]; x[1] = v[9]; x[2] = v[10]; x[3] = v[11]; GB(G, 3 * b + 2) = x;
        };
        int b = wave * EVM_G_SLOTS + G.g;
        for (; b + 2 * step < nb; b += 3 * step) {  // three bodies' reads in flight per lane
            float v0[12], v1[12], v2[12];
            body_read(b, v0); body_read(b + step, v1); body_read(b + 2 * step, v2);
            body_put(b, v0); body_put(b + step, v1); body_put(b + 2 * step, v2);
        }
        for (; b < nb; b += step) {
            float v0[12];
            body_read(b, v0);
            body_put(b, v0);
        }
    }
#ifdef EVM_GSTAMPS
    const unsigned long long gs_tc = __builtin_amdgcn_s_memtime();
#endif
    int ncontact = 0;
    unsigned cmask = 0;  // wave-uniform: members with a cached point in any of the 16 envs
    {
        int nn[EVM_MAX_MEMBERS];  // every count requested before the first vote waits for one
#pragma unroll
        for (int m = 0; m < EVM_MAX_MEMBERS; m++) nn[m] = m < nm ? GS(mfn, m) : 0;
#pragma unroll
        for (int m = 0; m < EVM_MAX_MEMBERS; m++) {
            ncontact += nn[m];
            if (__any(nn[m] > 0)) cmask |= 1u << m;
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the LDS-DMA requests of the record image have landed
    __syncthreads();

    // ---- member-vs-member mode: contact program, the owners' records into registers, split-impulse recovery, warm start ----
    // For the contact phases the lanes of a wave regroup: wave w serves the envs [w epw, (w + 1) epw) of the workgroup (epw =
    // 16 / waves), lane -> (env ce = w epw + lane % epw, slot cs = lane / epw).  All slots of an env sit in ONE wave, so the
    // rounds of the contact rows need no workgroup barrier between them — only the switches between joint rows and contact
    // rows do.  (LDS image: a row's 16 lanes now hit epw env columns, a 64 / (4 epw)... -way bank conflict on the b128 reads
    // of the bodies; cheaper than a barrier per round.)
    float *const crec_lane = cc.t.crec ? cc.t.crec + 4 * cc.lane : nullptr;
    float *ptl = imt + ((nb + 3) & ~3);  // [6 nm][16 envs] push / turn velocities of the members during the split-impulse phase
    auto slow = [&](int phase, int r) -> float { return g_slow_visit(phase, w1, r, G.qb, ce, imt, crec_lane, ptl); };
#ifdef EVM_GSTAMPS
    unsigned long long gs_c0 = __builtin_amdgcn_s_memtime(), gs_c1 = gs_c0, gs_c2 = gs_c0, gs_c3 = gs_c0, gs_cb = gs_c0;
#endif
    if (scm) {  // (program, words and inverse masses were initialised by every thread before the barrier above)
        {
            // every lane of the wave walks the ids of its env (the slots of an env redundantly: same values, same stores)
            const int left_out = g_build_program(GC, nslots, epw, nn_c, pw_c, pflags_c, prog, meta + 4 * wave);
            if (left_out > 0 && cs == 0 && clive) { atomicMax(d.resid, 0x7f800000); atomicAdd(&d.errs[1], left_out); }  // sticky
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef EVM_GSTAMPS
        __builtin_amdgcn_s_waitcnt(0);
        gs_cb = __builtin_amdgcn_s_memtime();
#endif
        nrounds = __builtin_amdgcn_readfirstlane(meta[4 * wave]);
        use_b1 = __builtin_amdgcn_readfirstlane(meta[4 * wave + 1]) != 0;
        if (use_b1) {
            // An env with more than 16 live manifolds (one or two of 4096 in most steps): its 17th.. manifold moves into a FREE
            // slot of another env of this wavefront (a typical env uses 7 of its 16) as a guest — the lane then works on the
            // guest's env column — instead of the overflow bank, whose every visit re-reads its record from global memory and
            // made that one wavefront, hence the whole launch, wait.  One lane rewrites the wave's program words; the overflow
            // bank stays for what finds no free slot.
            if (lane == 0) {
                int left = 0;
                for (int ea = 0; ea < epw; ea++)
                    for (int sa = 0; sa < nslots; sa++) {
                        const unsigned w = prog[((16 + sa) << 4) + wave * epw + ea];
                        if (w == 0xffffffffu) break;  // (an env's overflow words are contiguous from slot 0)
                        bool placed = false;
                        for (int eb = 0; eb < epw && !placed; eb++)
                            for (int sb = 0; sb < nslots && !placed; sb++)
                                if (prog[(sb << 4) + wave * epw + eb] == 0xffffffffu) {
                                    prog[(sb << 4) + wave * epw + eb] = w | (1u << 26) | ((unsigned) ea << 27);
                                    prog[((16 + sa) << 4) + wave * epw + ea] = 0xffffffffu;
                                    placed = true;
                                }
                        if (!placed) left++;
                    }
                // (a hole left in an env's overflow words would hide the ones behind it: keep them packed from slot 0)
                if (left > 0)
                    for (int ea = 0; ea < epw; ea++) {
                        int o = 0;
                        for (int sa = 0; sa < nslots; sa++) {
                            const unsigned w = prog[((16 + sa) << 4) + wave * epw + ea];
                            if (w == 0xffffffffu) continue;
                            prog[((16 + sa) << 4) + wave * epw + ea] = 0xffffffffu;
                            prog[((16 + o) << 4) + wave * epw + ea] = w;
                            o++;
                        }
                    }
                meta[4 * wave + 1] = left > 0 ? 1 : 0;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            use_b1 = __builtin_amdgcn_readfirstlane(meta[4 * wave + 1]) != 0;
        }
        {
            // this lane's manifold and the env it belongs to (its own env, or the guest's)
            const unsigned w0 = prog[(cs << 4) + ce];
            const bool guest = w0 != 0xffffffffu && ((w0 >> 26) & 1u) != 0;
            ce_k = guest ? wave * epw + (int) ((w0 >> 27) & 15u) : ce;
            GC.e = ce_k;
            ck.lane = sub * EVM_G_ENVS + ce_k;
            ck.env = tile64 * 64 + ck.lane;
            clive_k = ck.env < d.n_real && (!mask || mask[ck.env]);
            g_bank_load(ck, w0 == 0xffffffffu ? w0 : (w0 & 0x03ffffffu), K0);
        }
        if (use_b1) {
            w1 = prog[((16 + cs) << 4) + ce];
            if (w1 != 0xffffffffu) {  // its split-impulse accumulators start at zero
                reinterpret_cast<f32x4 *>(crec_lane + ((size_t) ((int) (w1 & 511u) * EVM_CR_STRIDE + 80) << 6))[0] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#ifdef EVM_GSTAMPS
        __builtin_amdgcn_s_waitcnt(0);   // (the owners' records have landed)
        gs_c1 = __builtin_amdgcn_s_memtime();
#endif
        if (__builtin_amdgcn_readfirstlane(meta[4 * wave + 2]) != 0) {
            // solveGroupCacheFriendlySplitImpulseIterations: the same rounds on the push / turn velocities; ten iterations like
            // the oracle (further ones would add exactly nothing once an iteration changes nothing)
            float pa0[4] = {0.f, 0.f, 0.f, 0.f};
            for (int k = cs; k < 6 * nm; k += nslots) ptl[(k << 4) + ce] = 0.f;
            for (int it = 0; it < NUM_ITER; it++)
                for (int r = 0; r < nrounds; r++) {
                    g_split_bank(GC, K0, K0.round == r, pa0, imt, ptl);
                    if (use_b1) slow(3, r);
                }
            if (clive)
                for (int k = cs; k < 6 * nm; k += nslots) cc.t.scratch[((size_t) (c_skel.sc_pt + k) << 6) + cc.lane] = ptl[(k << 4) + ce];  // for the integration kernel
        }
#ifdef EVM_GSTAMPS
        gs_c2 = __builtin_amdgcn_s_memtime();
#endif
        for (int r = 0; r < nrounds; r++) {
            g_contact_bank<0>(GC, K0, K0.round == r, imt);
            if (use_b1) slow(0, r);
        }
#ifdef EVM_GSTAMPS
        gs_c3 = __builtin_amdgcn_s_memtime();
#endif
        g_lds_barrier();  // the warm-started deltas of every env are in place before any wave's joint rows read them
    }

#if defined(EVM_GSTAMPS) || defined(EVM_GSTAMPS2)
    const unsigned long long gs_t1 = __builtin_amdgcn_s_memtime();
#endif
    // ---- the sweeps: this wave's group entries, NUM_ITER times over ----
    // A wave's list is sorted by the global order, which ends with the contact entries: per sweep the wave first walks its
    // joint entries (everything in LDS; the next entry's descriptor — header + this lane group's slot, three LDS reads — is
    // requested while the current entry runs), then its contact entries (records in global memory, each requested while the
    // previous one runs; the first one before the wave waits for that entry's turn).
    const int first = gs->first[wave], count = gs->count[wave];
    float res = 0.f;
    struct GDesc { f32x4 h, d0, d1; };  // raw: decoding (readfirstlane) waits for the reads, so it happens at the point of use
    auto fetch = [&](int j, GDesc &D) {
        D.h = hdr[first + j];
        D.d0 = tbl[((first + j) * EVM_G_SLOTS + G.g) * 2];
        D.d1 = tbl[((first + j) * EVM_G_SLOTS + G.g) * 2 + 1];
    };
    auto contact_load = [&](const GDesc &D, Blk42 &k) {
        if (!(__builtin_amdgcn_readfirstlane(__float_as_int(D.h[3])) & cmask)) return;
        const int rec = __float_as_int(D.d0[0]), m = __float_as_int(D.d0[1]);
        if (rec >= 0 && ((cmask >> m) & 1u)) blk_load_quads<0, EVM_CM_STRIDE / 4>(rec_quads(c, rec), k);
    };
    auto contact_run = [&](const GDesc &D, const Blk42 &k, int it) -> float {
        const int rec = __float_as_int(D.d0[0]), a = __float_as_int(D.d0[1]);
        const int need = __float_as_int(D.d1[1]), ps = __float_as_int(D.d1[2]);
        const int expA = it * (ps & 0xffff) + (need & 0xffff);
        float r = 0.f;
        if (rec >= 0) {
            g_wait2(G, c, a, expA, a, expA);  // also when the member touches nothing: the version is a visit count
            if ((cmask >> a) & 1u) r = g_contact(G, c, k, rec, a, D.d0[3], D.d1[3]);
            g_publish(G, a, expA + 1, a, expA + 1);
        }
        return r;
    };
    float res_c = 0.f;  // contact rows of the last sweep: belongs to env ce (member-vs-member mode)
#ifdef EVM_GSTAMPS3
    unsigned long long s3_prev_end = __builtin_amdgcn_s_memtime();
#endif
    for (int it = 0; it < NUM_ITER; it++) {
        float rs = 0.f, rc = 0.f;
        int j0 = 0;
        while (j0 < count) {  // runs of joint entries and runs of contact entries, as the wave's list has them
            GDesc cur;
            fetch(j0, cur);
            const int run = __builtin_amdgcn_readfirstlane(__float_as_int(cur.h[1])) >> 16;
            if (__builtin_amdgcn_readfirstlane(__float_as_int(cur.h[0])) != 4) {
                for (int j = j0; j < j0 + run; j++) {
                    GDesc nxt;
                    fetch(j + 1 < j0 + run ? j + 1 : j0, nxt);
#ifdef EVM_GSTAMPS
                    const unsigned long long gs_e0 = __builtin_amdgcn_s_memtime();
#endif
                    const int ty = __builtin_amdgcn_readfirstlane(__float_as_int(cur.h[0]));
                    const int rec = __float_as_int(cur.d0[0]), a = __float_as_int(cur.d0[1]), b = __float_as_int(cur.d0[2]);
                    if (ty >= 5) {
                        // chain entry (5: hinges, 6: p2p): every lane takes part in the hand-over, filled slot or not
#ifdef EVM_GSTAMPS3
                        __builtin_amdgcn_s_waitcnt(0xC07F); const unsigned long long s3_e0 = __builtin_amdgcn_s_memtime();
#endif
                        const int nact = __builtin_amdgcn_readfirstlane(__float_as_int(cur.h[2]));
                        const float imA = cur.d0[3], imB = cur.d1[0];
                        const int need = __float_as_int(cur.d1[1]), ps = __float_as_int(cur.d1[2]);
                        // the shared body: slot g's visit is the g-th of the run, so the run's first visit waits for
                        // (its own count) = expA - g and the last one publishes expA - g + nact
                        const int expA0 = it * (ps & 0xffff) + (need & 0xffff) - G.g, expB = it * (ps >> 16) + (need >> 16);
                        if (rec >= 0) g_wait2(G, c, a, expA0, b, expB);
#ifdef EVM_GSTAMPS3
                        __builtin_amdgcn_s_waitcnt(0xC07F); const unsigned long long s3_e1 = __builtin_amdgcn_s_memtime();
#endif
                        const float r = ty == 5 ? g_hinge_chain(G, rec, a, b, imA, imB, nact) : g_p2p_chain(G, rec, a, b, imA, imB, nact);
#ifdef EVM_GSTAMPS3
                        __builtin_amdgcn_s_waitcnt(0xC07F); const unsigned long long s3_e2 = __builtin_amdgcn_s_memtime();
#endif
                        if (rec >= 0) { g_publish(G, a, expA0 + nact, b, expB + 1); rs = fmaxf(rs, r); }
#ifdef EVM_GSTAMPS3
                        __builtin_amdgcn_s_waitcnt(0xC07F);
                        if (ty == 5 && (threadIdx.x & 63) == 0) {
                            const unsigned long long s3_e3 = __builtin_amdgcn_s_memtime();
                            atomicAdd(&G.st3[8], s3_e1 - s3_e0); atomicAdd(&G.st3[9], s3_e2 - s3_e1); atomicAdd(&G.st3[10], s3_e3 - s3_e2);
                            atomicAdd(&G.st3[11], s3_e0 - s3_prev_end); 
                        }
                        s3_prev_end = __builtin_amdgcn_s_memtime();
#endif
                    } else if (rec >= 0) {
                        const float imA = cur.d0[3], imB = cur.d1[0];
                        const int need = __float_as_int(cur.d1[1]), ps = __float_as_int(cur.d1[2]);
                        const int expA = it * (ps & 0xffff) + (need & 0xffff), expB = it * (ps >> 16) + (need >> 16);
                        g_wait2(G, c, a, expA, b, expB);
                        float r;
                        if (ty == 0) r = g_hinge(G, rec, a, b, imA, imB);
                        else if (ty == 1) r = g_fixed(G, rec, a, b, imA, imB);
                        else if (ty == 2) {
                            const bool iso = __builtin_amdgcn_readfirstlane(__float_as_int(cur.h[2])) != 0 && !any_pending;
                            r = iso ? g_slider<true>(G, rec, a, b, imA, imB) : g_slider<false>(G, rec, a, b, imA, imB);
                        } else r = g_p2p(G, rec, a, b, imA, imB);
                        g_publish(G, a, expA + 1, b, expB + 1);
                        rs = fmaxf(rs, r);
                    }
#ifdef EVM_GSTAMPS
#pragma unroll
                    for (int q = 0; q < 4; q++) if (ty == q || (q == 0 && ty == 5) || (q == 3 && ty == 6)) { gs_type[q] += __builtin_amdgcn_s_memtime() - gs_e0; gs_n[q]++; }
#endif
                    cur = nxt;
                }
            } else {
#ifdef EVM_GSTAMPS
                const unsigned long long gs_e0 = __builtin_amdgcn_s_memtime();
#endif
                GDesc db;
                Blk42 ka, kb;
                contact_load(cur, ka);
                for (int j = 0; j < run; j += 2) {
                    if (j + 1 < run) { fetch(j0 + j + 1, db); contact_load(db, kb); }
                    rs = fmaxf(rs, contact_run(cur, ka, it));
                    if (j + 1 < run) {
                        if (j + 2 < run) { fetch(j0 + j + 2, cur); contact_load(cur, ka); }
                        rs = fmaxf(rs, contact_run(db, kb, it));
                    }
                }
#ifdef EVM_GSTAMPS
                gs_type[4] += __builtin_amdgcn_s_memtime() - gs_e0; gs_n[4] += run;
#endif
            }
            j0 += run;
        }
        if (scm) {
            // contact rows of this sweep: all normal rows, then all friction rows, each as rounds of body-disjoint manifolds;
            // a wave's rounds touch the LDS columns of its own envs only
#ifdef EVM_GSTAMPS
            const unsigned long long gs_e0 = __builtin_amdgcn_s_memtime();
#endif
            g_lds_barrier();   // every joint row of this sweep is done
#ifdef EVM_GSTAMPS
            const unsigned long long gs_e1 = __builtin_amdgcn_s_memtime();
#endif
            for (int r = 0; r < nrounds; r++) {
                rc = fmaxf(rc, g_contact_rows<1>(GC, K0, K0.round == r, imt));
                if (use_b1) rc = fmaxf(rc, slow(1, r));
            }
            for (int r = 0; r < nrounds; r++) {
                rc = fmaxf(rc, g_contact_rows<2>(GC, K0, K0.round == r, imt));
                if (use_b1) rc = fmaxf(rc, slow(2, r));
            }
#ifdef EVM_GSTAMPS
            if (gs_n[2] == 0) { gs_type[2] += __builtin_amdgcn_s_memtime() - gs_e1; }  // (slider slot of a wave without sliders: the rounds alone, no barrier)
#endif
            g_lds_barrier();   // ... and every contact row, before the next sweep's joint rows
            if (it == NUM_ITER - 1) res_c = rc;
#ifdef EVM_GSTAMPS
            gs_type[4] += __builtin_amdgcn_s_memtime() - gs_e0; gs_n[4] += 2 * nrounds;
#endif
        }
        if (it == NUM_ITER - 1) res = rs;
    }
    if (live) atomicMax(&resmax[G.e], __float_as_int(res));
    if (scm && clive_k) atomicMax(&resmax[ce_k], __float_as_int(res_c));
#if defined(EVM_GSTAMPS) || defined(EVM_GSTAMPS2)
    const unsigned long long gs_t2 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();

    // ---- final deltas -> staging copy (the integration kernel reads them) ----
    {
        float *gt = d.gtile + (size_t) tile64 * d.tile_floats + c.lane;
        for (int b = wave * EVM_G_SLOTS + G.g; b < nb && live; b += EVM_G_SLOTS * nw) {
            const f32x4 q0 = GB(G, 3 * b), q1 = GB(G, 3 * b + 1);
            gt[(b * 6 + 0) << 6] = q0[0]; gt[(b * 6 + 1) << 6] = q0[1]; gt[(b * 6 + 2) << 6] = q0[2];
            gt[(b * 6 + 3) << 6] = q0[3]; gt[(b * 6 + 4) << 6] = q1[0]; gt[(b * 6 + 5) << 6] = q1[1];
        }
    }
    if (scm && clive_k) g_bank_writeback(ck, K0);
    if (scm && clive) {
        if (use_b1 && w1 != 0xffffffffu) {
            CBank K1;
            g_bank_load(cc, w1, K1);
            g_bank_writeback(cc, K1);
        }
    }
    // ---- contact impulses back into the manifolds (by the lane group that ran the member's contact rows) ----
    for (int j = 0; j < count; j++) {
        if (__builtin_amdgcn_readfirstlane(__float_as_int(hdr[first + j][0])) != 4) continue;
        const f32x4 d0 = tbl[((first + j) * EVM_G_SLOTS + G.g) * 2];
        const int rec = __float_as_int(d0[0]), m = __float_as_int(d0[1]);
        if (rec >= 0 && ((cmask >> m) & 1u)) contact_writeback(c, m, GS(mfn, m));
    }
    // ---- muscle readbacks: getAppliedImpulse() = impulse of the last row written back ----
    for (int mi = wave * EVM_G_SLOTS + G.g; mi < c_skel.nmus && live; mi += EVM_G_SLOTS * nw) {
        const int rs = G.QR + (c_skel.sc_s + EVM_S_STRIDE * mi - c_skel.sc_h) / 4;
        const f32x4 s6 = GQ(G, rs + 6), s9 = GQ(G, rs + 9), s10 = GQ(G, rs + 10);  // fields 24..27, 36..39, 40..43
        const float jd4 = s6[1], jd5 = s6[2], a3 = s9[3], a4 = s10[0], a5 = s10[1];
        SC(c_skel.sc_mobs + 4 * mi + 1) = jd5 != 0.f ? a5 : (jd4 != 0.f ? a4 : a3);
        const int rp = G.QR + (c_skel.sc_p + EVM_P_STRIDE * (2 * mi) - c_skel.sc_h) / 4;
        SC(c_skel.sc_mobs + 4 * mi + 2) = GQ(G, rp + 3)[2];       // field 14 of p2p_a
        SC(c_skel.sc_mobs + 4 * mi + 3) = GQ(G, rp + 4 + 3)[2];   // field 14 of p2p_b
    }
    if (wave == 0 && G.g == 0 && live) {
        {   // batch-level residual: max over the 16 envs in the wave, one atomic per workgroup
            if (lane == (int) __builtin_ctzll(__ballot(true))) {  // (envs that left the kernel keep a zero in resmax)
                int r = 0;
                for (int k = 0; k < EVM_G_ENVS; k++) r = max(r, resmax[k]);
                atomicMax(d.resid, r);
            }
        }
        GS(diag, 0) = __int_as_float(resmax[G.e]);
        GS(diag, 1) = (float) ncontact;
        if (fin) {
            // the one writer of a starting reset's bookkeeping (RobotWalk::reset_engine): the stream advances by the
            // three draws whose rotation the setup kernels already used, flags / counters / settle count are set
            const M33 E = repose_draw(c);
            repose_finish(c, E, flags_in & ~EVM_FLAG_DONE, false);  // the manifolds were dropped by the setup kernel
            d.settle_left[c.env] = c_skel.settle_steps;
            GS(stat, 1) += 1;
        }
        SC(c_skel.sc_snap) = __int_as_float(d.flags[c.env]);
        SC(c_skel.sc_snap + 1) = __int_as_float(d.settle_left[c.env]);
        // the root's motion-state origin after this step, for every member's observation block
        const int b = c_skel.root;
        F3 o = G3(pos, 3 * b);
        const f32x4 r0 = GB(G, 3 * b);
        const F3 dl = f3(r0[0], r0[1], r0[2]);
        F3 lin = G3(lin, 3 * b) + dl;
        const F3 push = SC3(c_skel.sc_pt + 6 * b), turn = SC3(c_skel.sc_pt + 6 * b + 3);
        const bool nz = push.x != 0.f || push.y != 0.f || push.z != 0.f || turn.x != 0.f || turn.y != 0.f || turn.z != 0.f;
        if (nz) o = integ_pos(o, push, DT_F);
        lin = lin + f3(0.f, c_skel.body[b].ext_force_y, 0.f);
        const F3 o2 = integ_pos(o, lin, DT_F);
        SSC3(c_skel.sc_rootms, integ_pos(o2, lin, 0.f - DT_F));
    }
    // ---- fused form: the integration / observation items of k_split_post for this workgroup's 16 envs, by its 16 (wave, lane
    // group) processors.  Everything they read was written above by this workgroup (deltas in the staging copy, snapshot, root
    // motion state, muscle read-backs) or by earlier kernels: one barrier (with its workgroup-scope fences) in between.
    if (pa.mode >= 0) {
        __syncthreads();
        if (live) {
            Ctx cp = c;
            cp.lds = d.gtile + (size_t) tile64 * d.tile_floats;
            post_items(cp, d, pa, wave * EVM_G_SLOTS + G.g, EVM_G_SLOTS * nw);
        }
    }
#ifdef EVM_GSTAMPS2  // per wave: cycles in the sweeps phase, cycles of it spent waiting for versions, entries run (tools/gstamps.py --waves)
    if (sub == 0 && lane == (int) __builtin_ctzll(__ballot(true)) && wave < 4) {
        unsigned long long *st = d.stamps + (size_t) tile64 * 16;
        st[4 * wave] = gs_t2 - gs_t1; st[4 * wave + 1] = G.wait_cycles; st[4 * wave + 2] = (unsigned long long) count * NUM_ITER; st[4 * wave + 3] = gs_t1 - gs_t0;
    }
#elif defined(EVM_GSTAMPS)
    if (wave == 0 && sub == 0 && lane == 0) {
        unsigned long long *st = d.stamps + (size_t) tile64 * 16;
#pragma unroll
        for (int q = 0; q < 5; q++) { st[2 * q] = gs_type[q]; st[2 * q + 1] = gs_n[q]; }
        st[10] = gs_t1 - gs_t0; st[11] = gs_t2 - gs_t1; st[12] = __builtin_amdgcn_s_memtime() - gs_t2;
        st[13] = gs_ta - gs_t0; st[14] = gs_tb - gs_ta; st[15] = gs_tc - gs_tb;
        if (scm) { st[13] = gs_cb - gs_c0; st[14] = gs_c1 - gs_cb; st[15] = gs_c3 - gs_c1; st[5] = gs_c0 - gs_t0; }  // program, owners' records, split impulse + warm start; [5] (no plain slider entries in this mode): before the contact set-up
    }
#endif
}

}  // namespace evm
