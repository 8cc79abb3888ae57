// Skeleton constants shared by the host loader (skeleton_host.cpp) and the dynamics kernel (env_kernels.hip).
// One copy lives in __constant__ memory: every access is wave-uniform, so the compiler reads it through
// the scalar cache (s_load) and the values occupy SGPRs, not VGPRs.
#pragma once
#include <stdint.h>

#define EVM_MAX_BODIES 64
#define EVM_MAX_MEMBERS 24
#define EVM_MAX_HINGES 24
#define EVM_MAX_FIXED 8
#define EVM_MAX_MUSCLES 20
#define EVM_MAX_HULL_PTS 1024
#define EVM_MAX_PAIRS (EVM_MAX_MEMBERS * (EVM_MAX_MEMBERS - 1) / 2)
#define EVM_PC_STRIDE (EVM_MAX_PAIRS + 2)   // narrowphase work-list counters per copy: the pairs', the big-hull list's, the urgent list's
#define EVM_DEEP_SOON_DEFAULT (-0.06f)  // EnvDev::deep_soon: distance (margins subtracted) below which a pair is expected to need the penetration solver next step (cores 2 cm apart)
#define EVM_SPEC_SLOTS 256             // urgent-list entries whose penetration query a speculation block runs beside the pair's own query (narrow_dev.h), one slot each
#define EVM_SPEC_WORDS 16              // ints per slot

// per-constraint scratch strides (floats per env)
// Constraint records in the scratch tile.  A record starts on a multiple of 4 slots and is stored in QUADS: fields
// 4q..4q+3 of lane l sit together at float offset (base + 4q) * 64 + 4 l, so a wave moves a quad with one
// global_load/store_dwordx4 (1 KiB contiguous).  The accumulated impulses ("applied") start on a quad boundary
// because every sweep visit writes them back.
#define EVM_H_STRIDE 36   // hinge : relS6 = (relA, -relB) interleaved | p3 q3 ax3 | jd6 15 | rhs6 21 | pad 27 | applied6 28 | lo 34 hi 35
#define EVM_F_STRIDE 44   // fixed : relS6 | angax9 linax9 | jd6 24 | rhs6 30 | applied6 36 | pad2
#define EVM_S_STRIDE 44   // slider: relS6 | p3 q3 ax3 p2_3 q2_3 | jd6 21 | rhs6 27 | lo 33 hi 34 | pad 35 | applied6 36 | pad2
#define EVM_P_STRIDE 16   // p2p   : a1_3 a2_3 | jd3 6 | rhs3 9 | applied3 12 | pad
#define EVM_C_STRIDE 10   // contact point: rel3 lat3 | jd_n rhs_n jd_f rhs_f
#define EVM_CM_STRIDE 48  // contact record of a member: 4 points x EVM_C_STRIDE, then 4 x (applied normal, applied lateral)
// Two-body contact record (member-vs-member mode, EvmSkelC::self_collision): one per manifold id (floor-vs-member m -> id m,
// pair p -> id nm + p), 4 points x 5 quads; read-only during the sweeps (the accumulated impulses live in LDS there):
//   quad 0: relA.xyz, jd_n   quad 1: relB.xyz, rhs_n   quad 2: normalOnB.xyz, applied_n (warm-started)
//   quad 3: lat.xyz, jd_f    quad 4: rhs_f, applied_f (warm-started), rhs_penetration (split impulse), mu
// and one spare quad (20) that only the overflow path of the sweeps kernel uses (its split-impulse accumulators)
#define EVM_CR_POINT 20
#define EVM_CR_STRIDE 84
#define EVM_BIG_HULL 64    // hulls above this many vertices get the quarter-wave narrowphase (one query per 16 lanes)
#define EVM_PM_STRIDE 48  // persistent pair manifold: 4 points x (localA3 localB3 normalOnB3 dist applied applied_lateral)

struct EvmBodyC {
    float inv_mass;
    float inv_inertia[3];
    float ext_force_y;  // (gravity_force * inv_mass) * dt, y component (x = z = 0)
    float mass, friction;
    int per_sweep;  // visits touching this body in one sweep (joint visits + 1 contact visit for members)
    int isotropic;  // inv_inertia x == y == z (the attach spheres): world inverse inertia = k * R R^T = k * identity
    float m0[9];  // first_model_matrix basis, btMatrix3x3 rows (may be non-orthonormal: SURVEY App. A)
    float t0[3];  // first_model_matrix origin
};
struct EvmMemberC {
    int hull_off, hull_n;  // scaled hull points in hull[] (hull_off is even: hulls start on a vertex pair)
    int scan_first, scan_count;  // this member's slices in EvmSkelC::scan (in vertex order)
    float break_thr;       // relative contact breaking threshold
    float mu;              // combined friction with the floor
    int contact_response;
    float aabb_c[3], aabb_h[3];  // local box of btPolyhedralConvexAabbCachingShape::getAabb: centre and half extents (the
                                 // margin added twice, like Bullet) + gContactBreakingThreshold: the broadphase cull of the pairs
};
// A member pair that may collide (all but constraint parent / child: constraint.cpp:65,147), a < b; solver order = table order
struct EvmPairC {
    uint16_t a, b;
    float thr;   // min of the two relative breaking thresholds (btCollisionDispatcher::getNewManifold)
    float mu;    // product of the two frictions, clamped to +-10 (btManifoldResult::calculateCombinedFriction)
};
struct EvmHingeC {
    int a, b;
    float fa[9], fao[3], fb[9], fbo[3];  // frames in A / B (basis rows, origin)
    float factA, factB;
    float center, half_range, bias, relaxation;
};
struct EvmFixedC {
    int a, b;
    float fa[9], fao[3], fb[9], fbo[3];
};
struct EvmMuscleC {
    int sa, sb;  // attach sphere bodies (slider A, B)
    int ma, mb;  // member bodies (p2p_a: ma-sa, p2p_b: mb-sb)
    float piv_a[3], piv_b[3];
    float upper_lin, max_impulse, speed;  // max_impulse = max_force / fps
    float max_force;
    float factA, factB;
};
// One Gauss-Seidel visit (a constraint) in Bullet's solve order, everything the sweep needs in one 32-byte
// scalar load so that it can be fetched a visit ahead.
struct EvmVisitC {
    int type;    // 0 hinge, 1 fixed, 2 slider, 3 p2p
    int slot;    // first scratch slot of the constraint's record
    int a, b;    // body indices
    float imA, imB;
    int nslots;  // record length
    int need;    // bodies' version counters this visit waits for: needA | needB << 16 (visits earlier in the sweep that touch the body)
};
#define EVM_MAX_VISITS (EVM_MAX_HINGES + EVM_MAX_FIXED + 3 * EVM_MAX_MUSCLES)
#ifndef EVM_NW
#define EVM_NW 8
#endif

#define EVM_MAX_SCHED (EVM_MAX_VISITS + 64)
#define EVM_SCHED_NONE 0x7fff
// One slice of a member's hull for the deepest-vertex scan.  Hulls are cut into slices of at most 64 vertices so that
// the scans of the four 451-vertex feet spread over every wave that is available; the slices' (minimum, index) pairs
// meet in the tile (LDS, or its global staging copy in the split pipeline).
struct EvmScanC {
    int member, begin, end;  // vertices [begin, end) of the member's hull, begin even
    int wave;
};
#define EVM_MAX_SCAN 96
#define EVM_SCHED_BARRIER 0x8000
#define EVM_SCHED_CONTACT 0x4000   // entry = EVM_SCHED_CONTACT | member: the member's contact rows of this sweep
#define EVM_SCHED_MUSCLE 0x2000    // host scheduler only: muscle k = its three visits (slider, p2p_a, p2p_b) as one item

// One entry of a wave's sweep stream, self-contained (64 bytes = one s_load_dwordx16 at an index that depends on
// nothing but the stream position):
//   type 0 / 1   hinge / fixed visit on bodies (a, b)
//   type 4       a member's contact rows (a = b = member)
//   type 5       a whole muscle: slider(sa, sb), p2p(a, sa), p2p(b, sb) in Bullet's order.  The two attach spheres are
//                touched by nothing else, so they need no version counters and stay in registers across the three; only
//                the members a and b are waited for, each right before its own p2p rows.  slot = the slider's record,
//                the p2p records follow at sc_p + 2 k EVM_P_STRIDE.
struct EvmEntryC {
    int type, slot, a, b;
    float imA, imB;     // inverse masses of a and b
    int nslots, need;   // need = versions a (low 16 bits) and b must have reached within the sweep
    int psA, psB;       // EvmBodyC::per_sweep of a and b (version stride per sweep)
    int iso;            // type 5: both spheres isotropic, scalar inverse inertias kA, kB
    float kA, kB;
    int spheres;        // type 5: sa | sb << 16
    float imSa, imSb;   // type 5: inverse masses of the spheres
};
#define EVM_MAX_WAVE_ENTRIES 48

struct EvmSkelC {
    int nb, nm, nh, nf, nmus, root;
    int obs_dim, act_dim;
    int state_member[EVM_MAX_MEMBERS];  // observation order: root first (skeleton.cpp:140-160)
    int state_index[EVM_MAX_MEMBERS];   // inverse: member -> position of its 19-value block
    int ncon;                                        // skeleton constraints in file order (skeleton.cpp:77-82)
    int con_type[EVM_MAX_HINGES + EVM_MAX_FIXED];    // 0 hinge, 1 fixed
    int con_idx[EVM_MAX_HINGES + EVM_MAX_FIXED];
    float floor_o[3], floor_top_y;
    float root_pos[3];
    float min_vel, target_vel;
    int max_steps, init_remaining, reset_frames;
    int env_kind;          // 0 robot_walk, 1 robot_jump (EvmEnvParams::env_kind)
    int self_collision;    // member-vs-member contacts (EvmEnvParams::self_collision)
    int npair;             // collidable member pairs (0 unless self_collision)
    int hull_pts;                  // vertices in use in hull[] (all hulls, pair-padded)
    int big_hull_off, big_hull_n;  // the largest hull above EVM_BIG_HULL vertices (hull table offset, vertex count; -1: none):
                                   // the narrowphase kernel's blocks keep it in LDS
    int settle_steps;      // physics steps inside reset(): 2 * reset_frames (robot_walk.cpp:98-103) or reset_frames (robot_jump.cpp:104)
    float reset_angle_limit;  // pi * 2 / 3 (robot_walk.cpp:80) or pi / 3 (robot_jump.cpp:89)
    // scratch layout (offsets in floats-per-env)
    int sc_r, sc_ext, sc_ms, sc_pt, sc_mobs, sc_h, sc_f, sc_s, sc_p, sc_c, sc_total;
    int sc_snap;           // 2 ints (as float bits): flags and settle_left as they were when the step began — the post kernel's
                           // waves read these while the root's wave rewrites the live values
    int sc_nexte;          // 9 floats: rotation of the reset an env that just finished will start at its next call (drawn
                           // read-only from its RNG stream by the post kernel; consumed by the next call's kernels)
    int sc_rootms;         // 3 floats: the root's motion-state origin after this step (split pipeline: written by the sweeps
                           // kernel, read by every observation block)
    float sched_cycles;  // host estimate of the 10 sweeps under the schedule's cost model (information only)
    int nvisit;
    EvmVisitC visit[EVM_MAX_VISITS];
    // Sweep schedule: visits that share no body commute exactly, so only the per-body order of visits matters.
    // The host list-schedules one sweep (joint visits + one contact visit per member) onto the EVM_NW waves with
    // a cost model, bottom-level priorities and a hop latency between waves; each wave's list is ordered by
    // simulated start time (a linear extension of the dependency order => no deadlock with the version counters).
    // Entry = joint visit index, or EVM_SCHED_CONTACT | member.
    int nlevels;
    int nsched[EVM_NW];
    int nwsched[EVM_NW];  // entries of wsched (a muscle is one entry there, three codes in sched)
    int sched[EVM_NW][EVM_MAX_SCHED];
    EvmEntryC wsched[EVM_NW][EVM_MAX_WAVE_ENTRIES];  // the same lists as self-contained descriptors (what the kernel walks)
    int member_wave[EVM_MAX_MEMBERS];  // which wave maintains the manifold and builds the contact rows of member m
    int nscan;
    EvmScanC scan[EVM_MAX_SCAN];
    EvmBodyC body[EVM_MAX_BODIES];
    EvmMemberC member[EVM_MAX_MEMBERS];
    EvmHingeC hinge[EVM_MAX_HINGES];
    EvmFixedC fixed[EVM_MAX_FIXED];
    EvmMuscleC muscle[EVM_MAX_MUSCLES];
    // hull vertices in PAIRS: vertex g = 2 P + s has x, y, z at hull[6 P + s], hull[6 P + 2 + s], hull[6 P + 4 + s]
    // (two vertices per packed multiply/add in the scan); a hull with an odd count repeats its last vertex
    float hull[EVM_MAX_HULL_PTS * 3];
    EvmPairC pair[EVM_MAX_PAIRS];
    uint16_t pair_order[EVM_MAX_PAIRS];  // the pairs by decreasing narrowphase cost (hull sizes): the pair kernel's item order
};

// ---------------------------------------------------------------------------------------------------------------------
// Lane-group sweep schedule (k_sweeps_g, the sweeps kernel of the split pipeline).  A wavefront there is
// EVM_G_SLOTS lane groups x 16 environments: every group works on a DIFFERENT constraint of the same type for the same 16
// environments, so one instruction stream serves up to four constraints (the four legs of the spider give exactly that).
// A "group entry" = up to four visits of one type that share no body; the host packs one sweep's visits (Bullet order) into
// group entries such that, for every body, the entries that touch it keep Bullet's order, and deals the entries to the
// kernel's waves.  The table lives in global memory (one copy per EvmEnv) and is staged into LDS by the kernel.
#define EVM_G_SLOTS 4
#define EVM_G_ENVS 16
#define EVM_G_MAX_ENTRIES 96   // group entries of one sweep, all waves together
#define EVM_G_MAX_WAVES 4
struct EvmGSlotC {      // 32 bytes = two quads, read (broadcast) by the 16 lanes of one group
    int rec;            // < 0: empty slot.  types 0-3: first quad of the constraint's record relative to the record image
                        // (scratch slot - sc_h) / 4; type 4: first scratch slot of the member's contact record
    int a, b;           // bodies (type 4: a = b = member; type 3: a = member, b = attach sphere)
    float imA;
    float imB;
    int need;           // versions of a (low 16 bits) and b the visit waits for within its sweep
    int ps;             // visits per sweep on a (low 16 bits) and on b
    float aux;          // type 4: combined friction mu; type 2: 1 if both bodies are isotropic
};
struct EvmGEntryC {
    int type;           // 0 hinge, 1 fixed, 2 slider, 3 p2p, 4 contact rows
    int order;          // low 16 bits: position in the one global order of the sweep's entries (every wave's list is sorted by
                        // it); high 16 bits: entries of the same class (joint / contact rows) that follow in the wave's list, + 1
    int iso;            // type 2: every filled slot joins two isotropic bodies
    int members;        // type 4: bit mask of the members in the entry
};
struct EvmGSchedC {
    int nwaves;
    int total;                          // entries of all waves; wave w owns entries [first[w], first[w] + count[w])
    int first[EVM_G_MAX_WAVES], count[EVM_G_MAX_WAVES];
    int nrq;                            // quads of the joint-record image per env: (sc_c - sc_h) / 4
    int lds_bytes;                      // dynamic LDS of the kernel for this skeleton
    float est_cycles;                   // host estimate of the ten sweeps (information only)
    int with_contacts;                  // 1: the schedule carries the members' contact entries (floor contacts only); 0: joint
                                        // entries only, the contact rows run as barrier-separated rounds (member-vs-member mode)
    EvmGEntryC entry[EVM_G_MAX_ENTRIES];
    EvmGSlotC slot[EVM_G_MAX_ENTRIES][EVM_G_SLOTS];
};

