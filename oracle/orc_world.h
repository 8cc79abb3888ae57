// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under evomotion_amd/ may include, link or call this.
//
// Scalar CPU restatement of the reference's robot_walk environment:
//   Environment::do_step / reset            evo_motion_model/src/environment.cpp:33-48
//   RobotWalk ctor / compute_step / reset   evo_motion_model/src/env/robot_walk.cpp:17-104
//   Skeleton body / constraint / state order evo_motion_model/src/robot/skeleton.cpp:77-160
//   RigidBodyItem                           evo_motion_model/src/item.cpp:17-86
//   Hinge / Fixed constraints               evo_motion_model/src/robot/constraint.cpp:52-69,137-150
//   Muscle (slider + 2 p2p)                 evo_motion_model/src/robot/muscle.cpp:14-85
//   Proprioception states                   evo_motion_model/src/robot/proprioception_state.cpp:21-129
// plus the third-party algorithm those files delegate to: Bullet3's btDiscreteDynamicsWorld::stepSimulation
// with the sequential-impulse solver [UPSTREAM — Bullet3 is an un-vendored, un-pinned system dependency
// (evo_motion_model/CMakeLists.txt:14); its algorithm is restated here from the published bullet3 3.x
// sources as remembered, it could not be compiled or run in this image].
//
// PARITY STATUS: *parity unpinned* for the physics half.  The reference holds no test, golden vector or
// fixture for evo_motion_model, and Bullet3 is unavailable, so this restatement is pinned
// only by (a) analytic invariants (tests/test_oracle_physics.py) and (b) the independently verifiable
// constants of SURVEY.md App. D (RNG stream, step counters, dimensions).
//
// Collision: member hull vs the floor plane (deepest hull vertex per step fed into a Bullet-style 4-point persistent
// manifold; the reference's floor is a 2000 x 2 x 2000 box hull, so GJK against it reduces to that up to the choice of
// the point on a flat face) and, with EnvParams::self_collision, every member-vs-member pair the reference lets collide
// (all but constraint parent/child: constraint.cpp:65,147) through the GJK narrowphase of orc_narrow.h into one
// persistent manifold per pair, solved as two-dynamic-body normal + friction rows.
#pragma once
#include <string>
#include <vector>

#include "orc_math.h"
#include "orc_narrow.h"

namespace orc {

// MEASUREMENT SWITCH (orc_set_floor_as_hull; default 0): 1 = member-vs-floor contacts through the convex-convex path with the floor as
// the reference builds it (a cube hull scaled (1000, 1, 1000), robot_walk.cpp:22-25), instead of deepest-vertex-vs-plane.
extern int g_floor_as_hull;
extern float g_floor_hull_half;

struct ShapeDef {
    std::string name;
    std::vector<V3> pts;  // unique hull points, first-occurrence order
};
struct MemberDef {
    std::string name;
    int shape;
    float mass, friction;
    V3 t;
    float qw, qx, qy, qz;
    V3 scale;
    int ignore_collision;
};
struct ConstraintDef {
    int type;  // 0 hinge, 1 fixed
    std::string name;
    int parent, child;
    // hinge
    V3 pivot_p, pivot_c, axis_p, axis_c;
    float lim_lo, lim_hi;
    // fixed
    V3 tp, tc;
    float qp[4], qc[4];  // w x y z
};
struct MuscleDef {
    std::string name;
    int a, b;
    float attach_mass;
    V3 attach_scale, pos_a, pos_b;
    float force, speed;
};
struct SkeletonDef {
    std::string robot_name, root_name;
    std::vector<MemberDef> members;
    std::vector<ConstraintDef> constraints;
    std::vector<MuscleDef> muscles;
    std::vector<ShapeDef> shapes;
    int shape_index(const std::string &n) const;
    int member_index(const std::string &n) const;
};
bool load_skeleton(const char *path, SkeletonDef &out, std::string &err);

struct EnvParams {
    float initial_remaining_seconds = 1.f;  // evo_motion_model/src/env/env_factory.cpp:80-82
    float max_episode_seconds = 30.f;
    float target_velocity = 0.5f;
    float minimal_velocity = 0.1f;
    int reset_frames = 30;
    int self_collision = 0;  // 1: member-vs-member contacts (the reference's behaviour); 0: floor contacts only
    int env_kind = 0;  // 0 robot_walk (robot_walk.cpp), 1 robot_jump (robot_jump.cpp:66-110): reward max(vy,0)+vz, fail on
                       // remaining < 0, reset yaw/roll/pitch within pi/3, `reset_frames` settle steps in ONE loop
};

struct ManifoldPoint {
    V3 localA, localB;  // A = floor (static), B = member
    V3 posA, posB, normalB;
    float dist;
    float applied, applied_lat;
    V3 latdir;
};
struct Manifold {
    int n = 0;
    ManifoldPoint p[4];
};
// One persistent manifold per member pair that may collide; body0 = a < b = body1 (btBroadphasePair orders its proxies by
// unique id = insertion order = member order, skeleton.cpp:92-103).  Pairs are kept in lexicographic (a, b) order, which is
// also the order of their rows in the solver: Bullet's own order (hashed pair cache, island sort) cannot be known.
struct PairManifold {
    int a, b;
    float break_thr;   // min of the two shapes' relative breaking thresholds (btCollisionDispatcher::getNewManifold)
    float friction;    // calculateCombinedFriction: product of the two, clamped to +-10
    Manifold mf;
};

struct Body {
    // constants
    float mass = 0, inv_mass = 0;
    V3 inv_inertia_local;
    float friction = 0.5f;
    int shape = 0;
    V3 scale;
    bool is_member = false;
    bool contact_response = true;
    float break_thr = 0.02f;
    Xf first_model;
    V3 gravity_force;
    // state
    Xf xf;
    Q q;  // quaternion whose matFromQuat() is xf.b (valid unless the env is in the reset-pending state)
    V3 lin, ang;
    M3 iinv_world;
    V3 ms_origin;  // btDefaultMotionState origin (lags one step)
    // solver scratch
    V3 dlin, dang, push, turn, extF, extT;
};

struct Hinge {
    int a, b;
    Xf frameA, frameB;
    float center, half_range, bias, relaxation;
    // per-step
    float angle, correction;
    bool solve_limit;
    float applied;
};
struct Fixed {
    int a, b;
    Xf frameA, frameB;
    float applied;
};
struct Slider {
    int a, b;
    float upper_lin;  // lower = 0; ang limits 0,0
    float max_force, max_speed;
    bool powered;
    float target_vel;
    // per-step
    float lin_pos, depth0, ang_depth;
    bool solve_lin, solve_ang;
    float applied;
};
struct P2P {
    int a, b;
    V3 pivotA, pivotB;
    float applied;
};
struct WorldConstraint {
    int type;  // 0 hinge 1 fixed 2 slider 3 p2p
    int idx;
};

struct Row {
    int a, b;  // body index, -1 = static
    V3 n1, c1, n2, c2, angA, angB;
    float jd, rhs, rhs_pen, cfm, lo, hi, applied, applied_push, friction;
    int fric_of;  // for friction rows: index of the normal row
    int owner;    // constraint index (joint rows) or manifold slot (contact rows)
};

class World {
public:
    SkeletonDef skel;
    EnvParams prm;
    std::vector<Body> bodies;  // members, then per muscle attach_a, attach_b  (skeleton.cpp:92-103)
    std::vector<Hinge> hinges;
    std::vector<Fixed> fixeds;
    std::vector<Slider> sliders;
    std::vector<P2P> p2ps;
    std::vector<WorldConstraint> order;  // skeleton.cpp:77-90
    std::vector<Manifold> manifolds;     // one per member (vs floor)
    std::vector<PairManifold> pairs;     // member-vs-member, lexicographic; empty unless prm.self_collision
    int root = 0;
    std::vector<int> state_members;  // root first, then non-root members in array order (skeleton.cpp:140-160)
    std::vector<V3> last_lin, last_ang;
    float floor_top_y = -1.f, floor_friction = 0.5f;
    Xf floor_xf;
    MT19937 rng;
    int rng_draws = 0;
    int curr_step = 0, max_steps = 0, remaining_steps = 0;
    bool reset_pending = false;  // world transforms are E*M0 (non-orthonormal) until the next integrate
    M3 reset_E;
    // diagnostics of the last step
    int last_num_contacts = 0, last_num_joint_rows = 0;
    int last_num_pair_contacts = 0;      // of last_num_contacts: points of member-vs-member manifolds
    int last_pair_tests = 0, last_pair_gjk_iters = 0, last_pair_penetration_calls = 0;  // narrowphase work of the last step
    int last_floor_gjk_iters = 0, last_floor_queries = 0, last_floor_pen_calls = 0, last_floor_ccd_hits = 0;   // floor-as-hull measurement mode only
    long long total_pair_tests = 0, total_pair_penetration_calls = 0, total_physics_steps = 0;  // ... and since creation (settle steps of reset() included)
    float last_max_pair_penetration = 0; // deepest pair contact distance (negative = penetrating) after the last collide()
    float last_residual = 0;

    bool init(const SkeletonDef &s, int seed, const EnvParams &p, std::string &err);
    int nb() const { return (int) bodies.size(); }
    int nmember() const { return (int) skel.members.size(); }
    int nmuscle() const { return (int) skel.muscles.size(); }
    int obs_dim() const { return 19 * nmember() + 4 * nmuscle(); }
    int act_dim() const { return nmuscle(); }

    void apply_action(const float *action);  // controllers: muscle_controller.cpp:10-12, muscle.cpp:82-85
    void physics_step();                     // Environment::step_world -> stepSimulation(1/60, n, 1/60)
    void compute_step(float *obs, float *reward, int *done);
    void reset_begin();  // RobotWalk::reset_engine up to (not including) the settle steps
    void reset(float *obs, float *reward, int *done);
    void do_step(const float *action, float *obs, float *reward, int *done);

    // canonical state blob shared with the HIP implementation (layout: include/evomotion.h)
    int state_size() const;
    void get_state(float *out) const;
    void set_state(const float *in);
    void get_poses(float *out) const;  // nb x (px py pz qx qy qz qw)

    int npairs() const { return (int) pairs.size(); }
    ConvexView convex_view(int member) const;

private:
    void collide();
    void collide_pairs();
    void solve();
    void integrate();
};

}  // namespace orc
