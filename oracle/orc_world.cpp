// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_world.h header for scope, citations and parity status).
#include <algorithm>

#include "orc_world.h"

#include <array>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

namespace orc {

static constexpr float DT = 1.f / 60.f;         // evo_motion_model/src/constants.h.in:8
static constexpr float MARGIN = 0.04f;          // [UPSTREAM] CONVEX_DISTANCE_MARGIN
static constexpr float G_BREAK = 0.02f;         // [UPSTREAM] gContactBreakingThreshold
// [UPSTREAM] btContactSolverInfo defaults
static constexpr int NUM_ITER = 10;
static constexpr float ERP = 0.2f, ERP2 = 0.2f, GLOBAL_CFM = 0.f, SOR = 1.f, DAMPING = 1.f;
static constexpr float SPLIT_THRESHOLD = -0.04f, SPLIT_TURN_ERP = 0.1f, LINEAR_SLOP = 0.f, WARMSTART = 0.85f;

// ------------------------------------------------------------------------------------------------
// fixture parsing
// ------------------------------------------------------------------------------------------------
int SkeletonDef::shape_index(const std::string &n) const {
    for (size_t i = 0; i < shapes.size(); i++)
        if (shapes[i].name == n) return (int) i;
    return -1;
}
int SkeletonDef::member_index(const std::string &n) const {
    for (size_t i = 0; i < members.size(); i++)
        if (members[i].name == n) return (int) i;
    return -1;
}

static bool rdf(std::istringstream &ss, float &f) {
    std::string t;
    if (!(ss >> t)) return false;
    f = std::strtof(t.c_str(), nullptr);
    return true;
}
static bool rdv(std::istringstream &ss, V3 &v) { return rdf(ss, v.x) && rdf(ss, v.y) && rdf(ss, v.z); }

bool load_skeleton(const char *path, SkeletonDef &out, std::string &err) {
    std::ifstream f(path);
    if (!f) { err = std::string("cannot open ") + path; return false; }
    std::string line;
    struct PendingMember { std::string shape; };
    std::vector<std::string> member_shapes;
    std::vector<std::array<std::string, 2>> con_names;
    std::vector<std::array<std::string, 2>> mus_names;
    int shape_left = 0;
    while (std::getline(f, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ss(line);
        if (shape_left > 0) {
            V3 p;
            if (!rdv(ss, p)) { err = "bad shape point"; return false; }
            out.shapes.back().pts.push_back(p);
            shape_left--;
            continue;
        }
        std::string kw;
        ss >> kw;
        if (kw == "skeleton") {
            std::string r;
            ss >> out.robot_name >> r >> out.root_name;
        } else if (kw == "members" || kw == "constraints" || kw == "muscles" || kw == "shapes") {
        } else if (kw == "member") {
            MemberDef m;
            std::string shape;
            ss >> m.name >> shape;
            member_shapes.push_back(shape);
            if (!(rdf(ss, m.mass) && rdf(ss, m.friction) && rdv(ss, m.t) && rdf(ss, m.qw) && rdf(ss, m.qx) &&
                  rdf(ss, m.qy) && rdf(ss, m.qz) && rdv(ss, m.scale))) { err = "bad member"; return false; }
            ss >> m.ignore_collision;
            m.shape = -1;
            out.members.push_back(m);
        } else if (kw == "hinge") {
            ConstraintDef c{};
            c.type = 0;
            std::string p, ch;
            ss >> c.name >> p >> ch;
            con_names.push_back({p, ch});
            if (!(rdv(ss, c.pivot_p) && rdv(ss, c.pivot_c) && rdv(ss, c.axis_p) && rdv(ss, c.axis_c) &&
                  rdf(ss, c.lim_lo) && rdf(ss, c.lim_hi))) { err = "bad hinge"; return false; }
            out.constraints.push_back(c);
        } else if (kw == "fixed") {
            ConstraintDef c{};
            c.type = 1;
            std::string p, ch;
            ss >> c.name >> p >> ch;
            con_names.push_back({p, ch});
            bool ok = rdv(ss, c.tp);
            for (int i = 0; i < 4; i++) ok = ok && rdf(ss, c.qp[i]);
            ok = ok && rdv(ss, c.tc);
            for (int i = 0; i < 4; i++) ok = ok && rdf(ss, c.qc[i]);
            if (!ok) { err = "bad fixed"; return false; }
            out.constraints.push_back(c);
        } else if (kw == "muscle") {
            MuscleDef m{};
            std::string a, b;
            ss >> m.name >> a >> b;
            mus_names.push_back({a, b});
            if (!(rdf(ss, m.attach_mass) && rdv(ss, m.attach_scale) && rdv(ss, m.pos_a) && rdv(ss, m.pos_b) &&
                  rdf(ss, m.force) && rdf(ss, m.speed))) { err = "bad muscle"; return false; }
            out.muscles.push_back(m);
        } else if (kw == "shape") {
            ShapeDef s;
            int n = 0, ndup = 0;
            ss >> s.name >> n >> ndup;
            out.shapes.push_back(s);
            shape_left = n;
        } else {
            err = "unknown keyword " + kw;
            return false;
        }
    }
    for (size_t i = 0; i < out.members.size(); i++) {
        out.members[i].shape = out.shape_index(member_shapes[i]);
        if (out.members[i].shape < 0) { err = "unknown shape " + member_shapes[i]; return false; }
    }
    for (size_t i = 0; i < out.constraints.size(); i++) {
        out.constraints[i].parent = out.member_index(con_names[i][0]);
        out.constraints[i].child = out.member_index(con_names[i][1]);
        // evo_motion_model/src/robot/skeleton.cpp:55-59 throws std::runtime_error for unknown members
        if (out.constraints[i].parent < 0 || out.constraints[i].child < 0) { err = "Member not found"; return false; }
    }
    for (size_t i = 0; i < out.muscles.size(); i++) {
        out.muscles[i].a = out.member_index(mus_names[i][0]);
        out.muscles[i].b = out.member_index(mus_names[i][1]);
        if (out.muscles[i].a < 0 || out.muscles[i].b < 0) { err = "Member not found"; return false; }
    }
    if (out.member_index(out.root_name) < 0) { err = "root member not found"; return false; }
    return true;
}

// ------------------------------------------------------------------------------------------------
// construction  (RigidBodyItem: item.cpp:17-41; btConvexHullShape / btRigidBody [UPSTREAM])
// ------------------------------------------------------------------------------------------------
static M3 inertia_world(const M3 &basis, const V3 &inv_local) {
    // btRigidBody::updateInertiaTensor
    return basis.scaled(inv_local) * basis.transpose();
}

static void hull_props(const ShapeDef &sh, const V3 &scale, float mass, V3 &inv_inertia_local, float &break_thr) {
    // btPolyhedralConvexAabbCachingShape::recalcLocalAabb: support in +-axes of the scaled points, +- margin
    V3 mx(-BT_LARGE_FLOAT, -BT_LARGE_FLOAT, -BT_LARGE_FLOAT), mn(BT_LARGE_FLOAT, BT_LARGE_FLOAT, BT_LARGE_FLOAT);
    for (const V3 &p : sh.pts) {
        V3 s = p * scale;
        for (int i = 0; i < 3; i++) {
            if (s[i] > mx[i]) mx.at(i) = s[i];
            if (s[i] < mn[i]) mn.at(i) = s[i];
        }
    }
    V3 lmax = mx + V3(MARGIN, MARGIN, MARGIN), lmin = mn - V3(MARGIN, MARGIN, MARGIN);
    // getAabb(identity): btTransformAabb(localMin, localMax, margin, I)
    V3 half = 0.5f * (lmax - lmin);
    half += V3(MARGIN, MARGIN, MARGIN);
    V3 center = 0.5f * (lmax + lmin);
    V3 amin = center - half, amax = center + half;
    // btPolyhedralConvexShape::calculateLocalInertia
    V3 he = (amax - amin) * 0.5f;
    float lx = 2.f * (he.x + MARGIN), ly = 2.f * (he.y + MARGIN), lz = 2.f * (he.z + MARGIN);
    float x2 = lx * lx, y2 = ly * ly, z2 = lz * lz;
    float scaledmass = mass * 0.08333333f;
    V3 inertia = scaledmass * V3(y2 + z2, x2 + z2, x2 + y2);
    inv_inertia_local = V3(inertia.x != 0.f ? 1.f / inertia.x : 0.f, inertia.y != 0.f ? 1.f / inertia.y : 0.f,
                           inertia.z != 0.f ? 1.f / inertia.z : 0.f);
    // btCollisionShape::getContactBreakingThreshold = getAngularMotionDisc() * gContactBreakingThreshold
    float radius = length(amax - amin) * 0.5f;
    V3 c = (amin + amax) * 0.5f;
    float disc = radius + length(c);
    break_thr = disc * G_BREAK;
}

bool World::init(const SkeletonDef &s, int seed, const EnvParams &p, std::string &err) {
    skel = s;
    prm = p;
    rng.seed((uint32_t) seed);
    rng_draws = 0;
    const V3 gravity(0, -9.8f, 0);  // environment.cpp:30

    // floor: cube hull scaled (1000,1,1000) at (0,-2,2), mass 0, friction 0.5 (robot_walk.cpp:22-25,33)
    floor_xf = Xf::identity();
    floor_xf.o = V3(0.f, -2.f, 2.f);
    floor_top_y = floor_xf.o.y + 1.0f * 1.f;
    floor_friction = 0.5f;

    bodies.clear();
    for (const MemberDef &m : skel.members) {
        Body b;
        b.is_member = true;
        b.shape = m.shape;
        b.scale = m.scale;
        b.mass = m.mass;
        b.friction = m.friction;  // member.cpp:28
        b.contact_response = !m.ignore_collision;
        b.first_model.b = glm_mat3_cast(m.qw, m.qx, m.qy, m.qz);  // member.cpp:26
        b.first_model.o = m.t;
        bodies.push_back(b);
    }
    for (const MuscleDef &m : skel.muscles) {  // muscle.cpp:21-28
        int sph = skel.shape_index("sphere");
        if (sph < 0) { err = "sphere shape missing"; return false; }
        for (int side = 0; side < 2; side++) {
            Body b;
            b.is_member = false;
            b.shape = sph;
            b.scale = m.attach_scale;
            b.mass = m.attach_mass;
            b.friction = 0.5f;
            b.contact_response = false;  // CF_NO_CONTACT_RESPONSE, muscle.cpp:57-60
            const Body &parent = bodies[side == 0 ? m.a : m.b];
            Xf tr = Xf::identity();
            tr.o = side == 0 ? m.pos_a : m.pos_b;
            b.first_model = glm_mul(parent.first_model, tr);
            bodies.push_back(b);
        }
    }
    for (Body &b : bodies) {
        b.inv_mass = b.mass == 0.f ? 0.f : 1.0f / b.mass;
        hull_props(skel.shapes[b.shape], b.scale, b.mass, b.inv_inertia_local, b.break_thr);
        b.xf = b.first_model;
        b.q = quatFromMat(b.xf.b);
        b.ms_origin = b.xf.o;
        b.lin = b.ang = V3();
        b.iinv_world = inertia_world(b.xf.b, b.inv_inertia_local);
        b.gravity_force = b.inv_mass != 0.f ? gravity * (1.0f / b.inv_mass) : V3();
    }

    hinges.clear(); fixeds.clear(); sliders.clear(); p2ps.clear(); order.clear();
    for (const ConstraintDef &c : skel.constraints) {
        if (c.type == 0) {
            // btHingeConstraint(rbA, rbB, pivotInA, pivotInB, axisInA, axisInB)  [UPSTREAM], constraint.cpp:59-68
            Hinge h{};
            h.a = c.parent; h.b = c.child;
            const M3 &basisA = bodies[h.a].xf.b;
            V3 axisInA = c.axis_p, axisInB = c.axis_c;
            V3 rbAxisA1 = basisA.col(0), rbAxisA2;
            float projection = dot(axisInA, rbAxisA1);
            if (projection >= 1.0f - SIMD_EPSILON) {
                rbAxisA1 = -basisA.col(2); rbAxisA2 = basisA.col(1);
            } else if (projection <= -1.0f + SIMD_EPSILON) {
                rbAxisA1 = basisA.col(2); rbAxisA2 = basisA.col(1);
            } else {
                rbAxisA2 = cross(axisInA, rbAxisA1);
                rbAxisA1 = cross(rbAxisA2, axisInA);
            }
            h.frameA.o = c.pivot_p;
            h.frameA.b = M3(rbAxisA1.x, rbAxisA2.x, axisInA.x, rbAxisA1.y, rbAxisA2.y, axisInA.y, rbAxisA1.z,
                            rbAxisA2.z, axisInA.z);
            Q arc = shortestArcQuat(axisInA, axisInB);
            V3 rbAxisB1 = quatRotate(arc, rbAxisA1);
            V3 rbAxisB2 = cross(axisInB, rbAxisB1);
            h.frameB.o = c.pivot_c;
            h.frameB.b = M3(rbAxisB1.x, rbAxisB2.x, axisInB.x, rbAxisB1.y, rbAxisB2.y, axisInB.y, rbAxisB1.z,
                            rbAxisB2.z, axisInB.z);
            // setLimit(lo, hi) -> btAngularLimit::set(low, high, 0.9, 0.3, 1.0)
            h.half_range = (c.lim_hi - c.lim_lo) / 2.0f;
            h.center = btNormalizeAngle(c.lim_lo + h.half_range);
            h.bias = 0.3f;
            h.relaxation = 1.0f;
            order.push_back({0, (int) hinges.size()});
            hinges.push_back(h);
        } else {
            // btFixedConstraint = btGeneric6DofSpring2Constraint with all limits locked, constraint.cpp:143-149
            Fixed f{};
            f.a = c.parent; f.b = c.child;
            f.frameA.b = glm_mat3_cast(c.qp[0], c.qp[1], c.qp[2], c.qp[3]); f.frameA.o = c.tp;
            f.frameB.b = glm_mat3_cast(c.qc[0], c.qc[1], c.qc[2], c.qc[3]); f.frameB.o = c.tc;
            order.push_back({1, (int) fixeds.size()});
            fixeds.push_back(f);
        }
    }
    for (size_t mi = 0; mi < skel.muscles.size(); mi++) {
        const MuscleDef &m = skel.muscles[mi];
        int ia = nmember() + 2 * (int) mi, ib = ia + 1;
        Slider sl{};
        sl.a = ia; sl.b = ib;
        sl.max_force = m.force; sl.max_speed = m.speed;
        V3 d = bodies[ia].xf.o - bodies[ib].xf.o;
        sl.upper_lin = 2.f * std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);  // muscle.cpp:43-49
        sl.powered = false; sl.target_vel = 0.f;
        order.push_back({2, (int) sliders.size()});
        sliders.push_back(sl);
        P2P pa{}; pa.a = m.a; pa.b = ia; pa.pivotA = m.pos_a; pa.pivotB = V3();
        order.push_back({3, (int) p2ps.size()});
        p2ps.push_back(pa);
        P2P pb{}; pb.a = m.b; pb.b = ib; pb.pivotA = m.pos_b; pb.pivotB = V3();
        order.push_back({3, (int) p2ps.size()});
        p2ps.push_back(pb);
    }

    manifolds.assign(nmember(), Manifold());
    // member pairs that may collide: all but constraint parent/child (setIgnoreCollisionCheck, constraint.cpp:65,147);
    // CF_NO_CONTACT_RESPONSE bodies (attach spheres, muscle.cpp:57-60; ignore_collision members, member.cpp:31-33) get no rows
    pairs.clear();
    if (prm.self_collision) {
        for (int i = 0; i < nmember(); i++)
            for (int j = i + 1; j < nmember(); j++) {
                if (!bodies[i].contact_response || !bodies[j].contact_response) continue;
                bool adjacent = false;
                for (const ConstraintDef &c : skel.constraints)
                    if ((c.parent == i && c.child == j) || (c.parent == j && c.child == i)) adjacent = true;
                if (adjacent) continue;
                PairManifold pm;
                pm.a = i; pm.b = j;
                pm.break_thr = std::min(bodies[i].break_thr, bodies[j].break_thr);
                float f = bodies[i].friction * bodies[j].friction;
                if (f < -10.f) f = -10.f;
                if (f > 10.f) f = 10.f;
                pm.friction = f;
                pairs.push_back(pm);
            }
    }
    root = skel.member_index(skel.root_name);
    state_members.clear();
    state_members.push_back(root);
    for (int i = 0; i < nmember(); i++)
        if (i != root) state_members.push_back(i);
    last_lin.assign(nmember(), V3());
    last_ang.assign(nmember(), V3());

    curr_step = 0;
    max_steps = (int) (prm.max_episode_seconds / DT);            // robot_walk.cpp:30
    remaining_steps = (int) (prm.initial_remaining_seconds / DT);  // robot_walk.cpp:31
    reset_pending = false;
    reset_E = M3::identity();
    return true;
}

// ------------------------------------------------------------------------------------------------
// controllers
// ------------------------------------------------------------------------------------------------
void World::apply_action(const float *action) {
    for (size_t i = 0; i < sliders.size(); i++) {
        sliders[i].powered = true;
        sliders[i].target_vel = action[i] * sliders[i].max_speed;
    }
}

// ------------------------------------------------------------------------------------------------
// collision: hull vs floor plane -> persistent manifold  ([UPSTREAM] btPersistentManifold semantics)
// ------------------------------------------------------------------------------------------------
static int get_cache_entry(const Manifold &m, const V3 &localA, float thr) {
    float shortest = thr * thr;
    int nearest = -1;
    for (int i = 0; i < m.n; i++) {
        V3 d = m.p[i].localA - localA;
        float dist = dot(d, d);
        if (dist < shortest) { shortest = dist; nearest = i; }
    }
    return nearest;
}
static int sort_cached_points(const Manifold &m, const ManifoldPoint &pt) {
    int maxPenIdx = -1;
    float maxPen = pt.dist;
    for (int i = 0; i < 4; i++)
        if (m.p[i].dist < maxPen) { maxPenIdx = i; maxPen = m.p[i].dist; }
    float res[4] = {0, 0, 0, 0};
    if (maxPenIdx != 0) { V3 a = pt.localA - m.p[1].localA, b = m.p[3].localA - m.p[2].localA; res[0] = length2(cross(a, b)); }
    if (maxPenIdx != 1) { V3 a = pt.localA - m.p[0].localA, b = m.p[3].localA - m.p[2].localA; res[1] = length2(cross(a, b)); }
    if (maxPenIdx != 2) { V3 a = pt.localA - m.p[0].localA, b = m.p[3].localA - m.p[1].localA; res[2] = length2(cross(a, b)); }
    if (maxPenIdx != 3) { V3 a = pt.localA - m.p[0].localA, b = m.p[2].localA - m.p[1].localA; res[3] = length2(cross(a, b)); }
    // btVector4::closestAxis4 = absolute4().maxAxis4()
    int maxIndex = -1;
    float maxVal = -BT_LARGE_FLOAT;
    for (int i = 0; i < 4; i++) {
        float v = std::fabs(res[i]);
        if (v > maxVal) { maxIndex = i; maxVal = v; }
    }
    return maxIndex;
}

// the reference's floor as what Bullet sees: the eight corners of cube.obj under local scaling (1000, 1, 1000)
// (evo_motion_model/src/env/robot_walk.cpp:22-25, src/item.cpp:17-41: a btConvexHullShape like every member)
static const V3 kFloorCube[8] = {V3(-1, -1, -1), V3(-1, -1, 1), V3(-1, 1, -1), V3(-1, 1, 1), V3(1, -1, -1), V3(1, -1, 1), V3(1, 1, -1), V3(1, 1, 1)};
int g_floor_as_hull = 0;
float g_floor_hull_half = 1000.f;   // 1000 = the reference's floor; smaller: a well-conditioned box of that half width kept under each member (measurement)

void World::collide() {
    ConvexView floor_view;
    V3 fmin, fmax;
    if (g_floor_as_hull) {
        floor_view.pts = kFloorCube; floor_view.n = 8;
        floor_view.scale = V3(1000.f, 1.f, 1000.f);
        floor_view.xf = floor_xf;
        floor_view.margin = MARGIN;
        world_aabb(floor_view, G_BREAK, fmin, fmax);
    }
    last_floor_gjk_iters = 0; last_floor_queries = 0; last_floor_pen_calls = 0; last_floor_ccd_hits = 0;
    for (int mi = 0; mi < nmember(); mi++) {
        Body &B = bodies[mi];
        if (!B.contact_response) continue;
        Manifold &mf = manifolds[mi];
        const ShapeDef &sh = skel.shapes[B.shape];
        bool have = false;
        ManifoldPoint np{};
        if (g_floor_as_hull) {
            // MEASUREMENT MODE (orc_set_floor_as_hull): the floor goes through the same convex-convex path as a member pair —
            // AABB overlap, one btGjkPairDetector query (body0 = floor, body1 = member), btManifoldResult::addContactPoint
            V3 mn, mx;
            world_aabb(convex_view(mi), G_BREAK, mn, mx);
            const bool overlap = fmin.x <= mx.x && fmax.x >= mn.x && fmin.y <= mx.y && fmax.y >= mn.y && fmin.z <= mx.z && fmax.z >= mn.z;
            if (overlap) {
                const float md = MARGIN + MARGIN + B.break_thr;   // the manifold's threshold = the smaller of the two shapes' (the member's)
                ConvexView fv = floor_view;
                if (g_floor_hull_half < 1000.f) {   // same top face, a box of human size centred under the member: no 2000 m operands
                    fv.scale = V3(g_floor_hull_half, 1.f, g_floor_hull_half);
                    fv.xf.o = V3(std::floor(B.xf.o.x + 0.5f), floor_xf.o.y, std::floor(B.xf.o.z + 0.5f));
                }
                const ClosestResult r = gjk_closest_points(fv, convex_view(mi), md * md);
                if (r.used_penetration) last_floor_pen_calls++;
                if (r.ccd_status == 0) last_floor_ccd_hits++;
                last_floor_gjk_iters += r.iterations; last_floor_queries++;
                if (r.has && !(r.distance > B.break_thr)) {
                    const V3 pointA = r.pointOnB + r.normalOnB * r.distance;
                    np.localA = floor_xf.invXform(pointA);
                    np.localB = B.xf.invXform(r.pointOnB);
                    np.posA = pointA; np.posB = r.pointOnB; np.normalB = r.normalOnB; np.dist = r.distance;
                    have = true;
                }
            }
        } else {
            // deepest hull vertex (first strict minimum of world y)
            float best = SIMD_INFINITY;
            V3 bestw;
            for (const V3 &p : sh.pts) {
                V3 w = B.xf(p * B.scale);
                if (w.y < best) { best = w.y; bestw = w; }
            }
            const V3 normal(0.f, -1.f, 0.f);  // on B (member), pointing towards A (floor)
            float dist_core = best - floor_top_y;
            float depth = dist_core - (MARGIN + MARGIN);
            if (!(depth > B.break_thr)) {
                // btManifoldResult::addContactPoint
                V3 pointOnB(bestw.x, bestw.y - MARGIN, bestw.z);
                V3 pointA = pointOnB + normal * depth;
                np.localA = floor_xf.invXform(pointA);
                np.localB = B.xf.invXform(pointOnB);
                np.posA = pointA; np.posB = pointOnB; np.normalB = normal; np.dist = depth;
                have = true;
            }
        }
        if (have) {
            np.applied = 0; np.applied_lat = 0;
            int idx = get_cache_entry(mf, np.localA, B.break_thr);
            if (idx >= 0) {  // replaceContactPoint keeps the accumulated impulses
                np.applied = mf.p[idx].applied; np.applied_lat = mf.p[idx].applied_lat;
                mf.p[idx] = np;
            } else {
                int ins = mf.n;
                if (ins == 4) ins = sort_cached_points(mf, np); else mf.n++;
                if (ins < 0) ins = 0;
                mf.p[ins] = np;
            }
        }
        // btPersistentManifold::refreshContactPoints(trA = floor, trB = member)
        for (int i = mf.n - 1; i >= 0; i--) {
            ManifoldPoint &mp = mf.p[i];
            mp.posA = floor_xf(mp.localA);
            mp.posB = B.xf(mp.localB);
            mp.dist = dot(mp.posA - mp.posB, mp.normalB);
        }
        for (int i = mf.n - 1; i >= 0; i--) {
            ManifoldPoint &mp = mf.p[i];
            bool remove = false;
            if (!(mp.dist <= B.break_thr)) remove = true;
            else {
                V3 projected = mp.posA - mp.normalB * mp.dist;
                V3 diff = mp.posB - projected;
                if (dot(diff, diff) > B.break_thr * B.break_thr) remove = true;
            }
            if (remove) {
                int last = mf.n - 1;
                if (i != last) mf.p[i] = mf.p[last];
                mf.n--;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// collision: member vs member  ([UPSTREAM] btConvexConvexAlgorithm::processCollision + btManifoldResult)
// ------------------------------------------------------------------------------------------------
ConvexView World::convex_view(int member) const {
    const Body &B = bodies[member];
    const ShapeDef &sh = skel.shapes[B.shape];
    ConvexView v;
    v.pts = sh.pts.data();
    v.n = (int) sh.pts.size();
    v.scale = B.scale;
    v.xf = B.xf;
    v.margin = MARGIN;
    return v;
}

void World::collide_pairs() {
    last_pair_tests = 0; last_pair_gjk_iters = 0; last_pair_penetration_calls = 0;
    last_max_pair_penetration = 0.f;
    total_physics_steps++;
    if (pairs.empty()) return;
    // world AABBs as btCollisionWorld::updateAabbs leaves them.  The broadphase only culls: a pair whose boxes are apart
    // cannot be within the breaking threshold, and a cached point is gone (refresh below) long before the boxes part, so
    // the contacts do not depend on Bullet's pair-cache hysteresis, which is not restated.
    std::vector<V3> amin(nmember()), amax(nmember());
    for (int i = 0; i < nmember(); i++) world_aabb(convex_view(i), G_BREAK, amin[i], amax[i]);
    for (PairManifold &pm : pairs) {
        const Body &A = bodies[pm.a], &B = bodies[pm.b];
        Manifold &mf = pm.mf;
        const bool overlap = amin[pm.a].x <= amax[pm.b].x && amax[pm.a].x >= amin[pm.b].x && amin[pm.a].y <= amax[pm.b].y &&
                             amax[pm.a].y >= amin[pm.b].y && amin[pm.a].z <= amax[pm.b].z && amax[pm.a].z >= amin[pm.b].z;
        if (overlap) {
            last_pair_tests++; total_pair_tests++;
            // btConvexConvexAlgorithm: m_maximumDistanceSquared = (marginA + marginB + breaking threshold)^2
            float md = MARGIN + MARGIN + pm.break_thr;
            const ClosestResult r = gjk_closest_points(convex_view(pm.a), convex_view(pm.b), md * md);
            last_pair_gjk_iters += r.iterations;
            if (r.used_penetration) { last_pair_penetration_calls++; total_pair_penetration_calls++; }
            if (r.has && !(r.distance > pm.break_thr)) {
                // btManifoldResult::addContactPoint(normalOnBInWorld, pointInWorld, depth)
                const V3 pointA = r.pointOnB + r.normalOnB * r.distance;
                ManifoldPoint np{};
                np.localA = A.xf.invXform(pointA);
                np.localB = B.xf.invXform(r.pointOnB);
                np.posA = pointA; np.posB = r.pointOnB; np.normalB = r.normalOnB; np.dist = r.distance;
                np.applied = 0; np.applied_lat = 0;
                int idx = get_cache_entry(mf, np.localA, pm.break_thr);
                if (idx >= 0) {
                    np.applied = mf.p[idx].applied; np.applied_lat = mf.p[idx].applied_lat;
                    mf.p[idx] = np;
                } else {
                    int ins = mf.n;
                    if (ins == 4) ins = sort_cached_points(mf, np); else mf.n++;
                    if (ins < 0) ins = 0;
                    mf.p[ins] = np;
                }
            }
        }
        // btPersistentManifold::refreshContactPoints(trA, trB): the stored normal stays, positions and distance follow the bodies
        for (int i = mf.n - 1; i >= 0; i--) {
            ManifoldPoint &mp = mf.p[i];
            mp.posA = A.xf(mp.localA);
            mp.posB = B.xf(mp.localB);
            mp.dist = dot(mp.posA - mp.posB, mp.normalB);
        }
        for (int i = mf.n - 1; i >= 0; i--) {
            ManifoldPoint &mp = mf.p[i];
            bool remove = false;
            if (!(mp.dist <= pm.break_thr)) remove = true;
            else {
                V3 projected = mp.posA - mp.normalB * mp.dist;
                V3 diff = mp.posB - projected;
                if (dot(diff, diff) > pm.break_thr * pm.break_thr) remove = true;
            }
            if (remove) {
                int last = mf.n - 1;
                if (i != last) mf.p[i] = mf.p[last];
                mf.n--;
            }
        }
        for (int i = 0; i < mf.n; i++) last_max_pair_penetration = std::fmin(last_max_pair_penetration, mf.p[i].dist);
    }
}

// ------------------------------------------------------------------------------------------------
// solver  ([UPSTREAM] btSequentialImpulseConstraintSolver, scalar reference row solvers)
// ------------------------------------------------------------------------------------------------
struct Info2 {
    V3 J1l[6], J1a[6], J2l[6], J2a[6];
    float err[6], lo[6], hi[6];
    int n;
    void clear() {
        n = 0;
        for (int i = 0; i < 6; i++) { J1l[i] = J1a[i] = J2l[i] = J2a[i] = V3(); err[i] = 0; lo[i] = -SIMD_INFINITY; hi[i] = SIMD_INFINITY; }
    }
};

static float motor_factor(float pos, float lowLim, float uppLim, float vel, float timeFact) {
    if (lowLim > uppLim) return 1.0f;
    if (lowLim == uppLim) return 0.0f;
    float lim_fact = 1.0f;
    float delta_max = vel / timeFact;
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

static void hinge_info(Hinge &h, const Body &A, const Body &B, Info2 &o) {
    const float fps = 1.f / DT;
    // getInfo1: testLimit
    {
        V3 refAxis0 = A.xf.b * h.frameA.b.col(0);
        V3 refAxis1 = A.xf.b * h.frameA.b.col(1);
        V3 swingAxis = B.xf.b * h.frameB.b.col(1);
        h.angle = std::atan2(dot(swingAxis, refAxis0), dot(swingAxis, refAxis1));  // m_referenceSign = +1
        h.correction = 0.f; h.solve_limit = false;
        if (h.half_range >= 0.f) {
            float dev = btNormalizeAngle(h.angle - h.center);
            if (dev < -h.half_range) { h.solve_limit = true; h.correction = -(dev + h.half_range); }
            else if (dev > h.half_range) { h.solve_limit = true; h.correction = h.half_range - dev; }
        }
    }
    // getInfo2InternalUsingFrameOffset
    Xf trA = A.xf * h.frameA, trB = B.xf * h.frameB;
    V3 ofs = trB.o - trA.o;
    float miA = A.inv_mass, miB = B.inv_mass, miS = miA + miB;
    float factA = miS > 0.f ? miB / miS : 0.5f;
    float factB = 1.0f - factA;
    V3 ax1A = trA.b.col(2), ax1B = trB.b.col(2);
    V3 ax1 = ax1A * factA + ax1B * factB;
    if (length2(ax1) < SIMD_EPSILON) { factA = 0.f; factB = 1.f; ax1 = ax1A * factA + ax1B * factB; }
    ax1 = normalized(ax1);
    V3 relB = trB.o - B.xf.o;
    V3 projB = ax1 * dot(relB, ax1);
    V3 orthoB = relB - projB;
    V3 relA = trA.o - A.xf.o;
    V3 projA = ax1 * dot(relA, ax1);
    V3 orthoA = relA - projA;
    V3 totalDist = projA - projB;
    relA = orthoA + totalDist * factA;
    relB = orthoB - totalDist * factB;
    V3 p = orthoB * factA + orthoA * factB;
    float len2 = length2(p);
    if (len2 > SIMD_EPSILON) p = p / std::sqrt(len2); else p = trA.b.col(1);
    V3 q = cross(ax1, p);
    o.J1a[0] = cross(relA, p); o.J2a[0] = -cross(relB, p);
    o.J1a[1] = cross(relA, q); o.J2a[1] = -cross(relB, q);
    o.J1a[2] = cross(relA, ax1); o.J2a[2] = -cross(relB, ax1);
    float k = fps * ERP;
    o.J1l[0] = p; o.J1l[1] = q; o.J1l[2] = ax1;
    o.J2l[0] = -p; o.J2l[1] = -q; o.J2l[2] = -ax1;
    o.err[0] = k * dot(p, ofs); o.err[1] = k * dot(q, ofs); o.err[2] = k * dot(ax1, ofs);
    o.J1a[3] = p; o.J1a[4] = q; o.J2a[3] = -p; o.J2a[4] = -q;
    V3 u = cross(ax1A, ax1B);
    o.err[3] = k * dot(u, p); o.err[4] = k * dot(u, q);
    o.n = 5;
    if (h.solve_limit) {
        float limit_err = h.correction;  // * m_referenceSign (+1)
        int limit = limit_err > 0.f ? 1 : 2;
        int r = 5;
        o.J1a[r] = ax1; o.J2a[r] = -ax1;
        // lostop == histop never holds for the robot_walk limits (half_range > 0)
        o.err[r] = 0.f;
        o.err[r] += k * limit_err;
        if (limit == 1) { o.lo[r] = 0; o.hi[r] = SIMD_INFINITY; } else { o.lo[r] = -SIMD_INFINITY; o.hi[r] = 0; }
        float bounce = h.relaxation;
        if (bounce > 0.f) {
            float vel = dot(A.ang, ax1);
            vel -= dot(B.ang, ax1);
            if (limit == 1) {
                if (vel < 0) { float newc = -bounce * vel; if (newc > o.err[r]) o.err[r] = newc; }
            } else {
                if (vel > 0) { float newc = -bounce * vel; if (newc < o.err[r]) o.err[r] = newc; }
            }
        }
        o.err[r] *= h.bias;
        o.n = 6;
    }
}

static void fixed_info(Fixed &f, const Body &A, const Body &B, Info2 &o) {
    const float fps = 1.f / DT;
    // btGeneric6DofSpring2Constraint::calculateTransforms (RO_XYZ), all six axes locked (m_currentLimit == 3)
    Xf cA = A.xf * f.frameA, cB = B.xf * f.frameB;
    V3 linDiff = cA.b.inverse() * (cB.o - cA.o);
    M3 rel = cA.b.inverse() * cB.b;
    V3 ang;
    {  // matrixToEulerXYZ; btGetMatrixElem(mat, idx) = mat[idx % 3][idx / 3]
        float fi = rel.r[2][0];
        if (fi < 1.0f) {
            if (fi > -1.0f) {
                ang.x = std::atan2(-rel.r[2][1], rel.r[2][2]);
                ang.y = std::asin(rel.r[2][0]);
                ang.z = std::atan2(-rel.r[1][0], rel.r[0][0]);
            } else {
                ang.x = -std::atan2(rel.r[0][1], rel.r[1][1]); ang.y = -SIMD_HALF_PI; ang.z = 0.f;
            }
        } else {
            ang.x = std::atan2(rel.r[0][1], rel.r[1][1]); ang.y = SIMD_HALF_PI; ang.z = 0.f;
        }
    }
    V3 axis0 = cB.b.col(0), axis2 = cA.b.col(2);
    V3 cax[3];
    cax[1] = cross(axis2, axis0);
    cax[0] = cross(cax[1], axis2);
    cax[2] = cross(axis0, cax[1]);
    for (int i = 0; i < 3; i++) cax[i] = normalized(cax[i]);
    int row = 0;
    // setAngularLimits first (rows 0..2), error sign -1 for rotational
    for (int i = 0; i < 3; i++) {
        o.J1a[row] = cax[i]; o.J2a[row] = -cax[i];
        float limit_error = ang[i] - 0.f;  // test_value - loLimit
        o.err[row] = fps * ERP * limit_error * -1.f;
        row++;
    }
    // setLinearLimits (rows 3..5)
    for (int i = 0; i < 3; i++) {
        V3 ax = cA.b.col(i);
        o.J1l[row] = ax; o.J2l[row] = -ax;
        V3 relB = cB.o - B.xf.o, relA = cA.o - A.xf.o;
        V3 tmpA = cross(relA, ax), tmpB = cross(relB, ax);
        o.J1a[row] = tmpA; o.J2a[row] = -tmpB;
        float limit_error = linDiff[i] - 0.f;
        o.err[row] = fps * ERP * limit_error * 1.f;
        row++;
    }
    o.n = 6;
}

static void slider_info(Slider &s, const Body &A, const Body &B, Info2 &o) {
    const float fps = 1.f / DT;
    // frames in A and B are identity, m_useLinearReferenceFrameA = true (muscle.cpp:30-36)
    const Xf &trA = A.xf, &trB = B.xf;
    // calculateTransforms
    V3 sliderAxis = trA.b.col(0);
    V3 delta = trB.o - trA.o;
    float depth[3];
    for (int i = 0; i < 3; i++) depth[i] = dot(delta, trA.b.col(i));
    (void) sliderAxis;
    // testAngLimits (lower = upper = 0)
    s.ang_depth = 0.f; s.solve_ang = false;
    {
        V3 axisA0 = trA.b.col(1), axisA1 = trA.b.col(2), axisB0 = trB.b.col(1);
        float rot = std::atan2(dot(axisB0, axisA1), dot(axisB0, axisA0));
        rot = btAdjustAngleToLimits(rot, 0.f, 0.f);
        if (rot < 0.f) { s.ang_depth = rot - 0.f; s.solve_ang = true; }
        else if (rot > 0.f) { s.ang_depth = rot - 0.f; s.solve_ang = true; }
    }
    // testLinLimits (lower 0 <= upper)
    s.solve_lin = false;
    s.lin_pos = depth[0];
    {
        const float lower = 0.f, upper = s.upper_lin;
        if (lower <= upper) {
            if (depth[0] > upper) { depth[0] -= upper; s.solve_lin = true; }
            else if (depth[0] < lower) { depth[0] -= lower; s.solve_lin = true; }
            else depth[0] = 0.f;
        } else depth[0] = 0.f;
    }
    s.depth0 = depth[0];
    // getInfo2NonVirtual (m_useOffsetForConstraintFrame = true)
    const float signFact = 1.0f;
    V3 ofs = trB.o - trA.o;
    float miA = A.inv_mass, miB = B.inv_mass, miS = miA + miB;
    float factA = miS > 0.f ? miB / miS : 0.5f;
    float factB = 1.0f - factA;
    V3 ax1A = trA.b.col(0), ax1B = trB.b.col(0);
    V3 ax1 = ax1A * factA + ax1B * factB;
    ax1 = normalized(ax1);
    V3 p, q;
    btPlaneSpace1(ax1, p, q);
    o.J1a[0] = p; o.J1a[1] = q; o.J2a[0] = -p; o.J2a[1] = -q;
    float currERP = 1.0f * ERP;  // m_softnessOrthoAng * info->erp
    float k = fps * currERP;
    V3 u = cross(ax1A, ax1B);
    o.err[0] = k * dot(u, p);
    o.err[1] = k * dot(u, q);
    int nrow = 1;
    nrow++; int s2 = nrow;
    nrow++; int s3 = nrow;
    V3 relB = trB.o - B.xf.o;  // = 0 (identity frame) but kept literal
    V3 projB = ax1 * dot(relB, ax1);
    V3 orthoB = relB - projB;
    V3 relA = trA.o - A.xf.o;
    V3 projA = ax1 * dot(relA, ax1);
    V3 orthoA = relA - projA;
    float sliderOffs = s.lin_pos - s.depth0;
    V3 totalDist = projA + ax1 * sliderOffs - projB;
    relA = orthoA + totalDist * factA;
    relB = orthoB - totalDist * factB;
    p = orthoB * factA + orthoA * factB;
    float len2 = length2(p);
    if (len2 > SIMD_EPSILON) p = p / std::sqrt(len2); else p = trA.b.col(1);
    q = cross(ax1, p);
    o.J1a[s2] = cross(relA, p); o.J2a[s2] = -cross(relB, p);
    o.J1a[s3] = cross(relA, q); o.J2a[s3] = -cross(relB, q);
    o.J1l[s2] = p; o.J1l[s3] = q; o.J2l[s2] = -p; o.J2l[s3] = -q;
    currERP = 1.0f * ERP;  // m_softnessOrthoLin * info->erp
    k = fps * currERP;
    o.err[s2] = k * dot(p, ofs);
    o.err[s3] = k * dot(q, ofs);
    // linear limit / motor row
    float limit_err = 0.f;
    int limit = 0;
    if (s.solve_lin) { limit_err = s.depth0 * signFact; limit = limit_err > 0.f ? 2 : 1; }
    bool powered = s.powered;
    if (limit || powered) {
        nrow++;
        int r = nrow;
        o.J1l[r] = ax1; o.J2l[r] = -ax1;
        o.J1a[r] = cross(relA, ax1); o.J2a[r] = -cross(relB, ax1);  // both bodies dynamic
        const float lostop = 0.f, histop = s.upper_lin;
        if (limit && (lostop == histop)) powered = false;
        o.err[r] = 0.f; o.lo[r] = 0.f; o.hi[r] = 0.f;
        currERP = ERP;
        if (powered) {
            float tag_vel = s.target_vel;
            float mot_fact = motor_factor(s.lin_pos, lostop, histop, tag_vel, fps * currERP);
            o.err[r] -= signFact * mot_fact * s.target_vel;
            o.lo[r] += -s.max_force / fps;
            o.hi[r] += s.max_force / fps;
        }
        if (limit) {
            k = fps * currERP;
            o.err[r] += k * limit_err;
            if (lostop == histop) { o.lo[r] = -SIMD_INFINITY; o.hi[r] = SIMD_INFINITY; }
            else if (limit == 1) { o.lo[r] = -SIMD_INFINITY; o.hi[r] = 0; }
            else { o.lo[r] = 0; o.hi[r] = SIMD_INFINITY; }
            // bounce = |1 - m_dampingLimLin| = 0 -> no restitution term
            o.err[r] *= 1.0f;  // m_softnessLimLin
        }
    }
    // angular limit row (lower == upper == 0)
    limit_err = 0.f; limit = 0;
    if (s.solve_ang) { limit_err = s.ang_depth; limit = limit_err > 0.f ? 1 : 2; }
    if (limit) {
        nrow++;
        int r = nrow;
        o.J1a[r] = ax1; o.J2a[r] = -ax1;
        k = fps * ERP;
        o.err[r] += k * limit_err;
        o.lo[r] = -SIMD_INFINITY; o.hi[r] = SIMD_INFINITY;  // lostop == histop
        o.err[r] *= 1.0f;  // m_softnessLimAng
    }
    o.n = nrow + 1;
}

static void p2p_info(const P2P &c, const Body &A, const Body &B, Info2 &o) {
    const float fps = 1.f / DT;
    o.J1l[0] = V3(1, 0, 0); o.J1l[1] = V3(0, 1, 0); o.J1l[2] = V3(0, 0, 1);
    V3 a1 = A.xf.b * c.pivotA;
    M3 s1 = skew(-a1);
    o.J1a[0] = s1.r[0]; o.J1a[1] = s1.r[1]; o.J1a[2] = s1.r[2];
    o.J2l[0] = V3(-1, 0, 0); o.J2l[1] = V3(0, -1, 0); o.J2l[2] = V3(0, 0, -1);
    V3 a2 = B.xf.b * c.pivotB;
    M3 s2 = skew(a2);
    o.J2a[0] = s2.r[0]; o.J2a[1] = s2.r[1]; o.J2a[2] = s2.r[2];
    float k = fps * ERP;
    for (int j = 0; j < 3; j++) o.err[j] = k * (a2[j] + B.xf.o[j] - a1[j] - A.xf.o[j]);
    o.n = 3;
}

static inline void apply_impulse(Body &b, const V3 &lin, const V3 &ang, float mag) {
    b.dlin += lin * mag;
    b.dang += ang * mag;
}
static inline void apply_push(Body &b, const V3 &lin, const V3 &ang, float mag) {
    b.push += lin * mag;
    b.turn += ang * mag;
}

void World::solve() {
    // ---- convertBodies ----
    for (Body &b : bodies) {
        b.dlin = b.dang = b.push = b.turn = V3();
        V3 totalForce = b.gravity_force;  // applyGravity; cleared after the step
        b.extF = totalForce * b.inv_mass * DT;
        b.extT = V3();
        // BT_ENABLE_GYROSCOPIC_FORCE_IMPLICIT_BODY: btRigidBody::computeGyroscopicImpulseImplicit_Body
        V3 idl(1.f / b.inv_inertia_local.x, 1.f / b.inv_inertia_local.y, 1.f / b.inv_inertia_local.z);
        V3 omega1 = b.ang;
        Q q = quatFromMat(b.xf.b);
        V3 omegab = quatRotate(inverse(q), omega1);
        M3 Ib(idl.x, 0, 0, 0, idl.y, 0, 0, 0, idl.z);
        V3 ibo = Ib * omegab;
        V3 f = DT * cross(omegab, ibo);
        M3 skew0 = skew(omegab);
        V3 om = Ib * omegab;
        M3 skew1 = skew(om);
        M3 J = Ib + (skew0 * Ib - skew1) * DT;
        V3 omega_div = solve33(J, f);
        omegab = omegab - omega_div;
        V3 omega2 = quatRotate(q, omegab);
        b.extT += omega2 - omega1;
    }

    // ---- convertJoints ----
    std::vector<Row> jrows;
    std::vector<int> first_row(order.size(), 0), num_rows(order.size(), 0);
    Info2 inf;
    for (size_t ci = 0; ci < order.size(); ci++) {
        inf.clear();
        int a, b;
        switch (order[ci].type) {
            case 0: { Hinge &h = hinges[order[ci].idx]; a = h.a; b = h.b; h.applied = 0; hinge_info(h, bodies[a], bodies[b], inf); break; }
            case 1: { Fixed &f = fixeds[order[ci].idx]; a = f.a; b = f.b; f.applied = 0; fixed_info(f, bodies[a], bodies[b], inf); break; }
            case 2: { Slider &s = sliders[order[ci].idx]; a = s.a; b = s.b; s.applied = 0; slider_info(s, bodies[a], bodies[b], inf); break; }
            default: { P2P &p = p2ps[order[ci].idx]; a = p.a; b = p.b; p.applied = 0; p2p_info(p, bodies[a], bodies[b], inf); break; }
        }
        first_row[ci] = (int) jrows.size();
        num_rows[ci] = inf.n;
        const Body &A = bodies[a], &B = bodies[b];
        for (int j = 0; j < inf.n; j++) {
            Row r{};
            r.a = a; r.b = b; r.owner = (int) ci;
            r.n1 = inf.J1l[j]; r.c1 = inf.J1a[j]; r.n2 = inf.J2l[j]; r.c2 = inf.J2a[j];
            r.lo = inf.lo[j]; r.hi = inf.hi[j];  // breaking threshold = SIMD_INFINITY: clamps are no-ops
            r.cfm = GLOBAL_CFM;
            r.angA = A.iinv_world * r.c1;
            r.angB = B.iinv_world * r.c2;
            V3 iMJlA = r.n1 * A.inv_mass, iMJaA = A.iinv_world * r.c1;
            V3 iMJlB = r.n2 * B.inv_mass, iMJaB = B.iinv_world * r.c2;
            float sum = dot(iMJlA, r.n1);
            sum += dot(iMJaA, r.c1);
            sum += dot(iMJlB, r.n2);
            sum += dot(iMJaB, r.c2);
            r.jd = std::fabs(sum) > SIMD_EPSILON ? SOR / sum : 0.f;
            float vel1Dotn = dot(r.n1, A.lin + A.extF) + dot(r.c1, A.ang + A.extT);
            float vel2Dotn = dot(r.n2, B.lin + B.extF) + dot(r.c2, B.ang + B.extT);
            float rel_vel = vel1Dotn + vel2Dotn;
            float positionalError = inf.err[j];
            float velocityError = 0.f - rel_vel * DAMPING;
            r.rhs = positionalError * r.jd + velocityError * r.jd;
            r.applied = 0.f;
            jrows.push_back(r);
        }
    }
    last_num_joint_rows = (int) jrows.size();

    // ---- convertContacts ----  manifolds in the order: floor-vs-member by member index (body0 = floor/static, body1 =
    // member), then member-vs-member pairs in lexicographic order (body0 = a, body1 = b).  Bullet's own manifold order
    // (dispatcher array, island sort) cannot be known; what is Bullet's is kept: per manifold point a normal row and one
    // friction row, all normal rows solved before all friction rows, both kinds in manifold order.
    std::vector<Row> crows, frows;
    struct CRef { Manifold *mf; int slot; };
    std::vector<CRef> cref;
    const float invTimeStep = 1.f / DT;
    last_num_pair_contacts = 0;
    auto convert_manifold = [&](int ia, int ib, Manifold &mf, float combinedFriction) {
        Body *A = ia >= 0 ? &bodies[ia] : nullptr;
        Body &B = bodies[ib];
        const V3 originA = A ? A->xf.o : floor_xf.o;
        for (int j = 0; j < mf.n; j++) {
            ManifoldPoint &cp = mf.p[j];
            // contact processing threshold = BT_LARGE_FLOAT: every cached point is processed
            const V3 n = cp.normalB;
            V3 rel_pos1 = cp.posA - originA;
            V3 rel_pos2 = cp.posB - B.xf.o;
            // getVelocityInLocalPointNoDelta
            V3 vel1 = A ? A->lin + A->extF + cross(A->ang + A->extT, rel_pos1) : V3(0, 0, 0);
            V3 vel2 = B.lin + B.extF + cross(B.ang + B.extT, rel_pos2);
            V3 vel = vel1 - vel2;
            float rel_vel = dot(n, vel);
            Row r{};
            r.a = ia; r.b = ib; r.owner = j;
            // setupContactConstraint
            V3 torqueAxis0 = cross(rel_pos1, n);
            V3 torqueAxis1 = cross(rel_pos2, n);
            r.angA = A ? A->iinv_world * torqueAxis0 : V3();
            r.angB = B.iinv_world * (-torqueAxis1);
            {
                float denom0 = 0.f, denom1 = 0.f;
                if (A) { V3 vec = cross(r.angA, rel_pos1); denom0 = A->inv_mass + dot(n, vec); }
                { V3 vec = cross(-r.angB, rel_pos2); denom1 = B.inv_mass + dot(n, vec); }
                r.jd = SOR / (denom0 + denom1 + GLOBAL_CFM * invTimeStep);
            }
            if (A) { r.n1 = n; r.c1 = torqueAxis0; } else { r.n1 = V3(); r.c1 = V3(); }
            r.n2 = -n; r.c2 = -torqueAxis1;
            float penetration = cp.dist + LINEAR_SLOP;
            r.friction = combinedFriction;
            float restitution = 0.f;  // combined restitution is 0
            // warm start
            r.applied = cp.applied * WARMSTART;
            if (A) apply_impulse(*A, r.n1 * A->inv_mass, r.angA, r.applied);
            apply_impulse(B, -r.n2 * B.inv_mass, -r.angB, -r.applied);
            r.applied_push = 0.f;
            {
                float vel1Dotn = A ? dot(r.n1, A->lin + A->extF) + dot(r.c1, A->ang + A->extT) : 0.f;
                float vel2Dotn = dot(r.n2, B.lin + B.extF) + dot(r.c2, B.ang + B.extT);
                float rv = vel1Dotn + vel2Dotn;
                float positionalError = 0.f;
                float velocityError = restitution - rv;
                if (penetration > 0) velocityError -= penetration * invTimeStep;
                else positionalError = -penetration * ERP2 * invTimeStep;
                float penetrationImpulse = positionalError * r.jd;
                float velocityImpulse = velocityError * r.jd;
                if (penetration > SPLIT_THRESHOLD) { r.rhs = penetrationImpulse + velocityImpulse; r.rhs_pen = 0.f; }
                else { r.rhs = velocityImpulse; r.rhs_pen = penetrationImpulse; }
                r.cfm = 0.f;
                r.lo = 0.f; r.hi = 1e10f;
            }
            int normal_index = (int) crows.size();
            crows.push_back(r);
            cref.push_back({&mf, j});
            if (A) last_num_pair_contacts++;
            // friction direction (velocity dependent, one direction)
            cp.latdir = vel - n * rel_vel;
            float lat_rel_vel = length2(cp.latdir);
            if (lat_rel_vel > SIMD_EPSILON) {
                cp.latdir *= 1.f / std::sqrt(lat_rel_vel);
            } else {
                V3 d2;
                btPlaneSpace1(n, cp.latdir, d2);
            }
            // setupFrictionConstraint
            Row fr{};
            fr.a = ia; fr.b = ib; fr.owner = j; fr.fric_of = normal_index;
            fr.friction = combinedFriction;
            if (A) {
                fr.n1 = cp.latdir;
                fr.c1 = cross(rel_pos1, fr.n1);
                fr.angA = A->iinv_world * fr.c1;
            } else { fr.n1 = V3(); fr.c1 = V3(); fr.angA = V3(); }
            fr.n2 = -cp.latdir;
            V3 ft = cross(rel_pos2, fr.n2);
            fr.c2 = ft;
            fr.angB = B.iinv_world * ft;
            {
                float denom0 = 0.f;
                if (A) { V3 vec = cross(fr.angA, rel_pos1); denom0 = A->inv_mass + dot(cp.latdir, vec); }
                V3 vec = cross(-fr.angB, rel_pos2);
                float denom1 = B.inv_mass + dot(cp.latdir, vec);
                fr.jd = SOR / (denom0 + denom1);
            }
            {
                float vel1Dotn = A ? dot(fr.n1, A->lin + A->extF) + dot(fr.c1, A->ang) : 0.f;  // no external torque impulse
                float vel2Dotn = dot(fr.n2, B.lin + B.extF) + dot(fr.c2, B.ang);
                float rv = vel1Dotn + vel2Dotn;
                float velocityError = 0.f - rv;
                fr.rhs = velocityError * fr.jd;
                fr.rhs_pen = 0.f; fr.cfm = 0.f;
                fr.lo = -fr.friction; fr.hi = fr.friction;
            }
            // setFrictionConstraintImpulse (warm start)
            fr.applied = cp.applied_lat * WARMSTART;
            if (A) apply_impulse(*A, fr.n1 * A->inv_mass, fr.angA, fr.applied);
            apply_impulse(B, -fr.n2 * B.inv_mass, -fr.angB, -fr.applied);
            frows.push_back(fr);
        }
    };
    for (int mi = 0; mi < nmember(); mi++) {
        Body &B = bodies[mi];
        if (!B.contact_response) continue;
        float combinedFriction = floor_friction * B.friction;
        if (combinedFriction < -10.f) combinedFriction = -10.f;
        if (combinedFriction > 10.f) combinedFriction = 10.f;
        convert_manifold(-1, mi, manifolds[mi], combinedFriction);
    }
    for (PairManifold &pm : pairs) convert_manifold(pm.a, pm.b, pm.mf, pm.friction);
    last_num_contacts = (int) crows.size();

    auto dl = [&](int i) -> V3 { return i >= 0 ? bodies[i].dlin : V3(); };
    auto da = [&](int i) -> V3 { return i >= 0 ? bodies[i].dang : V3(); };
    auto generic = [&](Row &c, bool lower_only) -> float {
        float deltaImpulse = c.rhs - c.applied * c.cfm;
        float d1 = dot(c.n1, dl(c.a)) + dot(c.c1, da(c.a));
        float d2 = dot(c.n2, dl(c.b)) + dot(c.c2, da(c.b));
        deltaImpulse -= d1 * c.jd;
        deltaImpulse -= d2 * c.jd;
        float sum = c.applied + deltaImpulse;
        if (sum < c.lo) { deltaImpulse = c.lo - c.applied; c.applied = c.lo; }
        else if (!lower_only && sum > c.hi) { deltaImpulse = c.hi - c.applied; c.applied = c.hi; }
        else c.applied = sum;
        if (c.a >= 0) apply_impulse(bodies[c.a], c.n1 * bodies[c.a].inv_mass, c.angA, deltaImpulse);
        if (c.b >= 0) apply_impulse(bodies[c.b], c.n2 * bodies[c.b].inv_mass, c.angB, deltaImpulse);
        return deltaImpulse * (1.f / c.jd);
    };

    // ---- split impulse iterations (penetration recovery, contacts only) ----
    for (int it = 0; it < NUM_ITER; it++) {
        float lsr = 0.f;
        for (Row &c : crows) {
            float deltaImpulse = 0.f;
            if (c.rhs_pen != 0.f) {
                deltaImpulse = c.rhs_pen - c.applied_push * c.cfm;
                float d1 = c.a >= 0 ? dot(c.n1, bodies[c.a].push) + dot(c.c1, bodies[c.a].turn) : 0.f;
                float d2 = dot(c.n2, bodies[c.b].push) + dot(c.c2, bodies[c.b].turn);
                deltaImpulse -= d1 * c.jd;
                deltaImpulse -= d2 * c.jd;
                float sum = c.applied_push + deltaImpulse;
                if (sum < c.lo) { deltaImpulse = c.lo - c.applied_push; c.applied_push = c.lo; }
                else c.applied_push = sum;
                if (c.a >= 0) apply_push(bodies[c.a], c.n1 * bodies[c.a].inv_mass, c.angA, deltaImpulse);
                apply_push(bodies[c.b], c.n2 * bodies[c.b].inv_mass, c.angB, deltaImpulse);
            }
            float res = deltaImpulse * (1.f / c.jd);
            lsr = std::fmax(lsr, res * res);
        }
        if (lsr <= 0.f || it >= NUM_ITER - 1) break;
    }

    // ---- velocity iterations ----
    for (int it = 0; it < NUM_ITER; it++) {
        float lsr = 0.f;
        for (Row &c : jrows) { float r = generic(c, false); lsr = std::fmax(lsr, r * r); }
        for (Row &c : crows) { float r = generic(c, true); lsr = std::fmax(lsr, r * r); }
        for (Row &c : frows) {
            float total = crows[c.fric_of].applied;
            if (total > 0.f) {
                c.lo = -(c.friction * total);
                c.hi = c.friction * total;
                float r = generic(c, false);
                lsr = std::fmax(lsr, r * r);
            }
        }
        last_residual = lsr;
        if (lsr <= 0.f || it >= NUM_ITER - 1) break;
    }

    // ---- finish ----
    for (size_t i = 0; i < crows.size(); i++) {
        ManifoldPoint &cp = cref[i].mf->p[cref[i].slot];
        cp.applied = crows[i].applied;
        cp.applied_lat = frows[i].applied;
    }
    for (size_t ci = 0; ci < order.size(); ci++) {
        if (num_rows[ci] == 0) continue;
        float last = jrows[first_row[ci] + num_rows[ci] - 1].applied;  // writeBackJoints: last row wins
        switch (order[ci].type) {
            case 0: hinges[order[ci].idx].applied = last; break;
            case 1: fixeds[order[ci].idx].applied = last; break;
            case 2: sliders[order[ci].idx].applied = last; break;
            default: p2ps[order[ci].idx].applied = last; break;
        }
    }
    for (Body &b : bodies) {
        // btSolverBody::writebackVelocityAndTransform
        V3 lin = b.lin + b.dlin;
        V3 ang = b.ang + b.dang;
        if (b.push.x != 0.f || b.push.y != 0.f || b.push.z != 0.f || b.turn.x != 0.f || b.turn.y != 0.f || b.turn.z != 0.f) {
            Q qn;
            b.xf = integrateTransform(b.xf, b.push, b.turn * SPLIT_TURN_ERP, DT, &qn);
            b.q = qn;
        }
        b.lin = lin + b.extF;
        b.ang = ang + b.extT;
    }
}

void World::integrate() {
    for (Body &b : bodies) {
        Q qn;
        Xf pred = integrateTransform(b.xf, b.lin, b.ang, DT, &qn);
        // proceedToTransform -> setCenterOfMassTransform: interpolation state := new state, updateInertiaTensor
        b.xf = pred;
        b.q = qn;
        b.iinv_world = inertia_world(b.xf.b, b.inv_inertia_local);
        // synchronizeSingleMotionState: integrateTransform(interp, v, w, localTime - fixedTimeStep), localTime == 0
        float t = 0.f - DT;
        b.ms_origin = b.xf.o + b.lin * t;
    }
    reset_pending = false;
}

void World::physics_step() {
    collide();
    collide_pairs();
    solve();
    integrate();
}

// ------------------------------------------------------------------------------------------------
// observation, reward, termination (robot_walk.cpp:56-74, proprioception_state.cpp:23-129)
// ------------------------------------------------------------------------------------------------
void World::compute_step(float *obs, float *reward, int *done) {
    int k = 0;
    const float PI_F = (float) M_PI;
    for (size_t si = 0; si < state_members.size(); si++) {
        int mi = state_members[si];
        Body &b = bodies[mi];
        float yaw, pitch, roll;
        getEulerZYX(quatFromMat(b.xf.b), yaw, pitch, roll);
        V3 lv = b.lin, av = b.ang;
        V3 la = last_lin[mi] - lv, aa = last_ang[mi] - av;
        last_lin[mi] = lv; last_ang[mi] = av;
        obs[k++] = yaw / PI_F; obs[k++] = pitch / PI_F; obs[k++] = roll / PI_F;
        obs[k++] = lv.x; obs[k++] = lv.y; obs[k++] = lv.z;
        obs[k++] = av.x / PI_F; obs[k++] = av.y / PI_F; obs[k++] = av.z / PI_F;
        obs[k++] = la.x; obs[k++] = la.y; obs[k++] = la.z;
        obs[k++] = aa.x / PI_F; obs[k++] = aa.y / PI_F; obs[k++] = aa.z / PI_F;
        obs[k++] = 0.f;  // floor_touched is never set after construction
        if (mi == root) {
            V3 c = b.xf.o;
            obs[k++] = std::log(length(c) + 1.f);
            obs[k++] = c.y;
            obs[k++] = std::atan2(c.z, c.x);
        } else {
            V3 c = b.ms_origin - bodies[root].ms_origin;
            obs[k++] = c.x; obs[k++] = c.y; obs[k++] = c.z;
        }
    }
    for (size_t i = 0; i < sliders.size(); i++) {
        obs[k++] = sliders[i].lin_pos;
        obs[k++] = sliders[i].applied;
        obs[k++] = p2ps[2 * i].applied;
        obs[k++] = p2ps[2 * i + 1].applied;
    }
    // robot_walk.cpp:61-68: the root's z velocity; robot_jump.cpp:71-80: max(vy, 0) + vz and a strict fail test
    float lin_vel_z = prm.env_kind == 1 ? std::max(bodies[root].lin.y, 0.f) + bodies[root].lin.z : bodies[root].lin.z;
    *reward = lin_vel_z;
    if (lin_vel_z < prm.minimal_velocity) remaining_steps -= 1;
    else if (lin_vel_z >= prm.target_velocity) remaining_steps += 1;
    bool win = curr_step >= max_steps;
    bool fail = prm.env_kind == 1 ? remaining_steps < 0 : remaining_steps <= 0;
    *done = (win | fail) ? 1 : 0;
    curr_step += 1;
}

// ------------------------------------------------------------------------------------------------
// reset (robot_walk.cpp:76-104, item.cpp:77-86)
// ------------------------------------------------------------------------------------------------
void World::reset_begin() {
    const V3 root_pos(1.f, 0.25f, 2.f);
    const float angle_limit = prm.env_kind == 1 ? (float) M_PI / 3.f : (float) M_PI * 2.f / 3.f;  // robot_jump.cpp:89
    float yaw = rng.uniform01() * angle_limit - angle_limit / 2.f;
    float roll = rng.uniform01() * angle_limit - angle_limit / 2.f;
    float pitch = rng.uniform01() * angle_limit - angle_limit / 2.f;
    rng_draws += 3;
    Xf model;
    model.b = glm_eulerAngleYXZ(yaw, pitch, roll);
    model.o = root_pos;
    for (Body &b : bodies) {
        b.xf = glm_mul(model, b.first_model);
        b.ms_origin = b.xf.o;
        b.lin = b.ang = V3();
        // NOTE: btRigidBody::setWorldTransform does not refresh m_invInertiaTensorWorld: the next step
        // runs on the previous transform's tensor.
    }
    for (Manifold &m : manifolds) m.n = 0;  // removeRigidBody/addRigidBody drops every persistent manifold
    for (PairManifold &pm : pairs) pm.mf.n = 0;
    reset_pending = true;
    reset_E = model.b;
}

void World::reset(float *obs, float *reward, int *done) {
    reset_begin();
    for (int i = 0; i < prm.reset_frames; i++) physics_step();
    curr_step = 0;
    remaining_steps = (int) (prm.initial_remaining_seconds / DT);
    if (prm.env_kind == 0)  // robot_walk.cpp:98-103 settles twice; robot_jump.cpp:104-107 once
        for (int i = 0; i < prm.reset_frames; i++) physics_step();
    compute_step(obs, reward, done);
}

void World::do_step(const float *action, float *obs, float *reward, int *done) {
    apply_action(action);
    physics_step();
    compute_step(obs, reward, done);
}

// ------------------------------------------------------------------------------------------------
// canonical state blob (layout documented in include/evomotion.h, EVM_STATE_*)
// ------------------------------------------------------------------------------------------------
int World::state_size() const {
    // (49 per member pair, right after the floor manifolds: count, 4 x (localA3 localB3 normalOnB3 dist applied applied_lateral))
    return 13 * nb() + 1 + 9 + 6 * nb() + 3 * nmember() + 6 * nmember() + 37 * nmember() + nmuscle() + 1 + 2 + 49 * npairs();
}
void World::get_state(float *o) const {
    int k = 0;
    for (const Body &b : bodies) {
        o[k++] = b.xf.o.x; o[k++] = b.xf.o.y; o[k++] = b.xf.o.z;
        o[k++] = b.q.x; o[k++] = b.q.y; o[k++] = b.q.z; o[k++] = b.q.w;
        o[k++] = b.lin.x; o[k++] = b.lin.y; o[k++] = b.lin.z;
        o[k++] = b.ang.x; o[k++] = b.ang.y; o[k++] = b.ang.z;
    }
    o[k++] = reset_pending ? 1.f : 0.f;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) o[k++] = reset_E.r[i][j];
    for (const Body &b : bodies) {
        o[k++] = b.iinv_world.r[0].x; o[k++] = b.iinv_world.r[0].y; o[k++] = b.iinv_world.r[0].z;
        o[k++] = b.iinv_world.r[1].y; o[k++] = b.iinv_world.r[1].z; o[k++] = b.iinv_world.r[2].z;
    }
    for (int i = 0; i < nmember(); i++) { o[k++] = bodies[i].ms_origin.x; o[k++] = bodies[i].ms_origin.y; o[k++] = bodies[i].ms_origin.z; }
    for (int i = 0; i < nmember(); i++) {
        o[k++] = last_lin[i].x; o[k++] = last_lin[i].y; o[k++] = last_lin[i].z;
        o[k++] = last_ang[i].x; o[k++] = last_ang[i].y; o[k++] = last_ang[i].z;
    }
    for (int i = 0; i < nmember(); i++) {
        const Manifold &m = manifolds[i];
        o[k++] = (float) m.n;
        for (int j = 0; j < 4; j++) {
            if (j < m.n) {
                const ManifoldPoint &p = m.p[j];
                o[k++] = p.localA.x; o[k++] = p.localA.y; o[k++] = p.localA.z;
                o[k++] = p.localB.x; o[k++] = p.localB.y; o[k++] = p.localB.z;
                o[k++] = p.dist; o[k++] = p.applied; o[k++] = p.applied_lat;
            } else for (int t = 0; t < 9; t++) o[k++] = 0.f;
        }
    }
    for (const PairManifold &pm : pairs) {
        o[k++] = (float) pm.mf.n;
        for (int j = 0; j < 4; j++) {
            if (j < pm.mf.n) {
                const ManifoldPoint &p = pm.mf.p[j];
                o[k++] = p.localA.x; o[k++] = p.localA.y; o[k++] = p.localA.z;
                o[k++] = p.localB.x; o[k++] = p.localB.y; o[k++] = p.localB.z;
                o[k++] = p.normalB.x; o[k++] = p.normalB.y; o[k++] = p.normalB.z;
                o[k++] = p.dist; o[k++] = p.applied; o[k++] = p.applied_lat;
            } else for (int t = 0; t < 12; t++) o[k++] = 0.f;
        }
    }
    for (const Slider &s : sliders) o[k++] = s.target_vel;
    o[k++] = (!sliders.empty() && sliders[0].powered) ? 1.f : 0.f;
    o[k++] = (float) curr_step;
    o[k++] = (float) remaining_steps;
}
void World::set_state(const float *in) {
    int k = 0;
    std::vector<V3> pos(nb());
    for (int i = 0; i < nb(); i++) {
        Body &b = bodies[i];
        pos[i] = V3(in[k], in[k + 1], in[k + 2]); k += 3;
        b.q = Q(in[k], in[k + 1], in[k + 2], in[k + 3]); k += 4;
        b.lin = V3(in[k], in[k + 1], in[k + 2]); k += 3;
        b.ang = V3(in[k], in[k + 1], in[k + 2]); k += 3;
    }
    reset_pending = in[k++] != 0.f;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) reset_E.r[i].at(j) = in[k++];
    for (int i = 0; i < nb(); i++) {
        Body &b = bodies[i];
        float xx = in[k], xy = in[k + 1], xz = in[k + 2], yy = in[k + 3], yz = in[k + 4], zz = in[k + 5];
        k += 6;
        if (reset_pending) {
            Xf model; model.b = reset_E; model.o = V3(1.f, 0.25f, 2.f);
            b.xf = glm_mul(model, b.first_model);
            b.iinv_world = M3(xx, xy, xz, xy, yy, yz, xz, yz, zz);
        } else {
            b.xf.b = matFromQuat(b.q);
            b.xf.o = pos[i];
            b.iinv_world = inertia_world(b.xf.b, b.inv_inertia_local);
        }
    }
    for (int i = 0; i < nmember(); i++) { bodies[i].ms_origin = V3(in[k], in[k + 1], in[k + 2]); k += 3; }
    for (int i = 0; i < nmember(); i++) {
        last_lin[i] = V3(in[k], in[k + 1], in[k + 2]); k += 3;
        last_ang[i] = V3(in[k], in[k + 1], in[k + 2]); k += 3;
    }
    for (int i = 0; i < nmember(); i++) {
        Manifold &m = manifolds[i];
        m.n = (int) in[k++];
        for (int j = 0; j < 4; j++) {
            ManifoldPoint &p = m.p[j];
            p.localA = V3(in[k], in[k + 1], in[k + 2]); k += 3;
            p.localB = V3(in[k], in[k + 1], in[k + 2]); k += 3;
            p.dist = in[k++]; p.applied = in[k++]; p.applied_lat = in[k++];
            p.normalB = V3(0.f, -1.f, 0.f);
        }
    }
    for (PairManifold &pm : pairs) {
        pm.mf.n = (int) in[k++];
        for (int j = 0; j < 4; j++) {
            ManifoldPoint &p = pm.mf.p[j];
            p.localA = V3(in[k], in[k + 1], in[k + 2]); k += 3;
            p.localB = V3(in[k], in[k + 1], in[k + 2]); k += 3;
            p.normalB = V3(in[k], in[k + 1], in[k + 2]); k += 3;
            p.dist = in[k++]; p.applied = in[k++]; p.applied_lat = in[k++];
        }
    }
    for (Slider &s : sliders) s.target_vel = in[k++];
    bool powered = in[k++] != 0.f;
    for (Slider &s : sliders) s.powered = powered;
    curr_step = (int) in[k++];
    remaining_steps = (int) in[k++];
}
void World::get_poses(float *o) const {
    int k = 0;
    for (const Body &b : bodies) {
        Q q = reset_pending ? quatFromMat(b.xf.b) : b.q;
        o[k++] = b.xf.o.x; o[k++] = b.xf.o.y; o[k++] = b.xf.o.z;
        o[k++] = q.x; o[k++] = q.y; o[k++] = q.z; o[k++] = q.w;
    }
}

}  // namespace orc
