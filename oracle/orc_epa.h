// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under evomotion_amd/ may include, link or call this.
//
// Penetration depth of two overlapping convex hulls the way the reference's world computes it: the world of
// evo_motion_model/src/environment.cpp:20-31 is built from btDefaultCollisionConfiguration, whose default
// btDefaultCollisionConstructionInfo::m_useEpaPenetrationAlgorithm = true hands btGjkEpaPenetrationDepthSolver to every
// btConvexConvexAlgorithm.
//
// [UPSTREAM — Bullet3, un-vendored and un-pinned (evo_motion_model/CMakeLists.txt:14); restated from the published bullet3 3.x
// sources as remembered, not compiled here]:
//   btGjkEpaPenetrationDepthSolver::calcPenDepth   nine guess vectors; Penetration() on the margin-inflated shapes, else Distance()
//                                                   on the cores
//   btGjkEpaSolver2::Penetration / Distance        (BulletCollision/NarrowPhaseCollision/btGjkEpa2.cpp)
//   gjkepa2_impl::MinkowskiDiff / GJK / EPA         the file's own GJK (projectorigin on 2/3/4 points, EncloseOrigin) and the
//                                                   expanding polytope: face list + stock list, findbest, expand (horizon),
//                                                   newface with getedgedist, at most EPA_MAX_VERTICES support points,
//                                                   2 x that many faces, EPA_MAX_ITERATIONS rounds
// Everything is computed in shape A's local frame, like the original; witnesses go back through A's world transform.
// Pointers of the original are indices here (vertex store / face store), list order and recursion order are kept, so that ties in
// findbest and the order in which horizon faces are made are the original's.
#pragma once
#include "orc_narrow.h"

namespace orc {

// single-precision constants of btGjkEpa2.cpp
static constexpr int GJK2_MAX_ITERATIONS = 128;
static constexpr float GJK2_ACCURACY = 0.0001f;
static constexpr float GJK2_MIN_DISTANCE = 0.0001f;
static constexpr float GJK2_DUPLICATED_EPS = 0.0001f;
static constexpr float GJK2_SIMPLEX2_EPS = 0.0f;
static constexpr float GJK2_SIMPLEX3_EPS = 0.0f;
static constexpr float GJK2_SIMPLEX4_EPS = 0.0f;
static constexpr int EPA_MAX_VERTICES = 128;
static constexpr int EPA_MAX_ITERATIONS = 255;
static constexpr float EPA_ACCURACY = 0.0001f;
static constexpr float EPA_PLANE_EPS = 0.00001f;
static constexpr int EPA_MAX_FACES = EPA_MAX_VERTICES * 2;

struct EpaResults {  // btGjkEpaSolver2::sResults
    enum Status { Separated, Penetrating, GJK_Failed, EPA_Failed };
    Status status = Separated;
    V3 witnesses[2];
    V3 normal;
    float distance = 0.f;
    // diagnostics
    int gjk_iterations = 0, epa_iterations = 0, epa_status = -1, epa_vertices = 0;
};

bool epa_penetration(const ConvexView &A, const Xf &wtrs0, const ConvexView &B, const Xf &wtrs1, const V3 &guess, EpaResults &results);
bool epa_distance(const ConvexView &A, const Xf &wtrs0, const ConvexView &B, const Xf &wtrs1, const V3 &guess, EpaResults &results);

// btGjkEpaPenetrationDepthSolver::calcPenDepth; returns Penetration()'s verdict, v = the solver's normal
bool epa_calc_pen_depth(const ConvexView &A, const ConvexView &B, const Xf &transA, const Xf &transB, V3 &v, V3 &witnessA,
                        V3 &witnessB, EpaResults *diag = nullptr);

}  // namespace orc
