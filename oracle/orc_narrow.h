// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under evomotion_amd/ may include, link or call this.
//
// Convex-convex narrowphase of the member-vs-member pairs the reference lets collide: every pair of members except
// constraint parent/child (evo_motion_model/src/robot/constraint.cpp:65,147 setIgnoreCollisionCheck; dispatcher and
// broadphase built in evo_motion_model/src/environment.cpp:20-31; shapes = btConvexHullShape of the OBJ vertices with
// local scaling, evo_motion_model/src/item.cpp:17-41).
//
// [UPSTREAM — Bullet3, un-vendored and un-pinned (evo_motion_model/CMakeLists.txt:14); restated from the published
// bullet3 3.x sources as remembered, not compiled here]:
//   btConvexConvexAlgorithm::processCollision     one btGjkPairDetector query per pair and step, result into a
//                                                  persistent manifold (no perturbation passes: the default
//                                                  btConvexConvexAlgorithm::CreateFunc has m_numPerturbationIterations = 0)
//   btGjkPairDetector::getClosestPointsNonVirtual  the main loop, its exits (m_degenerateSimplex codes), margins, the
//                                                  penetration branch and the final normal check
//   btVoronoiSimplexSolver                         closest(), closestPtPointTriangle / Tetrahedron, reduceVertices, inSimplex
//   btMinkowskiPenetrationDepthSolver              42 fixed directions + a second GJK on the displaced shape
//
// Stated deviations:
//   * penetration of the un-margined cores (deeper than both 0.04 margins together) is resolved with Bullet's
//     btMinkowskiPenetrationDepthSolver, not with btGjkEpaPenetrationDepthSolver, which is what
//     btDefaultCollisionConstructionInfo::m_useEpaPenetrationAlgorithm = true selects in the reference's configuration: the
//     sampled-direction solver is branch-free per direction and maps onto one-environment-per-lane execution, EPA's
//     growing polytope does not.  Both are Bullet's own answers to the same query; they differ in the direction found.
//   * the libccd-derived intersection pre-test at the top of getClosestPointsNonVirtual (status 0 forces the penetration
//     branch) is not restated: the Voronoi loop's own exits (a degenerate or full simplex, |v|^2 < REL_ERROR2) reach
//     the same branch for overlapping cores.
//   * the support vertex is the first maximum of dot(dir, scaled point) over the de-duplicated hull points in
//     first-occurrence order; Bullet takes dot(dir * scaling, unscaled point) over the duplicated list — the same vertex
//     whenever the products are exact (+-1 cube coordinates, power-of-two scalings: every shape of the reference's skeleton).
#pragma once
#include "orc_math.h"

namespace orc {

struct ConvexView {
    const V3 *pts;   // unique hull points, first-occurrence order (unscaled)
    int n;
    V3 scale;        // btCollisionShape local scaling
    Xf xf;           // world transform (basis may be non-orthonormal in the step that follows reset())
    float margin;    // CONVEX_DISTANCE_MARGIN = 0.04
};

struct ClosestResult {
    bool has = false;     // a point was handed to the manifold result (before btManifoldResult's own breaking-threshold test)
    V3 normalOnB;         // world, from B towards A
    V3 pointOnB;          // world
    float distance = 0;   // margins subtracted; negative = penetration
    // diagnostics
    int iterations = 0, degenerate = 0, method = -1;
    bool used_penetration = false;
    int ccd_status = -2, ccd_iterations = 0;   // the pre-test's verdict (0 intersect, -1 apart, -2 not run) and its loop trips
};

// which btConvexPenetrationDepthSolver the pair detector calls for overlapping cores: 0 = btGjkEpaPenetrationDepthSolver (orc_epa.cpp;
// what the reference's btDefaultCollisionConfiguration selects), 1 = btMinkowskiPenetrationDepthSolver (rounds 2-3 of this
// repository; kept so that tests can put a number on the difference)
extern int g_penetration_solver;
// the libccd-derived intersection pre-test at the top of btGjkPairDetector::getClosestPointsNonVirtual (bullet3 >= 2.88): 1 = run it
// and force the penetration branch when it reports intersecting cores (`status == 0`), 0 = leave it out (see the deviations above)
extern int g_ccd_pretest;

// btConvexHullShape::localGetSupportingVertexWithoutMargin
V3 local_support(const ConvexView &S, const V3 &dir);

// btGjkPairDetector::getClosestPoints with m_maximumDistanceSquared = max_dist2
ClosestResult gjk_closest_points(const ConvexView &A, const ConvexView &B, float max_dist2);

// btPolyhedralConvexAabbCachingShape::getAabb(t) + the gContactBreakingThreshold fattening of btCollisionWorld::updateSingleAabb
void world_aabb(const ConvexView &S, float contact_threshold, V3 &mn, V3 &mx);

}  // namespace orc
