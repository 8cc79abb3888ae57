"""Known-answer physics checks written ONCE against a tiny `World` interface and run on two backends:
the CPU oracle (tests/test_oracle_physics.py) and the HIP path through the C ABI (tests/test_gpu_physics.py).  They hold
both programs to physics — free fall, pendulum period, motor laws, momentum, Coulomb friction, resting contact — and not
only to each other.  Every world is set up through the canonical state blob (include/evomotion.h), so the two backends
start from bit-identical states."""
import math

import numpy as np

import blob as blobmod

DT = 1.0 / 60.0
G = 9.8
FLOOR_TOP = -1.0
MARGIN = 0.04


# ---- backends ---------------------------------------------------------------------------------------------------------
class OracleWorld:
    name = "oracle"

    def __init__(self, skeleton, seed=1, lib=None, self_collision=0):
        import orc
        self.e = orc.OracleEnv(seed=seed, skeleton=skeleton, lib=lib, self_collision=self_collision)
        self.nb, self.nm = self.e.nb, self.e.nm
        self.nmus = self.e.act_dim
        self.npairs = len(self.e.pairs())
        self.e.reset_begin()

    def state(self):
        return self.e.get_state()

    def set_state(self, s):
        self.e.set_state(s)

    def step(self, n=1):
        self.e.physics_step(n)

    def do_step(self, action):
        return self.e.do_step(np.asarray(action, np.float32))[0]

    def body_constants(self):
        return self.e.body_constants()


class HipWorld:
    name = "hip"

    def __init__(self, skeleton, seed=1, lib=None, self_collision=0):
        from evomotion_amd import VecRobotWalk
        self.env = VecRobotWalk(1, seed=seed, device=0, parameters={"skeleton_json_path": skeleton, "self_collision": self_collision})
        self.nb, self.nm, self.nmus = self.env.n_bodies, self.env.n_members, self.env.action_dim
        self.npairs = self.env.n_pairs
        self.env.debug_reset_begin()

    def state(self):
        return self.env.get_state()[0]

    def set_state(self, s):
        self.env.set_state(np.asarray(s, np.float32)[None])

    def step(self, n=1):
        self.env.debug_physics_steps(n)

    def do_step(self, action):
        import torch
        st = self.env.do_step(torch.from_numpy(np.asarray(action, np.float32)[None]))
        return st.state.cpu().numpy()[0]

    def body_constants(self):
        return self.env.body_constants()


# ---- state helpers ----------------------------------------------------------------------------------------------------
def fields(w):
    return blobmod.fields(w.nb, w.nm, w.nmus, getattr(w, "npairs", 0))


def clean_state(w, pos, quat=None, lin=None, ang=None):
    """A settled (not reset-pending) world state: bodies at `pos` [nb,3] with quaternions `quat` [nb,4] (x,y,z,w; identity
    by default), velocities lin / ang, no cached contacts, motors unpowered, motion states = positions."""
    s = w.state().copy()
    f = fields(w)
    b = s[f["bodies"]].reshape(w.nb, 13)
    b[:, 0:3] = np.asarray(pos, np.float32)
    b[:, 3:7] = np.asarray(quat, np.float32) if quat is not None else np.array([0, 0, 0, 1], np.float32)
    b[:, 7:10] = 0 if lin is None else np.asarray(lin, np.float32)
    b[:, 10:13] = 0 if ang is None else np.asarray(ang, np.float32)
    s[f["pending"]] = 0
    s[f["E"]] = np.eye(3, dtype=np.float32).ravel()
    s[f["ms"]] = b[: w.nm, 0:3].ravel()
    s[f["hist"]] = 0
    m = s[f["manifold"]].reshape(w.nm, 37)
    m[:] = 0
    s[f["target"]] = 0
    s[f["powered"]] = 0
    s[f["pairs"]] = 0
    w.set_state(s)
    return s


def bodies(w):
    return w.state()[fields(w)["bodies"]].reshape(w.nb, 13).astype(np.float64)


def manifold_counts(w):
    return w.state()[fields(w)["manifold"]].reshape(w.nm, 37)[:, 0]


def pair_manifolds(w):
    """[npairs, 49]: count, then 4 x (localA3 localB3 normalOnB3 dist applied applied_lateral)"""
    return w.state()[fields(w)["pairs"]].reshape(w.npairs, 49).astype(np.float64)


def quat_z(theta):
    return np.array([0.0, 0.0, math.sin(theta / 2), math.cos(theta / 2)])


def rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


# ---- skeletons --------------------------------------------------------------------------------------------------------
def skel_cube(write_skeleton, tmp_path, scale=(0.2, 0.2, 0.2), mass=1.0, friction=0.5, name="cube.skel"):
    return write_skeleton(tmp_path / name, [dict(name="body", mass=mass, scale=scale, friction=friction)])


# ---- the checks -------------------------------------------------------------------------------------------------------
def check_free_fall(w):
    """semi-implicit Euler: v += g dt first, then x += v dt with the NEW velocity (btRigidBody::integrateVelocities then
    predictIntegratedTransform); nothing else acts on a lone body far above the floor"""
    clean_state(w, [[0.0, 50.0, 0.0]])
    y, v = 50.0, 0.0
    for k in range(15):
        w.step()
        b = bodies(w)
        v += float(np.float32(-G) * np.float32(DT))
        y += v * DT
        assert abs(b[0, 8] - v) < 2e-6 and abs(b[0, 1] - y) < 2e-5
        assert abs(b[0, 7]) < 1e-7 and abs(b[0, 9]) < 1e-7


def check_momentum_free_spinning_box(w, steps=600):
    """a free box (unequal inertia, spinning about a non-principal axis) over 600 steps: horizontal linear momentum exactly
    constant, vertical = free fall, and the world angular momentum R I R^T w constant — Bullet's implicit gyroscopic term
    (BT_ENABLE_GYROSCOPIC_FORCE_IMPLICIT_BODY) is what keeps it from drifting"""
    c = w.body_constants()
    inv_i = c[0, 2:5].astype(np.float64)
    iloc = 1.0 / inv_i
    v0 = np.array([0.7, 0.0, -0.3])
    w0 = np.array([2.0, 1.0, 0.5])
    clean_state(w, [[0.0, 4000.0, 0.0]], lin=[v0], ang=[w0])
    b = bodies(w)
    L0 = rot(b[0, 3:7]) @ (iloc * (rot(b[0, 3:7]).T @ b[0, 10:13]))
    w.step(steps)
    b = bodies(w)
    assert np.isfinite(b).all()
    assert abs(b[0, 7] - v0[0]) < 1e-6 and abs(b[0, 9] - v0[2]) < 1e-6           # no horizontal force
    assert abs(b[0, 8] - (-G * DT * steps)) < 2e-3 * G * DT * steps                   # fp32 accumulation of g dt
    R = rot(b[0, 3:7])
    L1 = R @ (iloc * (R.T @ b[0, 10:13]))
    drift = np.linalg.norm(L1 - L0) / np.linalg.norm(L0)
    e0 = 0.5 * np.dot(w0, iloc * w0)
    wl = R.T @ b[0, 10:13]
    e1 = 0.5 * np.dot(wl, iloc * wl)
    # the implicit (backward-Euler) gyroscopic step damps the rotational energy slightly and keeps |L| to a few percent
    assert drift < 0.08, drift
    assert e1 <= e0 * 1.001 and e1 > 0.8 * e0, (e0, e1)
    return drift


def check_resting_box(w, half=0.2):
    """a cube set down just above its resting height: comes to rest at floor top + both 0.04 margins + half extent, with a
    full four-point persistent manifold"""
    rest = FLOOR_TOP + 2 * MARGIN + half
    clean_state(w, [[0.0, rest + 0.05, 0.0]])
    w.step(300)
    b = bodies(w)
    assert abs(b[0, 1] - rest) < 6e-3, b[0, 1]
    assert np.abs(b[0, 7:10]).max() < 2e-2 and np.abs(b[0, 10:13]).max() < 5e-2
    assert int(manifold_counts(w)[0]) == 4
    return float(b[0, 1])


def check_sliding_friction(w, half_y=0.1, v0=2.0, mass=1.0):
    """a flat box sliding on the floor, Coulomb friction with mu = 0.5 (box) x 0.5 (floor) = 0.25:
    (1) from the settled four-point manifold the first sliding step takes exactly mu g dt off the velocity;
    (2) while it slides every contact point's friction impulse is saturated at mu x its normal impulse (along the point's
        own sliding direction);
    (3) it stops near v0^2 / (2 mu g).  (Only near: at 3 cm per step the cached points drift past the manifold's breaking
        threshold every step, the box rides on the one fresh deepest-vertex point, rocks and sinks ~2 cm — the normal impulses
        of a step are then not m g dt.  That is the contact model's behaviour — the reference's convex-hull pair adds one point
        per step too, SURVEY App. B.7 — not a friction error, and (2) holds through all of it.)"""
    mu = 0.25
    rest = FLOOR_TOP + 2 * MARGIN + half_y
    clean_state(w, [[0.0, rest, 0.0]])
    w.step(120)  # settle: four-point manifold, normal impulses converged
    assert int(manifold_counts(w)[0]) == 4
    s = w.state().copy()
    f = fields(w)
    b = s[f["bodies"]].reshape(w.nb, 13)
    x_start = float(b[0, 0])
    b[0, 7] = v0
    w.set_state(s)
    prev = v0
    first = None
    for k in range(70):
        w.step()
        vx = bodies(w)[0, 7]
        m = w.state()[f["manifold"]].reshape(w.nm, 37)
        n = int(m[0, 0])
        pts = m[0, 1:].reshape(4, 9)[:n].astype(np.float64)
        if k == 0:
            first = prev - vx
            assert abs(first - mu * G * DT) < 5e-3 * mu * G * DT, (first, mu * G * DT)          # (1)
        live = pts[pts[:, 7] > 1e-6] if n > 0 else pts   # a point whose normal impulse ended at 0 has no friction row
        if vx > 0.3 and len(live):                          # (SolverMode: friction rows run only when the normal impulse > 0)
            assert np.all(np.abs(live[:, 8]) <= mu * live[:, 7] * (1 + 1e-5) + 1e-7)             # Coulomb cone
            assert abs(np.abs(live[:, 8]).sum() - mu * live[:, 7].sum()) < 1e-6 + 1e-4 * live[:, 7].sum()  # (2) saturated
        prev = vx
    assert abs(prev) < 2e-2                                                                      # it has stopped
    dist = bodies(w)[0, 0] - x_start
    ideal = v0 * v0 / (2 * mu * G)
    assert 0.85 * ideal < dist < 1.05 * ideal, (dist, ideal)                                     # (3)
    return first, dist, ideal


PEND_L = 0.5
PEND_PIVOT_LOCAL = (1.3, 0.9, 0.0)


def skel_pendulum(write_skeleton, tmp_path):
    """a heavy base standing on the floor and a light bob hanging from a z-axis hinge on its side"""
    base_h = 1.0
    base_y = FLOOR_TOP + 2 * MARGIN + base_h
    members = [dict(name="body", mass=2000.0, scale=(1.0, base_h, 1.0)),
               dict(name="bob", mass=0.5, t=(PEND_PIVOT_LOCAL[0], PEND_PIVOT_LOCAL[1] - PEND_L, 0.0), scale=(0.05, 0.05, 0.05))]
    cons = [dict(type="hinge", name="h", parent="body", child="bob", pivot_p=PEND_PIVOT_LOCAL, pivot_c=(0.0, PEND_L, 0.0),
                 axis_p=(0, 0, 1), axis_c=(0, 0, 1), lo=-3.0, hi=3.0)]
    return write_skeleton(tmp_path / "pendulum.skel", members, cons), base_y


def check_pendulum_period(w, base_y, theta0=0.1):
    """small-angle period of a physical pendulum: T = 2 pi sqrt((I_cm + m d^2) / (m g d)), I_cm from the loader's own
    (margin-inflated box) inertia; measured from the zero crossings of the swing angle over several periods"""
    c = w.body_constants()
    m = float(c[1, 0])
    i_cm = 1.0 / float(c[1, 4])  # about z
    t_pred = 2 * math.pi * math.sqrt((i_cm + m * PEND_L ** 2) / (m * G * PEND_L))
    base = np.array([0.0, base_y, 0.0])
    pivot = base + np.array(PEND_PIVOT_LOCAL)
    q = quat_z(theta0)
    bob = pivot + rot(q) @ np.array([0.0, -PEND_L, 0.0])
    clean_state(w, [base, bob], quat=[[0, 0, 0, 1], q])
    ang = []
    for _ in range(420):
        w.step()
        b = bodies(w)
        d = b[1, 0:3] - (b[0, 0:3] + rot(b[0, 3:7]) @ np.array(PEND_PIVOT_LOCAL))
        ang.append(math.atan2(d[0], -d[1]))
        assert abs(np.linalg.norm(d) - PEND_L) < 5e-3      # the hinge holds the bob on its circle
    ang = np.array(ang)
    assert ang.max() < 1.05 * theta0 and ang.min() > -1.05 * theta0   # no energy gain
    # upward zero crossings, linearly interpolated
    t = []
    for k in range(1, len(ang)):
        if ang[k - 1] < 0 <= ang[k]:
            t.append((k - 1 + ang[k - 1] / (ang[k - 1] - ang[k])) * DT)
    assert len(t) >= 4
    period = (t[-1] - t[0]) / (len(t) - 1)
    assert abs(period - t_pred) < 0.01 * t_pred, (period, t_pred)
    return period, t_pred


def skel_two_masses(write_skeleton, tmp_path, mass, force, name):
    """two equal cubes one metre apart on the x axis, joined by ONE muscle attached at both centres of mass: the slider
    axis is sphere A's x axis = the line of centres, no torque on either member"""
    members = [dict(name="body", mass=mass, scale=(0.2, 0.2, 0.2)),
               dict(name="other", mass=mass, t=(1.0, 0.0, 0.0), scale=(0.2, 0.2, 0.2))]
    mus = [dict(name="m0", a="body", b="other", pos_a=(0, 0, 0), pos_b=(0, 0, 0), force=force, speed=8.0)]
    return write_skeleton(tmp_path / name, members, [], mus)


def _muscle_world(w):
    pos = np.zeros((w.nb, 3))
    pos[:, 1] = 2000.0          # free fall, far from the floor: gravity acts on all four bodies alike
    pos[1, 0] = 1.0             # member "other"
    pos[3, 0] = 1.0             # its attach sphere (bodies: members, then per muscle sphere a, sphere b)
    clean_state(w, pos)


def check_motor_reaches_target_velocity(w, action=0.25):
    """force-unlimited slider motor: the relative velocity of the two attach points along the axis reaches
    action x speed (8 m/s) — Muscle::contract -> setTargetLinMotorVelocity (muscle.cpp:82-85)"""
    _muscle_world(w)
    for _ in range(6):
        w.do_step([action])
    b = bodies(w)
    rel = b[3, 7] - b[2, 7]      # sphere b - sphere a along x
    assert abs(rel - action * 8.0) < 0.02 * abs(action) * 8.0, rel
    rel_m = b[1, 7] - b[0, 7]    # the members follow their spheres (p2p joints)
    assert abs(rel_m - action * 8.0) < 0.05 * abs(action) * 8.0, rel_m
    return rel


def check_motor_saturates_at_max_force(w, steps=30, mass=1000.0, sphere_mass=0.1875, force=64.0):
    """64 N is far too little to reach 8 m/s on 1000 kg: the motor row saturates at max_force x dt per step and the two halves
    accelerate apart at F / m each"""
    _muscle_world(w)
    for _ in range(steps):
        w.do_step([1.0])
    b = bodies(w)
    rel = b[1, 7] - b[0, 7]
    pred = 2.0 * force / (mass + sphere_mass) * steps * DT
    assert abs(rel - pred) < 0.02 * pred, (rel, pred)
    assert abs((b[0, 7] + b[1, 7]) * mass + (b[2, 7] + b[3, 7]) * sphere_mass) < 1e-3 * mass * pred   # no net momentum
    return rel, pred


# ---- second batch: welded pair, hinge limit, impact, static friction -----------------------------------------------------------
def skel_welded_pair(write_skeleton, tmp_path):
    """two unequal boxes joined by ONE fixed constraint whose frame sits between them"""
    members = [dict(name="body", mass=2.0, scale=(0.2, 0.1, 0.15)),
               dict(name="other", mass=0.5, t=(0.6, 0.0, 0.0), scale=(0.1, 0.1, 0.1))]
    cons = [dict(type="fixed", name="f", parent="body", child="other", tp=(0.3, 0.0, 0.0), tc=(-0.3, 0.0, 0.0))]
    return write_skeleton(tmp_path / "welded.skel", members, cons)


def check_welded_pair_moves_as_one_body(w, steps=240):
    """a fixed constraint transmits only internal forces: the pair's total linear momentum follows free fall exactly (Newton's
    third law row by row), the relative pose of the two boxes stays what the constraint frames say, and the spin is shared"""
    c = w.body_constants()
    m = c[:2, 0].astype(np.float64)
    v0 = np.array([0.4, 0.0, -0.2])
    w0 = np.array([0.0, 1.5, 0.8])
    pos = np.array([[0.0, 3000.0, 0.0], [0.6, 3000.0, 0.0]])
    com = (m[:, None] * pos).sum(0) / m.sum()
    lin = [v0 + np.cross(w0, p - com) for p in pos]          # rigid rotation about the common centre of mass
    clean_state(w, pos, lin=lin, ang=[w0, w0])
    p0 = (m[:, None] * np.array(lin)).sum(0)
    w.step(steps)
    b = bodies(w)
    assert np.isfinite(b).all()
    p1 = (m[:, None] * b[:2, 7:10]).sum(0)
    assert abs(p1[0] - p0[0]) < 2e-5 * m.sum() and abs(p1[2] - p0[2]) < 2e-5 * m.sum()       # no external horizontal force
    assert abs(p1[1] - (p0[1] - m.sum() * G * DT * steps)) < 2e-3 * m.sum() * G * DT * steps   # gravity only (fp32 accumulation)
    # relative pose: child's frame origin seen from the parent
    Rp = rot(b[0, 3:7])
    rel = Rp.T @ (b[1, 0:3] - b[0, 0:3])
    assert np.abs(rel - np.array([0.6, 0.0, 0.0])).max() < 2e-3, rel
    Rrel = Rp.T @ rot(b[1, 3:7])
    angle = 0.5 * math.sqrt((Rrel[2, 1] - Rrel[1, 2]) ** 2 + (Rrel[0, 2] - Rrel[2, 0]) ** 2 + (Rrel[1, 0] - Rrel[0, 1]) ** 2)  # sin(angle)
    assert angle < 5e-3, angle
    assert np.abs(b[0, 10:13] - b[1, 10:13]).max() < 2e-2                                      # one angular velocity
    return float(np.abs(rel - np.array([0.6, 0.0, 0.0])).max()), angle


def skel_limited_pendulum(write_skeleton, tmp_path, lim=0.3):
    base_h = 1.0
    base_y = FLOOR_TOP + 2 * MARGIN + base_h
    members = [dict(name="body", mass=2000.0, scale=(1.0, base_h, 1.0)),
               dict(name="bob", mass=0.5, t=(PEND_PIVOT_LOCAL[0], PEND_PIVOT_LOCAL[1] - PEND_L, 0.0), scale=(0.05, 0.05, 0.05))]
    cons = [dict(type="hinge", name="h", parent="body", child="bob", pivot_p=PEND_PIVOT_LOCAL, pivot_c=(0.0, PEND_L, 0.0),
                 axis_p=(0, 0, 1), axis_c=(0, 0, 1), lo=-lim, hi=lim)]
    return write_skeleton(tmp_path / "pendulum_limited.skel", members, cons), base_y


def check_hinge_limit_holds(w, base_y, lim=0.3):
    """a pendulum thrown at its hinge limit: the swing angle never passes the limit by more than the velocity-level row's
    one-step overshoot, and the bob stays on its circle"""
    base = np.array([0.0, base_y, 0.0])
    pivot = base + np.array(PEND_PIVOT_LOCAL)
    bob = pivot + np.array([0.0, -PEND_L, 0.0])
    omega = 3.0                                   # rad/s about z: would reach ~0.7 rad without the limit
    clean_state(w, [base, bob], lin=[[0, 0, 0], [omega * PEND_L, 0.0, 0.0]], ang=[[0, 0, 0], [0.0, 0.0, omega]])
    worst, reached = 0.0, False
    for _ in range(240):
        w.step()
        b = bodies(w)
        d = b[1, 0:3] - (b[0, 0:3] + rot(b[0, 3:7]) @ np.array(PEND_PIVOT_LOCAL))
        ang = math.atan2(d[0], -d[1])
        worst = max(worst, abs(ang))
        reached = reached or abs(ang) > 0.9 * lim
        assert abs(np.linalg.norm(d) - PEND_L) < 1e-2
    assert reached                                 # it did run into the limit
    assert worst < lim + omega * DT + 0.02, worst  # at most one step's travel (+ the limit's softness) beyond it
    return worst


def check_impact_does_not_bounce(w, half=0.2):
    """restitution is zero on both sides: whatever the impact speed (3.1 and 6.3 m/s here), the only upward velocity a landing
    box ever gets is the solver's position correction of a shallow penetration, erp x depth / dt with erp = 0.2 and depth below
    the 0.04 m split-impulse threshold (deeper penetrations are pushed out without touching the velocity): under 0.48 m/s,
    not proportional to the impact; and the box ends at its resting height"""
    rest = FLOOR_TOP + 2 * MARGIN + half
    bound = 0.2 * 0.04 / DT + 0.02
    out = []
    for drop in (0.5, 2.0):
        clean_state(w, [[0.0, rest + drop, 0.0]])
        low, up, hit = 1e9, 0.0, False
        for k in range(200):
            w.step()
            b = bodies(w)
            if int(manifold_counts(w)[0]) > 0:
                hit = True
            if hit:
                up = max(up, b[0, 8])
                low = min(low, b[0, 1])
        assert hit
        assert up < bound, (drop, up, bound)
        assert low > rest - 0.12, (low, rest)          # one step of travel at the impact speed, at most
        b = bodies(w)
        assert abs(b[0, 1] - rest) < 8e-3 and np.abs(b[0, 7:10]).max() < 3e-2
        out.append((math.sqrt(2 * G * drop), up, low - rest))
    return out


def check_static_friction_holds(w, half_y=0.1):
    """inside the Coulomb cone the friction rows cancel a small sideways velocity completely in ONE step: a resting box
    kicked with less than mu g dt stays where it is"""
    mu = 0.25
    rest = FLOOR_TOP + 2 * MARGIN + half_y
    clean_state(w, [[0.0, rest, 0.0]])
    w.step(120)
    assert int(manifold_counts(w)[0]) == 4
    s = w.state().copy()
    f = fields(w)
    b = s[f["bodies"]].reshape(w.nb, 13)
    x0 = float(b[0, 0])
    kick = 0.5 * mu * G * DT
    b[0, 7] = kick
    w.set_state(s)
    w.step()
    b1 = bodies(w)
    assert abs(b1[0, 7]) < 0.05 * kick, (b1[0, 7], kick)
    w.step(30)
    b2 = bodies(w)
    assert abs(b2[0, 0] - x0) < 2 * kick * DT + 1e-4 and abs(b2[0, 7]) < 0.05 * kick
    return float(b1[0, 7]), kick


def check_slider_limits(w, steps=40):
    """the muscle's slider runs between 0 and twice the initial distance of its attach points (muscle.cpp:43-49): a
    force-unlimited motor driven at +8 m/s ends at 2 m, driven back at -8 m/s ends at 0, and never passes either limit — the
    motor's target speed is cut to what still fits before the stop (the approach is geometric, ratio 1 - erp = 0.8 per step)"""
    _muscle_world(w)
    seps = []
    for _ in range(steps):
        w.do_step([1.0])
        b = bodies(w)
        seps.append(b[3, 0] - b[2, 0])
    hi = max(seps)
    assert hi < 2.0 + 2e-3, hi
    assert abs(seps[-1] - 2.0) < 5e-3, seps[-1]                   # held at the upper limit
    near = [2.0 - x for x in seps if 0.005 < 2.0 - x < 0.2]
    ratios = [b_ / a_ for a_, b_ in zip(near[:-1], near[1:])]
    assert len(ratios) >= 5 and all(abs(r - 0.8) < 0.03 for r in ratios), ratios
    for _ in range(steps):
        w.do_step([-1.0])
        b = bodies(w)
        seps.append(b[3, 0] - b[2, 0])
    lo = min(seps)
    assert lo > -2e-3, lo
    assert abs(seps[-1]) < 5e-3, seps[-1]                         # held at the lower limit
    return hi, lo


def check_hinge_removes_off_axis_rotation(w, base_y):
    """a hinge leaves ONE relative rotation free: a bob kicked about an axis perpendicular to the hinge axis loses that spin to
    the two angular rows at once (the 2000 kg base barely moves), and the two hinge axes stay parallel"""
    base = np.array([0.0, base_y, 0.0])
    pivot = base + np.array(PEND_PIVOT_LOCAL)
    bob = pivot + np.array([0.0, -PEND_L, 0.0])
    clean_state(w, [base, bob], ang=[[0, 0, 0], [2.0, 1.0, 0.0]])
    w.step()
    b = bodies(w)
    rel = b[1, 10:13] - b[0, 10:13]
    assert abs(rel[0]) < 2e-2 and abs(rel[1]) < 2e-2, rel      # gone after one step
    worst = 0.0
    for _ in range(120):
        w.step()
        b = bodies(w)
        za, zb = rot(b[0, 3:7])[:, 2], rot(b[1, 3:7])[:, 2]
        worst = max(worst, 1.0 - float(np.dot(za, zb)))
    assert worst < 1e-4, worst
    return float(np.abs(rel[:2]).max()), worst


# ---- member-vs-member contacts (EvmEnvParams::self_collision; the reference lets every pair of members collide except
# constraint parent / child: constraint.cpp:65,147) ----------------------------------------------------------------------
def skel_two_free_boxes(write_skeleton, tmp_path, half_a=(0.5, 0.2, 0.5), half_b=(0.15, 0.15, 0.15), mass_a=4.0, mass_b=0.5,
                        name="two_boxes.skel"):
    """two members and no constraint between them: one collidable pair"""
    members = [dict(name="body", mass=mass_a, scale=half_a), dict(name="other", mass=mass_b, t=(0.0, 1.0, 0.0), scale=half_b)]
    return write_skeleton(tmp_path / name, members)


def check_box_rests_on_box(w, half_a=(0.5, 0.2, 0.5), half_b=(0.15, 0.15, 0.15)):
    """a small box set down on a big one that lies on the floor: both hulls carry the 0.04 margin, so the cores come to rest
    0.08 apart (exactly like a box on the floor), the pair's persistent manifold fills to four points (one new point per step,
    like every convex-convex pair of the reference), and nothing sinks or drifts"""
    assert w.npairs == 1
    ya = FLOOR_TOP + 2 * MARGIN + half_a[1]
    yb = ya + half_a[1] + 2 * MARGIN + half_b[1]
    clean_state(w, [[0.0, ya, 0.0], [0.1, yb + 0.03, -0.05]])
    w.step(400)
    b = bodies(w)
    assert np.isfinite(b).all()
    assert abs(b[0, 1] - ya) < 8e-3, (b[0, 1], ya)
    gap = (b[1, 1] - half_b[1]) - (b[0, 1] + half_a[1])
    assert abs(gap - 2 * MARGIN) < 8e-3, gap
    assert np.abs(b[:2, 7:10]).max() < 3e-2 and np.abs(b[:2, 10:13]).max() < 8e-2
    assert abs(b[1, 0] - 0.1) < 2e-2 and abs(b[1, 2] + 0.05) < 2e-2          # friction holds it where it was set down
    pm = pair_manifolds(w)
    assert int(pm[0, 0]) == 4, pm[0, 0]
    pts = pm[0, 1:].reshape(4, 12)
    assert np.all(pts[:, 7] < -0.999)                                         # normal on B (the upper box) points down, at A
    assert np.all(pts[:, 10] >= 0) and pts[:, 10].sum() > 0                   # normal impulses push, never pull
    # the four normal impulses of a step carry the upper box's weight
    mb = float(w.body_constants()[1, 0])
    assert abs(pts[:, 10].sum() - mb * G * DT) < 0.05 * mb * G * DT, (pts[:, 10].sum(), mb * G * DT)
    return gap, pts[:, 10].sum()


def check_free_boxes_collide_inelastically(w, half=0.15, mass_a=4.0, mass_b=0.5):
    """two boxes meet head-on far above the floor: every contact row pushes the two bodies with equal and opposite impulses, so
    the total momentum is the free-fall momentum to rounding; restitution is zero, so along the contact normal they separate no
    faster than the solver's position correction; and the cores never get closer than the two margins allow (minus one step of
    approach)"""
    v = 1.5
    pos = np.array([[0.0, 3000.0, 0.0], [1.2, 3000.0, 0.02]])
    lin = np.array([[0.0, 0.0, 0.0], [-v, 0.0, 0.0]])
    clean_state(w, pos, lin=lin)
    m = np.array([mass_a, mass_b])
    p0 = (m[:, None] * lin).sum(0)
    touched, min_gap = False, 1e9
    hx_a = float(w.body_constants()[0, 0]) and 0.5   # half extent of "body" along x (skel_two_free_boxes default)
    for k in range(90):
        w.step()
        b = bodies(w)
        gap = (b[1, 0] - half) - (b[0, 0] + hx_a)
        min_gap = min(min_gap, gap)
        touched = touched or int(pair_manifolds(w)[0, 0]) > 0
        p = (m[:, None] * b[:2, 7:10]).sum(0)
        assert abs(p[0] - p0[0]) < 2e-5 * m.sum() and abs(p[2] - p0[2]) < 2e-5 * m.sum(), (k, p, p0)
    assert touched
    assert min_gap > 2 * MARGIN - v * DT - 5e-3, min_gap                      # never deeper than one step of approach
    b = bodies(w)
    rel = b[1, 7] - b[0, 7]                                                    # separation speed along x afterwards
    assert -1e-3 < rel < 0.2 * 0.04 / DT + 0.02, rel                           # no bounce beyond the erp correction
    v_common = p0[0] / m.sum()
    assert abs(b[0, 7] - v_common) < 0.12 and abs(b[1, 7] - v_common) < 0.6    # both near the common velocity
    return min_gap, rel


def skel_folding_arm(write_skeleton, tmp_path):
    """body - upper - lower chained by two z hinges, laid out straight along +x: `lower` (the forearm, 1 m long) and `body` share
    no constraint, so they may collide when the elbow folds the forearm back over the 0.8 m upper arm"""
    members = [dict(name="body", mass=50.0, scale=(0.3, 0.3, 0.3)),
               dict(name="upper", mass=0.5, t=(0.7, 0.0, 0.0), scale=(0.4, 0.05, 0.05)),
               dict(name="lower", mass=0.5, t=(1.6, 0.0, 0.0), scale=(0.5, 0.05, 0.05))]
    cons = [dict(type="hinge", name="h0", parent="body", child="upper", pivot_p=(0.3, 0.0, 0.0), pivot_c=(-0.4, 0.0, 0.0),
                 axis_p=(0, 0, 1), axis_c=(0, 0, 1), lo=-0.1, hi=0.1),
            dict(type="hinge", name="h1", parent="upper", child="lower", pivot_p=(0.4, 0.06, 0.0), pivot_c=(-0.5, 0.06, 0.0),
                 axis_p=(0, 0, 1), axis_c=(0, 0, 1), lo=-3.1, hi=3.1)]
    return write_skeleton(tmp_path / "folding_arm.skel", members, cons)


def check_folded_arm_stops_at_the_body(w, omega=2.0):
    """the forearm is swung about its elbow, up and over, back towards the body.  With member-vs-member contacts its tip
    lands on the body's top face and stays outside it — cores 0.08 apart, less one step of approach; with floor contacts only
    it sinks into the body until the elbow's own limit stops it"""
    y0 = 3000.0
    pos = np.array([[0.0, y0, 0.0], [0.7, y0, 0.0], [1.6, y0, 0.0]])
    elbow = np.array([1.1, y0 + 0.06, 0.0])
    lin = np.zeros((3, 3))
    lin[2] = np.cross([0, 0, omega], pos[2] - elbow)
    ang = np.zeros((3, 3))
    ang[2] = [0, 0, omega]
    clean_state(w, pos, lin=lin, ang=ang)
    # sample the forearm's box densely (the closest feature is the body's top edge against the forearm's underside)
    gx, gy, gz = np.meshgrid(np.linspace(-1, 1, 81), np.linspace(-1, 1, 5), np.linspace(-1, 1, 5), indexing="ij")
    corners = np.stack([gx.ravel(), gy.ravel(), gz.ravel()], 1)
    ha, hc = np.array([0.3, 0.3, 0.3]), np.array([0.5, 0.05, 0.05])
    deepest, touched = 1e9, False
    for k in range(170):
        w.step()
        b = bodies(w)
        assert np.isfinite(b).all()
        # signed distance of the forearm's sample points to the body's box (in the body's frame)
        Ra, Rc = rot(b[0, 3:7]), rot(b[2, 3:7])
        pc = (Rc @ (corners * hc).T).T + b[2, 0:3]
        loc = (Ra.T @ (pc - b[0, 0:3]).T).T
        q = np.abs(loc) - ha
        sd = np.linalg.norm(np.maximum(q, 0), axis=1) + np.minimum(q.max(1), 0)
        deepest = min(deepest, sd.min())
        if w.npairs:
            touched = touched or pair_manifolds(w)[:, 0].sum() > 0
    return deepest, touched
