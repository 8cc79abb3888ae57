"""The parallel Gauss-Seidel schedule must be a pure re-ordering of Bullet's visit list that never swaps two
visits sharing a body (those are the only pairs that do not commute)."""
import ctypes

import numpy as np

from conftest import write_skeleton


def schedule(hip_lib, path):
    dims = (ctypes.c_int * 4)()
    cap = 256
    visits = np.zeros((128, 4), np.int32)
    sched = np.zeros((16, cap), np.int32)
    ip = ctypes.POINTER(ctypes.c_int)
    hip_lib.check(hip_lib.lib.evm_skeleton_schedule(path.encode(), dims, visits.ctypes.data_as(ip), sched.ctypes.data_as(ip), cap))
    nv, nlev, nw, _ = list(dims)
    return visits[:nv], sched.reshape(-1)[: nw * cap].reshape(nw, cap), nlev, nw


CONTACT = 0x4000


def check(visits, sched, nlev, nw, nm):
    """Every joint visit and every member's contact visit appears exactly once; the per-body `need` counters are
    Bullet-order prefix counts; and (dependency edges) U (each wave's list order) is acyclic, i.e. there is a global
    linear extension, so the version-counter waits cannot deadlock."""
    nv = len(visits)
    lists = [[int(e) for e in sched[w] if e >= 0] for w in range(nw)]
    flat = [e for l in lists for e in l]
    assert sorted(e for e in flat if not e & CONTACT) == list(range(nv))
    assert sorted(e & (CONTACT - 1) for e in flat if e & CONTACT) == list(range(nm))
    cnt = {}
    for i in range(nv):
        t, a, b, need = visits[i]
        assert need & 0xFFFF == cnt.get(a, 0) and need >> 16 == cnt.get(b, 0)
        cnt[a] = cnt.get(a, 0) + 1
        cnt[b] = cnt.get(b, 0) + 1
    # items in dependency order: joints (Bullet order) then contacts; edge = previous item touching the same body
    bodies = {i: {int(visits[i][1]), int(visits[i][2])} for i in range(nv)}
    items = list(range(nv)) + [CONTACT | m for m in range(nm)]
    for m in range(nm):
        bodies[CONTACT | m] = {m}
    edges = {e: set() for e in items}
    last = {}
    for e in items:
        for bd in bodies[e]:
            if bd in last:
                edges[last[bd]].add(e)
            last[bd] = e
    for l in lists:
        for x, y in zip(l, l[1:]):
            edges[x].add(y)
    indeg = {e: 0 for e in items}
    for e in items:
        for y in edges[e]:
            indeg[y] += 1
    ready = [e for e in items if indeg[e] == 0]
    seen = 0
    while ready:
        e = ready.pop()
        seen += 1
        for y in edges[e]:
            indeg[y] -= 1
            if indeg[y] == 0:
                ready.append(y)
    assert seen == len(items), "cycle: the schedule could deadlock"


def test_spider_schedule(hip_lib):
    visits, sched, nlev, nw = schedule(hip_lib, hip_lib.DEFAULT_SKELETON)
    assert len(visits) == 52 and nlev == 8  # 12 hinges + 4 fixed + 12 x (slider, p2p, p2p); depth = the root-body chain
    assert (visits[:16, 0] <= 1).all() and list(visits[16:19, 0]) == [2, 3, 3]  # Bullet order (skeleton.cpp:77-90)
    check(visits, sched, nlev, nw, 17)


def test_generic_skeleton_schedules(hip_lib, tmp_path):
    members = [dict(name="body", mass=2.0, scale=(0.4, 0.2, 0.5))]
    cons, mus = [], []
    for k in range(5):  # a chain with interleaved hinge / fixed constraints and two muscles
        members.append(dict(name=f"seg{k}", mass=0.25, t=(0.5 * (k + 1), 0, 0), scale=(0.2, 0.1, 0.1)))
        parent = "body" if k == 0 else f"seg{k-1}"
        if k % 2 == 0:
            cons.append(dict(type="hinge", name=f"c{k}", parent=parent, child=f"seg{k}", pivot_p=(0.25, 0, 0), pivot_c=(-0.25, 0, 0),
                             axis_p=(0, 0, 1), axis_c=(0, 0, 1), lo=-1.0, hi=1.0))
        else:
            cons.append(dict(type="fixed", name=f"c{k}", parent=parent, child=f"seg{k}", tp=(0.25, 0, 0), tc=(-0.25, 0, 0)))
    mus.append(dict(name="m0", a="body", b="seg0", pos_a=(0.1, 0.1, 0), pos_b=(0, 0.1, 0)))
    mus.append(dict(name="m1", a="seg1", b="seg3", pos_a=(0, 0.1, 0), pos_b=(0, 0.1, 0)))
    path = write_skeleton(tmp_path / "chain.skel", members, cons, mus)
    visits, sched, nlev, nw = schedule(hip_lib, path)
    assert len(visits) == 5 + 6
    check(visits, sched, nlev, nw, 6)
    # single body, no constraints: empty schedule
    path = write_skeleton(tmp_path / "cube.skel", [dict(name="body", mass=1.0, scale=(0.2, 0.2, 0.2))])
    visits, sched, nlev, nw = schedule(hip_lib, path)
    assert len(visits) == 0 and nlev == 0
    check(visits, sched, nlev, nw, 1)


# ---- lane-group schedule of the split pipeline's sweeps kernel (EvmGSchedC) -------------------------------------------
def group_schedule(hip_lib, path, n_waves):
    dims = (ctypes.c_int * 4)()
    cap = 96
    ent = np.full((cap, 22), -7, np.int32)
    ip = ctypes.POINTER(ctypes.c_int)
    hip_lib.check(hip_lib.lib.evm_skeleton_group_schedule(path.encode(), n_waves, dims, ent.ctypes.data_as(ip), cap))
    total, nw, lds, cyc = list(dims)
    return ent[:total], nw, lds, cyc


def check_groups(visits, ent, nw, nm_contact):
    """Group entries are a re-packing of Bullet's visit list: every visit exactly once; the four visits of an entry have one
    type and share no body; per body the visits come in Bullet's order with the prefix-count versions the kernel waits for;
    (dependencies) U (each wave's list order) has a linear extension (the global entry index), so waits cannot deadlock."""
    nv = len(visits)
    # Bullet order key of every visit: joints by (type, a, b) in visit-list order, contacts after
    key = {}
    for i, (t, a, b, need) in enumerate(visits):
        key.setdefault((int(t), int(a), int(b)), []).append(i)
    seen_joint, seen_contact = [], []
    body_seq = {}   # body -> list of (global entry index, need, visit order key)
    per_wave_last = {}
    assert sorted(int(r[0]) >> 8 for r in ent) == list(range(len(ent)))
    for row in ent:
        w, k, ty = int(row[0]) & 0xFF, int(row[0]) >> 8, int(row[1])   # k = position in the one global order
        assert 0 <= w < nw and 0 <= ty <= 6
        assert per_wave_last.get(w, -1) < k      # a wave's list follows the global order
        per_wave_last[w] = k
        used = set()
        filled = 0
        chain = ty >= 5   # chain entry: consecutive visits on ONE shared body a, run as phases in slot order
        ty = {5: 0, 6: 3}.get(ty, ty)
        chain_a, chain_need = None, None
        for q in range(4):
            rec, a, b, need, ps = [int(v) for v in row[2 + 5 * q: 7 + 5 * q]]
            if rec < 0:
                continue
            assert filled == q, "filled slots come first"
            filled += 1
            if chain:
                if chain_a is None:
                    chain_a, chain_need = a, need & 0xFFFF
                assert a == chain_a and (need & 0xFFFF) == chain_need + q, "chain slots = consecutive visits on the shared body"
                bodies = {b}
            else:
                bodies = {a, b}
            assert not (bodies & used), "two visits of one entry share a body"
            used |= bodies
            if ty == 4:
                assert a == b
                seen_contact.append(a)
                order = nv + a
                body_seq.setdefault(a, []).append((k, need & 0xFFFF, order, ps & 0xFFFF))
            else:
                order = key[(ty, a, b)].pop(0)
                seen_joint.append(order)
                body_seq.setdefault(a, []).append((k + q / 8.0, need & 0xFFFF, order, ps & 0xFFFF))
                body_seq.setdefault(b, []).append((k, need >> 16, order, ps >> 16))
        assert filled >= 1
    assert sorted(seen_joint) == list(range(nv))
    assert len(seen_contact) == len(set(seen_contact)) == nm_contact
    for body, seq in body_seq.items():
        seq.sort()
        assert [s[2] for s in seq] == sorted(s[2] for s in seq), "Bullet's order broken on body %d" % body
        assert len({s[0] for s in seq}) == len(seq)          # strictly increasing entries (slots of a chain entry: in slot order)
        assert [s[1] for s in seq] == list(range(len(seq)))  # version each visit waits for = visits before it
        assert all(s[3] == len(seq) for s in seq)            # visits per sweep


def test_spider_group_schedule(hip_lib):
    visits, _, _, _ = schedule(hip_lib, hip_lib.DEFAULT_SKELETON)
    for nw in (1, 2, 4):
        ent, w, lds, cyc = group_schedule(hip_lib, hip_lib.DEFAULT_SKELETON, nw)
        assert w == nw and lds <= 160 * 1024
        check_groups(visits, ent, nw, 17)
        # the four legs pack: far fewer entries than the 52 + 17 visits
        assert len(ent) <= 40, len(ent)
        print("group schedule, %d wave(s): %d entries, LDS %d B, estimate %d cycles" % (nw, len(ent), lds, cyc))


def test_generic_group_schedule(hip_lib, tmp_path):
    members = [dict(name="body", mass=2.0, scale=(0.4, 0.2, 0.5))]
    cons, mus = [], []
    for k in range(5):
        members.append(dict(name=f"seg{k}", mass=0.25, t=(0.5 * (k + 1), 0, 0), scale=(0.2, 0.1, 0.1)))
        parent = "body" if k == 0 else f"seg{k-1}"
        if k % 2 == 0:
            cons.append(dict(type="hinge", name=f"c{k}", parent=parent, child=f"seg{k}", pivot_p=(0.25, 0, 0), pivot_c=(-0.25, 0, 0),
                             axis_p=(0, 0, 1), axis_c=(0, 0, 1), lo=-1.0, hi=1.0))
        else:
            cons.append(dict(type="fixed", name=f"c{k}", parent=parent, child=f"seg{k}", tp=(0.25, 0, 0), tc=(-0.25, 0, 0)))
    mus.append(dict(name="m0", a="body", b="seg0", pos_a=(0.1, 0.1, 0), pos_b=(0, 0.1, 0)))
    mus.append(dict(name="m1", a="seg1", b="seg3", pos_a=(0, 0.1, 0), pos_b=(0, 0.1, 0)))
    path = write_skeleton(tmp_path / "chain.skel", members, cons, mus)
    visits, _, _, _ = schedule(hip_lib, path)
    for nw in (1, 3):
        ent, w, lds, cyc = group_schedule(hip_lib, path, nw)
        check_groups(visits, ent, nw, 6)
    path = write_skeleton(tmp_path / "cube.skel", [dict(name="body", mass=1.0, scale=(0.2, 0.2, 0.2))])
    visits, _, _, _ = schedule(hip_lib, path)
    ent, w, lds, cyc = group_schedule(hip_lib, path, 1)
    check_groups(visits, ent, 1, 1)


def _random_skeleton(rng, tmp_path, tag):
    """a random tree of 2..14 members joined by hinges / fixed constraints, with 0..6 muscles between random member pairs
    (several may share a member: the chain entries of the group schedule)"""
    nm = int(rng.integers(2, 15))
    members = [dict(name="body", mass=2.0, scale=(0.3, 0.2, 0.3))]
    cons, mus = [], []
    for k in range(1, nm):
        parent = int(rng.integers(0, k))
        members.append(dict(name=f"m{k}", mass=float(rng.uniform(0.1, 1.0)), t=(0.4 * k, 0.1 * parent, 0.0), scale=(0.15, 0.1, 0.1)))
        pname = members[parent]["name"]
        if rng.random() < 0.7:
            ax = [(1, 0, 0), (0, 1, 0), (0, 0, 1)][int(rng.integers(0, 3))]
            cons.append(dict(type="hinge", name=f"c{k}", parent=pname, child=f"m{k}", pivot_p=(0.2, 0, 0), pivot_c=(-0.2, 0, 0),
                             axis_p=ax, axis_c=ax, lo=-1.0, hi=1.0))
        else:
            cons.append(dict(type="fixed", name=f"c{k}", parent=pname, child=f"m{k}", tp=(0.2, 0, 0), tc=(-0.2, 0, 0)))
    for j in range(int(rng.integers(0, 7))):
        a, b = rng.choice(nm, 2, replace=False)
        mus.append(dict(name=f"mu{j}", a=members[int(a)]["name"], b=members[int(b)]["name"], pos_a=(0.05, 0.05, 0), pos_b=(0, 0.05, 0.05)))
    return write_skeleton(tmp_path / f"rand{tag}.skel", members, cons, mus), nm, len(cons), len(mus)


def test_random_skeletons_schedule_validly(hip_lib, tmp_path):
    """40 random articulated trees: both schedules (level schedule of the tile kernel, lane-group entries of k_sweeps_g at 1, 2
    and 4 waves) stay pure re-orderings of Bullet's visit list that never swap two visits sharing a body, or the loader says
    why it cannot build one (a skeleton whose records do not fit the 160 KB of LDS falls back to the tile kernel)."""
    rng = np.random.default_rng(2024)
    built = 0
    for tag in range(40):
        path, nm, nc, nmu = _random_skeleton(rng, tmp_path, tag)
        visits, sched, nlev, nw = schedule(hip_lib, path)
        assert len(visits) == nc + 3 * nmu            # per muscle: the slider and its two p2p joints
        check(visits, sched, nlev, nw, nm)
        for gw in (1, 2, 4):
            try:
                ent, w, lds, cyc = group_schedule(hip_lib, path, gw)
            except Exception as e:                    # refused with a reason (too many records / entries for the kernel's tables)
                assert "group schedule" in str(e) or "LDS" in str(e) or "entries" in str(e), e
                continue
            assert lds <= 160 * 1024
            check_groups(visits, ent, gw, nm)
            built += 1
    assert built >= 100, built


def _star(write, tmp_path, nm, nmus, name):
    """a root with nm - 1 members hinged to it in a ring and nmus muscles from the root to the first members"""
    members = [dict(name="body", mass=3.0, scale=(0.3, 0.15, 0.3))]
    cons, mus = [], []
    for k in range(1, nm):
        a = 2 * np.pi * k / nm
        d = np.array([0.5 * np.cos(a), 0.0, 0.5 * np.sin(a)])
        members.append(dict(name=f"m{k}", mass=0.3, t=tuple(d), scale=(0.08, 0.06, 0.08)))
        cons.append(dict(type="hinge", name=f"c{k}", parent="body", child=f"m{k}", pivot_p=tuple(d / 2), pivot_c=tuple(-d / 2),
                         axis_p=(0, 1, 0), axis_c=(0, 1, 0), lo=-0.5, hi=0.5))
    for j in range(nmus):
        mus.append(dict(name=f"mu{j}", a="body", b=f"m{1 + j % (nm - 1)}", pos_a=(0.02, 0.08, 0.0), pos_b=(0.0, 0.05, 0.0)))
    return write(tmp_path / name, members, cons, mus)


def test_capacity_limits_are_errors_not_crashes(hip_lib, tmp_path):
    """skel_const.h compiles the tables for 24 members / 24 hinges / 20 muscles / 64 bodies: the largest skeleton that fits is
    scheduled, one member or one muscle more is refused with a message"""
    import pytest
    path = _star(write_skeleton, tmp_path, 24, 20, "max.skel")        # 24 members + 40 attach spheres = 64 bodies
    visits, sched, nlev, nw = schedule(hip_lib, path)
    assert len(visits) == 23 + 60
    check(visits, sched, nlev, nw, 24)
    for nm, nmus, name in ((25, 4, "members.skel"), (12, 21, "muscles.skel")):
        with pytest.raises(Exception) as e:
            schedule(hip_lib, _star(write_skeleton, tmp_path, nm, nmus, name))
        assert "capacity" in str(e.value) or "exceed" in str(e.value), e.value
