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


def check(visits, sched, nlev, nw):
    nv = len(visits)
    level = -np.ones(nv, int)
    pos = {}
    for w in range(nw):
        l = 0
        for i, e in enumerate(sched[w]):
            if e < 0:
                break
            v = e & 0x7FFF
            if v != 0x7FFF:
                assert level[v] == -1, "visit scheduled twice"
                level[v] = l
                pos[v] = (w, i)
            if e & 0x8000:
                l += 1
        assert l == nlev, "every wave closes every level exactly once"
    assert (level >= 0).all(), "every visit is scheduled"
    cnt = {}
    for i in range(nv):
        t, a, b, need = visits[i]
        # the version each body must have reached = number of earlier visits (Bullet order) touching it
        assert need & 0xFFFF == cnt.get(a, 0) and need >> 16 == cnt.get(b, 0)
        cnt[a] = cnt.get(a, 0) + 1
        cnt[b] = cnt.get(b, 0) + 1
        for j in range(i + 1, nv):
            if {a, b} & {visits[j][1], visits[j][2]}:
                assert level[i] < level[j], (i, j)  # dependent visits keep Bullet's order
    # deadlock freedom: each wave's list is sorted by level
    for w in range(nw):
        ls = [level[e & 0x7FFF] for e in sched[w] if e >= 0 and (e & 0x7FFF) != 0x7FFF]
        assert ls == sorted(ls)


def test_spider_schedule(hip_lib):
    visits, sched, nlev, nw = schedule(hip_lib, hip_lib.DEFAULT_SKELETON)
    assert len(visits) == 52 and nlev == 8  # 12 hinges + 4 fixed + 12 x (slider, p2p, p2p); depth = the root-body chain
    assert (visits[:16, 0] <= 1).all() and list(visits[16:19, 0]) == [2, 3, 3]  # Bullet order (skeleton.cpp:77-90)
    check(visits, sched, nlev, nw)


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
    check(visits, sched, nlev, nw)
    # single body, no constraints: empty schedule
    path = write_skeleton(tmp_path / "cube.skel", [dict(name="body", mass=1.0, scale=(0.2, 0.2, 0.2))])
    visits, sched, nlev, nw = schedule(hip_lib, path)
    assert len(visits) == 0 and nlev == 0
