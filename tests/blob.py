"""Named views of the canonical per-env state blob (layout: include/evomotion.h)."""
import numpy as np


def fields(nb, nm, nmus, npairs=0):
    """npairs > 0: the env runs with member-vs-member contacts; its blob ends with one 49-float manifold per pair"""
    out, k = {}, 0

    def add(name, n):
        nonlocal k
        out[name] = slice(k, k + n)
        k += n

    add("bodies", 13 * nb)
    add("pending", 1)
    add("E", 9)
    add("iinv", 6 * nb)
    add("ms", 3 * nm)
    add("hist", 6 * nm)
    add("manifold", 37 * nm)
    add("pairs", 49 * npairs)   # per pair: count, 4 x (localA3 localB3 normalOnB3 dist applied applied_lateral)
    add("target", nmus)
    add("powered", 1)
    add("counters", 2)
    out["_size"] = k
    return out


def body_view(blob, nb):
    b = blob[..., : 13 * nb].reshape(blob.shape[:-1] + (nb, 13))
    return dict(pos=b[..., 0:3], quat=b[..., 3:7], lin=b[..., 7:10], ang=b[..., 10:13])


def compare(a, b, nb, nm, nmus, npairs=None):
    """max abs difference per meaningful field (quat/iinv are only defined in one of the two pending modes).  npairs: member
    pairs of an env with member-vs-member contacts; by default taken from the blob's width."""
    if npairs is None:
        extra = a.shape[-1] - fields(nb, nm, nmus)["_size"]
        assert extra >= 0 and extra % 49 == 0, (a.shape, nb, nm, nmus)
        npairs = extra // 49
    assert a.shape[-1] == b.shape[-1]
    f = fields(nb, nm, nmus, npairs)
    res = {}
    pend = a[..., f["pending"]][..., 0] != 0
    va, vb = body_view(a, nb), body_view(b, nb)
    for k in ("pos", "lin", "ang"):
        res[k] = float(np.abs(va[k] - vb[k]).max())
    if (~pend).any():
        qa, qb = va["quat"][~pend], vb["quat"][~pend]
        res["quat"] = float(np.minimum(np.abs(qa - qb).max(-1), np.abs(qa + qb).max(-1)).max())
    if pend.any():
        res["iinv(pending)"] = float(np.abs(a[pend][:, f["iinv"]] - b[pend][:, f["iinv"]]).max())
        res["E(pending)"] = float(np.abs(a[pend][:, f["E"]] - b[pend][:, f["E"]]).max())
    for k in ("pending", "ms", "hist", "target", "powered", "counters"):
        d = np.abs(a[..., f[k]] - b[..., f[k]])
        res[k] = float(d.max()) if d.size else 0.0
    ma = a[..., f["manifold"]].reshape(a.shape[:-1] + (nm, 37))
    mb = b[..., f["manifold"]].reshape(b.shape[:-1] + (nm, 37))
    res["mf_count"] = float(np.abs(ma[..., 0] - mb[..., 0]).max())
    same = ma[..., 0] == mb[..., 0]
    res["mf_points"] = float(np.abs(ma[same][:, 1:] - mb[same][:, 1:]).max()) if same.any() else 0.0
    if npairs:
        pa = a[..., f["pairs"]].reshape(a.shape[:-1] + (npairs, 49))
        pb = b[..., f["pairs"]].reshape(b.shape[:-1] + (npairs, 49))
        res["pm_count"] = float(np.abs(pa[..., 0] - pb[..., 0]).max())
        res["pm_count_mismatches"] = int((pa[..., 0] != pb[..., 0]).sum())
        res["pm_live"] = int((pa[..., 0] > 0).sum())
        same = (pa[..., 0] == pb[..., 0]) & (pa[..., 0] > 0)
        if same.any():
            qa, qb = pa[same][:, 1:].reshape(-1, 4, 12), pb[same][:, 1:].reshape(-1, 4, 12)
            res["pm_geom"] = float(np.abs(qa[..., :10] - qb[..., :10]).max())      # local points, normal, distance
            res["pm_impulse"] = float(np.abs(qa[..., 10:] - qb[..., 10:]).max())   # accumulated normal / friction impulses
        else:
            res["pm_geom"] = res["pm_impulse"] = 0.0
    return res
