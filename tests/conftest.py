import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc_lib():
    import orc
    if not os.path.exists(orc.ORC_LIB):
        subprocess.check_call(["make", "-s", "-C", orc.ORC_DIR])
    return orc.load()


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; CPU tests only use its host-only entry points."""
    so = os.path.join(ROOT, "evomotion_amd", "libevomotion_hip.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "evomotion_amd", "csrc")])
    from evomotion_amd import _lib
    return _lib


def write_skeleton(path, members, constraints=(), muscles=(), shapes_from=None):
    """Write a small skeleton fixture for analytic tests.  members: list of dicts."""
    import orc

    def hx(v):
        import numpy as np
        return float(np.float32(v)).hex()

    lines = ["# test skeleton", "skeleton test root %s" % members[0]["name"], "members %d" % len(members)]
    for m in members:
        vals = [m["mass"], m.get("friction", 0.5)] + list(m.get("t", (0, 0, 0))) + list(m.get("q", (1, 0, 0, 0))) + list(m["scale"])
        lines.append("member %s %s %s %d" % (m["name"], m.get("shape", "cube"), " ".join(hx(v) for v in vals), 0))
    lines.append("constraints %d" % len(constraints))
    for c in constraints:
        if c["type"] == "hinge":
            vals = list(c["pivot_p"]) + list(c["pivot_c"]) + list(c["axis_p"]) + list(c["axis_c"]) + [c["lo"], c["hi"]]
            lines.append("hinge %s %s %s %s" % (c["name"], c["parent"], c["child"], " ".join(hx(v) for v in vals)))
        else:
            vals = list(c["tp"]) + list(c.get("qp", (1, 0, 0, 0))) + list(c["tc"]) + list(c.get("qc", (1, 0, 0, 0)))
            lines.append("fixed %s %s %s %s" % (c["name"], c["parent"], c["child"], " ".join(hx(v) for v in vals)))
    lines.append("muscles %d" % len(muscles))
    for m in muscles:
        vals = [m.get("mass", 0.1875)] + list(m.get("scale", (0.0625,) * 3)) + list(m["pos_a"]) + list(m["pos_b"]) + [m.get("force", 64.0), m.get("speed", 8.0)]
        lines.append("muscle %s %s %s %s" % (m["name"], m["a"], m["b"], " ".join(hx(v) for v in vals)))
    # copy the shape tables from the committed fixture
    src = open(orc.SKEL).read().split("\n")
    i = next(k for k, l in enumerate(src) if l.startswith("shapes "))
    lines += [l for l in src[i:] if l]
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return str(path)


def write_skeleton_json(folder, members, constraints=(), muscles=(), robot="test"):
    """The same spec as write_skeleton(), in the REFERENCE's own format: <folder>/skeleton/robot.json with every float a
    32-character bit string (converter.cpp:138-147) + <folder>/obj/<shape>.obj hulls (shapes.cpp:24-56).  The hull
    vertices come from the committed fixture's shape tables, emitted as triangles in table order."""
    import json
    import os
    import struct
    import numpy as np
    import orc

    def bits(v):
        return format(struct.unpack(">I", struct.pack(">f", float(np.float32(v))))[0], "032b")

    def v3(v):
        return dict(x=bits(v[0]), y=bits(v[1]), z=bits(v[2]))

    def q4(q):  # (w, x, y, z)
        return dict(w=bits(q[0]), x=bits(q[1]), y=bits(q[2]), z=bits(q[3]))

    os.makedirs(os.path.join(folder, "skeleton"), exist_ok=True)
    os.makedirs(os.path.join(folder, "obj"), exist_ok=True)
    doc = dict(robot_name=robot, root_name=members[0]["name"], members=[], constraints=[], muscles=[])
    for m in members:
        doc["members"].append(dict(name=m["name"], shape=m.get("shape", "cube"), mass=bits(m["mass"]), friction=bits(m.get("friction", 0.5)),
                                   translation=v3(m.get("t", (0, 0, 0))), rotation=q4(m.get("q", (1, 0, 0, 0))), scale=v3(m["scale"]),
                                   ignore_collision=False))
    for c in constraints:
        if c["type"] == "hinge":
            doc["constraints"].append(dict(type="hinge", name=c["name"], parent_name=c["parent"], child_name=c["child"],
                                           pivot_in_parent=v3(c["pivot_p"]), pivot_in_child=v3(c["pivot_c"]), axis_in_parent=v3(c["axis_p"]),
                                           axis_in_child=v3(c["axis_c"]), limit_radian=dict(min=bits(c["lo"]), max=bits(c["hi"]))))
        else:
            doc["constraints"].append(dict(type="fixed", name=c["name"], parent_name=c["parent"], child_name=c["child"],
                                           frame_in_parent=dict(translation=v3(c["tp"]), rotation=q4(c.get("qp", (1, 0, 0, 0)))),
                                           frame_in_child=dict(translation=v3(c["tc"]), rotation=q4(c.get("qc", (1, 0, 0, 0))))))
    for m in muscles:
        doc["muscles"].append(dict(name=m["name"], item_a=m["a"], item_b=m["b"], attach_mass=bits(m.get("mass", 0.1875)),
                                   attach_scale=v3(m.get("scale", (0.0625,) * 3)), pos_in_a=v3(m["pos_a"]), pos_in_b=v3(m["pos_b"]),
                                   force=bits(m.get("force", 64.0)), speed=bits(m.get("speed", 8.0))))
    path = os.path.join(folder, "skeleton", robot + ".json")
    with open(path, "w") as f:
        json.dump(doc, f, indent=4, sort_keys=True)
    # hull tables of the committed fixture -> OBJ
    src = open(orc.SKEL).read().split("\n")
    i = next(k for k, l in enumerate(src) if l.startswith("shapes ")) + 1
    while i < len(src) and src[i].startswith("shape "):
        _, name, n, _ = src[i].split()
        pts = [[float.fromhex(t) for t in src[i + 1 + k].split()] for k in range(int(n))]
        with open(os.path.join(folder, "obj", name + ".obj"), "w") as f:
            f.write("# test hull\n")
            for p in pts:
                f.write("v %r %r %r\n" % (p[0], p[1], p[2]))
            f.write("vn 0.0 1.0 0.0\n")
            idx = list(range(1, len(pts) + 1))
            while len(idx) % 3:
                idx.append(len(pts))
            for k in range(0, len(idx), 3):
                f.write("f %d//1 %d//1 %d//1\n" % (idx[k], idx[k + 1], idx[k + 2]))
        i += 1 + int(n)
    return path
