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
