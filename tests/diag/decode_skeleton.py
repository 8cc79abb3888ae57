#!/usr/bin/env python3
"""Decode the reference's robot_walk skeleton + collision hulls into one text fixture.

Reads (data files only, no reference source):
  <ref>/evo_motion_model/resources/skeleton/new_format_spider.json
  <ref>/evo_motion_model/resources/obj/{cube,feet,sphere}.obj
and writes evomotion_amd/data/robot_walk_spider.skel.

JSON floats are 32-char IEEE-754 bit strings (reference: evo_motion_model/src/converter.cpp:138-147).
Hull points follow evo_motion_model/src/shapes.cpp:42-56 (one point per face-vertex, parsed with stof),
then duplicates are dropped keeping FIRST-occurrence order, which preserves the result of any
first-strict-extremum support scan over the duplicated list.

Every float is written as a C99 hex-float of the exact float32 value, so both loaders
(oracle/ and evomotion_amd/csrc/) read bit-identical constants.
"""
import json, struct, sys, os
import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(__file__), "..", "..", "evomotion_amd", "data", "robot_walk_spider.skel")
RES = os.path.join(REF, "evo_motion_model", "resources")


def bits(s):
    return struct.unpack(">f", int(s, 2).to_bytes(4, "big"))[0]


def hx(f):
    return float(np.float32(f)).hex()


def v3(o):
    return [bits(o["x"]), bits(o["y"]), bits(o["z"])]


def q4(o):
    return [bits(o["w"]), bits(o["x"]), bits(o["y"]), bits(o["z"])]


def fl(vals):
    return " ".join(hx(v) for v in vals)


def load_hull(name):
    verts, order = [], []
    for line in open(os.path.join(RES, "obj", name + ".obj")):
        s = line.rstrip("\n").split(" ")
        if s[0] == "v":
            verts.append([np.float32(float(t)) for t in s[1:4]])
        elif s[0] == "f":
            order += [int(t.split("/")[0]) - 1 for t in s[1:4]]
    seen, uniq = set(), []
    for i in order:
        k = tuple(float(c) for c in verts[i])
        if k not in seen:
            seen.add(k)
            uniq.append(verts[i])
    return len(order), uniq


def main():
    j = json.load(open(os.path.join(RES, "skeleton", "new_format_spider.json")))
    out = []
    out.append("# robot_walk skeleton fixture (decoded from the reference's data files by tests/diag/decode_skeleton.py)")
    out.append(f"skeleton {j['robot_name']} root {j['root_name']}")
    out.append(f"members {len(j['members'])}")
    for m in j["members"]:
        out.append("member %s %s %s %s %s %s %s %d" % (
            m["name"], m["shape"], hx(bits(m["mass"])), hx(bits(m["friction"])),
            fl(v3(m["translation"])), fl(q4(m["rotation"])), fl(v3(m["scale"])), 1 if m["ignore_collision"] else 0))
    out.append(f"constraints {len(j['constraints'])}")
    for c in j["constraints"]:
        if c["type"] == "hinge":
            out.append("hinge %s %s %s %s %s %s %s %s %s" % (
                c["name"], c["parent_name"], c["child_name"],
                fl(v3(c["pivot_in_parent"])), fl(v3(c["pivot_in_child"])),
                fl(v3(c["axis_in_parent"])), fl(v3(c["axis_in_child"])),
                hx(bits(c["limit_radian"]["min"])), hx(bits(c["limit_radian"]["max"]))))
        elif c["type"] == "fixed":
            fp, fc = c["frame_in_parent"], c["frame_in_child"]
            out.append("fixed %s %s %s %s %s %s %s" % (
                c["name"], c["parent_name"], c["child_name"],
                fl(v3(fp["translation"])), fl(q4(fp["rotation"])),
                fl(v3(fc["translation"])), fl(q4(fc["rotation"]))))
        else:
            raise SystemExit("unknown constraint type " + c["type"])
    out.append(f"muscles {len(j['muscles'])}")
    for m in j["muscles"]:
        out.append("muscle %s %s %s %s %s %s %s %s %s" % (
            m["name"], m["item_a"], m["item_b"], hx(bits(m["attach_mass"])), fl(v3(m["attach_scale"])),
            fl(v3(m["pos_in_a"])), fl(v3(m["pos_in_b"])), hx(bits(m["force"])), hx(bits(m["speed"]))))
    shapes = ["cube", "feet", "sphere"]
    out.append(f"shapes {len(shapes)}")
    for s in shapes:
        ndup, uniq = load_hull(s)
        out.append(f"shape {s} {len(uniq)} {ndup}")
        for p in uniq:
            out.append(fl(p))
    with open(OUT, "w") as f:
        f.write("\n".join(out) + "\n")
    print("wrote", os.path.abspath(OUT), len(out), "lines")


if __name__ == "__main__":
    main()
