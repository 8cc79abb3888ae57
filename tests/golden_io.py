import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "agent_golden.txt")


def load(path=GOLDEN):
    out, params = {}, []
    lines = open(path).read().split("\n")
    i = 0
    while i < len(lines):
        l = lines[i]
        if l.startswith("param "):
            t = l.split()
            params.append((t[1], t[2], tuple(int(v) for v in t[3:])))
        if l.startswith("tensor "):
            t = l.split()
            shape = tuple(int(v) for v in t[3:3 + int(t[2])])
            n = int(np.prod(shape))
            vals = []
            i += 1
            while len(vals) < n:
                vals += [float(v) for v in lines[i].split()]
                i += 1
            out[t[1]] = np.array(vals, np.float32).reshape(shape)
            continue
        i += 1
    out["_params"] = params
    return out
