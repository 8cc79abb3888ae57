import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import orc, blob
from evomotion_amd import VecRobotWalk
N = 2
env = VecRobotWalk(N, seed=1234)
oracles = [orc.OracleEnv(seed=1234 + i) for i in range(N)]
nb, nm, nmus = env.n_bodies, env.n_members, env.n_muscles
env.debug_reset_begin()
for o in oracles: o.reset_begin()
env.debug_physics_steps(1)
for o in oracles: o.physics_step()
sg = env.get_state(); so = np.stack([o.get_state() for o in oracles])
bg, bo = blob.body_view(sg, nb), blob.body_view(so, nb)
np.set_printoptions(precision=5, suppress=True, linewidth=200)
print("ang diff per body env0:", np.abs(bg["ang"][0] - bo["ang"][0]).max(-1))
print("lin diff per body env0:", np.abs(bg["lin"][0] - bo["lin"][0]).max(-1))
print("gpu ang body0..4", bg["ang"][0][:5]); print("orc ang body0..4", bo["ang"][0][:5])
