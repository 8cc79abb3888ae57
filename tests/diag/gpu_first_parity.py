"""End-to-end check on a GPU box: HIP env vs CPU oracle (teacher-forced and free-running)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import orc, blob
from evomotion_amd import VecRobotWalk

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
env = VecRobotWalk(N, seed=1234)
oracles = [orc.OracleEnv(seed=1234 + i) for i in range(N)]
nb, nm, nmus = env.n_bodies, env.n_members, env.n_muscles
cg, co = env.body_constants(), oracles[0].body_constants()
cg[nm:, 6] = co[nm:, 6]
print("loader constants max abs diff", np.abs(cg - co).max())
def cmp(tag):
    sg = env.get_state(); so = np.stack([o.get_state() for o in oracles])
    print(tag, {k: ("%.3g" % v) for k, v in blob.compare(so, sg, nb, nm, nmus).items()})
cmp("creation")
env.debug_reset_begin()
for o in oracles: o.reset_begin()
cmp("after reset_begin")
pg = env.body_poses().cpu().numpy(); po = np.stack([o.poses() for o in oracles])
print("pose diff after reset_begin: pos", np.abs(pg[..., :3] - po[..., :3]).max(), "quat", np.abs(pg[..., 3:] - po[..., 3:]).max())
for k in range(60):
    env.debug_physics_steps(1)
    for o in oracles: o.physics_step()
    if k < 3 or k % 10 == 9: cmp("settle %d" % k)
rng = np.random.default_rng(0)
F = blob.fields(nb, nm, nmus)
for k in range(100):
    so = np.stack([o.get_state() for o in oracles])
    env.set_state(so)
    a = rng.uniform(-1, 1, (N, 12)).astype(np.float32)
    st = env.do_step(torch.from_numpy(a))
    og = st.state.cpu().numpy(); rg = st.reward.cpu().numpy(); dg = st.done.cpu().numpy()
    outs = [o.do_step(a[i]) for i, o in enumerate(oracles)]
    oo = np.stack([x[0] for x in outs]); ro = np.array([x[1] for x in outs]); do = np.array([x[2] for x in outs])
    d = np.abs(og - oo)
    if k < 4 or k % 25 == 24:
        i = np.unravel_index(d.argmax(), d.shape)
        print("TF step", k, "obs diff %.3g at env %d col %d (gpu %.5g oracle %.5g)" % (d.max(), i[0], i[1], og[i], oo[i]),
              "members %.3g muscles %.3g" % (d[:, :323].max(), d[:, 323:].max()), "rew %.3g" % np.abs(rg - ro).max(), "done mism", int((dg != do).sum()))
        cmp("   state")
