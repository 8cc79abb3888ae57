import sys, numpy as np, torch
sys.path.insert(0,'tests')
import orc, blob, test_schedule as ts
from conftest import write_skeleton
import pathlib, tempfile
from evomotion_amd import VecRobotWalk
tmp=pathlib.Path(tempfile.mkdtemp())
path = ts._star(write_skeleton, tmp, 18, 13, "large.skel")
n=8
env=VecRobotWalk(n,seed=1234,parameters=dict(skeleton_json_path=path))
orcs=[orc.OracleEnv(seed=900+i,skeleton=path) for i in range(n)]
print('pairs',env.n_pairs)
for o in orcs: o.reset()
rng=np.random.default_rng(9)
for k in range(3):
    so=np.stack([o.get_state() for o in orcs]); env.set_state(so)
    a=rng.uniform(-1,1,(n,13)).astype(np.float32)
    env.do_step(torch.from_numpy(a))
    for i,o in enumerate(orcs): o.do_step(a[i])
    s1=np.stack([o.get_state() for o in orcs]); d=blob.compare(s1, env.get_state(), 44,18,13)
    f=blob.fields(44,18,13,env.n_pairs)
    act=[(s1[i][f['manifold']].reshape(18,37)[:,0]>0).sum()+(s1[i][f['pairs']].reshape(-1,49)[:,0]>0).sum() for i in range(n)]
    print(k, {kk:(round(v,7) if isinstance(v,float) else v) for kk,v in d.items() if kk in('pos','lin','ang','pm_count_mismatches','mf_count','pm_live')}, 'active', act, 'resid', env.residual(clear=True), 'errs', env.errors())
