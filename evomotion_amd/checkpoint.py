"""`*.th` module checkpoints in the reference's wire format.

The reference writes networks with `torch::serialize::OutputArchive` (evo_motion_networks/include/evo_motion_networks/
saver.h:13-25) and reads them with `InputArchive` + `Module::load` (saver.h:27-39): a TorchScript zip archive whose
module hierarchy carries the parameters under the reference's names (`head.0.weight … mu.0.weight, sigma.0.weight`
for the actor, actor.cpp:9-28).  A scripted mirror module has exactly that hierarchy, so `torch.jit.save` of it is a
file the reference's `load_torch` accepts, and `torch.jit.load` reads what `save_torch` wrote.

Optimiser archives (`*_optimizer.th`, ppo_gae.cpp:194,196, soft_actor_critic.cpp:186-199) are the output of
`torch::optim::Adam::save` (torch/csrc/api/include/torch/optim/serialize.h, format "1.5.0"): a TorchScript archive with
  pytorch_version = "1.5.0"
  state/<key>/{step: int, exp_avg: Tensor, exp_avg_sq: Tensor}          one sub-archive per parameter that has state
  param_groups/"param_groups/size" = tensor(1), param_groups/"param_groups/0"/{"params/size" = tensor(n),
      "params/<i>" = "<key>" (i-th parameter of the group), options/{lr, betas, eps, weight_decay, amsgrad}}
where <key> is a decimal string — the saving process's parameter address in the reference, which the loader only uses to
match `state` entries to positions in the group (`std::stoull`); any distinct decimal strings do.  `save_adam_th` /
`load_adam_th` write and read that format, so optimiser state moves between this package and the reference in both
directions (tests/test_checkpoint.py checks both against the compiled reference).
"""
import collections
import os

import torch


def load_th(path):
    """Named fp32 tensors of a `.th` module archive, in the archive's (= named_parameters) order."""
    if not os.path.isfile(path):
        raise RuntimeError("Could not find " + os.path.dirname(path))  # std::runtime_error of saver.h:33-34
    m = torch.jit.load(path, map_location="cpu")
    out = collections.OrderedDict()
    for n, t in m.named_parameters():
        out[n] = t.detach().clone()
    for n, t in m.named_buffers():
        out[n] = t.detach().clone()
    return out


def load_into(module, path, strict=True):
    """`load_torch(folder, module, file)`: copies the archive's tensors into `module` by name."""
    sd = load_th(path)
    own = dict(module.named_parameters())
    own.update(dict(module.named_buffers()))
    missing = [k for k in own if k not in sd]
    unexpected = [k for k in sd if k not in own]
    if strict and (missing or unexpected):
        raise RuntimeError("checkpoint %s does not match the module: missing %s, unexpected %s" % (path, missing, unexpected))
    with torch.no_grad():
        for k, v in sd.items():
            if k in own:
                if tuple(own[k].shape) != tuple(v.shape):
                    raise RuntimeError("checkpoint %s: %s has shape %s, module expects %s" % (path, k, tuple(v.shape), tuple(own[k].shape)))
                own[k].copy_(v.to(own[k].device))
    return module


def save_th(module, path):
    """`save_torch(folder, module, file)`: the folder must exist (saver.h:17-18)."""
    folder = os.path.dirname(os.path.abspath(path))
    if not os.path.isdir(folder):
        raise RuntimeError("Could not find " + folder)
    was_training = module.training
    cpu = torch.jit.script(_cpu_copy(module))
    cpu.train(was_training)
    torch.jit.save(cpu, path)


def _cpu_copy(module):
    import copy
    return copy.deepcopy(module).to("cpu")


# ---- torch::optim::Adam archives -------------------------------------------------------------------------------------
def save_adam_th(path, states, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False):
    """`save_torch(folder, optimizer, file)` for a torch::optim::Adam with ONE parameter group.
    states: per parameter, in the group's (= named_parameters) order, `(step, exp_avg, exp_avg_sq)` or None for a parameter
    without state (no step taken yet)."""
    folder = os.path.dirname(os.path.abspath(path))
    if not os.path.isdir(folder):
        raise RuntimeError("Could not find " + folder)
    C = torch._C
    cu = C.CompilationUnit()

    def archive():  # torch::serialize::OutputArchive = a script module without methods
        return C.ScriptModule("__torch__.Module", cu, True)

    def put_module(parent, name, child):
        parent._register_attribute(name, child._type(), child)

    root = archive()
    root._register_attribute("pytorch_version", C.StringType.get(), "1.5.0")
    state = archive()
    keys = [str(1000 + i) for i in range(len(states))]
    for key, st in zip(keys, states):
        if st is None:
            continue
        step, m, v = st
        a = archive()
        a._register_attribute("step", C.IntType.get(), int(step))
        a._register_attribute("exp_avg", C.TensorType.get(), m.detach().to("cpu", torch.float32).contiguous().clone())
        a._register_attribute("exp_avg_sq", C.TensorType.get(), v.detach().to("cpu", torch.float32).contiguous().clone())
        put_module(state, key, a)
    put_module(root, "state", state)
    groups = archive()
    groups._register_attribute("param_groups/size", C.TensorType.get(), torch.tensor(1, dtype=torch.int64))
    g0 = archive()
    g0._register_attribute("params/size", C.TensorType.get(), torch.tensor(len(states), dtype=torch.int64))
    for i, key in enumerate(keys):
        g0._register_attribute("params/%d" % i, C.StringType.get(), key)
    opt = archive()
    opt._register_attribute("lr", C.FloatType.get(), float(lr))
    opt._register_attribute("betas", C.TupleType([C.FloatType.get(), C.FloatType.get()]), (float(betas[0]), float(betas[1])))
    opt._register_attribute("eps", C.FloatType.get(), float(eps))
    opt._register_attribute("weight_decay", C.FloatType.get(), float(weight_decay))
    opt._register_attribute("amsgrad", C.BoolType.get(), bool(amsgrad))
    put_module(g0, "options", opt)
    put_module(groups, "param_groups/0", g0)
    put_module(root, "param_groups", groups)
    root.save(path)


def load_adam_th(path):
    """`load_torch(folder, optimizer, file)`: returns (states, options) with states[i] = (step, exp_avg, exp_avg_sq) or None
    for the i-th parameter of the (single) group, options = dict(lr, betas, eps, weight_decay, amsgrad)."""
    if not os.path.isfile(path):
        raise RuntimeError("Could not find " + os.path.dirname(path))
    m = torch.jit.load(path, map_location="cpu")
    if m.pytorch_version != "1.5.0":
        raise RuntimeError("%s: unsupported optimiser archive version %r" % (path, m.pytorch_version))
    groups = m.param_groups
    if int(getattr(groups, "param_groups/size")) != 1:
        raise RuntimeError("%s: exactly one parameter group is supported" % path)
    g0 = getattr(groups, "param_groups/0")
    n = int(getattr(g0, "params/size"))
    by_key = dict(m.state.named_children())
    states = []
    for i in range(n):
        a = by_key.get(getattr(g0, "params/%d" % i))
        states.append(None if a is None else (int(a.step), a.exp_avg.detach().clone(), a.exp_avg_sq.detach().clone()))
    o = g0.options
    return states, dict(lr=float(o.lr), betas=tuple(o.betas), eps=float(o.eps), weight_decay=float(o.weight_decay), amsgrad=bool(o.amsgrad))


def adam_states_from_flat(module, step, exp_avg_flat, exp_avg_sq_flat):
    """flat moment vectors (named_parameters order, the HIP trainers' layout) -> per-parameter states of save_adam_th"""
    out, o = [], 0
    for p in module.parameters():
        n = p.numel()
        out.append(None if step == 0 else (step, exp_avg_flat[o:o + n].reshape(p.shape), exp_avg_sq_flat[o:o + n].reshape(p.shape)))
        o += n
    return out


def adam_flat_from_states(module, states):
    """per-parameter states -> (step, flat exp_avg, flat exp_avg_sq); parameters without state contribute zeros"""
    ps = list(module.parameters())
    if len(states) != len(ps):
        raise RuntimeError("optimiser archive holds %d parameters, the module has %d" % (len(states), len(ps)))
    ms, vs, step = [], [], 0
    for p, st in zip(ps, states):
        if st is None:
            ms.append(torch.zeros(p.numel())); vs.append(torch.zeros(p.numel()))
            continue
        if tuple(st[1].shape) != tuple(p.shape):
            raise RuntimeError("optimiser archive: moment shape %s does not match parameter shape %s" % (tuple(st[1].shape), tuple(p.shape)))
        step = max(step, st[0])
        ms.append(st[1].reshape(-1).float()); vs.append(st[2].reshape(-1).float())
    return step, torch.cat(ms), torch.cat(vs)
