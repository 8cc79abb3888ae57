"""`*.th` module checkpoints in the reference's wire format.

The reference writes networks with `torch::serialize::OutputArchive` (evo_motion_networks/include/evo_motion_networks/
saver.h:13-25) and reads them with `InputArchive` + `Module::load` (saver.h:27-39): a TorchScript zip archive whose
module hierarchy carries the parameters under the reference's names (`head.0.weight … mu.0.weight, sigma.0.weight`
for the actor, actor.cpp:9-28).  A scripted mirror module has exactly that hierarchy, so `torch.jit.save` of it is a
file the reference's `load_torch` accepts, and `torch.jit.load` reads what `save_torch` wrote.

Optimiser archives (`*_optimizer.th`, ppo_gae.cpp:194,196) are not produced: their keys are parameter addresses of
the saving process (torch::optim::serialize) and they are not needed to run a policy.
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
