"""seeded_params(): deterministic weights by state_dict key -- shared by make_golden.py (applied to the REFERENCE
model) and by the tests (applied to the mirror / handed to the oracle), so that large models (PCN: 15 MB of weights)
can be pinned without storing their parameters."""
import numpy as np
import torch


def seeded_params(module, seed):
    """Overwrite every floating-point entry of module.state_dict(), in sorted key order, with U(-a, a) values from a
    per-key seeded CPU generator (a = 1/sqrt(fan_in) for weights, 0.1 for 1-d tensors; BatchNorm `running_var` U(0.5, 1.5),
    BatchNorm `weight` U(0.5, 1.5)).  Depends only on the key names and shapes, not on construction order."""
    sd = module.state_dict()
    for i, k in enumerate(sorted(sd.keys())):
        v = sd[k]
        if not v.dtype.is_floating_point:
            continue
        a = 0.1 if v.dim() < 2 else 1.0 / float(np.sqrt(v[0].numel()))
        g = torch.Generator().manual_seed(seed + i)
        r = torch.rand(v.shape, generator=g)
        prefix = k.rsplit(".", 1)[0]
        if k.endswith("running_var") or (k.endswith(".weight") and v.dim() == 1 and prefix + ".running_var" in sd):
            v.copy_(0.5 + r)                     # BatchNorm variance / scale: positive, order one
        else:
            v.copy_((r * 2 - 1) * a)
    return module
