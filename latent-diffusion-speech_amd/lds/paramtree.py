"""nn.Module shells whose only job is to own parameters under the reference's state_dict keys."""
import torch
from torch import nn

from . import init_weights


class ParamTree(nn.Module):
    """Nested modules built from dotted parameter names, e.g.
    'down_blocks.0.resnets.0.conv1.weight' -> self.down_blocks.0.resnets.0.conv1.weight"""

    def __init__(self, shapes=None, seed=0, prefix=""):
        super().__init__()
        if shapes:
            for key, shape in shapes.items():
                node = self
                parts = key.split(".")
                for p in parts[:-1]:
                    if p not in node._modules:
                        node.add_module(p, ParamTree())
                    node = node._modules[p]
                val = torch.from_numpy(init_weights.init_tensor(prefix + key, tuple(shape), seed))
                node.register_parameter(parts[-1], nn.Parameter(val, requires_grad=False))
