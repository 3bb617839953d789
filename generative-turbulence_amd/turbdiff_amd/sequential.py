"""nn.Sequential that hands keyword arguments only to the members that can take them.

Interface of the reference's ``turbdiff/sequential.py:10-39`` (``decode`` and
``center_block`` of the DenoisingModel are built from it, so its presence is part of the
module tree / state_dict schema).
"""

import inspect

import torch.nn as nn


def _accepted_keywords(module: nn.Module):
    params = inspect.signature(module.forward).parameters.values()
    if any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params):
        return None  # takes everything
    return frozenset(p.name for p in params)


class KwargsSequential(nn.Sequential):
    def __init__(self, *modules):
        super().__init__(*modules)
        self._keywords = [_accepted_keywords(m) for m in modules]

    def forward(self, input, *args, **kwargs):
        out = input
        for module, accepted in zip(self, self._keywords):
            kw = kwargs if accepted is None else {k: v for k, v in kwargs.items() if k in accepted}
            out = module(out, *args, **kw)
        return out
