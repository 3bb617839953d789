"""Conditioning dictionary helpers (interface of turbdiff/models/conditioning.py:13-93).

``C`` maps a conditioning type to an unbatched (c, X, Y, Z) tensor.  The keys only need a
boolean ``local`` / ``global_`` attribute, so the reference's own ``Conditioning.Type`` enum
members work unchanged when this package is used as a drop-in.
"""

import enum

import torch


class Conditioning:
    class Type(enum.Enum):
        CELL_TYPE = enum.auto()
        CELL_POS = enum.auto()

        @property
        def local(self):
            return True

        @property
        def global_(self):
            return False


def _gather(C, attr):
    parts = [v for k, v in C.items() if getattr(k, attr)]
    return torch.cat(parts, dim=0) if parts else None


def local_conditioning(C):
    """Channel concatenation (dict order) of the per-cell conditionings, or None."""
    return _gather(C, "local")


def global_conditioning(C):
    return _gather(C, "global_")
