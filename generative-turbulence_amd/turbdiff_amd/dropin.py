"""Register turbdiff_amd's modules under the reference's import paths.

    import turbdiff_amd.dropin as dropin; dropin.install()

After this, ``import turbdiff.models.ddpm`` (as done by the reference's
``turbdiff/models/diffusion.py:16`` and ``turbdiff/config.py``) resolves to
``turbdiff_amd.models.ddpm`` and so on, while every other reference module (data loading,
Lightning task, metrics, ...) is imported from the reference unchanged.  Must run before the
reference's ``turbdiff.models.diffusion`` is imported.
"""

from __future__ import annotations

import importlib
import sys

_MAP = {
    "turbdiff.models.ddpm": "turbdiff_amd.models.ddpm",
    "turbdiff.models.attention": "turbdiff_amd.models.attention",
    "turbdiff.models.utils": "turbdiff_amd.models.utils",
    "turbdiff.sequential": "turbdiff_amd.sequential",
}


def install() -> dict[str, object]:
    installed = {}
    for ref_name, ours in _MAP.items():
        if ref_name in sys.modules and sys.modules[ref_name].__name__ != ours:
            raise RuntimeError(f"{ref_name} is already imported from the reference; call dropin.install() first")
        mod = importlib.import_module(ours)
        sys.modules[ref_name] = mod
        installed[ref_name] = mod
    return installed


def uninstall():
    for ref_name, ours in _MAP.items():
        if ref_name in sys.modules and sys.modules[ref_name].__name__ == ours:
            del sys.modules[ref_name]
