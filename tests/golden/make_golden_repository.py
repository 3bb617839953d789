#!/usr/bin/env python3
"""Golden vectors for the case-file reader, from the reference's own OpenFOAMDataRepository.

Build container only.  h5py is not installed here, so the reference's reader (turbdiff/data/ofles.py:320-421,
unmodified) is run with ``h5py`` replaced by tests/h5fake.py -- an in-memory stand-in with h5py's File / Group / Dataset
interface, incl. its rule that fancy indices be increasing and unique -- on seeded case trees laid out as
scripts/foam2h5.py + scripts/grid-embedding.py write them.  Recorded in tests/golden/repository.npz: times, every
metadata field, and read_data / read for unsorted index lists with duplicates.

    python tests/golden/make_golden_repository.py
"""

import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent))
import h5fake  # noqa: E402
from make_golden import REF, _stub, install_stubs  # noqa: E402

REQUESTS = [[5, 2, 2, 7], [0], [10, 9, 8, 0, 10], [3, 4, 5]]


def main():
    install_stubs()
    _stub("h5py", File=h5fake.File, Group=h5fake.Group)
    import cachetools  # the stub's cachedmethod is the identity decorator: fine for one-shot reads
    sys.path.insert(0, str(REF))
    from turbdiff.data.ofles import OpenFOAMDataRepository, Variable

    files = h5fake.install_cases()
    out = {}
    for phase, paths in files.items():
        repo = OpenFOAMDataRepository(paths, (Variable.U, Variable.P, Variable.NUT))
        for i, t in enumerate(repo.times):
            out[f"{phase}/{i}/times"] = np.asarray(t)
            m = repo.read_metadata(i)
            out[f"{phase}/{i}/cell_counts"] = np.asarray(m.cell_counts)
            out[f"{phase}/{i}/cell_idx"] = m.cell_idx.numpy()
            out[f"{phase}/{i}/h"] = m.h.numpy()
            out[f"{phase}/{i}/nu"] = np.array(m.nu)
            out[f"{phase}/{i}/case_name"] = np.array(m.case_name)
            out[f"{phase}/{i}/holes"] = np.stack([np.concatenate((h.pos, h.size)) for h in m.holes])
            for name, desc in m.boundaries.items():
                out[f"{phase}/{i}/boundary/{name}/idx"] = desc["idx"].numpy()
                out[f"{phase}/{i}/boundary/{name}/type"] = np.array(desc["type"])
            for var, per in m.boundary_conditions.items():
                for bname, bc in per.items():
                    out[f"{phase}/{i}/bc/{var.name}/{bname}/type"] = np.array(bc.type.name)
                    if bc.value is not None:
                        out[f"{phase}/{i}/bc/{var.name}/{bname}/value"] = bc.value.numpy()
            for r, req in enumerate(REQUESTS):
                data = repo.read(i, req)
                out[f"{phase}/{i}/read/{r}/t"] = data.t.numpy()
                for v, x in data.samples.items():
                    out[f"{phase}/{i}/read/{r}/{v.name}"] = x.numpy()
    np.savez_compressed(HERE / "repository.npz", **out)
    print("wrote", HERE / "repository.npz", len(out), "arrays")


if __name__ == "__main__":
    main()
