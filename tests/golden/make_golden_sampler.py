#!/usr/bin/env python3
"""Generate tests/golden/samplers.npz from the *reference's* OpenFOAMDataset / OpenFOAMSampler /
OpenFOAMEvaluationSampler (turbdiff/data/ofles.py:424-540), run in the build container on a fake repository
(three cases with 37 / 12 / 23 time steps, the first seconds discarded).

    python tests/golden/make_golden_sampler.py
"""
import random
import sys
from pathlib import Path

import numpy as np

OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(OUT))
from make_golden import REF, install_stubs  # noqa: E402

TIMES = [np.arange(37) * 0.01, np.arange(12) * 0.01 + 0.05, np.arange(23) * 0.02]
DISCARD = 0.075


class FakeRepo:
    def __init__(self):
        self.times = TIMES
        self.n_cases = len(TIMES)

    def reset_caches(self):
        pass

    def read(self, file_idx, samples):
        return ("case", file_idx, [int(s) for s in samples])


def main():
    install_stubs()
    import more_itertools  # stub from install_stubs: chunked must really chunk here

    def chunked(it, n):
        it = list(it)
        return [it[i:i + n] for i in range(0, len(it), n)]

    more_itertools.chunked = chunked
    sys.path.insert(0, str(REF))
    import torch.utils.data

    # the reference targets torch 2.6, whose Sampler.__init__ still takes (and ignores) the data source
    torch.utils.data.Sampler.__init__ = lambda self, *a, **k: None
    import turbdiff.data.ofles as R

    R.chunked = chunked
    ds = R.OpenFOAMDataset(FakeRepo(), stats="STATS", discard_first_seconds=DISCARD)
    out = {"discard": np.array(DISCARD), "len": np.array(len(ds))}
    for i, t in enumerate(TIMES):
        out[f"times/{i}"] = t
        out[f"valid_steps/{i}"] = np.asarray(ds.valid_steps[i])
    flat = lambda batches: (np.array([len(b) for b in batches]), np.array([i for b in batches for i in b]))
    for bs in (1, 4, 6):
        s = R.OpenFOAMSampler(ds, batch_size=bs, shuffle=False)
        out[f"train/bs{bs}/len"] = np.array(len(s))
        out[f"train/bs{bs}/plain/sizes"], out[f"train/bs{bs}/plain/idx"] = flat(list(s))
        random.seed(1000 + bs)
        s = R.OpenFOAMSampler(ds, batch_size=bs, shuffle=True)
        out[f"train/bs{bs}/shuffled/sizes"], out[f"train/bs{bs}/shuffled/idx"] = flat(list(s))
    for bs, spf in ((8, 8), (3, 5)):
        s = R.OpenFOAMEvaluationSampler(ds, batch_size=bs, samples_per_file=spf)
        out[f"eval/bs{bs}_spf{spf}/len"] = np.array(len(s))
        out[f"eval/bs{bs}_spf{spf}/sizes"], out[f"eval/bs{bs}_spf{spf}/idx"] = flat(list(s))
    # __getitem__: which (file, steps) a batch of flat indices reads
    reads = []
    for batch in ([0], [3, 1, 2], [29 + 2, 29 + 0], [29 + 7 + 5]):
        b = ds[list(batch)]
        reads.append((b.data[1], b.data[2]))
    out["getitem/files"] = np.array([r[0] for r in reads])
    for i, r in enumerate(reads):
        out[f"getitem/steps/{i}"] = np.array(r[1])
    b = ds.get_times(2, [0.1, 0.3])
    out["get_times/steps"] = np.array(b.data[2])
    np.savez_compressed(OUT / "samplers.npz", **out)
    print("wrote", OUT / "samplers.npz", len(out), "arrays; len(ds) =", len(ds))


if __name__ == "__main__":
    main()
