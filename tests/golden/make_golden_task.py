#!/usr/bin/env python3
"""Golden expectations for config-keyed construction of the training task.

Runs only in the build container: imports the *reference's* ``DiffusionTraining`` (third-party packages that are
not installed are replaced by the same empty stand-ins as in make_golden.py), builds it the way
``turbdiff/config.py:64-101`` does from a run configuration -- the ``model`` group is read from the reference's
``config/model/diffusion.yaml``, variants override single keys -- and records what the build must reproduce:

    tests/golden/task_configs.json
        per case: the run configuration (input) and, from the reference task, the state_dict key -> shape map,
        the optimiser class, the LR-schedule factors at a few steps, the conditioning widths and the
        hyper-parameters the task keeps as attributes.

    python tests/golden/make_golden_task.py
"""

import json
import sys
from pathlib import Path

import torch
import yaml
from torch import nn

sys.path.insert(0, str(Path(__file__).resolve().parent))
from make_golden import REF, install_stubs  # noqa: E402

OUT = Path(__file__).resolve().parent


def build_reference_task(run):
    """What instantiate_data_and_task does for the diffusion task (config.py:64-101), without the datamodule:
    max_train_steps = max_epochs * len(train_dataloader) with a given loader length."""
    import turbdiff.models.diffusion as Dm
    from turbdiff.data.ofles import Variable

    class _NoMetrics(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    m = run["model"]
    variables = tuple(Variable) if m["variables"] is None else tuple(Variable.from_str(v) for v in m["variables"].split(","))
    kw = dict(
        data_dir=Path(run["data"]["root"]) / "data", samples_root=Path(run["samples_root"]) / "explore" / "debug",
        dim=m["dim"], cell_type_embedding_type=m["cell_type_embedding_type"],
        cell_type_embedding_dim=m["cell_type_embedding_dim"], normalization_mode=m["normalization_mode"],
        variables=variables, beta_schedule=m["beta_schedule"], timesteps=m["timesteps"],
        learning_rate=float(m["learning_rate"]), min_learning_rate=float(m["min_learning_rate"]), lr_decay=m["lr_decay"],
        max_train_steps=m["max_epochs"] * run["steps_per_epoch"], loss=m["loss"],
        cell_type_features=m["cell_type_features"], cell_pos_features=m.get("cell_pos_features", False),
        clip_denoised=m["clip_denoised"], noise_bcs=m["noise_bcs"], learned_variances=m["learned_variances"],
        elbo_weight=float(m["elbo_weight"]) if m["elbo_weight"] is not None else None,
        detach_elbo_mean=m["detach_elbo_mean"], time_embedding=m["time_embedding"], actfn=m["actfn"],
        optimizer=m["optimizer"], norm_type=m["norm_type"], with_geometry_embedding=m["with_geometry_embedding"],
    )
    try:
        return Dm.DiffusionTraining(**kw), True
    except Exception:  # the metric collection needs data files on disk: build the task without it
        Dm.SampleMetricsCollection = _NoMetrics
        Dm.SampleStore = lambda *a, **k: None
        return Dm.DiffusionTraining(**kw), False


def build_reference_defaults():
    import turbdiff.models.diffusion as Dm

    class _NoMetrics(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    kw = dict(data_dir=Path("data/shapes/data"), samples_root=Path("data/samples/explore/debug"))
    try:
        return Dm.DiffusionTraining(**kw), True
    except Exception:
        Dm.SampleMetricsCollection = _NoMetrics
        Dm.SampleStore = lambda *a, **k: None
        return Dm.DiffusionTraining(**kw), False


def main():
    install_stubs()
    sys.path.insert(0, str(REF))
    model_group = yaml.safe_load((REF / "config" / "model" / "diffusion.yaml").read_text())
    base = {"model": model_group, "data": {"root": "data/shapes", "name": "shapes"}, "samples_root": "data/samples",
            "trainer": {"gradient_clip_val": 0.1}, "matmul_precision": "medium", "steps_per_epoch": 37}
    variants = {
        "shipped": {},
        "onehot_cellpos": {"cell_type_embedding_type": "onehot", "cell_pos_features": True},
        "learned_variances": {"learned_variances": True, "elbo_weight": 0.1, "loss": "l1", "clip_denoised": True},
        "instance_gelu_adam": {"norm_type": "instance", "actfn": "gelu", "optimizer": "adam", "lr_decay": None,
                               "variables": "u", "normalization_mode": "mean-std", "dim": 16, "timesteps": 100},
        "no_cell_types": {"cell_type_features": False, "cell_pos_features": True, "optimizer": "adamw"},
    }
    cases = {}
    for name, over in list(variants.items()) + [("ctor_defaults", None)]:
        run = json.loads(json.dumps(base))
        torch.manual_seed(0)
        if over is None:
            # the constructor's OWN defaults (diffusion.py:42-70): only the two required paths are passed
            run = None
            task, with_metrics = build_reference_defaults()
        else:
            run["model"].update(over)
            task, with_metrics = build_reference_task(run)
        cfg = task.configure_optimizers()
        opt = cfg["optimizer"]
        factors = None
        if "lr_scheduler" in cfg:
            lam = cfg["lr_scheduler"]["scheduler"].lr_lambdas[0]
            factors = {str(s): lam(s) for s in (0, 1, 10, task.max_train_steps // 2, task.max_train_steps, task.max_train_steps + 5)}
        cases[name] = {
            "run_config": run,
            "expect": {
                "state_dict": {k: list(v.shape) for k, v in task.state_dict().items()},
                "metric_buffers_included": with_metrics,
                "optimizer": type(opt).__name__, "lr": opt.param_groups[0]["lr"], "lr_factors": factors,
                "max_train_steps": task.max_train_steps,
                "local_conditioning_dim": task.conditioning.local_conditioning_dim,
                "global_conditioning_dim": task.conditioning.global_conditioning_dim,
                "variables": [v.name for v in task.variables],
                "attrs": {k: getattr(task, k) for k in ("dim", "timesteps", "beta_schedule", "loss", "clip_denoised", "noise_bcs",
                                                        "learned_variances", "elbo_weight", "detach_elbo_mean", "actfn",
                                                        "optimizer", "lr_decay", "normalization_mode")},
                "n_parameters": sum(p.numel() for p in task.parameters()),
            },
        }
        print(name, len(cases[name]["expect"]["state_dict"]), "keys,", cases[name]["expect"]["n_parameters"], "parameters")
    (OUT / "task_configs.json").write_text(json.dumps(cases, indent=1, sort_keys=True) + "\n")


if __name__ == "__main__":
    main()
