"""The callers of the hot path, without Lightning / hydra / h5py.

``DiffusionTrainer`` reproduces what the reference's ``DiffusionTraining`` task
(turbdiff/models/diffusion.py:41-243) does around ``GaussianDiffusion``:

* construction from the keys of ``config/model/diffusion.yaml`` (dim, timesteps, beta_schedule,
  loss, noise_bcs, learned_variances, clip_denoised, norm_type, actfn, optimizer, learning_rate,
  min_learning_rate, lr_decay, cell_type_embedding_dim, ...) -- u_net_levels = 4 is hard-coded
  there (diffusion.py:120) and here;
* ``_model_input`` (diffusion.py:237-242): for an ``OpenFOAMBatch`` (sparse cell lists, metadata, stats) the
  fused ingress kernels -- grid embedding + normalisation in one pass, cell-type embedding
  (``turbdiff_amd.gridio``; ofles.py:220-240, normalization.py:19-23, cell_type_embeddings.py:62-70); for a
  dense batch the same arithmetic as torch glue;
* ``training_step`` = ``GaussianDiffusion.forward``; ``sample`` = ``p_sample_loop`` + denormalise
  (diffusion.py:152-165); ``configure_optimizers`` = RAdam + per-step exponential LambdaLR
  (diffusion.py:210-235); ``fit_step`` adds the trainer's clip-by-norm 0.1 (train.yaml:30-31);
* ``measure_sample_time``: the reference's only timing harness (scripts/evaluate-runtime.py:54-96:
  synchronize, perf_counter_ns around ``task.sample`` at batch size 1).

The ``state_dict`` has the reference task's 149 keys (``model.model.*``, the two aliases of the
cell-type embedding, and the 8 metric buffers as placeholders), so the reference's checkpoints load
with ``strict=True``.  A batch is an ``OpenFOAMBatch`` (``turbdiff_amd.data.ofles`` or the reference's own,
on the device) or any object with ``x`` (B, F, X, Y, Z) in physical units, ``cell_idx`` (flat in-domain
cell indices), ``cell_types`` (X, Y, Z) int64 in [0, 6), and the per-feature ``mean`` / ``std``.
"""

from __future__ import annotations

import math
import os
import time
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

from . import ops
from .data.ofles import Variable
from .models.cell_type_embeddings import CellTypeEmbedding
from .models.conditioning import Conditioning
from .models.ddpm import DenoisingModel, GaussianDiffusion
from .models.normalization import Normalization

ACTFNS = {"silu": nn.SiLU, "gelu": nn.GELU, "relu": nn.ReLU, "softplus": nn.Softplus, "tanh": nn.Tanh}
N_CELL_TYPES = 6  # inside, outside, walls, inlets, outlets, empties (cell_type_embeddings.py:30-38)


class _Holder(nn.Module):
    """Empty module used to reproduce nested state_dict paths."""


def _metric_placeholders():
    """`{val,test}_sample_metrics.metrics.0.distance.*` buffers of the reference task
    (models/metrics.py); evaluation metrics are out of scope, the buffers only keep checkpoints
    strict-loadable (legendre_* hold the real Gauss-Legendre rule, tke_spectrum the device implementation of
    the reference's TurbulentKineticEnergySpectrum with the same Lebedev rule)."""
    nodes, weights = np.polynomial.legendre.leggauss(64)
    coll = _Holder()
    coll.metrics = nn.ModuleList([_Holder()])
    dist = _Holder()
    dist.register_buffer("legendre_nodes", torch.tensor(nodes, dtype=torch.float32))
    dist.register_buffer("legendre_weights", torch.tensor(weights, dtype=torch.float32))
    from .models.metrics import TurbulentKineticEnergySpectrum

    dist.tke_spectrum = TurbulentKineticEnergySpectrum()  # 5810-node Lebedev rule, the device spectrum (§8 f3)
    coll.metrics[0].distance = dist
    return coll


# torch.set_float32_matmul_precision(config.matmul_precision) of the reference's entry points (train.py,
# scripts/eval_ckpt.py:57) -> compute mode here: "highest" keeps IEEE fp32 products (fp32 MFMA convs); "high" and
# "medium" allow reduced-precision products on fp32 tensors (TF32 / bf16 in stock PyTorch) -> split-precision convs
# (bf16 hi + lo: 16 significand bits, more than TF32's 10).  16-bit STORAGE is never implied by a reference config;
# ask for it explicitly: compute_mode="fp16" (IEEE half tensors + fp16 MFMA operands: TF32's 11 significand bits at the
# bf16 kernels' speed; training runs under a power-of-two loss scale, optim.ClipRAdam) or "bf16" (8 bits, no scale needed).
MATMUL_PRECISION_TO_MODE = {"highest": "f32", "high": "f32s", "medium": "f32s"}
COMPUTE_MODES = ("f32", "f32s", "bf16", "fp16")
_MODE_DTYPE = {"f32": torch.float32, "f32s": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}


def apply_compute_mode(net: DenoisingModel, mode: str) -> None:
    """f32: fp32 tensors, IEEE fp32 MFMA convs; f32s: fp32 tensors, split-precision convs; bf16 / fp16: 16-bit tensors."""
    if mode not in COMPUTE_MODES:
        raise ValueError(f"compute mode {mode!r} not in {COMPUTE_MODES}")
    net.set_compute_dtype(_MODE_DTYPE[mode])
    # per model, not per process: a second trainer in another mode leaves this one's arithmetic alone
    net.conv_impl = "split" if mode == "f32s" else "auto"


def _get(cfg, key, default=None):
    """cfg[key] for dict-like (dict, OmegaConf DictConfig) and attribute-style (SimpleNamespace) configs."""
    if cfg is None:
        return default
    if hasattr(cfg, "get") and callable(cfg.get):
        v = cfg.get(key, default)
        return default if v is None and default is not None else v
    return getattr(cfg, key, default)


def _has(cfg, key) -> bool:
    return key in cfg if hasattr(cfg, "__contains__") else hasattr(cfg, key)


def _raw(cfg, key):
    return cfg[key] if hasattr(cfg, "__getitem__") else getattr(cfg, key)


class DiffusionTrainer(nn.Module):
    """Constructor surface = the reference task's (``DiffusionTraining.__init__``, diffusion.py:42-70: the 27
    keywords ``config.py:74-101`` forwards), plus ``gradient_clip_val`` (the trainer's, train.yaml:30-31),
    ``u_net_levels`` (hard-coded 4 in the reference, diffusion.py:120) and ``compute_mode``."""

    # defaults = the reference constructor's own (diffusion.py:42-70), NOT the shipped YAML's: a caller that relies on
    # defaults gets the same task from both; the shipped configuration is SHIPPED_CONFIG / from_config
    def __init__(self, data_dir=None, samples_root=None, dim: int = 32, cell_type_embedding_type: str = "learned",
                 cell_type_embedding_dim: int = 4, normalization_mode: str = "mean-std", variables=tuple(Variable),
                 beta_schedule: str = "sigmoid", timesteps: int = 100, learning_rate: float = 1e-3,
                 min_learning_rate: float = 1e-3, lr_decay: str | None = None, max_train_steps: int = 1000,
                 loss: str = "l1", cell_type_features: bool = True, cell_pos_features: bool = False,
                 clip_denoised: bool = False, noise_bcs: bool = False, learned_variances: bool = False,
                 elbo_weight: float | None = None, detach_elbo_mean: bool = True, time_embedding: str = "nyquist",
                 actfn: str = "silu", optimizer: str = "adam", norm_type: str = "instance",
                 with_geometry_embedding: bool = True, *, gradient_clip_val: float = 0.1, u_net_levels: int = 4,
                 compute_dtype: torch.dtype | None = None, compute_mode: str | None = None, n_features: int | None = None):
        super().__init__()
        self.data_dir, self.samples_root = data_dir, samples_root
        self.variables = tuple(v if isinstance(v, Variable) else Variable.from_str(v) for v in variables)
        vars_dim = sum(v.dims for v in self.variables)
        assert n_features is None or n_features == vars_dim, "n_features follows from the variables"
        self.normalization_mode = normalization_mode
        self.normalization = Normalization(self.variables, normalization_mode)
        self.dim, self.timesteps, self.beta_schedule, self.loss = dim, timesteps, beta_schedule, loss
        self.cell_type_embedding_type, self.cell_type_embedding_dim = cell_type_embedding_type, cell_type_embedding_dim
        self.cell_type_features, self.cell_pos_features = cell_type_features, cell_pos_features
        self.time_embedding = time_embedding  # accepted and unused, as in the reference (diffusion.py:64, never read)
        self.cell_type_embedding = (CellTypeEmbedding.create(cell_type_embedding_type, cell_type_embedding_dim)
                                    if cell_type_features else None)
        # holds the embedding a second time, as in the reference (diffusion.py:89-93) -> two state_dict aliases
        self.conditioning = Conditioning(self.variables, self.cell_type_embedding, cell_pos_features)
        net = DenoisingModel(in_features=vars_dim, out_features=vars_dim * (2 if learned_variances else 1),
                             c_local_features=self.conditioning.local_conditioning_dim,
                             c_global_features=self.conditioning.global_conditioning_dim, timesteps=timesteps, dim=dim,
                             u_net_levels=u_net_levels, actfn=ACTFNS[actfn], norm_type=norm_type,
                             with_geometry_embedding=with_geometry_embedding)
        if compute_mode is None:
            compute_mode = {torch.bfloat16: "bf16", torch.float16: "fp16"}.get(compute_dtype)
        self.compute_mode = compute_mode
        if compute_mode is not None:
            apply_compute_mode(net, compute_mode)
        elif compute_dtype is not None:
            net.set_compute_dtype(compute_dtype)
        self.model = GaussianDiffusion(net, timesteps=timesteps, beta_schedule=beta_schedule, loss_type=loss,
                                       clip_denoised=clip_denoised, noise_bcs=noise_bcs,
                                       learned_variances=learned_variances, elbo_weight=elbo_weight,
                                       detach_elbo_mean=detach_elbo_mean)
        self.val_sample_metrics = _metric_placeholders()
        self.test_sample_metrics = _metric_placeholders()
        self.learning_rate, self.min_learning_rate = learning_rate, min_learning_rate
        self.lr_decay, self.max_train_steps, self.optimizer = lr_decay, max_train_steps, optimizer
        self.clip_denoised, self.noise_bcs, self.learned_variances = clip_denoised, noise_bcs, learned_variances
        self.elbo_weight, self.detach_elbo_mean, self.actfn = elbo_weight, detach_elbo_mean, actfn
        self.gradient_clip_val = gradient_clip_val
        self.fused_optimizer = True  # ClipRAdam on GPU; False -> clip_grad_norm_ + torch.optim.RAdam
        # fp16 training: initial loss scale (a power of two), or None = chosen from the first batch so that the seed of the
        # backward pass, 2 (eps_hat - eps) / (B F n_cells), lands near 2^-3 (initial_loss_scale)
        self.loss_scale = None
        self._opt = self._sched = None
        self.ddp = None  # set to a parallel.BucketedDataParallel(self) for multi-GPU training
        self.stats = None

    # the model group of the shipped run configuration (config/model/diffusion.yaml:1-41), as keywords
    SHIPPED_CONFIG = dict(dim=32, variables=("u", "p"), beta_schedule="log-snr-linear", timesteps=500, learning_rate=1e-4,
                          min_learning_rate=1e-6, lr_decay="exp", loss="l2", noise_bcs=True, optimizer="radam",
                          norm_type="group", with_geometry_embedding=False)

    # the keys config.py:74-101 reads off ``config.model`` and forwards under the same name
    CONFIG_KEYS = ("dim", "cell_type_embedding_type", "cell_type_embedding_dim", "normalization_mode", "beta_schedule",
                   "timesteps", "learning_rate", "min_learning_rate", "lr_decay", "loss", "cell_type_features",
                   "clip_denoised", "noise_bcs", "learned_variances", "elbo_weight", "detach_elbo_mean", "time_embedding",
                   "actfn", "optimizer", "norm_type", "with_geometry_embedding")

    @classmethod
    def from_config(cls, config, *, max_train_steps: int | None = None, steps_per_epoch: int | None = None,
                    compute_mode: str | None = None, **overrides):
        """Build the task from a run configuration, as ``instantiate_data_and_task`` does (config.py:64-101).

        ``config`` is either the whole run config (``config.model``, ``config.data.root``, ``config.samples_root``,
        ``config.trainer.gradient_clip_val``, ``config.matmul_precision`` -- what a checkpoint's ``ckpt["config"]``
        holds) or just its ``model`` group (the keys of ``config/model/diffusion.yaml``); dicts, OmegaConf nodes and
        namespaces all work.  ``max_train_steps`` = ``model.max_epochs * len(train_dataloader)`` in the reference:
        pass it, or ``steps_per_epoch``.  ``compute_mode`` overrides the mapping of ``matmul_precision``."""
        model = _get(config, "model")
        root = config if model is not None else None
        model = model if model is not None else config
        name = _get(model, "name", "diffusion")
        if not str(name).startswith("diffusion"):
            raise ValueError(f"model.name = {name!r}: only the diffusion task is built here")
        # a key that is present is forwarded as is -- None included (lr_decay: ~, elbo_weight: ~), as config.py does
        kw = {k: _raw(model, k) for k in cls.CONFIG_KEYS if _has(model, k)}
        var = _get(model, "variables")
        kw["variables"] = tuple(Variable) if var is None else tuple(
            Variable.from_str(v) for v in (var.split(",") if isinstance(var, str) else var))
        kw["cell_pos_features"] = bool(_get(model, "cell_pos_features", False))
        if max_train_steps is None and steps_per_epoch is not None:
            max_train_steps = int(_get(model, "max_epochs", 1)) * int(steps_per_epoch)
        if max_train_steps is not None:
            kw["max_train_steps"] = int(max_train_steps)
        if root is not None:
            data_root = _get(_get(root, "data"), "root")
            if data_root is not None:
                from pathlib import Path

                kw["data_dir"] = Path(data_root) / "data"
            if _get(root, "samples_root") is not None:
                kw["samples_root"] = _get(root, "samples_root")
            clip = _get(_get(root, "trainer"), "gradient_clip_val")
            if clip is not None:
                kw["gradient_clip_val"] = float(clip)
            if compute_mode is None and _get(root, "matmul_precision") is not None:
                compute_mode = MATMUL_PRECISION_TO_MODE[str(_get(root, "matmul_precision"))]
        kw["compute_mode"] = compute_mode or "f32"
        for k in ("learning_rate", "min_learning_rate", "elbo_weight"):  # YAML 1.1 reads "1e-4" as a string
            if isinstance(kw.get(k), str):
                kw[k] = float(kw[k])
        kw.update(overrides)
        return cls(**kw)

    # ---- the glue around the hot path ------------------------------------------------------
    @staticmethod
    def normalize_grid(x, mean, std):
        m, s = mean.view(-1, 1, 1, 1), std.view(-1, 1, 1, 1)
        return torch.addcmul(-m / s, torch.reciprocal(s), x)

    @staticmethod
    def denormalize_grid(x, mean, std):
        return torch.addcmul(mean.view(-1, 1, 1, 1), std.view(-1, 1, 1, 1), x)

    @staticmethod
    def _is_openfoam_batch(batch):
        return hasattr(batch, "data") and hasattr(batch, "stats")

    def _model_input(self, batch):
        """diffusion.py:237-242.  An ``OpenFOAMBatch`` (sparse cell lists + metadata + stats) goes through
        the fused ingress kernels; a dense batch (``x``, ``cell_types``, ``mean``, ``std``) through torch glue."""
        if self._is_openfoam_batch(batch):
            x = self.normalization.normalized_grid_embedding(batch.data, batch.stats)
            return x, self.conditioning(batch.data)
        x = self.normalize_grid(batch.x, batch.mean, batch.std)
        C = {Conditioning.Type.CELL_TYPE: torch.movedim(self.cell_type_embedding.embedding(batch.cell_types), -1, 0)}
        return x, C

    def _cell_idx(self, batch):
        return batch.data.metadata.cell_idx if self._is_openfoam_batch(batch) else batch.cell_idx

    def training_step(self, batch):
        x, C = self._model_input(batch)
        loss, _ = self.model(x, C, SimpleNamespace(cell_idx=self._cell_idx(batch)), None)
        return loss

    @torch.no_grad()
    def sample(self, batch, start_from=None, noise_fn=None):
        """diffusion.py:152-158: dense denormalised samples (B, F, X, Y, Z)."""
        x = self._sample_normalized(batch, start_from, noise_fn)
        if self._is_openfoam_batch(batch):
            return self.normalization.denormalize_grid(x, batch.stats)
        return self.denormalize_grid(x, batch.mean, batch.std)

    @torch.no_grad()
    def sample_cells(self, batch, start_from=None, noise_fn=None):
        """The samples as ``SampleStore.add_samples`` stores them (metrics.py:52-58): per variable the
        denormalised in-domain values, channels-last (B, n_cells, dims) -- one fused egress kernel."""
        x = self._sample_normalized(batch, start_from, noise_fn)
        return self.normalization.denormalized_cells(x, batch.data.metadata, batch.stats)

    def _sample_normalized(self, batch, start_from, noise_fn):
        x, C = self._model_input(batch)
        kw = {}
        if noise_fn is None and not self.model.learned_variances:
            # identically seeded ranks draw the same per-call nonce: the trajectory ids keep their noise streams apart
            # (rank r samples trajectories r B ... r B + B - 1 of the global set, whatever the sharding)
            kw["trajectory_ids"] = self._global_trajectory_ids(x.shape[0])
        return self.model.p_sample_loop(x, C, self._cell_idx(batch), pbar=False, start_from=start_from, noise_fn=noise_fn, **kw)

    @staticmethod
    def _global_trajectory_ids(B: int):
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return list(range(B))
        # offset = the batch sizes of the lower ranks (ranks may hold different B: a ragged last validation batch), not
        # rank * local B, which made ids overlap -- and identically seeded ranks then drew identical noise streams
        sizes = [None] * dist.get_world_size()
        dist.all_gather_object(sizes, int(B))
        first = sum(sizes[: dist.get_rank()])
        return [first + i for i in range(B)]

    @torch.no_grad()
    def validation_step(self, batch, store):
        """diffusion.py:167-175: remember the statistics, sample, hand the samples to the store."""
        if self.stats is None:
            self.stats = batch.stats
        store.add_cells(self.sample_cells(batch), batch.data.metadata)
        return {}

    def configure_optimizers(self):
        klass = {"adam": torch.optim.Adam, "adamw": torch.optim.AdamW, "radam": torch.optim.RAdam}.get(self.optimizer)
        if klass is None:
            raise RuntimeError(f"Unknown optimizer {self.optimizer}")
        params = list(self.parameters())
        if self.optimizer == "radam" and self.fused_optimizer and params[0].is_cuda:
            from .optim import ClipRAdam  # clip + RAdam fused; same arithmetic as the two torch calls

            opt = ClipRAdam(params, lr=self.learning_rate, max_norm=self.gradient_clip_val or None,
                            loss_scale=self._loss_scale_for_optimizer())
        else:
            if self.model.model.compute_dtype == torch.float16:
                raise RuntimeError("fp16 training needs the loss-scaling optimiser: optimizer='radam' with fused_optimizer on a GPU")
            opt = klass(params, lr=self.learning_rate)
        sched = None
        if self.lr_decay == "exp":
            rate = math.log(self.min_learning_rate / self.learning_rate) / self.max_train_steps
            sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda step: math.exp(rate * min(step, self.max_train_steps)))
        return opt, sched

    @staticmethod
    def initial_loss_scale(n_loss_elements: int) -> float:
        """Power-of-two loss scale for an error mean over `n_loss_elements` = B F n_cells values: the backward pass of
        S * loss then starts from ~(eps_hat - eps) / 8 instead of from 1e-7, far below fp16's smallest normal number."""
        return float(2.0 ** max(0, math.floor(math.log2(max(n_loss_elements, 1))) - 4))

    def _loss_scale_for_optimizer(self):
        if self.model.model.compute_dtype != torch.float16:
            return None
        return self.loss_scale if self.loss_scale is not None else 2.0**16

    def enable_graph_step(self, on: bool = True):
        """fit_step runs forward + backward as ONE captured hipGraph per input signature (GraphedTrainingStep) instead of
        ~700 eager launches: the host's share of a step drops from ~9 ms (B = 6 at 192x64x48) / ~6 ms (48x32x32, where
        it was the whole step) to the optimiser's ~1 ms.  With `self.ddp` (parallel.BucketedDataParallel) the hooks' staging
        kernels are captured and every bucket's all-reduce is started by the host when the captured backward passes the
        bucket's boundary (BucketedDataParallel.replay_launch): overlap as in the eager step."""
        self._graph_step = GraphedTrainingStep(self) if on else None
        return self

    def fit_step(self, batch):
        """zero_grad -> training_step -> backward -> [grad all-reduce] -> clip -> optimizer -> LR schedule."""
        if self._opt is None:
            if self.loss_scale is None and self.model.model.compute_dtype == torch.float16:
                with torch.no_grad():
                    x0, _ = self._model_input(batch)  # (first step only: the batch size behind the loss's mean)
                self.loss_scale = self.initial_loss_scale(int(x0.shape[0]) * int(x0.shape[1]) * int(self._cell_idx(batch).numel()))
            self._opt, self._sched = self.configure_optimizers()
        scale_loss = getattr(self._opt, "scale_loss", None)
        gs = getattr(self, "_graph_step", None)
        if gs is not None:
            loss = gs(batch)  # leaves the gradients in the tensors the captured backward owns
        else:
            self._opt.zero_grad(set_to_none=True)
            loss = self.training_step(batch)
            (loss if scale_loss is None else scale_loss(loss)).backward()
        if self.ddp is not None:
            self.ddp.finish()
        if self.gradient_clip_val and not hasattr(self._opt, "max_norm"):
            torch.nn.utils.clip_grad_norm_(self.parameters(), self.gradient_clip_val)
        self._opt.step()
        if self._sched is not None:
            self._sched.step()
        return loss.detach()

    @torch.no_grad()
    def measure_sample_time(self, batch, repeats: int = 1):
        """Seconds per `sample(batch)` with the reference's protocol (evaluate-runtime.py:63-84)."""
        best = float("inf")
        for _ in range(repeats):
            torch.cuda.synchronize()
            t0 = time.perf_counter_ns()
            self.sample(batch)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter_ns() - t0) * 1e-9)
        return best


class GraphedTrainingStep:
    """`DiffusionTrainer.training_step` + `backward()` as one hipGraph per input signature (shapes, arithmetic mode).

    Everything the captured step reads lives in buffers this object owns: the normalised input x, the conditioning
    tensors, the in-domain mask and the number of in-domain cells (a device scalar: `tdx_masked_loss_dyn`), so batches
    of ANY geometry of the same grid size replay the same graph after four small copies.  What stays eager around the
    replay: `_model_input` (fused ingress kernels, cell-type embedding) before it, and after it the short chain from the
    conditioning tensors' gradients back into whatever produced them (the learned cell-type table), then clip +
    optimiser as usual.  The draws of t and of the noise happen inside the graph (torch's graph-safe Philox
    generator: fresh per replay); weight re-packing is captured too, so every replay packs the CURRENT weights.

    Parameter gradients are the tensors the captured backward allocated: every call re-attaches them to `p.grad` and the
    replay overwrites them (`fit_step` skips `zero_grad` in this mode)."""

    MAX_SLOTS = 2

    def __init__(self, trainer, inject: bool = False):
        """inject=True (tests): the captured step reads its timesteps and noise from `self.t` / `self.noise` of the slot
        (set through `set_draws`) instead of drawing them -- the same arithmetic as an eager `p_losses(..., noise=)`."""
        from collections import OrderedDict

        self.tr = trainer
        self.slots = OrderedDict()
        self.stream = None
        self.inject = inject
        self._draws = None

    def set_draws(self, t, noise):
        self._draws = (t, noise)

    @staticmethod
    def _tensors(C):
        return {k: v for k, v in C.items() if torch.is_tensor(v)} if isinstance(C, dict) else {}

    def _signature(self, x, C):
        from . import _lib as L

        m = self.tr.model.model
        return (tuple(x.shape), str(x.device), tuple(sorted((str(k), tuple(v.shape), str(v.dtype), bool(v.requires_grad))
                                                            for k, v in self._tensors(C).items())),
                getattr(m, "compute_dtype", None), getattr(m, "conv_impl", None), L.conv_impl(),
                # what else the capture bakes in: parameter storage and which of them take gradients, the loss
                # configuration, whether the backward runs on a scaled loss
                tuple((id(p), p.data_ptr(), bool(p.requires_grad)) for p in self.tr.parameters()),
                self.tr.model.loss_type, bool(self.tr.model.noise_bcs), getattr(self.tr._opt, "loss_scale", None) is not None)

    def __call__(self, batch):
        tr = self.tr
        x, C = tr._model_input(batch)
        mask, n_cells = tr.model.domain_mask(tr._cell_idx(batch), x[0, 0].numel())
        if int(n_cells) <= 0:
            raise ValueError("no in-domain cells: the loss is a mean over an empty set (tdx_masked_loss rejects it too)")
        sig = self._signature(x, C)
        slot = self.slots.get(sig)
        if slot is None:
            slot = self.slots[sig] = self._capture(x, C, mask, n_cells)
            while len(self.slots) > self.MAX_SLOTS:
                self.slots.popitem(last=False)
        else:
            self.slots.move_to_end(sig)
        with torch.no_grad():
            slot.x.copy_(x)
            for k, v in self._tensors(C).items():
                slot.C[k].copy_(v)
            slot.mask.copy_(mask)
            slot.n.fill_(int(n_cells))
            slot.scale.fill_(float(getattr(tr._opt, "loss_scale", None) or 1.0))  # fp16: the optimiser's current loss scale
            if self.inject:
                slot.t.copy_(self._draws[0])
                slot.noise.copy_(self._draws[1])
        for p, g in slot.grads:  # (re-attached: a zero_grad(set_to_none=True) in between cannot detach them)
            p.grad = g
        for p in slot.chained:  # their gradients come from the eager chain below, accumulated: start from nothing
            p.grad = None
        slot.graph.replay()
        if slot.ddp_plan is not None:
            slot.replays += 1
            tr.ddp.replay_launch(slot.ddp_plan, slot.replays)  # the buckets' all-reduces, as the captured backward reaches them
        chain = [(v, slot.C[k].grad) for k, v in self._tensors(C).items() if v.requires_grad]
        if chain:
            torch.autograd.backward([v for v, _ in chain], [g for _, g in chain])
        # a fresh tensor per step (a 4-byte copy enqueued behind the replay): slot.loss is overwritten by the next replay
        return slot.loss.detach().clone()

    def _capture(self, x, C, mask, n_cells):
        from . import _lib as L

        tr, dev = self.tr, x.device
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=dev)
        slot = SimpleNamespace()
        slot.x = x.detach().clone()
        slot.C = dict(C) if isinstance(C, dict) else C
        for k, v in self._tensors(C).items():
            slot.C[k] = v.detach().clone().requires_grad_(v.requires_grad)
        slot.mask = mask.clone()
        slot.n = torch.tensor([int(n_cells)], dtype=torch.int64, device=dev)
        slot.scale = torch.ones((), dtype=torch.float32, device=dev)  # loss scale, read when the graph runs
        scaled = getattr(tr._opt, "loss_scale", None) is not None
        md = SimpleNamespace(cell_idx=None, domain_mask=(slot.mask, slot.n))
        params = [p for p in tr.parameters() if p.requires_grad]
        leaves = params + [v for v in self._tensors(slot.C).values() if v.requires_grad]

        if self.inject:
            slot.t = self._draws[0].detach().clone()
            slot.noise = self._draws[1].detach().clone()

        def body():
            if self.inject:
                loss, _ = tr.model.p_losses(slot.x, slot.t, slot.C, md, None, noise=slot.noise)
            else:
                loss, _ = tr.model(slot.x, slot.C, md, None)
            (loss * slot.scale if scaled else loss).backward()
            return loss

        import warnings

        ddp = getattr(tr, "ddp", None)
        ddp = ddp if (ddp is not None and ddp.active) else None
        s = self.stream
        s.wait_stream(torch.cuda.current_stream(dev))
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            with torch.cuda.stream(s):
                for _ in range(2):  # allocator, per-stream workspaces and the scratch arena of this stream settle
                    for t in leaves:
                        t.grad = None
                    body()
                    if ddp is not None:
                        ddp.finish()  # (eager data-parallel steps: every rank runs the same two; they also fix the bucket order)
                slot.arena = L.scratch_arena(dev)  # held while the graph lives (the arena map is bounded)
        torch.cuda.current_stream(dev).wait_stream(s)
        if any("AccumulateGrad node's stream" in str(w.message) for w in caught):
            # gradient-accumulation nodes of an EARLIER eager backward are still alive (somebody holds that step's loss or
            # another tensor of its autograd graph) and belong to the stream it ran on: captured, they would pull that
            # stream into the capture -- on this ROCm build hipStreamEndCapture then crashes instead of reporting it
            raise RuntimeError("GraphedTrainingStep: an autograd graph of an earlier eager step is still alive (a kept loss "
                               "tensor?); drop it before the first captured step")
        for t in leaves:
            t.grad = None  # the captured backward allocates them, in the graph's own pool
        # stale packed operands: the captured forward then CONTAINS the re-packing of every weight (each replay packs the
        # weights of that moment); and again afterwards, so that no eager forward takes the capture's not-yet-written
        # buffers for packed operands of the current version
        torch.autograd.graph.increment_version([p for p in tr.model.parameters()])
        slot.graph = torch.cuda.CUDAGraph()
        slot.ddp_plan, slot.replays = None, 0
        if ddp is not None:
            slot.gen = torch.zeros(1, dtype=torch.int32, device=dev)  # replay counter, read by the captured bucket marks
            ddp.begin_capture(slot.gen)

        # "thread_local": the backward's launches come from autograd's thread and are captured all the same (capture is a
        # property of the stream); what the mode relaxes is the CHECK of other threads' runtime calls -- ProcessGroupNCCL's
        # watchdog polls the events of the warm-up steps' collectives, and under the default "global" mode that hipEventQuery
        # raises hipErrorStreamCaptureUnsupported inside our capture and aborts the process (2 of 6 runs)
        if ddp is not None:
            torch.cuda.synchronize(dev)  # the warm-up steps' collectives are done before the capture begins
        # The weight gradients' side stream (ops._WgradSide) pays in eager mode (+0.2-0.4 ms hidden) and COSTS in a replayed graph
        # (B = 6: 21.09 ms captured with the fork, 20.81 without; eager 20.29 / 20.46): the captured step keeps everything on one
        # stream unless TDX_GRAPH_WGRAD_STREAM=1 asks for the fork (A/B switch)
        fork = ops.WGRAD_STREAM
        if os.environ.get("TDX_GRAPH_WGRAD_STREAM", "0") != "1":
            ops.WGRAD_STREAM = False
        try:
            with torch.cuda.graph(slot.graph, stream=s, capture_error_mode="thread_local"):
                if ddp is not None:
                    slot.gen.add_(1)
                slot.loss = body().detach()  # (no autograd graph outlives the capture: its nodes belong to this stream)
        finally:
            ops.WGRAD_STREAM = fork
        if ddp is not None:
            slot.ddp_plan = ddp.end_capture()
        torch.autograd.graph.increment_version([p for p in tr.model.parameters()])
        # parameters the captured backward did not reach get their gradient from the eager chain (cell-type table)
        # the packed operands the captured launches read and refresh live in the model's pack plans: held here, so that a
        # plan the model rebuilds later (other parameter storage, other operand format) cannot free them under the graph
        slot.pack_plans = list(tr.model.model._static_lists()[3].values())
        slot.chained = [p for p in params if p.grad is None]
        slot.grads = [(p, p.grad) for p in params if p.grad is not None]  # the tensors every replay writes
        return slot
