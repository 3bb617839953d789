"""ctypes binding of libtdx_hip.so -- the C ABI declared in include/tdx.h.

The library is built in-tree by ``__graft_entry__.build()`` (``make -C csrc``) and lives next
to this file.  There is no CPU implementation behind this module: if the shared object is
missing, or a kernel launch fails, a RuntimeError is raised.
"""

from __future__ import annotations

import ctypes as C
import os
import threading
from pathlib import Path

import torch

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("TDX_LIB", _HERE / "libtdx_hip.so"))  # TDX_LIB: kernel-development builds

F32, BF16 = 0, 1
F16 = 3  # TDX_F16: IEEE half tensors (fp16 MFMA operands, fp32 accumulation)
CONV_AUTO, CONV_DIRECT, CONV_MFMA, CONV_SPLIT = 0, 1, 2, 3
F32_SPLIT = 2  # TDX_F32_SPLIT: pack code of fp32 weights for CONV_SPLIT
WS_CLEAN = 0x100  # TDX_WS_CLEAN (include/tdx.h)

_vp, _i, _i64, _u64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_size_t

# name -> (restype, argtypes); must list every symbol of include/tdx.h
SIGNATURES = {
    "tdx_version": (_i, []),
    "tdx_arch": (C.c_char_p, []),
    "tdx_set_scratch": (_i, [_vp, _sz]),
    "tdx_ncv_to_nvc": (_i, [_vp, _vp, _i, _i, _i64, _i, _i, _vp]),
    "tdx_nvc_to_ncv": (_i, [_vp, _vp, _i, _i, _i64, _i, _i, _vp]),
    "tdx_cast": (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    "tdx_conv3_pack_weight": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "tdx_conv3_pack_weights": (_i, [_vp, _i, _i, _vp]),
    "tdx_transpose_many": (_i, [_vp, _i, _vp]),
    "tdx_conv3_fwd": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "tdx_conv3_uses_ring": (_i, [_i] * 7),
    "tdx_conv3_ring_brick_depth": (_i, [_i] * 7),
    "tdx_conv3_fwd_kernel": (_i, [_i] * 9),
    "tdx_conv3_fwd_gn": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _f, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "tdx_conv3_fwd_partial": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _f, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "tdx_conv3_bwd_data_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "tdx_conv3_bwd_data": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "tdx_conv3_bwd_data_add": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "tdx_conv3_bwd_weight_workspace_bytes": (_sz, [_i, _i, _i]),
    "tdx_conv3_bwd_weight": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "tdx_conv1_fwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i64, _i, _i, _vp]),
    "tdx_conv1_fwd_gn": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i64, _i, _i, _vp]),
    "tdx_conv1_bwd_weight": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i64, _i, _vp]),
    "tdx_conv1_bwd_weight_oc": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i64, _i, _vp]),
    "tdx_encode_fwd": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i64, _i, _i, _vp]),
    "tdx_encode_bwd": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp]),
    "tdx_decode_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _vp]),
    "tdx_decode_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _vp]),
    "tdx_gn_workspace_bytes": (_sz, [_i, _i]),
    "tdx_gn_stats": (_i, [_vp, _vp, _i, _i64, _i, _i, _f, _i, _vp, _vp]),
    "tdx_gn_apply": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _i, _vp]),
    "tdx_gn_apply_decode": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _i, _vp]),
    "tdx_gn_apply_encoded": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _vp]),
    "tdx_gn_bwd": (_i, [_vp] * 12 + [_i, _i64, _i, _i, _i, _i, _vp, _vp]),
    "tdx_resize_fwd": (_i, [_vp, _vp] + [_i] * 9 + [_vp]),
    "tdx_resize_bwd": (_i, [_vp, _vp, _vp] + [_i] * 9 + [_vp]),
    "tdx_attn_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "tdx_attn_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "tdx_attn_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "tdx_cell_mask": (_i, [_vp, _i64, _vp, _i64, _vp]),
    "tdx_q_sample": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _i64, _vp]),
    "tdx_p_sample_step": (_i, [_vp] * 7 + [_i, _vp, _i, _i, _vp, _i, _i, _i64, _vp]),
    "tdx_p_sample_step_rng": (_i, [_vp] * 5 + [_i, _vp, _i, _i, _vp, _i, _i, _i64, _u64, _vp, _vp, _vp]),
    "tdx_masked_loss": (_i, [_vp, _vp, _vp, _i64, _i, _vp, _vp, _i, _i, _i64, _vp, _vp]),
    "tdx_masked_loss_dyn": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i64, _vp, _vp]),
    "tdx_masked_loss_workspace_bytes": (_sz, []),
    "tdx_grid_embed": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i64, _vp]),
    "tdx_grid_select": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i64, _i64, _vp]),
    "tdx_cell_embed_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i64, _vp]),
    "tdx_cell_embed_bwd_workspace_bytes": (_sz, [_i, _i]),
    "tdx_cell_embed_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i64, _vp, _vp]),
    "tdx_tke_energy": (_i, [_vp, _vp, _i, _i64, _vp]),
    "tdx_tke_sphere": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "tdx_randn": (_i, [_vp, _i64, _u64, _u64, _vp, _vp]),
    "tdx_randn_batched": (_i, [_vp, _i, _i64, _u64, _vp, _vp, _vp]),
    "tdx_convg_apply": (_i, [_vp, _vp, _vp, _vp] + [_i] * 16 + [_vp]),
    "tdx_convg_fold_clamp": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "tdx_convg_bwd_weight": (_i, [_vp, _vp, _vp, _vp] + [_i] * 15 + [_vp]),
    "tdx_film_supported": (_i, [_i, _i]),
    "tdx_film_fwd": (_i, [_vp, _i, _i, _vp, _i, _vp]),
    "tdx_film_bwd_workspace_bytes": (_sz, [_i, _i, _vp, _i]),
    "tdx_film_bwd": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "tdx_opt_chunk_elems": (_i64, []),
    "tdx_grad_norm": (_i, [_vp, _vp, _vp, _i, _f, _vp, _vp, _vp]),
    "tdx_radam_step": (_i, [_vp, _vp, _vp, _i, _vp, _i64, _f, _f, _f, _f, _i, _vp]),
    "tdx_signal_host": (_i, [_vp, _vp, C.c_uint32, _vp]),
    "tdx_stage_scaled": (_i, [_vp, _i, _f, _vp]),
    "tdx_grad_norm_scaled": (_i, [_vp, _vp, _vp, _i, _f, _f, _vp, _vp, _vp]),
    "tdx_radam_step_scaled": (_i, [_vp, _vp, _vp, _i, _vp, _i64, _f, _f, _f, _f, _i, _vp]),
}

FILM_MAX_LAYERS = 32  # TDX_FILM_MAX_LAYERS


class FilmLayer(C.Structure):  # TdxFilmLayer
    _fields_ = [("weight", _vp), ("bias", _vp), ("out", _vp), ("channels", _i)]


class PackJob(C.Structure):  # TdxPackJob
    _fields_ = [("w", _vp), ("wf", _vp), ("wb", _vp), ("Cin", _i), ("Cout", _i)]


class TransposeJob(C.Structure):  # TdxTransposeJob
    _fields_ = [("src", _vp), ("dst", _vp), ("rows", _i), ("cols", _i)]


class StageItem(C.Structure):  # TdxStageItem
    _fields_ = [("src", _vp), ("dst", _vp), ("n", _i64)]


class FilmGrad(C.Structure):  # TdxFilmGrad
    _fields_ = [("weight", _vp), ("grad_out", _vp), ("grad_weight", _vp), ("grad_bias", _vp), ("channels", _i)]


_lib = None


def load() -> C.CDLL:
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP kernels have not been built. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C generative-turbulence_amd/csrc`)."
            )
        lib = C.CDLL(str(LIB_PATH))
        lax = "TDX_LIB" in os.environ and os.environ.get("TDX_LIB_LAX") == "1"  # A/B runs against older builds
        for name, (res, args) in SIGNATURES.items():
            if lax and not hasattr(lib, name):
                continue
            fn = getattr(lib, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


_SCRATCH: "OrderedDict" = None  # (device index, stream handle) -> arena tensor, least recently bound first
_ACTIVE = None  # key of the arena the library currently points at
SCRATCH_BYTES = int(os.environ.get("TDX_SCRATCH_MB", "96")) << 20
SCRATCH_SMALL_BYTES = 4096  # streams that only need the zero block at the arena's head (the weight-gradient side stream)
SCRATCH_MAX_ARENAS = int(os.environ.get("TDX_SCRATCH_MAX_ARENAS", "6"))
# entry points that may use the arena (K-split slabs of the small-grid kernels, the zero block of the DMA kernels)
ARENA_USERS = {"tdx_conv3_fwd", "tdx_conv3_fwd_gn", "tdx_conv3_fwd_partial", "tdx_conv3_bwd_data", "tdx_conv3_bwd_data_add",
               "tdx_conv3_bwd_weight", "tdx_attn_fwd"}  # tdx_attn_fwd: partial (O, m, l) of the stream-K schedule
# TDX_DETERMINISTIC=1: these merge their parameter gradients through per-split slabs in the launching stream's arena as well
DET_ARENA_USERS = {"tdx_conv1_bwd_weight", "tdx_conv1_bwd_weight_oc", "tdx_encode_bwd", "tdx_decode_bwd", "tdx_gn_stats"}


def deterministic() -> bool:
    """TDX_DETERMINISTIC=1 in the environment (read per call, like the library does): run-to-run reproducible gradients."""
    return os.environ.get("TDX_DETERMINISTIC", "0") not in ("", "0")


# bind + launch of an arena user is one critical section: the library keeps ONE arena pointer per process and ctypes
# releases the GIL during a foreign call, so a second launching thread could otherwise re-bind between the two
_LAUNCH_LOCK = threading.RLock()
_SMALL_STREAMS = set()  # (device index, stream handle) of streams declared zero-block-only


def declare_zero_block_only(stream) -> None:
    """The given torch stream will only launch arena users that read the 16-byte zero block (tdx_conv3_bwd_weight):
    it gets a 4-KiB arena instead of TDX_SCRATCH_MB.  (torch draws stream handles from a pool of 32 per device and priority: a
    stream a caller creates much later can be the same handle -- its convs then run without the small-grid kernels, correct but
    on the brick kernels.  A high-priority side stream would avoid the collision but slows every kernel beside a live RCCL
    communicator -- ops._WgradSide, TDX_WGRAD_STREAM_PRIORITY.)"""
    _SMALL_STREAMS.add((stream.device.index, stream.cuda_stream))


def ensure_scratch(device=None) -> None:
    """Hand the library its scratch arena for (`device`, the CURRENT stream) -- tdx_set_scratch: K-split slabs of the
    small-grid conv kernels on the deep U-Net levels, zero block of the LDS-DMA kernels.  The library keeps ONE arena
    pointer per process and its kernels get it as a launch argument, so the binding is per launch: every stream that
    launches convs owns its own arena (a hipGraph captured on a side stream keeps replaying on that stream's arena while
    eager work on another stream uses another), and `call` re-binds whenever the launching stream changes.  At most
    TDX_SCRATCH_MAX_ARENAS arenas are kept (least recently bound dropped first; whoever captured a graph on an arena
    holds its tensor -- `scratch_arena()` -- so a dropped arena stays allocated while its graph lives; torch's pooled
    stream handles wrap around after 32 streams, which is why long-lived users keep ONE stream instead of making new ones).
    TDX_SCRATCH_MB=0: no arena (those layers then run on the brick kernels)."""
    global _SCRATCH, _ACTIVE
    if _SCRATCH is None:
        from collections import OrderedDict

        _SCRATCH = OrderedDict()
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    key = (idx, stream(idx))
    if _ACTIVE == key:
        return
    if SCRATCH_BYTES <= 0:
        load().tdx_set_scratch(None, 0)
        _ACTIVE = key
        return
    buf = _SCRATCH.get(key)
    if buf is None:
        nbytes = SCRATCH_SMALL_BYTES if key in _SMALL_STREAMS else SCRATCH_BYTES
        with torch.cuda.device(idx):
            buf = _SCRATCH[key] = torch.zeros(nbytes, dtype=torch.uint8, device=torch.device("cuda", idx))
        while len(_SCRATCH) > SCRATCH_MAX_ARENAS:
            _SCRATCH.popitem(last=False)
    else:
        _SCRATCH.move_to_end(key)
    rc = load().tdx_set_scratch(buf.data_ptr(), buf.numel())
    if rc != 0:
        raise RuntimeError(f"tdx_set_scratch failed: {rc}")
    _ACTIVE = key


def scratch_arena(device=None):
    """The arena tensor bound to (`device`, current stream), or None.  A graph captured on this stream must hold it."""
    ensure_scratch(device)
    return _SCRATCH.get(_ACTIVE) if _SCRATCH is not None else None


H16_DTYPES = (torch.bfloat16, torch.float16)  # 16-bit storage formats that share the matrix-core kernels


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float16:
        return F16
    raise TypeError(f"tdx kernels support float32, bfloat16 and float16 activations, got {dt}")


_CONV_IMPLS = {"auto": CONV_AUTO, "direct": CONV_DIRECT, "mfma": CONV_MFMA, "split": CONV_SPLIT}
_conv_impl_override: str | None = None


def set_conv_impl(name: str | None) -> None:
    """Process-wide default of the 3x3x3 conv implementation ("auto", "direct", "mfma", "split"; None: "auto") for code
    that runs outside any model's conv_impl_scope and without TDX_CONV_IMPL.  "split" = split-precision MFMA convs on
    fp32 tensors."""
    global _conv_impl_override
    if name is not None and name not in _CONV_IMPLS:
        raise ValueError(f"unknown conv implementation {name!r}")
    _conv_impl_override = name


_conv_impl_scopes: list = []  # innermost model whose forward is running (DenoisingModel.conv_impl)


class conv_impl_scope:
    """`with conv_impl_scope(name):` -- the conv implementation of ONE model while its forward runs (name None: no-op).
    The autograd nodes created inside remember it for their backward, so two models with different arithmetic (an
    f32s trainer next to an f32 one) no longer share a process-wide switch."""

    def __init__(self, name: str | None):
        if name is not None and name not in _CONV_IMPLS:
            raise ValueError(f"unknown conv implementation {name!r}")
        self.name = name

    def __enter__(self):
        if self.name is not None:
            _conv_impl_scopes.append(self.name)

    def __exit__(self, *exc):
        if self.name is not None:
            _conv_impl_scopes.pop()


def conv_impl() -> int:
    """TDX_CONV_IMPL (the explicit test / benchmark knob) > the running model's own choice > set_conv_impl > "auto"."""
    env = os.environ.get("TDX_CONV_IMPL")
    if env:
        return _CONV_IMPLS[env]
    if _conv_impl_scopes:
        return _CONV_IMPLS[_conv_impl_scopes[-1]]
    return _CONV_IMPLS[_conv_impl_override or "auto"]


def pack_code(dt: torch.dtype) -> int:
    """dtype code for tdx_conv3_pack_weight: fp32 weights are packed as bf16 hi + lo images under
    TDX_CONV_IMPL=split (split-precision MFMA convs for fp32 tensors)."""
    if dt == torch.float32 and conv_impl() == CONV_SPLIT:
        return F32_SPLIT
    return dtype_code(dt)


def ptr(t: torch.Tensor | None):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("tdx kernels need device tensors (no CPU path exists in the product)")
    if not t.is_contiguous():
        raise RuntimeError("tdx kernels need contiguous tensors")
    return t.data_ptr()


# the launching stream's handle, ~700 times per training step: torch.cuda.current_stream() builds a Python Stream object
# per call (2.65 us measured), the raw query returns the same handle in a tenth of that
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream(device_index=None):
    if _raw_stream is None:
        return torch.cuda.current_stream(device_index).cuda_stream
    return _raw_stream(torch.cuda.current_device() if device_index is None else device_index)


_ERR = {-1: "TDX_EINVAL (bad argument)", -2: "TDX_ESHAPE (unsupported shape)", -3: "TDX_EDTYPE (unsupported dtype)"}


class KernelTimer:
    """HIP-event timing of selected entry points on the stream they are launched on
    (bench.py's roofline leg).  `work` is an optional per-call amount (flops or bytes)."""

    def __init__(self, names):
        self.names = set(names)
        self.records = []  # (name, start_event, end_event, work, meta)
        self.pending_work = 0.0

    def summary(self, where=None):
        """{name: launches / ms / work / bytes}; `where(meta)` selects records (meta = what the caller attached:
        e.g. the kernel family and the algorithmic bytes of a conv call)."""
        torch.cuda.synchronize()
        out = {}
        for name, s, e, work, meta in self.records:
            if where is not None and not where(meta or {}):
                continue
            d = out.setdefault(name, {"launches": 0, "ms": 0.0, "work": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["ms"] += s.elapsed_time(e)
            d["work"] += work
            d["bytes"] += (meta or {}).get("bytes", 0.0)
        return out


TIMER: KernelTimer | None = None


def call(name: str, *args, work: float = 0.0, meta=None):
    """meta: None, a dict, or a zero-argument callable returning one (evaluated only while a timer is attached)."""
    if name in ARENA_USERS or (name in DET_ARENA_USERS and deterministic()):
        with _LAUNCH_LOCK:
            ensure_scratch()  # the arena of the launching stream
            rc = _launch(name, args, work, meta)
    else:
        rc = _launch(name, args, work, meta)
    if rc != 0:
        raise RuntimeError(f"{name} failed: {_ERR.get(rc, f'hipError {rc}')}")


def _launch(name, args, work, meta):
    t = TIMER
    if t is not None and name in t.names:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = getattr(load(), name)(*args)
        e.record()
        t.records.append((name, s, e, work, meta() if callable(meta) else meta))
        return rc
    return getattr(load(), name)(*args)


def query(name: str, *args) -> int:
    return int(getattr(load(), name)(*args))


def kernel_sources_fingerprint(pattern: str = "tdx_conv*") -> str | None:
    """sha256 (16 hex digits) over the conv kernels' sources next to the library (csrc/tdx_conv*.hip|h, in name order):
    what a committed profile describes.  tools/summarize_profiles.py stores it in profiles/*_traffic.json and bench.py
    quotes `roofline.traffic` only from a file whose fingerprint equals the tree's."""
    import hashlib

    src = _HERE.parent / "csrc"
    files = sorted(f for f in src.glob(pattern) if f.suffix in (".hip", ".h"))
    if not files:
        return None
    h = hashlib.sha256()
    for f in files:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


KERNEL_DIRECT, KERNEL_BRICK, KERNEL_SMALL, KERNEL_RING = 0, 1, 2, 3  # TDX_KERNEL_* (tdx_conv3_fwd_kernel)


def conv3_fwd_meta(C1, C2, Cout, B, X, Y, Z, dt, real=None):
    """Bookkeeping of one forward conv call for the timers: the kernel family that serves it and its algorithmic HBM
    bytes (input + output + weights once; `real` = input channels that carry data)."""
    es = 2 if dt in H16_DTYPES else 4
    cin = real or (C1 + C2)
    return {"kind": query("tdx_conv3_fwd_kernel", C1, C2, Cout, B, X, Y, Z, dtype_code(dt), conv_impl()),
            "bytes": float((cin + Cout) * B * X * Y * Z * es + 27 * cin * Cout * es)}
