"""Differentiable operators of the hot path, executed by the HIP kernels in libtdx_hip.so.

Activations are NDHWC tensors ``(B, X, Y, Z, C)`` in float32, bfloat16 or float16; parameters stay in
the reference's shapes and float32 (so state_dicts round-trip).  Each
``torch.autograd.Function`` here replaces one stock PyTorch call of the reference:

    conv3            nn.Conv3d(k=3, padding_mode="replicate")            ddpm.py:164
    conv1            nn.Conv3d(k=1) incl. the torch.cat that feeds it     ddpm.py:188,292,293,433,436,459
    gn_film_silu     GroupNorm -> addcmul(shift, scale+1, x) -> SiLU (+x)  ddpm.py:165-176,197
    resize           F.interpolate(trilinear, align_corners=True)         ddpm.py:359-369
    attention        F.scaled_dot_product_attention                       attention.py:9-15
    q_sample / p_sample_step / masked_loss                                ddpm.py:745-852

There is no CPU implementation: tensors must live on the GPU.
"""

from __future__ import annotations

import weakref

import os

import torch
from torch.autograd.function import once_differentiable

from . import _lib as L


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


_CLEAN = {}
WS_CLEAN = 0 if os.environ.get("TDX_WS_CLEAN", "1") == "0" else L.WS_CLEAN


def _clean_ws(nbytes: int, device, tag=None) -> torch.Tensor:
    """Persistent all-zero workspace for the calls that take TDX_WS_CLEAN (include/tdx.h): they find
    it zero and leave it zero, so a training step launches no memsets for accumulator buffers."""
    if not WS_CLEAN:
        return _ws(nbytes, device)
    # tag: buffers are shared only by calls with the same internal layout; and only by calls issued from the same
    # stream: two streams (two captured graphs, two threads) running the same layer concurrently must not meet in one
    # accumulator buffer
    key = (device, L.stream(device.index), int(nbytes), tag)
    buf = _CLEAN.get(key)
    if buf is None:
        buf = _CLEAN[key] = torch.zeros(max(int(nbytes), 16), dtype=torch.uint8, device=device)
    return buf


def _grid(x: torch.Tensor):
    assert x.dim() == 5, "expected an NDHWC tensor (B, X, Y, Z, C)"
    return x.shape[0], x.shape[1], x.shape[2], x.shape[3], x.shape[4]


# --------------------------------------------------------------------------- layout


class _ToNVC(torch.autograd.Function):
    """(B, C, X, Y, Z) float32 -> (B, X, Y, Z, C) compute dtype."""

    @staticmethod
    def forward(ctx, x, dtype):
        B, Cc, X, Y, Z = x.shape
        x = x.contiguous()
        y = torch.empty((B, X, Y, Z, Cc), dtype=dtype, device=x.device)
        L.call("tdx_ncv_to_nvc", L.ptr(x), L.ptr(y), B, Cc, X * Y * Z, L.dtype_code(x.dtype), L.dtype_code(dtype), L.stream())
        ctx.in_dtype = x.dtype
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        B, X, Y, Z, Cc = gy.shape
        gy = gy.contiguous()
        gx = torch.empty((B, Cc, X, Y, Z), dtype=ctx.in_dtype, device=gy.device)
        L.call("tdx_nvc_to_ncv", L.ptr(gy), L.ptr(gx), B, Cc, X * Y * Z, L.dtype_code(gy.dtype), L.dtype_code(ctx.in_dtype), L.stream())
        return gx, None


class _ToNCV(torch.autograd.Function):
    """(B, X, Y, Z, C) -> (B, C, X, Y, Z) in `dtype`."""

    @staticmethod
    def forward(ctx, x, dtype):
        B, X, Y, Z, Cc = x.shape
        x = x.contiguous()
        y = torch.empty((B, Cc, X, Y, Z), dtype=dtype, device=x.device)
        L.call("tdx_nvc_to_ncv", L.ptr(x), L.ptr(y), B, Cc, X * Y * Z, L.dtype_code(x.dtype), L.dtype_code(dtype), L.stream())
        ctx.in_dtype = x.dtype
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        B, Cc, X, Y, Z = gy.shape
        gy = gy.contiguous()
        gx = torch.empty((B, X, Y, Z, Cc), dtype=ctx.in_dtype, device=gy.device)
        L.call("tdx_ncv_to_nvc", L.ptr(gy), L.ptr(gx), B, Cc, X * Y * Z, L.dtype_code(gy.dtype), L.dtype_code(ctx.in_dtype), L.stream())
        return gx, None


def to_nvc(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    return _ToNVC.apply(x, dtype)


def to_ncv(x: torch.Tensor, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    return _ToNCV.apply(x, dtype)


# --------------------------------------------------------------------------- model boundary


class _Encode(torch.autograd.Function):
    """(B,4,X,Y,Z) f32 [+ (4,X,Y,Z) conditioning] -> (B,X,Y,Z,D or 2D): both 1x1 encoders, the
    layout change and the channel concat in one pass."""

    @staticmethod
    def forward(ctx, x, c_local, wx, bx, wc, bc, dtype, lazy=None):
        B, Fx, X, Y, Z = x.shape
        V, D = X * Y * Z, wx.shape[0]
        x = x.contiguous().float()
        has_c = c_local is not None
        c = c_local.contiguous().float() if has_c else None
        wx2, bx2 = wx.detach().reshape(D, Fx).contiguous(), bx.detach().contiguous()
        wc2 = wc.detach().reshape(D, -1).contiguous() if has_c else None
        bc2 = bc.detach().contiguous() if has_c else None
        if lazy is not None:
            # encode_deferred: nothing is computed -- the one consumer (the first ResnetBlock's identity skip) evaluates
            # the encoders itself from these operands (tdx_gn_apply_encoded).  The returned tensor is a shape-only stand-in
            # (one element, expanded) that ties the consumer's gradient back to this node.
            lazy.extend((x, Fx, wx2, bx2, c, c.shape[0] if has_c else 0, wc2, bc2, D))
            y = torch.empty(1, dtype=dtype, device=x.device).expand(B, X, Y, Z, 2 * D if has_c else D)
        else:
            y = torch.empty((B, X, Y, Z, 2 * D if has_c else D), dtype=dtype, device=x.device)
            L.call("tdx_encode_fwd", L.ptr(x), Fx, L.ptr(wx2), L.ptr(bx2), L.ptr(c), c.shape[0] if has_c else 0, L.ptr(wc2),
                   L.ptr(bc2), L.ptr(y), B, V, D, L.dtype_code(dtype), L.stream())
        ctx.save_for_backward(x, c, wc2)
        ctx.shapes = (tuple(wx.shape), tuple(wc.shape) if has_c else None, tuple(c_local.shape) if has_c else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, c, wc2 = ctx.saved_tensors
        wx_shape, wc_shape, c_shape = ctx.shapes
        B, Fx = x.shape[:2]
        V = x[0, 0].numel()
        D = wx_shape[0]
        gy = gy.contiguous()
        dev = gy.device
        dwx = torch.empty(wx_shape, dtype=torch.float32, device=dev)
        dbx = torch.empty(D, dtype=torch.float32, device=dev)
        has_c = c is not None
        dwc = torch.empty(wc_shape, dtype=torch.float32, device=dev) if has_c else None
        dbc = torch.empty(D, dtype=torch.float32, device=dev) if has_c else None
        dc = torch.empty(c_shape, dtype=torch.float32, device=dev) if (has_c and ctx.needs_input_grad[1]) else None
        L.call("tdx_encode_bwd", L.ptr(gy), L.ptr(x), Fx, L.ptr(c), c.shape[0] if has_c else 0, L.ptr(wc2), L.ptr(dwx),
               L.ptr(dbx), L.ptr(dwc), L.ptr(dbc), L.ptr(dc), B, V, D, L.dtype_code(gy.dtype), L.stream())
        return None, dc, dwx, dbx, dwc, dbc, None, None


class DeferredEncoding:
    """What `encode_deferred` returns: `.standin` -- a shape-only tensor (never read) that carries the autograd edge to
    the encoders -- and `.operands`, from which `resnet_block(..., skip_encoded=...)` evaluates the encoder output inside
    its tail kernel."""

    def __init__(self, standin, operands):
        self.standin, self.operands = standin, tuple(operands)


def encode_deferred(x, c_local, wx, bx, wc, bc, dtype) -> DeferredEncoding:
    """`encode` without the (B, X, Y, Z, 2D) tensor: for the case where its only reader is the identity skip of a fused
    ResnetBlock whose conv runs on another input (DenoisingModel.compose_first_conv)."""
    ops_ = []
    standin = _Encode.apply(x, c_local, wx, bx, wc, bc, dtype, ops_)
    return DeferredEncoding(standin, ops_)


def encode_supported(x, c_local, wx):
    D = wx.shape[0]
    n = (2 * D if c_local is not None else D) // 8
    return (x.shape[1] == 4 and (c_local is None or c_local.shape[0] == 4) and D % 8 == 0 and n & (n - 1) == 0
            and n <= 64 and not x.requires_grad)


def encode(x, c_local, wx, bx, wc, bc, dtype):
    return _Encode.apply(x, c_local, wx, bx, wc, bc, dtype)


class _Decode(torch.autograd.Function):
    """(B,X,Y,Z,D) -> (B,4,X,Y,Z) f32: decode.1's 1x1 conv fused with the layout change."""

    @staticmethod
    def forward(ctx, h, w, bias):
        B, X, Y, Z, D = h.shape
        F = w.shape[0]
        h = h.contiguous()
        w2 = w.detach().reshape(F, D).contiguous()
        y = torch.empty((B, F, X, Y, Z), dtype=torch.float32, device=h.device)
        L.call("tdx_decode_fwd", L.ptr(h), L.ptr(w2), L.ptr(bias.detach().contiguous()), L.ptr(y), B, X * Y * Z, D, F,
               L.dtype_code(h.dtype), L.stream())
        ctx.save_for_backward(h, w2)
        ctx.wshape = tuple(w.shape)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        h, w2 = ctx.saved_tensors
        B, X, Y, Z, D = h.shape
        F = w2.shape[0]
        gy = gy.contiguous().float()
        dh = torch.empty_like(h)
        dw = torch.empty(ctx.wshape, dtype=torch.float32, device=h.device)
        db = torch.empty(F, dtype=torch.float32, device=h.device)
        L.call("tdx_decode_bwd", L.ptr(gy), L.ptr(h), L.ptr(w2), L.ptr(dh), L.ptr(dw), L.ptr(db), B, X * Y * Z, D, F,
               L.dtype_code(h.dtype), L.stream())
        return dh, dw, db


def decode_supported(h, w):
    n = h.shape[-1] // 8
    return w.shape[0] == 4 and h.shape[-1] % 8 == 0 and n & (n - 1) == 0 and n <= 64


def decode(h, w, bias):
    return _Decode.apply(h, w, bias)


def decode_fused_supported(C: int, w) -> bool:
    """resnet_block(..., decode_wb=(w, bias)) on a block with C output channels: inference only."""
    n = C // 8
    return (not torch.is_grad_enabled() and w.shape[0] == 4 and w.shape[1] == C and C % 8 == 0 and n & (n - 1) == 0
            and n <= 64)


# --------------------------------------------------------------------------- conv 3x3x3

# packed operands are cached per (parameter, version, dtype): repacked once per optimiser
# step in training, once per model in sampling
_pack_cache: dict = {}


def _packed_conv3(weight: torch.Tensor, dtype: torch.dtype):
    code = L.pack_code(dtype)
    key = (id(weight), dtype, code)
    hit = _pack_cache.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == weight._version and hit[2] == weight.data_ptr():
        return hit[3], hit[4]
    Cout, Cin = weight.shape[0], weight.shape[1]
    w = weight.detach().contiguous()
    wf = torch.empty(27 * Cin * Cout, dtype=dtype, device=w.device)
    wb = torch.empty(27 * Cin * Cout, dtype=dtype, device=w.device)
    L.call("tdx_conv3_pack_weight", L.ptr(w), L.ptr(wf), L.ptr(wb), Cin, Cout, code, L.stream())
    if len(_pack_cache) > 4096:
        _pack_cache.clear()
    _pack_cache[key] = (weakref.ref(weight), weight._version, weight.data_ptr(), wf, wb)
    return wf, wb


def _packed_conv3_cin_slice(weight: torch.Tensor, lo: int, hi: int, dtype: torch.dtype):
    """Forward operand of conv3 restricted to input channels [lo, hi) of `weight`, cached like
    _packed_conv3 (keyed on the full parameter's version)."""
    code = L.pack_code(dtype)
    key = (id(weight), dtype, lo, hi, code)
    hit = _pack_cache.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == weight._version and hit[2] == weight.data_ptr():
        return hit[3]
    Cout, Cin = weight.shape[0], hi - lo
    w = weight.detach()[:, lo:hi].contiguous()
    wf = torch.empty(27 * Cin * Cout, dtype=dtype, device=w.device)
    L.call("tdx_conv3_pack_weight", L.ptr(w), L.ptr(wf), None, Cin, Cout, code, L.stream())
    _pack_cache[key] = (weakref.ref(weight), weight._version, weight.data_ptr(), wf, None)
    return wf


def _cache_fresh(key, weight) -> bool:
    hit = _pack_cache.get(key)
    return hit is not None and hit[0]() is weight and hit[1] == weight._version and hit[2] == weight.data_ptr()


class PackPlan:
    """Packed operands of a FIXED list of 3x3x3 weights and transposed copies of a fixed list of 1x1 weights, kept in
    buffers the plan owns and refreshed IN PLACE by one launch each when a weight changed (every optimiser step in
    training; never while sampling).  The job tables are built once: a refresh costs the host a version check per weight
    and two foreign calls -- the per-step loop that allocated 2 x 22 operand tensors and rebuilt both tables was 0.3 ms
    of device idle time at the head of every training step (tools/timeline_gaps.py).  `_packed_conv3` / `_conv1_wt`
    find the operands through the shared cache, as before."""

    def __init__(self, conv3_weights, conv1_weights, dtype: torch.dtype):
        self.dtype, self.code = dtype, L.pack_code(dtype)
        self.w3, self.w1 = list(conv3_weights), list(conv1_weights)
        for w in self.w3 + self.w1:
            assert w.is_cuda and w.dtype == torch.float32 and w.is_contiguous(), "fp32 contiguous device parameters"
        self.ops3 = [(torch.empty(27 * w.shape[1] * w.shape[0], dtype=dtype, device=w.device),
                      torch.empty(27 * w.shape[1] * w.shape[0], dtype=dtype, device=w.device)) for w in self.w3]
        self.ops1 = [torch.empty((w[0].numel(), w.shape[0]), dtype=torch.float32, device=w.device) for w in self.w1]
        self.tab3 = (L.PackJob * max(len(self.w3), 1))(*[L.PackJob(w.data_ptr(), wf.data_ptr(), wb.data_ptr(), w.shape[1], w.shape[0])
                                                         for w, (wf, wb) in zip(self.w3, self.ops3)])
        self.tab1 = (L.TransposeJob * max(len(self.w1), 1))(*[L.TransposeJob(w.data_ptr(), wt.data_ptr(), w.shape[0], w[0].numel())
                                                              for w, wt in zip(self.w1, self.ops1)])
        self.ptrs = [w.data_ptr() for w in self.w3 + self.w1]
        self.seen3, self.seen1 = [None] * len(self.w3), [None] * len(self.w1)

    def valid_for(self, conv3_weights, conv1_weights, dtype) -> bool:
        """Same parameter tensors at the same addresses, same operand format?"""
        ws = list(conv3_weights) + list(conv1_weights)
        return (dtype == self.dtype and L.pack_code(dtype) == self.code and len(ws) == len(self.ptrs)
                and all(a is b for a, b in zip(ws, self.w3 + self.w1)) and all(w.data_ptr() == p for w, p in zip(ws, self.ptrs)))

    def refresh(self) -> None:
        if self.w3 and any(w._version != v for w, v in zip(self.w3, self.seen3)):
            L.call("tdx_conv3_pack_weights", self.tab3, len(self.w3), self.code, L.stream())
            for i, (w, (wf, wb)) in enumerate(zip(self.w3, self.ops3)):
                self.seen3[i] = w._version
                _pack_cache[(id(w), self.dtype, self.code)] = (weakref.ref(w), w._version, w.data_ptr(), wf, wb)
        if self.w1 and any(w._version != v for w, v in zip(self.w1, self.seen1)):
            L.call("tdx_transpose_many", self.tab1, len(self.w1), L.stream())
            for i, (w, wt) in enumerate(zip(self.w1, self.ops1)):
                self.seen1[i] = w._version
                _pack_cache[(id(w), "wt")] = (weakref.ref(w), w._version, w.data_ptr(), wt, None)


def prefetch_weights(conv3_weights, conv1_weights, dtype: torch.dtype, plan: "PackPlan | None" = None) -> "PackPlan | None":
    """Refresh the packed operands of all given 3x3x3 weights (as _packed_conv3 would, one by one) and the transposed
    copies of all given 1x1 weights (as _conv1_wt would) in ONE launch each -- after an optimiser step every weight is
    stale, and a model forward would otherwise start with ~30 tiny launches.  Returns the PackPlan to pass back in next
    time (it is rebuilt when the parameter tensors or the operand format changed); the later _packed_conv3 / _conv1_wt
    calls hit the cache."""
    if plan is None or not plan.valid_for(conv3_weights, conv1_weights, dtype):
        plan = PackPlan(conv3_weights, conv1_weights, dtype)
    plan.refresh()
    if len(_pack_cache) > 4096:
        _pack_cache.clear()
    return plan


def conv3_partial_supported(x, weight, n_lead: int) -> bool:
    """tdx_conv3_fwd_partial: 16-bit MFMA path on the leading n_lead input channels."""
    return (x.dtype in L.H16_DTYPES and n_lead % 16 == 0 and weight.shape[0] % 32 == 0 and x.shape[-1] % 8 == 0
            and L.conv_impl() != L.CONV_DIRECT)


def conv3_shared_tail(e, weight, n_lead: int):
    """conv3 of the batch-shared channels [n_lead, Cin) of the first U-Net conv: e is (1, X, Y, Z, Cin - n_lead);
    no bias (the per-sample conv adds it).  Result (1, X, Y, Z, Cout), the `init` of conv3_partial."""
    Cin = weight.shape[1]
    B, X, Y, Z, Ce = _grid(e)
    assert B == 1 and Ce == Cin - n_lead
    wf = _packed_conv3_cin_slice(weight, n_lead, Cin, e.dtype)
    y = torch.empty((1, X, Y, Z, weight.shape[0]), dtype=e.dtype, device=e.device)
    L.call("tdx_conv3_fwd", L.ptr(e.contiguous()), Ce, None, 0, L.ptr(wf), None, L.ptr(y), 1, X, Y, Z, weight.shape[0],
           L.dtype_code(e.dtype), L.conv_impl(), L.stream())
    return y


class _Conv3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2, weight, bias, gn_groups=0, gn_eps=1e-5):
        B, X, Y, Z, C1 = _grid(x1)
        C2 = 0 if x2 is None else x2.shape[-1]
        Cout = weight.shape[0]
        assert weight.shape[1] == C1 + C2 and tuple(weight.shape[2:]) == (3, 3, 3)
        x1 = x1.contiguous()
        x2 = None if x2 is None else x2.contiguous()
        dt = x1.dtype
        wf, wb = _packed_conv3(weight, dt)
        y = torch.empty((B, X, Y, Z, Cout), dtype=dt, device=x1.device)
        flops = 54.0 * (C1 + C2) * Cout * B * X * Y * Z
        ctx.save_for_backward(x1, x2, wb)
        ctx.has_bias = bias is not None
        ctx.wshape = tuple(weight.shape)
        impl = ctx.impl = L.conv_impl()  # the backward runs outside the model's conv_impl_scope: it reuses this
        if gn_groups:
            stats = torch.empty((B, gn_groups, 2), dtype=torch.float32, device=x1.device)
            ws = _clean_ws(L.query("tdx_gn_workspace_bytes", B, Cout), x1.device)
            L.call("tdx_conv3_fwd_gn", L.ptr(x1), C1, L.ptr(x2), C2, L.ptr(wf), L.ptr(bias), L.ptr(y), L.ptr(stats),
                   gn_groups, float(gn_eps), L.ptr(ws), B, X, Y, Z, Cout, L.dtype_code(dt), impl | WS_CLEAN,
                   L.stream(), work=flops, meta=lambda: L.conv3_fwd_meta(C1, C2, Cout, B, X, Y, Z, dt))
            ctx.mark_non_differentiable(stats)
            return y, stats
        L.call("tdx_conv3_fwd", L.ptr(x1), C1, L.ptr(x2), C2, L.ptr(wf), L.ptr(bias), L.ptr(y), B, X, Y, Z, Cout,
               L.dtype_code(dt), impl, L.stream(), work=flops,
               meta=lambda: L.conv3_fwd_meta(C1, C2, Cout, B, X, Y, Z, dt))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, _gstats=None):
        x1, x2, wb = ctx.saved_tensors
        B, X, Y, Z, C1 = _grid(x1)
        C2 = 0 if x2 is None else x2.shape[-1]
        Cout, Cin = ctx.wshape[0], ctx.wshape[1]
        gy = gy.contiguous()
        dt, dev = gy.dtype, gy.device
        code, impl, st = L.dtype_code(dt), ctx.impl, L.stream()
        gx1 = gx2 = gw = gb = None
        if ctx.needs_input_grad[0] or (x2 is not None and ctx.needs_input_grad[1]):
            gx1 = torch.empty_like(x1)
            gx2 = None if x2 is None else torch.empty_like(x2)
            ws = _ws(L.query("tdx_conv3_bwd_data_workspace_bytes", B, X, Y, Z, Cin, code, impl), dev)
            L.call("tdx_conv3_bwd_data", L.ptr(gy), L.ptr(wb), L.ptr(gx1), C1, L.ptr(gx2), C2, 0, B, X, Y, Z, Cout,
                   code, impl, L.ptr(ws), st, work=54.0 * Cin * Cout * B * X * Y * Z)
        if ctx.needs_input_grad[2]:
            gw = torch.empty(ctx.wshape, dtype=torch.float32, device=dev)
            gb = torch.empty(Cout, dtype=torch.float32, device=dev) if ctx.has_bias else None
            ws = _clean_ws(L.query("tdx_conv3_bwd_weight_workspace_bytes", Cin, Cout, impl), dev, ("w3", Cin, Cout))
            L.call("tdx_conv3_bwd_weight", L.ptr(x1), C1, L.ptr(x2), C2, L.ptr(gy), L.ptr(gw), L.ptr(gb), B, X, Y, Z,
                   Cout, code, impl | WS_CLEAN, L.ptr(ws), st, work=54.0 * Cin * Cout * B * X * Y * Z)
        return gx1, gx2, gw, gb, None, None


def conv3(x1, weight, bias=None, x2=None):
    """Replicate-padded 3x3x3 convolution of the channel concatenation [x1 | x2]."""
    return _Conv3.apply(x1, x2, weight, bias)


def conv3_gn_stats(x1, weight, bias, groups, eps=1e-5, x2=None):
    """conv3 plus the GroupNorm(groups) statistics (B, groups, 2) of its output, accumulated in
    the conv epilogue.  Returns (y, stats); feed stats to gn_film_silu(..., stats=stats)."""
    return _Conv3.apply(x1, x2, weight, bias, groups, eps)


# --------------------------------------------------------------------------- conv 1x1x1


def _conv1_wt(weight: torch.Tensor) -> torch.Tensor:
    """[Cin][Cout] f32 copy of a 1x1 conv weight (the layout tdx_conv1_fwd reads), cached until the
    parameter changes (Tensor._version) instead of being re-transposed on every call."""
    key = (id(weight), "wt")
    hit = _pack_cache.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == weight._version and hit[2] == weight.data_ptr():
        return hit[3]
    wt = weight.detach().reshape(weight.shape[0], -1).t().contiguous()
    _pack_cache[key] = (weakref.ref(weight), weight._version, weight.data_ptr(), wt, None)
    return wt


def _conv1_weight_grad(x1, C1, x2, C2, gy, Cout, has_bias, rows, code, st):
    """(dW (Cout, C1 + C2), dbias (Cout) or None) of a 1x1 conv in nn.Conv3d's own layout (tdx_conv1_bwd_weight_oc):
    one zeroed allocation for both, no transposition afterwards."""
    Cin = C1 + C2
    off = (Cout * Cin + 63) // 64 * 64  # the bias gradient starts on a 256-B boundary
    buf = torch.zeros(off + (Cout if has_bias else 0), dtype=torch.float32, device=gy.device)
    gw = buf[: Cout * Cin].view(Cout, Cin)
    gb = buf[off:] if has_bias else None
    L.call("tdx_conv1_bwd_weight_oc", L.ptr(x1), C1, L.ptr(gy), Cout, gw.data_ptr(), Cin, L.ptr(gb), 1, rows, code, st)
    if x2 is not None:
        L.call("tdx_conv1_bwd_weight_oc", L.ptr(x2), C2, L.ptr(gy), Cout, gw.data_ptr() + 4 * C1, Cin, None, 1, rows, code, st)
    return gw, gb


class _Conv1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2, weight, bias, add):
        C1 = x1.shape[-1]
        C2 = 0 if x2 is None else x2.shape[-1]
        Cout = weight.shape[0]
        w2 = weight.detach().reshape(Cout, -1)
        assert w2.shape[1] == C1 + C2
        x1 = x1.contiguous()
        x2 = None if x2 is None else x2.contiguous()
        add = None if add is None else add.contiguous()
        rows = x1.numel() // C1
        wt = _conv1_wt(weight)  # [Cin][Cout]
        y = torch.empty(x1.shape[:-1] + (Cout,), dtype=x1.dtype, device=x1.device)
        L.call("tdx_conv1_fwd", L.ptr(x1), C1, L.ptr(x2), C2, L.ptr(wt), Cout, L.ptr(bias), L.ptr(add), L.ptr(y), rows,
               Cout, L.dtype_code(x1.dtype), L.stream())
        ctx.save_for_backward(x1, x2, w2.contiguous())
        ctx.has_bias = bias is not None
        ctx.has_add = add is not None
        ctx.wshape = tuple(weight.shape)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x1, x2, w2 = ctx.saved_tensors  # w2: [Cout][Cin]
        C1 = x1.shape[-1]
        C2 = 0 if x2 is None else x2.shape[-1]
        Cout, Cin = w2.shape
        gy = gy.contiguous()
        rows = gy.numel() // Cout
        code, st, dev = L.dtype_code(gy.dtype), L.stream(), gy.device
        gx1 = gx2 = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx1 = torch.empty_like(x1)
            L.call("tdx_conv1_fwd", L.ptr(gy), Cout, None, 0, w2.data_ptr(), Cin, None, None, L.ptr(gx1), rows, C1, code, st)
        if x2 is not None and ctx.needs_input_grad[1]:
            gx2 = torch.empty_like(x2)
            L.call("tdx_conv1_fwd", L.ptr(gy), Cout, None, 0, w2.data_ptr() + 4 * C1, Cin, None, None, L.ptr(gx2), rows, C2, code, st)
        if ctx.needs_input_grad[2]:
            gw, gb = _conv1_weight_grad(x1, C1, x2, C2, gy, Cout, ctx.has_bias, rows, code, st)
            gw = gw.view(ctx.wshape)
        return gx1, gx2, gw, gb, (gy if ctx.has_add else None)


def conv1(x1, weight, bias=None, x2=None, add=None):
    """Per-voxel channel GEMM of [x1 | x2] (+ add); weight is (Cout, Cin[,1,1,1])."""
    return _Conv1.apply(x1, x2, weight, bias, add)


# --------------------------------------------------------------------------- GroupNorm + FiLM + SiLU


class _GnFilmSilu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, scale, shift, res, groups, act, eps, stats=None):
        B, X, Y, Z, Cc = _grid(x)
        V = X * Y * Z
        x = x.contiguous()
        res = None if res is None else res.contiguous()
        dev, code, st = x.device, L.dtype_code(x.dtype), L.stream()
        gamma, beta = gamma.detach().contiguous(), beta.detach().contiguous()
        if scale is not None:
            scale = scale.detach().reshape(B, Cc).float().contiguous()
            shift = shift.detach().reshape(B, Cc).float().contiguous()
        if stats is None:
            stats = torch.empty((B, groups, 2), dtype=torch.float32, device=dev)
            ws = _ws(L.query("tdx_gn_workspace_bytes", B, Cc), dev)
            L.call("tdx_gn_stats", L.ptr(x), L.ptr(stats), B, V, Cc, groups, float(eps), code, L.ptr(ws), st)
        y = torch.empty_like(x)
        L.call("tdx_gn_apply", L.ptr(x), L.ptr(stats), L.ptr(gamma), L.ptr(beta), L.ptr(scale), L.ptr(shift), L.ptr(res),
               L.ptr(y), B, V, Cc, groups, int(act), code, st)
        ctx.save_for_backward(x, stats, gamma, beta, scale, shift)
        ctx.cfg = (groups, int(act), res is not None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, stats, gamma, beta, scale, shift = ctx.saved_tensors
        groups, act, has_res = ctx.cfg
        B, X, Y, Z, Cc = _grid(x)
        V = X * Y * Z
        gy = gy.contiguous()
        dev, code, st = x.device, L.dtype_code(x.dtype), L.stream()
        gx = torch.empty_like(x)
        dgamma = torch.empty(Cc, dtype=torch.float32, device=dev)
        dbeta = torch.empty(Cc, dtype=torch.float32, device=dev)
        dscale = dshift = None
        if scale is not None:
            dscale = torch.empty((B, Cc), dtype=torch.float32, device=dev)
            dshift = torch.empty((B, Cc), dtype=torch.float32, device=dev)
        ws = _ws(L.query("tdx_gn_workspace_bytes", B, Cc), dev)
        L.call("tdx_gn_bwd", L.ptr(x), L.ptr(gy), L.ptr(stats), L.ptr(gamma), L.ptr(beta), L.ptr(scale), L.ptr(shift),
               L.ptr(gx), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dscale), L.ptr(dshift), B, V, Cc, groups, act, code,
               L.ptr(ws), st)
        return gx, dgamma, dbeta, dscale, dshift, (gy if has_res else None), None, None, None, None


def gn_film_silu(x, gamma, beta, groups, scale=None, shift=None, res=None, act=True, eps=1e-5, stats=None):
    """y = [silu]( GN(x) * (1 + scale) + shift ) + res;  scale/shift are (B, C) or None.
    `stats` (B, groups, 2) may come from conv3_gn_stats (else a statistics pass is run)."""
    return _GnFilmSilu.apply(x, gamma, beta, scale, shift, res, groups, act, eps, stats)


# --------------------------------------------------------------------------- trilinear resize


class _Resize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size):
        B, Xi, Yi, Zi, Cc = _grid(x)
        Xo, Yo, Zo = (int(s) for s in size)
        x = x.contiguous()
        y = torch.empty((B, Xo, Yo, Zo, Cc), dtype=x.dtype, device=x.device)
        L.call("tdx_resize_fwd", L.ptr(x), L.ptr(y), B, Xi, Yi, Zi, Xo, Yo, Zo, Cc, L.dtype_code(x.dtype), L.stream())
        ctx.geom = (B, Xi, Yi, Zi, Xo, Yo, Zo, Cc)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        B, Xi, Yi, Zi, Xo, Yo, Zo, Cc = ctx.geom
        gy = gy.contiguous()
        gx = torch.empty((B, Xi, Yi, Zi, Cc), dtype=gy.dtype, device=gy.device)
        L.call("tdx_resize_bwd", L.ptr(gy), None, L.ptr(gx), B, Xi, Yi, Zi, Xo, Yo, Zo, Cc, L.dtype_code(gy.dtype), L.stream())
        return gx, None


def resize(x, size):
    """Trilinear resample (align_corners=True) of an NDHWC tensor to grid `size`."""
    return _Resize.apply(x, tuple(size))


class _SkipAndResize(torch.autograd.Function):
    """x -> (x as skip connection, resize(x)): the two uses of a U-Net level's output (reference
    ddpm.py:355-358) as ONE node, so that the backward adds the skip gradient inside the resize adjoint
    instead of autograd running a separate add over the level's activation."""

    @staticmethod
    def forward(ctx, x, size):
        B, Xi, Yi, Zi, Cc = _grid(x)
        Xo, Yo, Zo = (int(s) for s in size)
        x = x.contiguous()
        y = torch.empty((B, Xo, Yo, Zo, Cc), dtype=x.dtype, device=x.device)
        L.call("tdx_resize_fwd", L.ptr(x), L.ptr(y), B, Xi, Yi, Zi, Xo, Yo, Zo, Cc, L.dtype_code(x.dtype), L.stream())
        ctx.geom = (B, Xi, Yi, Zi, Xo, Yo, Zo, Cc)
        ctx.set_materialize_grads(False)  # an unused output arrives as None, not as a zero tensor
        return x.view_as(x), y

    @staticmethod
    @once_differentiable
    def backward(ctx, g_skip, gy):
        B, Xi, Yi, Zi, Xo, Yo, Zo, Cc = ctx.geom
        if gy is None:
            return g_skip, None
        gy = gy.contiguous()
        add = None if g_skip is None else g_skip.contiguous()
        gx = torch.empty((B, Xi, Yi, Zi, Cc), dtype=gy.dtype, device=gy.device)
        L.call("tdx_resize_bwd", L.ptr(gy), L.ptr(add), L.ptr(gx), B, Xi, Yi, Zi, Xo, Yo, Zo, Cc, L.dtype_code(gy.dtype),
               L.stream())
        return gx, None


def skip_and_resize(x, size):
    """(skip, resized) for a U-Net down level; the gradients of both uses are merged in one kernel."""
    return _SkipAndResize.apply(x, tuple(size))


# --------------------------------------------------------------------------- attention


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, heads):
        B, N, C3 = qkv.shape
        D = C3 // (3 * heads)
        qkv = qkv.contiguous()
        out = torch.empty((B, N, heads * D), dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty((B, heads, N), dtype=torch.float32, device=qkv.device)
        L.call("tdx_attn_fwd", L.ptr(qkv), L.ptr(out), L.ptr(lse), B, N, heads, D, L.dtype_code(qkv.dtype), L.stream())
        ctx.save_for_backward(qkv, out, lse)
        ctx.heads = heads
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        qkv, out, lse = ctx.saved_tensors
        B, N, C3 = qkv.shape
        H = ctx.heads
        D = C3 // (3 * H)
        gout = gout.contiguous()
        dqkv = torch.empty_like(qkv)
        ws = _ws(L.query("tdx_attn_bwd_workspace_bytes", B, N, H, D), qkv.device)
        L.call("tdx_attn_bwd", L.ptr(qkv), L.ptr(out), L.ptr(lse), L.ptr(gout), L.ptr(dqkv), B, N, H, D,
               L.dtype_code(qkv.dtype), L.ptr(ws), L.stream())
        return dqkv, None


def attention(qkv, heads):
    """softmax(q k^T / sqrt(d)) v on a token-major (B, N, 3*heads*d) q|k|v tensor."""
    return _Attention.apply(qkv, heads)


# --------------------------------------------------------------------------- DDPM arithmetic


def cell_mask(cell_idx: torch.Tensor, V: int) -> torch.Tensor:
    """Dense uint8 in-domain mask from the reference's flat cell index list (utils.py:22-28)."""
    cell_idx = cell_idx.to(torch.int64).contiguous()
    mask = torch.empty(V, dtype=torch.uint8, device=cell_idx.device)
    L.call("tdx_cell_mask", L.ptr(cell_idx), cell_idx.numel(), L.ptr(mask), V, L.stream())
    return mask


def q_sample(x0, noise, sqrt_ac, sqrt_1mac, t, mask=None, keep_bcs=False):
    """sqrt(abar_t) x0 + sqrt(1 - abar_t) noise on (B, F, X, Y, Z) float32 tensors."""
    B, F = x0.shape[:2]
    V = x0[0, 0].numel()
    x0, noise = x0.contiguous(), noise.contiguous()
    t = t.to(torch.int64).contiguous()
    out = torch.empty_like(x0)
    L.call("tdx_q_sample", L.ptr(x0), L.ptr(noise), L.ptr(sqrt_ac), L.ptr(sqrt_1mac), L.ptr(t), 1 if t.numel() > 1 else 0,
           L.ptr(mask), int(keep_bcs), L.ptr(out), B, F, V, L.stream())
    return out


def p_sample_step(x_t, eps, z, z2, x_bcs, mask, sched, T, t_dev, noise_bcs, clip, out=None):
    """One fused reverse-diffusion update; t_dev is a device int64 scalar tensor."""
    B, F = x_t.shape[:2]
    V = x_t[0, 0].numel()
    if out is None:
        out = torch.empty_like(x_t)
    L.call("tdx_p_sample_step", L.ptr(x_t), L.ptr(eps), L.ptr(z), L.ptr(z2), L.ptr(x_bcs), L.ptr(mask), L.ptr(sched), T,
           L.ptr(t_dev), int(noise_bcs), int(clip), L.ptr(out), B, F, V, L.stream())
    return out


def p_sample_step_rng_supported(x_t) -> bool:
    return x_t.dtype == torch.float32 and x_t[0, 0].numel() % 4 == 0 and x_t.is_contiguous()


def p_sample_step_rng(x_t, eps, x_bcs, mask, sched, T, t_dev, noise_bcs, clip, seed, stream_ids, offset_dev, out=None):
    """The reverse step with z (and z2) drawn in the kernel; advances offset_dev and decrements t_dev on the device."""
    B, F = x_t.shape[:2]
    V = x_t[0, 0].numel()
    if out is None:
        out = torch.empty_like(x_t)
    L.call("tdx_p_sample_step_rng", L.ptr(x_t), L.ptr(eps), L.ptr(x_bcs), L.ptr(mask), L.ptr(sched), T, L.ptr(t_dev),
           int(noise_bcs), int(clip), L.ptr(out), B, F, V, seed, L.ptr(stream_ids), L.ptr(offset_dev), L.stream())
    return out


class _MaskedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eps_hat, noise, mask, n_cells, l1):
        B, F = eps_hat.shape[:2]
        V = eps_hat[0, 0].numel()
        eps_hat, noise = eps_hat.contiguous(), noise.contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=eps_hat.device)
        grad = torch.empty_like(eps_hat) if eps_hat.requires_grad else None
        ws = _ws(L.query("tdx_masked_loss_workspace_bytes"), eps_hat.device)
        if torch.is_tensor(n_cells):  # device int64 scalar, read when the kernels run (captured training step)
            L.call("tdx_masked_loss_dyn", L.ptr(eps_hat), L.ptr(noise), L.ptr(mask), L.ptr(n_cells), int(l1), L.ptr(loss),
                   L.ptr(grad), B, F, V, L.ptr(ws), L.stream())
        else:
            L.call("tdx_masked_loss", L.ptr(eps_hat), L.ptr(noise), L.ptr(mask), n_cells, int(l1), L.ptr(loss), L.ptr(grad),
                   B, F, V, L.ptr(ws), L.stream())
        if grad is not None:
            ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None


def masked_loss(eps_hat, noise, mask, n_cells, l1=False):
    """mean over batch of the mean error over (features, in-domain cells), ddpm.py:845-852.  n_cells: the number of
    in-domain cells as a Python int, or as an int64 device scalar read at run time."""
    if torch.is_tensor(n_cells):
        assert n_cells.dtype == torch.int64 and n_cells.numel() == 1 and n_cells.is_cuda
        return _MaskedLoss.apply(eps_hat, noise, mask, n_cells, l1)
    return _MaskedLoss.apply(eps_hat, noise, mask, int(n_cells), l1)


def randn_philox(out: torch.Tensor, seed: int, stream_id: int, offset_dev: torch.Tensor):
    """Fill `out` (float32) with N(0,1) draws; advances the device-side offset counter."""
    L.call("tdx_randn", L.ptr(out), out.numel(), seed, stream_id, L.ptr(offset_dev), L.stream())
    return out


def randn_philox_batched(out: torch.Tensor, seed: int, stream_ids: torch.Tensor, offset_dev: torch.Tensor):
    """Row b of `out` (B, ...) gets trajectory stream `stream_ids[b]` (int64 device tensor);
    all rows share the offset counter, advanced once.  Graph-capturable."""
    B = out.shape[0]
    L.call("tdx_randn_batched", L.ptr(out), B, out[0].numel(), seed, L.ptr(stream_ids), L.ptr(offset_dev), L.stream())
    return out


# --------------------------------------------------------------------------- FiLM projections of all blocks


class _FilmProjections(torch.autograd.Function):
    """scale | shift of every ResnetBlock from the conditioning vector in ONE launch (tdx_film_fwd; the reference runs
    nn.Linear(c_dim, 2 * dim_out) + chunk per block, ddpm.py:184,191-192), and the three gradients of all of them in
    two (tdx_film_bwd).  Arguments: c (B, T), then weight, bias of every layer; results: one (2, B, C_i) f32 tensor
    per layer, [0] = scale, [1] = shift."""

    @staticmethod
    def forward(ctx, c, *wb):
        n = len(wb) // 2
        c32 = c.detach().float().contiguous()
        B, T = c32.shape
        # fp32 contiguous parameters are used as they are (no per-layer detach / cast / copy calls: with 11 layers they
        # were a tenth of a millisecond of host time in front of the first conv of every forward)
        plain = lambda t: t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()
        ws = [plain(wb[2 * i].detach()) for i in range(n)]
        bs = [None if wb[2 * i + 1] is None else plain(wb[2 * i + 1].detach()) for i in range(n)]
        halves = [w.shape[0] // 2 for w in ws]
        flat = torch.empty(2 * B * sum(halves), dtype=torch.float32, device=c.device)  # all layers' (2, B, C_i) outputs
        outs, off = [], 0
        for h in halves:
            outs.append(flat[off : off + 2 * B * h].view(2, B, h))
            off += 2 * B * h
        base = flat.data_ptr()
        off = 0
        for lo in range(0, n, L.FILM_MAX_LAYERS):
            hi = min(n, lo + L.FILM_MAX_LAYERS)
            tab = (L.FilmLayer * (hi - lo))()
            for j, i in enumerate(range(lo, hi)):
                assert ws[i].shape == (2 * halves[i], T)
                tab[j] = L.FilmLayer(ws[i].data_ptr(), None if bs[i] is None else bs[i].data_ptr(), base + 4 * off, halves[i])
                off += 2 * B * halves[i]
            L.call("tdx_film_fwd", L.ptr(c32), B, T, tab, hi - lo, L.stream())
        ctx.save_for_backward(c32, *ws)
        ctx.has_bias = [b is not None for b in bs]
        ctx.c_dtype = c.dtype
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *gs):
        c32, *ws = ctx.saved_tensors
        n = len(ws)
        B, T = c32.shape
        dev = c32.device
        grads, dc = [], None
        for lo in range(0, n, L.FILM_MAX_LAYERS):
            hi = min(n, lo + L.FILM_MAX_LAYERS)
            tab = (L.FilmGrad * (hi - lo))()
            chans = (L.C.c_int * (hi - lo))()
            keep = []
            for j, i in enumerate(range(lo, hi)):
                C2 = ws[i].shape[0]
                g = gs[i]
                g = torch.zeros((2, B, C2 // 2), dtype=torch.float32, device=dev) if g is None else g.float().contiguous()
                dw = torch.empty_like(ws[i])
                db = torch.empty(C2, dtype=torch.float32, device=dev) if ctx.has_bias[i] else None
                tab[j] = L.FilmGrad(L.ptr(ws[i]), L.ptr(g), L.ptr(dw), L.ptr(db), C2 // 2)
                chans[j] = C2 // 2
                keep.append(g)
                grads += [dw, db]
            part = _ws(L.query("tdx_film_bwd_workspace_bytes", B, T, chans, hi - lo), dev)
            d = torch.empty_like(c32)
            L.call("tdx_film_bwd", L.ptr(c32), B, T, tab, hi - lo, L.ptr(d), L.ptr(part), L.stream())
            dc = d if dc is None else dc + d
        return (dc.to(ctx.c_dtype), *grads)


def film_projections(c: torch.Tensor, linears) -> list:
    """[(2, B, C_i) f32 tensor: scale, shift] for each nn.Linear(c_dim, 2 * C_i) in `linears`, all in one launch."""
    B, T = c.shape
    if not L.query("tdx_film_supported", B, T):
        # batches whose conditioning vectors do not fit the kernels' LDS (B * T * 4 > 160 KiB, i.e. hundreds of
        # samples): the reference's own per-block nn.Linear + chunk (ddpm.py:191-192) through torch
        out = []
        for lin in linears:
            f = torch.nn.functional.linear(c.float(), lin.weight.float(), lin.bias.float())
            out.append(torch.stack(f.chunk(2, dim=1)))
        return out
    args = []
    for lin in linears:
        args += [lin.weight, lin.bias]
    return list(_FilmProjections.apply(c, *args))


# --------------------------------------------------------------------------- fused ResnetBlock


_SIDE = {}  # device index -> side stream of the weight gradients


# the projected skip of a fused ResnetBlock runs as tdx_conv1_fwd_gn's ONE launch; False: two (1x1 conv into a temporary,
# then the GroupNorm / SiLU / add pass) -- a module constant for tests, not an environment switch any more
FUSE_SKIP_TAIL = True
WGRAD_STREAM = os.environ.get("TDX_WGRAD_STREAM", "1") != "0"  # read once (bench.py / tests set it before the import)


class _WgradSide:
    """The 3x3x3 weight gradients of a block run on a side stream next to the rest of its backward: they feed nothing
    before the optimiser and are bound by the matrix cores, what runs beside them (GroupNorm backward, 1x1 convs, the
    halo-shell launches) is bound by memory or latency: 22.94 -> 22.59 ms per step (four alternating pairs on one
    box).  The block joins the streams before it returns -- also when its backward raises (`_ResnetBlock.backward` joins
    in a `finally`) -- so gradient hooks and the optimiser see finished tensors; every tensor the side stream touches is
    marked with `record_stream`, so one that the block drops early (`del dh2`) goes back to the allocator only once the
    side stream is past its last use.  One join at the end of the whole backward pass instead (engine callback) was
    measured too: no faster (22.77 vs 22.71 ms); so was putting the 1x1 skip conv (beside the forward conv chain) and its
    weight gradient on the side stream as well: -0.07 ms, and the forward convs' own durations grow by 9 % under the overlap.
    The side stream only needs the zero block of the scratch arena (`_lib.declare_zero_block_only`: 4 KiB, not 96 MiB).
    TDX_WGRAD_STREAM=0 (read at import; `ops.WGRAD_STREAM`): everything on the launching stream."""

    def __init__(self, device):
        self.on = WGRAD_STREAM
        self.pending = False
        if self.on:
            idx = device.index
            if idx not in _SIDE:
                # Default priority.  torch hands out default-priority handles round-robin from 32 per device, so the 32nd stream a
                # caller creates later IS this one (and inherits its 4-KiB arena: no small-grid kernels on that stream -- correct,
                # slower).  Taking the side stream from the high-priority pool avoids that and costs nothing on one GPU (21.07 /
                # 21.01 / 21.00 vs 21.05 / 21.01 / 21.00 ms) -- but beside a live RCCL communicator it takes the step from 21.2 to
                # 33.0 ms (every kernel slower: forward convs 0.40 -> 0.30 of the peak), so it stays an opt-in (TDX_WGRAD_STREAM_PRIORITY=-1)
                _SIDE[idx] = torch.cuda.Stream(device=device, priority=int(os.environ.get("TDX_WGRAD_STREAM_PRIORITY", "0")))
                L.declare_zero_block_only(_SIDE[idx])
            self.side, self.main = _SIDE[idx], torch.cuda.current_stream(device)

    def run(self, fn, *tensors):
        if not self.on:
            return fn()
        for t in tensors:
            if t is not None:
                t.record_stream(self.side)
        self.side.wait_stream(self.main)
        self.pending = True
        with torch.cuda.stream(self.side):
            fn()

    def join(self):
        if self.on and self.pending:
            self.main.wait_stream(self.side)
            self.pending = False


class _ResnetBlock(torch.autograd.Function):
    """ResnetBlock (reference ddpm.py:180-197) as ONE autograd node:

        h1 = conv3([x1|x2]; w1) ; a1 = silu(GN(h1) * (1 + scale) + shift)
        h2 = conv3(a1; w2)      ; y  = silu(GN(h2)) + res,   res = [x1|x2] (identity) or conv1x1([x1|x2]; wr)

    Doing the backward by hand (instead of chaining the per-op Functions) lets the gradient that
    arrives over the residual path be added inside the data-gradient conv's epilogue / the 1x1
    conv's epilogue, so autograd never runs a separate three-pass add on the block input, and it
    cuts the number of autograd nodes per block from 6-7 to 1."""

    @staticmethod
    def forward(ctx, x1, x2, film, w1, b1, g1, be1, w2, b2, g2, be2, wr, br, groups, eps, partial=None, xc=None,
                xc_real=None, enc=None, dec=None):
        """xc: optional separate input of block1's conv (w1 then has xc's channel count); the residual
        path still uses x1.  Used for the U-Net's first block, whose conv is composed with the 1x1
        encoders and therefore runs on the raw input channels (models/ddpm.py, compose_first_conv)."""
        B, X, Y, Z, C1 = x1.shape if enc is not None else _grid(x1)
        C2 = 0 if x2 is None else x2.shape[-1]
        Cin, Cout = C1 + C2, w1.shape[0]
        if xc is not None:
            assert x2 is None and wr is None and partial is None and xc.shape[:4] == x1.shape[:4]
            xc = xc.contiguous()
        if enc is not None:
            # enc: operands of the encoders whose output x1 stands for (encode_deferred); x1 itself is never read
            assert xc is not None and wr is None and C1 == Cout
        Cc = xc.shape[-1] if xc is not None else Cin
        assert w1.shape[1] == Cc
        V = X * Y * Z
        dev, dt = x1.device, x1.dtype
        code, impl, st = L.dtype_code(dt), L.conv_impl(), L.stream()
        x1 = x1.contiguous() if enc is None else None
        x2 = None if x2 is None else x2.contiguous()
        wf1, wb1 = _packed_conv3(w1, dt)
        wf2, wb2 = _packed_conv3(w2, dt)
        f32c = lambda t: t.detach().float().contiguous()
        g1, be1, g2, be2 = f32c(g1), f32c(be1), f32c(g2), f32c(be2)
        film = f32c(film.reshape(2, B, Cout))  # (scale | shift) of film_projections: dense (B, Cout) halves, no copies
        scale, shift = film[0], film[1]
        gws = _clean_ws(L.query("tdx_gn_workspace_bytes", B, Cout), dev)

        def conv_gn(xa, Ca, xb, Cb, wf, bias, real=None):
            # `real`: channels that carry data (the composed first conv runs on zero-padded raw channels; the
            # timers' work figure counts the algorithmic channels only)
            y = torch.empty((B, X, Y, Z, Cout), dtype=dt, device=dev)
            stats = torch.empty((B, groups, 2), dtype=torch.float32, device=dev)
            L.call("tdx_conv3_fwd_gn", L.ptr(xa), Ca, L.ptr(xb), Cb, L.ptr(wf), L.ptr(bias), L.ptr(y), L.ptr(stats), groups,
                   float(eps), L.ptr(gws), B, X, Y, Z, Cout, code, impl | WS_CLEAN, st,
                   work=54.0 * (real or (Ca + Cb)) * Cout * B * V,
                   meta=lambda: L.conv3_fwd_meta(Ca, Cb, Cout, B, X, Y, Z, dt, real))
            return y, stats

        if partial is not None:
            # inference only: conv1 over the leading n_lead channels of x1, continued from the
            # precomputed convolution of the batch-shared tail (tdx_conv3_fwd_partial)
            n_lead, init = partial
            assert x2 is None and tuple(init.shape) == (1, X, Y, Z, Cout)
            h1 = torch.empty((B, X, Y, Z, Cout), dtype=dt, device=dev)
            st1 = torch.empty((B, groups, 2), dtype=torch.float32, device=dev)
            L.call("tdx_conv3_fwd_partial", L.ptr(x1), n_lead, C1, L.ptr(_packed_conv3_cin_slice(w1, 0, n_lead, dt)),
                   L.ptr(b1), L.ptr(init), 1, L.ptr(h1), L.ptr(st1), groups, float(eps), L.ptr(gws), B, X, Y, Z, Cout, code,
                   impl | WS_CLEAN, st, work=54.0 * n_lead * Cout * B * V)
        elif xc is not None:
            h1, st1 = conv_gn(xc, Cc, None, 0, wf1, b1, real=xc_real)
        else:
            h1, st1 = conv_gn(x1, C1, x2, C2, wf1, b1)
        a1 = torch.empty_like(h1)
        L.call("tdx_gn_apply", L.ptr(h1), L.ptr(st1), L.ptr(g1), L.ptr(be1), L.ptr(scale), L.ptr(shift), None, L.ptr(a1),
               B, V, Cout, groups, 1, code, st)
        h2, st2 = conv_gn(a1, Cout, None, 0, wf2, b2)
        wr2 = None
        fused_tail = False
        if dec is not None:
            # inference only: the block output goes straight through the model's 1x1 decoder (tdx_gn_apply_decode) and is
            # never written; returns the decoded (B, F, X, Y, Z) f32 tensor instead of the block output
            # (grad mode is always off inside Function.forward: the no-autograd condition is the caller's,
            # decode_fused_supported -- nothing is saved for a backward pass on this route)
            assert wr is None and enc is None and x2 is None and Cin == Cout
            wd, bd = dec
            F = wd.shape[0]
            out = torch.empty((B, F, X, Y, Z), dtype=torch.float32, device=dev)
            L.call("tdx_gn_apply_decode", L.ptr(h2), L.ptr(st2), L.ptr(g2), L.ptr(be2), L.ptr(x1),
                   L.ptr(wd.detach().reshape(F, Cout).float().contiguous()), L.ptr(bd.detach().float().contiguous()), L.ptr(out),
                   B, V, Cout, groups, F, code, st)
            return out
        y = torch.empty_like(h1)
        if enc is not None:
            xr, Fx, wx2, bx2, cr, Fc, wc2, bc2, D = enc
            L.call("tdx_gn_apply_encoded", L.ptr(h2), L.ptr(st2), L.ptr(g2), L.ptr(be2), L.ptr(xr), Fx, L.ptr(wx2), L.ptr(bx2),
                   L.ptr(cr), Fc, L.ptr(wc2), L.ptr(bc2), L.ptr(y), B, V, D, groups, code, st)
            fused_tail = True
        elif wr is None:
            assert x2 is None and Cin == Cout
            res = x1
        else:
            wr2 = wr.detach().reshape(Cout, Cin).contiguous()
            if FUSE_SKIP_TAIL and dt in L.H16_DTYPES and C1 % 32 == 0 and C2 % 32 == 0 and Cout % 32 == 0:
                # y = silu(GN(h2)) + conv1x1([x1|x2]) in one pass: the skip tensor is never written or re-read
                L.call("tdx_conv1_fwd_gn", L.ptr(x1), C1, L.ptr(x2), C2, L.ptr(_conv1_wt(wr)), Cout, L.ptr(br), L.ptr(h2),
                       L.ptr(st2), L.ptr(g2), L.ptr(be2), groups, L.ptr(y), B, V, Cout, code, st)
                fused_tail = True
            else:
                res = torch.empty_like(h1)
                L.call("tdx_conv1_fwd", L.ptr(x1), C1, L.ptr(x2), C2, L.ptr(_conv1_wt(wr)), Cout, L.ptr(br), None,
                       L.ptr(res), B * V, Cout, code, st)
        if not fused_tail:
            L.call("tdx_gn_apply", L.ptr(h2), L.ptr(st2), L.ptr(g2), L.ptr(be2), None, None, L.ptr(res), L.ptr(y), B, V, Cout,
                   groups, 1, code, st)
        ctx.save_for_backward(x1, x2, h1, st1, a1, h2, st2, film, g1, be1, g2, be2, wb1, wb2, wr2, xc)
        ctx.cfg = (groups, tuple(w1.shape), tuple(w2.shape), None if wr is None else tuple(wr.shape),
                   b1 is not None, b2 is not None, br is not None)
        ctx.xc_real, ctx.impl, ctx.c1 = xc_real, impl, C1
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        side = _WgradSide(gy.device)
        try:
            return _ResnetBlock._backward(ctx, gy, side)
        finally:
            side.join()  # also on an exception: no side-stream work may outlive the tensors it touches

    @staticmethod
    def _backward(ctx, gy, side):
        x1, x2, h1, st1, a1, h2, st2, film, g1, be1, g2, be2, wb1, wb2, wr2, xc = ctx.saved_tensors
        scale, shift = film[0], film[1]
        groups, w1s, w2s, wrs, hb1, hb2, hbr = ctx.cfg
        B, X, Y, Z, _ = _grid(h1)
        C1 = ctx.c1  # (x1 is None when the skip was evaluated from the encoders' operands)
        C2 = 0 if x2 is None else x2.shape[-1]
        Cin, Cout = C1 + C2, w1s[0]
        Cc = w1s[1]  # input channels of block1's conv (= Cin unless the conv has its own input xc)
        V = X * Y * Z
        gy = gy.contiguous()
        dev, dt = gy.device, gy.dtype
        code, impl, st = L.dtype_code(dt), ctx.impl, L.stream()
        f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        gws = _ws(L.query("tdx_gn_workspace_bytes", B, Cout), dev)
        flops = lambda ci: 54.0 * ci * Cout * B * V

        # ---- block2: GroupNorm + SiLU (+ residual: its gradient is gy itself)
        dh2, dg2, dbe2 = torch.empty_like(h2), f32(Cout), f32(Cout)
        L.call("tdx_gn_bwd", L.ptr(h2), L.ptr(gy), L.ptr(st2), L.ptr(g2), L.ptr(be2), None, None, L.ptr(dh2), L.ptr(dg2),
               L.ptr(dbe2), None, None, B, V, Cout, groups, 1, code, L.ptr(gws), st)
        dw2, db2 = f32(*w2s), (f32(Cout) if hb2 else None)
        # one workspace per (Cin, Cout): the accumulator / slab layout inside depends on both
        wws2 = _clean_ws(L.query("tdx_conv3_bwd_weight_workspace_bytes", Cout, Cout, impl), dev, ("w3", Cout, Cout))
        wws1 = _clean_ws(L.query("tdx_conv3_bwd_weight_workspace_bytes", Cc, Cout, impl), dev, ("w3", Cc, Cout))
        side.run(lambda: L.call("tdx_conv3_bwd_weight", L.ptr(a1), Cout, None, 0, L.ptr(dh2), L.ptr(dw2), L.ptr(db2), B, X, Y, Z,
                                Cout, code, impl | WS_CLEAN, L.ptr(wws2), L.stream(), work=flops(Cout)), a1, dh2, dw2, db2)
        da1 = torch.empty_like(a1)
        dws = _ws(L.query("tdx_conv3_bwd_data_workspace_bytes", B, X, Y, Z, max(Cin, Cout, Cc), code, impl), dev)
        L.call("tdx_conv3_bwd_data", L.ptr(dh2), L.ptr(wb2), L.ptr(da1), Cout, None, 0, 0, B, X, Y, Z, Cout, code, impl,
               L.ptr(dws), st, work=flops(Cout))
        del dh2
        # ---- block1: GroupNorm + FiLM + SiLU
        dh1, dg1, dbe1, dfilm = torch.empty_like(h1), f32(Cout), f32(Cout), f32(2, B, Cout)
        dscale, dshift = dfilm[0], dfilm[1]
        L.call("tdx_gn_bwd", L.ptr(h1), L.ptr(da1), L.ptr(st1), L.ptr(g1), L.ptr(be1), L.ptr(scale), L.ptr(shift), L.ptr(dh1),
               L.ptr(dg1), L.ptr(dbe1), L.ptr(dscale), L.ptr(dshift), B, V, Cout, groups, 1, code, L.ptr(gws), st)
        del da1
        dw1, db1 = f32(*w1s), (f32(Cout) if hb1 else None)
        if xc is not None:
            # block1's conv has its own input: weight gradient w.r.t. that input; the block input x1 only
            # feeds the identity skip, so its gradient is gy; the data gradient of the conv is needed only
            # if xc itself requires one (e.g. a learned cell-type embedding behind the raw conditioning)
            side.run(lambda: L.call("tdx_conv3_bwd_weight", L.ptr(xc), Cc, None, 0, L.ptr(dh1), L.ptr(dw1), L.ptr(db1), B, X, Y, Z,
                                    Cout, code, impl | WS_CLEAN, L.ptr(wws1), L.stream(), work=flops(ctx.xc_real or Cc)),
                     xc, dh1, dw1, db1)
            dxc = None
            if ctx.needs_input_grad[16]:
                dxc = torch.empty_like(xc)
                L.call("tdx_conv3_bwd_data", L.ptr(dh1), L.ptr(wb1), L.ptr(dxc), Cc, None, 0, 0, B, X, Y, Z, Cout, code, impl,
                       L.ptr(dws), st, work=flops(ctx.xc_real or Cc))
            side.join()
            return (gy, None, dfilm, dw1, db1, dg1, dbe1, dw2, db2, dg2, dbe2, None, None, None, None, None, dxc, None, None,
                    None)
        side.run(lambda: L.call("tdx_conv3_bwd_weight", L.ptr(x1), C1, L.ptr(x2), C2, L.ptr(dh1), L.ptr(dw1), L.ptr(db1), B, X, Y, Z,
                                Cout, code, impl | WS_CLEAN, L.ptr(wws1), L.stream(), work=flops(Cin)), x1, x2, dh1, dw1, db1)
        # ---- input gradient = conv1 data gradient + residual-path gradient
        gx1 = torch.empty_like(x1)
        gx2 = None if x2 is None else torch.empty_like(x2)
        dwr = dbr = None
        if wr2 is None:  # identity skip: add gy inside the data-gradient epilogue
            L.call("tdx_conv3_bwd_data_add", L.ptr(dh1), L.ptr(wb1), L.ptr(gx1), C1, None, 0, L.ptr(gy), None, B, X, Y, Z, Cout,
                   code, impl, L.ptr(dws), st, work=flops(Cin))
        else:
            t1 = torch.empty_like(x1)
            t2 = None if x2 is None else torch.empty_like(x2)
            L.call("tdx_conv3_bwd_data", L.ptr(dh1), L.ptr(wb1), L.ptr(t1), C1, L.ptr(t2), C2, 0, B, X, Y, Z, Cout, code, impl,
                   L.ptr(dws), st, work=flops(Cin))
            # 1x1 skip: dx = gy @ wr (+ t) -- the add rides in the 1x1 kernel's epilogue
            L.call("tdx_conv1_fwd", L.ptr(gy), Cout, None, 0, wr2.data_ptr(), Cin, None, L.ptr(t1), L.ptr(gx1), B * V, C1, code, st)
            if x2 is not None:
                L.call("tdx_conv1_fwd", L.ptr(gy), Cout, None, 0, wr2.data_ptr() + 4 * C1, Cin, None, L.ptr(t2), L.ptr(gx2),
                       B * V, C2, code, st)
            dwr, dbr = _conv1_weight_grad(x1, C1, x2, C2, gy, Cout, hbr, B * V, code, st)
            dwr = dwr.view(wrs)
        side.join()
        return (gx1, gx2, dfilm, dw1, db1, dg1, dbe1, dw2, db2, dg2, dbe2, dwr, dbr, None, None, None, None, None, None, None)


def resnet_block(x1, x2, scale, shift, conv1_wb, norm1_wb, conv2_wb, norm2_wb, skip_wb, groups, eps=1e-5, partial=None,
                 conv1_input=None, conv1_real_channels=None, film=None, skip_encoded=None, decode_wb=None):
    """Fused ResnetBlock; *_wb are (weight, bias) pairs, skip_wb is None for an identity skip.
    Requires SiLU activations and bf16/f32 NDHWC inputs; scale/shift are (B, Cout), or film = the (2, B, Cout)
    [scale, shift] tensor of film_projections (then scale and shift are ignored: no slicing copies either way).
    partial = (n_lead, init): no-grad only, see conv3_shared_tail.
    conv1_input: separate input tensor of block1's conv (conv1_wb then matches ITS channel count);
    conv1_real_channels: how many of its channels carry data (the rest is zero padding) -- bookkeeping for the
    kernel timers only.
    skip_encoded: a DeferredEncoding whose `.standin` is x1 -- the identity skip is then evaluated from the encoders'
    operands inside the tail kernel (needs conv1_input: x1 has no data to convolve).
    decode_wb = (weight (F, C, 1, 1, 1), bias): no-grad only, identity-skip blocks only -- returns
    decode(block output) as (B, F, X, Y, Z) f32 (ops.decode's result) without writing the block output
    (decode_fused_supported says when)."""
    wr, br = skip_wb if skip_wb is not None else (None, None)
    assert decode_wb is None or not torch.is_grad_enabled(), "decode_wb: inference only (the block output is not kept)"
    assert partial is None or not torch.is_grad_enabled(), "partial: inference only (the backward needs the full conv)"
    enc = None
    if skip_encoded is not None:
        assert x1 is skip_encoded.standin and conv1_input is not None and skip_wb is None and x2 is None
        enc = skip_encoded.operands
    if film is None:
        B, Cout = x1.shape[0], conv1_wb[0].shape[0]
        film = torch.stack((scale.reshape(B, Cout).float(), shift.reshape(B, Cout).float()))
    return _ResnetBlock.apply(x1, x2, film, conv1_wb[0], conv1_wb[1], norm1_wb[0], norm1_wb[1], conv2_wb[0],
                              conv2_wb[1], norm2_wb[0], norm2_wb[1], wr, br, groups, eps, partial, conv1_input,
                              conv1_real_channels, enc, decode_wb)


# --------------------------------------------------------------------------- baseline conv variants


def _taps_first(w5: torch.Tensor, in_axis: int, out_axis: int) -> torch.Tensor:
    """(.., k, k, k) weight with its in / out channel axes named -> [k^3][in][out] f32 (tdx_convg_* operand)."""
    k = w5.shape[-1]
    return w5.detach().float().permute(2, 3, 4, in_axis, out_axis).reshape(k**3, w5.shape[in_axis], w5.shape[out_axis]).contiguous()


def _convg_apply(x, w_t, bias, out_grid, Cout, k, stride, dilation, pad, replicate, transposed):
    B, Xi, Yi, Zi, Cin = _grid(x)
    y = torch.empty((B, *out_grid, Cout), dtype=x.dtype, device=x.device)
    L.call("tdx_convg_apply", L.ptr(x), L.ptr(w_t), L.ptr(bias), L.ptr(y), B, Xi, Yi, Zi, Cin, *out_grid, Cout, k, stride,
           dilation, pad, int(replicate), int(transposed), L.dtype_code(x.dtype), L.stream())
    return y


class _ConvG(torch.autograd.Function):
    """nn.Conv3d(k, stride, dilation, padding, padding_mode in {"zeros", "replicate"}) on NDHWC tensors: the dilated
    convs of DilatedCNNBlock (dilresnet.py:28-35) and the strided convs of tfnet's conv() (tfnet.py:187-193)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, dilation, padding, replicate):
        B, Xi, Yi, Zi, Cin = _grid(x)
        Cout, k = weight.shape[0], weight.shape[-1]
        assert weight.shape[1] == Cin and not (replicate and stride != 1), "replicate padding: stride 1 only"
        x = x.contiguous()
        out = tuple((e + 2 * padding - dilation * (k - 1) - 1) // stride + 1 for e in (Xi, Yi, Zi))
        y = _convg_apply(x, _taps_first(weight, 1, 0), None if bias is None else bias.detach().float().contiguous(), out, Cout, k,
                         stride, dilation, padding, replicate, False)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, dilation, padding, replicate, bias is not None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        stride, dilation, pad, replicate, has_bias = ctx.cfg
        B, Xi, Yi, Zi, Cin = _grid(x)
        Cout, k = weight.shape[0], weight.shape[-1]
        gy = gy.contiguous()
        Xo, Yo, Zo = gy.shape[1:4]
        code, st = L.dtype_code(x.dtype), L.stream()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            w_b = _taps_first(weight, 0, 1)  # [k^3][Cout][Cin]
            if replicate:  # adjoint on the padded grid, then every padded position onto the voxel it clamps to
                dpad = _convg_apply(gy, w_b, None, (Xi + 2 * pad, Yi + 2 * pad, Zi + 2 * pad), Cin, k, stride, dilation, 0, False, True)
                gx = torch.empty_like(x)
                L.call("tdx_convg_fold_clamp", L.ptr(dpad), L.ptr(gx), B, Xi, Yi, Zi, pad, Cin, code, st)
            else:
                gx = _convg_apply(gy, w_b, None, (Xi, Yi, Zi), Cin, k, stride, dilation, pad, False, True)
        if ctx.needs_input_grad[1]:
            dw = torch.zeros((k**3, Cin, Cout), dtype=torch.float32, device=x.device)
            gb = torch.zeros(Cout, dtype=torch.float32, device=x.device) if has_bias else None
            L.call("tdx_convg_bwd_weight", L.ptr(x), L.ptr(gy), L.ptr(dw), L.ptr(gb), B, Xi, Yi, Zi, Cin, Xo, Yo, Zo, Cout, k,
                   stride, dilation, pad, int(replicate), code, st)
            gw = dw.reshape(k, k, k, Cin, Cout).permute(4, 3, 0, 1, 2).contiguous()
        return gx, gw, gb, None, None, None, None


def conv3d(x, weight, bias=None, stride=1, dilation=1, padding=0, padding_mode="zeros"):
    """General NDHWC 3-D convolution (weight in nn.Conv3d's layout (Cout, Cin, k, k, k)); padding_mode "zeros" or
    "replicate".  Channel counts must be multiples of 8.  Vector-ALU kernels: baseline models only -- the U-Net's
    3x3x3 convs go through conv3 / resnet_block."""
    if padding_mode not in ("zeros", "replicate"):
        raise ValueError(f"padding_mode {padding_mode!r}")
    return _ConvG.apply(x, weight, bias, int(stride), int(dilation), int(padding), padding_mode == "replicate")


class _ConvTransposeG(torch.autograd.Function):
    """nn.ConvTranspose3d(k, stride, padding) (tfnet.py:203-205: k = 4, stride 2, padding 1) on NDHWC tensors;
    weight in its layout (Cin, Cout, k, k, k)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding):
        B, Xi, Yi, Zi, Cin = _grid(x)
        Cout, k = weight.shape[1], weight.shape[-1]
        assert weight.shape[0] == Cin
        x = x.contiguous()
        out = tuple((e - 1) * stride - 2 * padding + k for e in (Xi, Yi, Zi))
        y = _convg_apply(x, _taps_first(weight, 0, 1), None if bias is None else bias.detach().float().contiguous(), out, Cout, k,
                         stride, 1, padding, False, True)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, padding, bias is not None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        stride, pad, has_bias = ctx.cfg
        B, Xi, Yi, Zi, Cin = _grid(x)
        Cout, k = weight.shape[1], weight.shape[-1]
        gy = gy.contiguous()
        Xo, Yo, Zo = gy.shape[1:4]
        code, st = L.dtype_code(x.dtype), L.stream()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:  # a strided conv of dy
            gx = _convg_apply(gy, _taps_first(weight, 1, 0), None, (Xi, Yi, Zi), Cin, k, stride, 1, pad, False, False)
        if ctx.needs_input_grad[1]:
            # dW[ci][co][t] = sum_i x[i][ci] dy[i*s - p + t][co]: the weight gradient of that strided conv, roles swapped
            dw = torch.zeros((k**3, Cout, Cin), dtype=torch.float32, device=x.device)
            L.call("tdx_convg_bwd_weight", L.ptr(gy), L.ptr(x), L.ptr(dw), None, B, Xo, Yo, Zo, Cout, Xi, Yi, Zi, Cin, k, stride,
                   1, pad, 0, code, st)
            gw = dw.reshape(k, k, k, Cout, Cin).permute(4, 3, 0, 1, 2).contiguous()
            if has_bias:
                gb = gy.float().sum(dim=(0, 1, 2, 3))
        return gx, gw, gb, None, None


def conv_transpose3d(x, weight, bias=None, stride=2, padding=1):
    """NDHWC transposed 3-D convolution (weight (Cin, Cout, k, k, k) as in nn.ConvTranspose3d)."""
    return _ConvTransposeG.apply(x, weight, bias, int(stride), int(padding))
