"""DenoisingModel (3D U-Net) and GaussianDiffusion (DDPM) on the MI355X HIP kernels.

Drop-in for the public surface of the reference's ``turbdiff/models/ddpm.py``:
same class names, constructor keywords, method names / arguments and -- because the
parameter containers are the same torch modules in the same attribute tree -- the same
``state_dict`` keys and default initialisation.  What differs is the
execution: module ``forward``s do not call ``nn.Conv3d``/``nn.GroupNorm``; they hand their
parameters to the fused NDHWC operators in ``turbdiff_amd.ops``.

Layout contract: ``DenoisingModel.forward`` and every ``GaussianDiffusion`` method take and
return the reference's NCDHW float32 tensors ``(B, F, X, Y, Z)``.  The building blocks below
(``Block``, ``ResnetBlock``, ``Attention``, ``UNet``) run on NDHWC tensors ``(B, X, Y, Z, C)``
in the model's compute dtype (float32 or bfloat16 storage, fp32 accumulation).
"""

from __future__ import annotations

import math
import os
from dataclasses import dataclass
from functools import partial

import torch
import torch.nn.functional as F
from torch import nn

from .. import _lib, ops, schedules
from ..sequential import KwargsSequential
from .conditioning import global_conditioning, local_conditioning
from .utils import broadcast_right


@dataclass
class ModelPrediction:
    noise: torch.Tensor
    x_start: torch.Tensor
    mean: torch.Tensor
    log_var: torch.Tensor


# --------------------------------------------------------------------------- time embedding


class NyquistFrequencyEmbedding(nn.Module):
    """sin(bias + scale t) with k = dim/2 geometric frequencies, each as sin and cos
    (reference ddpm.py:103-148).  B x dim values: stays a torch op."""

    def __init__(self, dim: int, timesteps: int):
        super().__init__()
        scale, bias = schedules.nyquist_embedding_tables(dim, timesteps)
        self.register_buffer("scale", scale, persistent=False)
        self.register_buffer("bias", bias, persistent=False)

    def forward(self, t):
        return torch.addcmul(self.bias, self.scale, t[..., None]).sin()


class SinusoidalPosEmb(nn.Module):
    """Transformer-style embedding; unused by the default model, kept for API parity
    (reference ddpm.py:88-100)."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, t):
        half = self.dim // 2
        freq = torch.exp(torch.arange(half, device=t.device) * (-math.log(10000) / (half - 1)))
        ang = t[:, None] * freq[None, :]
        return torch.cat((ang.sin(), ang.cos()), dim=-1)


# --------------------------------------------------------------------------- blocks (NDHWC)


# one autograd node per ResnetBlock (hand-written backward) instead of one per operator;
# FUSE_BLOCKS = False selects the per-operator composition (same kernels, used as a cross-check by the tests)

FUSE_BLOCKS = True
# Module constants, not environment switches (they were, while each route was being compared with its alternative;
# tests flip them through monkeypatch):
# COMPOSE_FIRST_CONV = False: the first U-Net conv runs on the 64 encoded channels as written in the reference
COMPOSE_FIRST_CONV = True
# DEFER_ENCODE = False: the encoder output is written by tdx_encode_fwd and read back by the first block's skip
DEFER_ENCODE = True
# FUSE_DECODE = False: inference writes the last block's output and runs tdx_decode_fwd on it, as training does
FUSE_DECODE = True
# CACHE_COND_CONV = False: sampling recomputes the conditioning half of the first conv every step
CACHE_COND_CONV = True


def _norm_groups(norm: nn.GroupNorm) -> int:
    return norm.num_groups


class Block(nn.Module):
    """conv3x3x3(replicate) -> GroupNorm -> [FiLM] -> activation  (reference ddpm.py:154-177).

    ``forward`` fuses GroupNorm, the FiLM affine, SiLU and an optional residual add into one
    stats pass + one apply pass over the conv output."""

    def __init__(self, dim, dim_out, actfn, norm_klass=None):
        super().__init__()
        self.conv = nn.Conv3d(dim, dim_out, 3, padding=1, padding_mode="replicate")
        self.norm = norm_klass(dim_out)
        self.act = actfn()
        self._fused_act = isinstance(self.act, nn.SiLU)

    def forward(self, x, scale_shift=None, x2=None, res=None):
        G = _norm_groups(self.norm)
        h, stats = ops.conv3_gn_stats(x, self.conv.weight, self.conv.bias, G, self.norm.eps, x2=x2)
        scale, shift = scale_shift if scale_shift is not None else (None, None)
        if self._fused_act:
            return ops.gn_film_silu(h, self.norm.weight, self.norm.bias, G, scale, shift, res=res, act=True,
                                    eps=self.norm.eps, stats=stats)
        h = ops.gn_film_silu(h, self.norm.weight, self.norm.bias, G, scale, shift, act=False, eps=self.norm.eps,
                             stats=stats)
        h = self.act(h)
        return h if res is None else h + res


class ResnetBlock(nn.Module):
    """Two Blocks, FiLM-conditioned on `c` in the first, plus a 1x1-projected (or identity)
    skip of the block input (reference ddpm.py:180-197).  The input may be given as two
    tensors whose channel concatenation is never materialised (`x2`)."""

    def __init__(self, dim_in, dim_out, *, c_dim: int, actfn, norm_klass):
        super().__init__()
        self.project_onto_scale_shift = nn.Linear(c_dim, dim_out * 2)
        self.block1 = Block(dim_in, dim_out, actfn=actfn, norm_klass=norm_klass)
        self.block2 = Block(dim_out, dim_out, actfn=actfn, norm_klass=norm_klass)
        self.conv = nn.Conv3d(dim_in, dim_out, 1) if dim_in != dim_out else nn.Identity()
        self.dim_out = dim_out

    def fused(self, x2=None):
        identity = isinstance(self.conv, nn.Identity)
        return self.block1._fused_act and self.block2._fused_act and not (identity and x2 is not None) and FUSE_BLOCKS

    def forward(self, x, c, x2=None, partial=None, conv1=None, films=None, skip_encoded=None, decode_wb=None):
        """partial = (n_lead, init): inference only -- block1's conv runs over the leading n_lead channels
        of x and continues from `init`, the precomputed conv of the batch-shared remaining channels.
        conv1 = (input, weight, bias): block1's conv replaced by an equivalent conv on another input
        (DenoisingModel.compose_first_conv); x still feeds the identity skip -- or, with skip_encoded (an
        ops.DeferredEncoding whose stand-in x is), the skip is evaluated from the encoders' operands in the tail kernel.
        decode_wb = (weight, bias) of the model's final 1x1 conv: inference only -- returns that conv's (B, F, X, Y, Z)
        output, computed in the tail kernel (ops.decode_fused_supported).
        films = {id(block): (2, B, dim_out) scale | shift}: the projections of all blocks computed up front in one
        launch (DenoisingModel.film_table); without it the block projects `c` itself."""
        film = films.get(id(self)) if films is not None else None
        if film is None:
            film = ops.film_projections(c, [self.project_onto_scale_shift])[0]  # (2, B, dim_out): scale, shift
        scale, shift = film[0], film[1]
        identity = isinstance(self.conv, nn.Identity)
        if self.fused(x2):
            b1, b2 = self.block1, self.block2
            if conv1 is not None:
                assert identity and x2 is None
                return ops.resnet_block(x, None, scale, shift, (conv1[1], conv1[2]), (b1.norm.weight, b1.norm.bias),
                                        (b2.conv.weight, b2.conv.bias), (b2.norm.weight, b2.norm.bias), None,
                                        _norm_groups(b1.norm), b1.norm.eps, conv1_input=conv1[0],
                                        conv1_real_channels=conv1[3] if len(conv1) > 3 else None, film=film,
                                        skip_encoded=skip_encoded)
            return ops.resnet_block(x, x2, scale, shift, (b1.conv.weight, b1.conv.bias), (b1.norm.weight, b1.norm.bias),
                                    (b2.conv.weight, b2.conv.bias), (b2.norm.weight, b2.norm.bias),
                                    None if identity else (self.conv.weight, self.conv.bias), _norm_groups(b1.norm),
                                    b1.norm.eps, partial=partial, film=film, decode_wb=decode_wb)
        assert partial is None and conv1 is None and skip_encoded is None and decode_wb is None
        h = self.block1(x, scale_shift=(scale, shift), x2=x2)
        if isinstance(self.conv, nn.Identity):
            skip = x if x2 is None else torch.cat((x, x2), dim=-1)
        else:
            skip = ops.conv1(x, self.conv.weight, self.conv.bias, x2=x2)
        return self.block2(h, res=skip)


class Attention(nn.Module):
    """Multi-head self-attention over all voxels (reference ddpm.py:286-308).  to_qkv's NDHWC
    output already is the token-major q|k|v matrix the attention kernel reads, and the
    surrounding residual add is folded into the to_out projection."""

    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        self.heads = heads
        hidden = dim_head * heads
        self.to_qkv = nn.Conv3d(dim, hidden * 3, 1, bias=False)
        self.to_out = nn.Conv3d(hidden, dim, 1)

    def forward(self, x, residual=None):
        B, X, Y, Z, _ = x.shape
        qkv = ops.conv1(x, self.to_qkv.weight)
        out = ops.attention(qkv.reshape(B, X * Y * Z, -1), self.heads)
        return ops.conv1(out.reshape(B, X, Y, Z, -1), self.to_out.weight, self.to_out.bias, add=residual)


class PreNorm(nn.Module):
    def __init__(self, norm: nn.Module, fn: nn.Module, enabled: bool = True):
        super().__init__()
        self.norm = norm
        self.fn = fn

    def forward(self, x, residual=None):
        xn = ops.gn_film_silu(x, self.norm.weight, self.norm.bias, _norm_groups(self.norm), act=False, eps=self.norm.eps)
        return self.fn(xn) if residual is None else self.fn(xn, residual=residual)


class Residual(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x):
        if isinstance(self.fn, PreNorm) and isinstance(self.fn.fn, Attention):
            return self.fn(x, residual=x)
        return self.fn(x) + x


def down_size(spatial):
    """Halve each extent, never below the kernel size 3 (reference ddpm.py:358)."""
    return [max(int(s * 0.5), 3) for s in spatial]


class UNet(nn.Module):
    """Encoder / bottleneck / decoder with trilinear resampling between levels and skip
    concatenation (reference ddpm.py:326-372).  Skips are passed to the decoder blocks as a
    second conv input instead of being concatenated."""

    def __init__(self, downsampling_blocks, upsampling_blocks, center_block, *, downsampling_factor: float = 2.0):
        super().__init__()
        assert len(downsampling_blocks) == len(upsampling_blocks)
        self.downsampling_blocks = nn.ModuleList(downsampling_blocks)
        self.upsampling_blocks = nn.ModuleList(upsampling_blocks)
        self.center_block = center_block
        self.downsampling_factor = downsampling_factor
        self.scale_factor = 1 / downsampling_factor

    def forward(self, x, c, first_partial=None, first_conv=None, films=None, first_skip=None):
        skips = []
        assert first_skip is None or first_conv is not None
        for i, blk in enumerate(self.downsampling_blocks):
            if i == 0 and first_conv is not None:
                x = blk(x, c, conv1=first_conv, films=films, skip_encoded=first_skip)
            elif i == 0 and first_partial is not None:
                x = blk(x, c, partial=first_partial, films=films)
            else:
                x = blk(x, c, films=films)
            skip, x = ops.skip_and_resize(x, [max(int(s * self.scale_factor), 3) for s in x.shape[1:4]])
            skips.append(skip)
        x = self.center_block(x, c=c, films=films)
        for blk in self.upsampling_blocks:
            skip = skips.pop()
            x = blk(ops.resize(x, skip.shape[1:4]), c, x2=skip, films=films)
        return x


class GeometryEmbedding(nn.Module):
    """Strided-conv summary of the obstacle region (reference ddpm.py:375-395); off in the
    shipped configuration (config/model/diffusion.yaml:30), kept on stock torch ops."""

    def __init__(self, in_features, out_features, actfn):
        super().__init__()
        self.in_features, self.out_features, self.actfn = in_features, out_features, actfn
        self.extract_features = nn.Sequential(
            nn.Conv3d(in_features, out_features, kernel_size=5, stride=5),
            actfn(),
            nn.Conv3d(out_features, out_features, kernel_size=5, stride=1),
            actfn(),
            nn.Conv3d(out_features, out_features, kernel_size=5, stride=5),
        )

    def forward(self, c_local):
        front = torch.narrow(c_local, dim=-3, start=0, length=50)
        return self.extract_features(front).mean(dim=(-3, -2, -1))


# --------------------------------------------------------------------------- the denoiser


class DenoisingModel(nn.Module):
    """eps_theta(x_t, t, C): the turbdiff 3D U-Net (reference ddpm.py:398-505)."""

    def __init__(self, *, in_features: int, out_features: int, c_local_features: int, c_global_features: int,
                 timesteps: int, dim: int, u_net_levels: int, actfn=nn.SiLU, norm_type: str = "instance",
                 with_geometry_embedding: bool = False):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.c_local_features = c_local_features
        self.c_global_features = c_global_features
        self.dim = dim
        self.timesteps = timesteps
        self.u_net_levels = u_net_levels
        self.with_geometry_embedding = with_geometry_embedding
        self.compute_dtype = torch.float32
        self.conv_impl = None  # "auto" / "split" / "direct" / "mfma": this model's 3x3x3 conv arithmetic (_lib.conv_impl_scope)

        groups_of = {"instance": lambda ch: ch, "layer": lambda ch: 1, "group": lambda ch: 8}
        if norm_type not in groups_of:
            raise RuntimeError(f"Unknown norm type {norm_type}")
        norm_klass = lambda ch: nn.GroupNorm(groups_of[norm_type](ch), ch)

        # construction order follows the reference so that seeded default init matches
        self.encode_x = nn.Conv3d(in_features, dim, 1)
        c_local_dim = 0
        if c_local_features > 0:
            self.encode_c_local = nn.Conv3d(c_local_features, dim, 1)
            c_local_dim = dim
        c_dim = dim
        self.encode_t = NyquistFrequencyEmbedding(dim, timesteps)
        if c_global_features > 0:
            self.encode_c_global = nn.Linear(c_global_features, dim)
            c_dim += dim
        if with_geometry_embedding and c_local_features > 0:
            self.geometry_embedding = GeometryEmbedding(c_local_features, dim, actfn)
            c_dim += dim
        self.process_c = nn.Sequential(nn.Linear(c_dim, 4 * c_dim), actfn(), nn.Linear(4 * c_dim, c_dim), actfn())

        rb = partial(ResnetBlock, c_dim=c_dim, actfn=actfn, norm_klass=norm_klass)
        self.decode = KwargsSequential(rb(dim, dim), nn.Conv3d(dim, out_features, 1))
        down = [rb(dim + c_local_dim, dim * 2)] + [rb(dim * 2**i, dim * 2 ** (i + 1)) for i in range(1, u_net_levels)]
        up = [rb(2 * dim * 2 ** (i + 1), dim * 2**i) for i in reversed(range(u_net_levels))]
        mid = dim * 2**u_net_levels
        center = KwargsSequential(rb(mid, mid), Residual(PreNorm(norm_klass(mid), Attention(mid))), rb(mid, mid))
        self.u_net = UNet(down, up, center)

    def grad_ready_order(self):
        """Parameters in the order their gradients become ready in backward (the reverse of the forward's
        execution order, which is NOT the registration order): decode, up path from the finest level down,
        bottleneck, down path from the deepest level up, encoders, and last the conditioning MLP, whose
        input gradient collects a term from every block.  ``parallel.BucketedDataParallel`` lays out its
        all-reduce buckets in this order, so the first training step already overlaps."""
        mods = [self.decode[1], self.decode[0]]
        mods += list(reversed(self.u_net.upsampling_blocks))
        mods += list(reversed(list(self.u_net.center_block)))
        mods += list(reversed(self.u_net.downsampling_blocks))
        mods += [getattr(self, n) for n in ("encode_c_local", "encode_x") if hasattr(self, n)]
        # the FiLM projections of all blocks are one autograd node (film_table): their gradients arrive together,
        # after the last block's backward and right before the conditioning MLP's
        film = {id(p) for m in self.modules() if isinstance(m, ResnetBlock) for p in m.project_onto_scale_shift.parameters()}
        late = []
        for m in mods:
            for p in reversed(list(m.parameters())):
                if id(p) in film:
                    late.append(p)
                else:
                    yield p
        yield from late
        for n in ("geometry_embedding", "encode_c_global", "process_c"):
            if hasattr(self, n):
                yield from reversed(list(getattr(self, n).parameters()))

    def __getstate__(self):
        # copy.deepcopy / pickle: the cached module lists and pack plans (device buffers, foreign job tables) stay behind
        state = super().__getstate__()
        state.pop("_lists", None)
        return state

    def _static_lists(self):
        """Module lists the forward needs every call (the module tree does not change after construction): the 3x3x3 and
        1x1 weights whose operands are prefetched, and the ResnetBlocks in FiLM order."""
        hit = self.__dict__.get("_lists")
        if hit is None:
            conv3, conv1, blocks = [], [], []
            for m in self.modules():
                if isinstance(m, Block):
                    conv3.append(m.conv)
                elif isinstance(m, ResnetBlock):
                    blocks.append(m)
                    if not isinstance(m.conv, nn.Identity):
                        conv1.append(m.conv)
                elif isinstance(m, Attention):
                    conv1 += [m.to_qkv, m.to_out]
            hit = self.__dict__["_lists"] = (conv3, conv1, blocks, {})
        return hit

    def prefetch_weights(self, skip_first_conv: bool):
        """Packed operands of all 3x3x3 weights and transposed copies of all 1x1 weights the forward is about to
        ask for, refreshed in one launch each (ops.prefetch_weights) instead of one launch per layer on first use."""
        conv3, conv1, _, plans = self._static_lists()
        first = self.u_net.downsampling_blocks[0].block1.conv if skip_first_conv else None
        key = (bool(skip_first_conv), self.compute_dtype, _lib.pack_code(self.compute_dtype))
        plans[key] = ops.prefetch_weights([m.weight for m in conv3 if m is not first], [m.weight for m in conv1],
                                          self.compute_dtype, plans.get(key))

    def film_table(self, c):
        """{id(block): (2, B, dim_out) [scale, shift]} for every ResnetBlock, projected from the conditioning vector
        in one launch (the reference projects inside each block, ddpm.py:191-192)."""
        blocks = self._static_lists()[2]
        films = ops.film_projections(c, [b.project_onto_scale_shift for b in blocks])
        return {id(b): f for b, f in zip(blocks, films)}

    def set_compute_dtype(self, dtype: torch.dtype):
        """float32 (parity modes), bfloat16 or float16 (activation storage + MFMA operands; float16 = 11 significand bits,
        the arithmetic of the reference's TF32 GPU runs, at the bfloat16 kernels' speed -- training in it needs a loss scale:
        training.DiffusionTrainer(compute_mode="fp16"))."""
        assert dtype in (torch.float32, torch.bfloat16, torch.float16)
        self.compute_dtype = dtype
        return self

    def conditioning_vector(self, t, C, batch_size):
        parts = [self.encode_t(t)]
        c_global = global_conditioning(C)
        if c_global is not None:
            parts.append(self.encode_c_global(c_global))
        if self.with_geometry_embedding:
            c_local = local_conditioning(C)
            if c_local is not None:
                parts.append(self.geometry_embedding(c_local).expand((batch_size, -1)))
        # (one part is the common case: torch.cat of a single tensor is a device-to-device memcpy -- a memcpy NODE in a captured step)
        return self.process_c(parts[0] if len(parts) == 1 else torch.cat(parts, dim=-1))

    def conditioning_table(self, C, timesteps: int):
        """(timesteps, c_dim) tensor whose row t is conditioning_vector(t, C, 1) -- when that vector depends on t alone (no
        global conditioning, no geometry embedding: the shipped configuration, config/model/diffusion.yaml) -- else None.
        A sampler computes it once and passes its row as `cond=` instead of running the time MLP every reverse step."""
        if global_conditioning(C) is not None or self.with_geometry_embedding:
            return None
        dev = next(self.parameters()).device
        return self.conditioning_vector(torch.arange(timesteps, device=dev), C, timesteps)

    def encode_local(self, C):
        """encode_c_local(c_local) as a (1, X, Y, Z, dim) NDHWC tensor (None without local
        conditioning).  Independent of x and t: sampling computes it once per trajectory batch
        (the reference recomputes it every step, TODO at ddpm.py:480)."""
        with _lib.conv_impl_scope(self.conv_impl):
            return self._encode_local(C)

    def _encode_local(self, C):
        c_local = local_conditioning(C)
        if c_local is None:
            return None
        cl = ops.to_nvc(c_local[None].float(), self.compute_dtype)
        enc = ops.conv1(cl, self.encode_c_local.weight, self.encode_c_local.bias)
        # The conditioning half of the first U-Net conv does not depend on x or t either: without
        # autograd it is computed here once and the per-step conv continues from it.
        first = self.u_net.downsampling_blocks[0]
        n_lead = self.encode_x.out_channels
        if (CACHE_COND_CONV and not torch.is_grad_enabled() and isinstance(first, ResnetBlock) and first.fused()
                and first.block1.conv.in_channels == n_lead + enc.shape[-1]
                and ops.conv3_partial_supported(enc, first.block1.conv.weight, n_lead)):
            enc.first_conv_partial = (n_lead, ops.conv3_shared_tail(enc, first.block1.conv.weight, n_lead))
        return enc

    # The first U-Net conv reads cat(encode_x(x), encode_c_local(c)) (ddpm.py:495-501): two 1x1 convs of
    # 4 + 4 raw channels followed by a 3x3x3 conv of 64 channels.  Both maps are linear and replicate
    # padding commutes with a per-voxel map, so their composition is ONE 3x3x3 conv of the 8 raw channels
    # with weights W1 . W_enc and bias b1 + sum_taps W1 . b_enc -- 1/4 of the MFMA work of the widest layer
    # of the net in the forward pass (8 raw channels padded to 16), no data gradient at all in the backward
    # pass (x needs none), and a weight gradient on 32 instead of 64 input channels.  The composition is
    # done with autograd-tracked tensor ops on the (tiny) weights, so the gradients of conv.weight,
    # encode_x and encode_c_local follow from the composed conv's weight gradient by the chain rule.
    # each raw-channel group (x, c_local) is padded to 8 channels (16 in total: one K slice of the MFMA conv,
    # a half-filled tile of its weight gradient), or to 16 (32 in total) when c_local itself needs a gradient
    # (learned cell-type embedding): the data gradient w.r.t. the raw input then also runs on the MFMA path

    def compose_first_conv(self, x, c_local):
        """(raw NDHWC input (B, X, Y, Z, 2P), composed weight (Cout, 2P, 3, 3, 3), composed bias, number of raw
        channels that carry data) or None."""
        first = self.u_net.downsampling_blocks[0]
        D = self.encode_x.out_channels
        if not (COMPOSE_FIRST_CONV and isinstance(first, ResnetBlock) and first.fused() and c_local is not None
                and isinstance(first.conv, nn.Identity) and self.in_features <= 8 and self.c_local_features <= 8
                and first.block1.conv.in_channels == D + self.encode_c_local.out_channels):
            return None
        dev, P = x.device, (16 if c_local.requires_grad else 8)
        eye = getattr(self, "_raw_eye", None)
        if eye is None or eye[0].device != dev or eye[0].shape[0] != P:
            ex = torch.zeros(P, self.in_features, 1, 1, 1, device=dev)
            ec = torch.zeros(P, self.c_local_features, 1, 1, 1, device=dev)
            ex[: self.in_features, :, 0, 0, 0] = torch.eye(self.in_features, device=dev)
            ec[: self.c_local_features, :, 0, 0, 0] = torch.eye(self.c_local_features, device=dev)
            Co_ = first.block1.conv.out_channels
            nx, nc = self.in_features, self.c_local_features
            # column gather [x .. | zero pad | c .. | zero pad]: index nx + nc is the appended zero column
            idx = list(range(nx)) + [nx + nc] * (P - nx) + list(range(nx, nx + nc)) + [nx + nc] * (P - nc)
            eye = self._raw_eye = (ex, ec, torch.zeros(P, device=dev), torch.zeros(Co_, 1, 3, 3, 3, device=dev),
                                   torch.tensor(idx, dtype=torch.long, device=dev))
        if not ops.encode_supported(x, c_local, eye[0]):
            return None
        raw = ops.encode(x, c_local, eye[0], eye[2], eye[1], eye[2], self.compute_dtype)  # [x | 0 | c | 0]
        W1, b1 = first.block1.conv.weight, first.block1.conv.bias
        # without autograd (sampling: T forwards on the same weights) the composed weight is kept until one of the six
        # tensors it is made of changes: no einsum / gather / repack launches per reverse step
        frozen = not torch.is_grad_enabled()
        if frozen:
            made_of = (W1, b1, self.encode_x.weight, self.encode_x.bias, self.encode_c_local.weight, self.encode_c_local.bias)
            key = (P,) + tuple((id(p), p._version, p.data_ptr()) for p in made_of)
            hit = getattr(self, "_composed_frozen", None)
            if hit is not None and hit[0] == key:
                return raw, hit[1], hit[2], self.in_features + self.c_local_features
        # one contraction of the full weight with the block-diagonal encoder matrix (few autograd nodes on
        # the 3x3x3 weight), then the raw channels are spread to their padded positions by a constant gather
        w_enc = torch.block_diag(self.encode_x.weight.flatten(1), self.encode_c_local.weight.flatten(1))  # (2D, Fx + Fc)
        b_enc = torch.cat((self.encode_x.bias, self.encode_c_local.bias))
        w8 = torch.einsum("octuv,ck->oktuv", W1, w_enc)
        w_eff = torch.cat((w8, eye[3]), dim=1).index_select(1, eye[4])
        b_eff = b1 + torch.einsum("octuv,c->o", W1, b_enc)
        if frozen:
            self._composed_frozen = (key, w_eff, b_eff)
        return raw, w_eff, b_eff, self.in_features + self.c_local_features

    def forward(self, x: torch.Tensor, t: torch.Tensor, C, encoded_local=None, cond=None):
        """cond: the (B, c_dim) conditioning vectors, if the caller already has them (rows of conditioning_table)."""
        # conv_impl: this model's own choice of the 3x3x3 conv arithmetic (None: the process-wide default)
        with _lib.conv_impl_scope(self.conv_impl):
            return self._forward(x, t, C, encoded_local, cond)

    def _forward(self, x: torch.Tensor, t: torch.Tensor, C, encoded_local=None, cond=None):
        B = x.shape[0]
        c = self.conditioning_vector(t, C, B) if cond is None else cond
        c_local = local_conditioning(C) if self.c_local_features > 0 else None
        first_conv = first_skip = None
        if ops.encode_supported(x, c_local, self.encode_x.weight):
            # both encoders + NCDHW->NDHWC + concat in one kernel
            wc = self.encode_c_local.weight if c_local is not None else None
            bc = self.encode_c_local.bias if c_local is not None else None
            first_conv = self.compose_first_conv(x, c_local)
            if first_conv is not None and DEFER_ENCODE:
                # the encoder output's only reader is then the first block's identity skip: it is evaluated inside that
                # block's tail kernel and the (B, X, Y, Z, 2 dim) tensor never exists
                first_skip = ops.encode_deferred(x, c_local, self.encode_x.weight, self.encode_x.bias, wc, bc,
                                                 self.compute_dtype)
                h = first_skip.standin
            else:
                h = ops.encode(x, c_local, self.encode_x.weight, self.encode_x.bias, wc, bc, self.compute_dtype)
        else:
            h = ops.conv1(ops.to_nvc(x, self.compute_dtype), self.encode_x.weight, self.encode_x.bias)
            e = encoded_local if encoded_local is not None else self.encode_local(C)
            if e is not None:
                h = torch.cat((h, e.expand(B, -1, -1, -1, -1)), dim=-1)
        partial = getattr(encoded_local, "first_conv_partial", None) if not torch.is_grad_enabled() else None
        self.prefetch_weights(skip_first_conv=first_conv is not None)
        films = self.film_table(c)
        h = self.u_net(h, c, first_partial=partial, first_conv=first_conv, films=films, first_skip=first_skip)
        last = self.decode[0]
        if (FUSE_DECODE and isinstance(last, ResnetBlock) and last.fused() and isinstance(last.conv, nn.Identity)
                and self.decode[1].bias is not None and ops.decode_fused_supported(last.dim_out, self.decode[1].weight)):
            # inference: the last block's output exists only inside its tail kernel, which applies the decoder
            return last(h, c, films=films, decode_wb=(self.decode[1].weight, self.decode[1].bias))
        h = last(h, c, films=films)
        if ops.decode_supported(h, self.decode[1].weight):
            return ops.decode(h, self.decode[1].weight, self.decode[1].bias)
        y = ops.conv1(h, self.decode[1].weight, self.decode[1].bias)
        return ops.to_ncv(y, torch.float32)


# --------------------------------------------------------------------------- schedules (API)


def linear_beta_schedule(timesteps):
    return schedules.betas_for("linear", timesteps)


def log_linear_beta_schedule(timesteps):
    return schedules.betas_for("log-linear", timesteps)


def log_snr_linear_beta_schedule(timesteps, snr_1=1e3, snr_T=1e-5):
    assert (snr_1, snr_T) == (1e3, 1e-5), "only the reference's default SNR end points are tabulated"
    return schedules.betas_for("log-snr-linear", timesteps)


def cosine_beta_schedule(timesteps, s=0.008):
    assert s == 0.008
    return schedules.betas_for("cosine", timesteps)


def sigmoid_beta_schedule(timesteps, start=-3, end=3, tau=1, clamp_min=1e-5):
    assert (start, end, tau) == (-3, 3, 1)
    return schedules.betas_for("sigmoid", timesteps)


def normal_kl(mean1, logvar1, mean2, logvar2):
    """KL(N(mean1, e^logvar1) || N(mean2, e^logvar2)) elementwise (reference ddpm.py:597-607)."""
    return 0.5 * (logvar2 - logvar1 - 1.0 + torch.exp(logvar1 - logvar2) + (mean1 - mean2) ** 2 * torch.exp(-logvar2))


def normal_log_lk(x, mean, log_var):
    return -0.5 * (log_var + math.log(2 * math.pi) + (x - mean) ** 2 * torch.exp(-log_var))


def batch_mean(x: torch.Tensor):
    return x.flatten(1).mean(dim=1)


# --------------------------------------------------------------------------- the diffusion


GRAPH_SAMPLER = os.environ.get("TDX_GRAPH_SAMPLER", "1") != "0"
MAX_GRAPH_SAMPLERS = 3  # captured samplers kept per diffusion, least recently used dropped first


class GaussianDiffusion(nn.Module):
    """DDPM training loss and ancestral sampler around a DenoisingModel
    (reference ddpm.py:620-882).

    The q_sample / loss / reverse-step arithmetic runs in fused HIP kernels over a dense
    in-domain mask built once per ``cell_idx`` tensor; the learned-variance + ELBO branch
    (off in the shipped configuration) is evaluated with torch ops on the same tensors.
    """

    def __init__(self, model, *, timesteps: int = 1000, loss_type: str = "l2", beta_schedule: str = "sigmoid",
                 clip_denoised: bool = False, noise_bcs: bool = False, learned_variances: bool = False,
                 elbo_weight: float | None = None, detach_elbo_mean: bool = True):
        super().__init__()
        self.model = model
        self.clip_denoised = clip_denoised
        self.noise_bcs = noise_bcs
        self.learned_variances = learned_variances
        self.elbo_weight = elbo_weight
        self.detach_elbo_mean = detach_elbo_mean
        self.num_timesteps = timesteps
        self.loss_type = loss_type
        if loss_type not in ("l1", "l2"):
            raise ValueError(f"invalid loss type {loss_type}")
        tables = schedules.diffusion_tables(beta_schedule, timesteps)
        for name, tab in tables.items():
            self.register_buffer(name, tab, persistent=False)
        self.register_buffer("step_tables", schedules.pack_step_tables(tables), persistent=False)
        self._mask_cache = None
        self._graph_samplers = None  # signature -> sampling.GraphSampler (built on first use, see graph_samplers)

    # ---- the captured samplers: owned by the diffusion they sample from, so they die with it
    def graph_samplers(self):
        """signature -> GraphSampler, least recently used first.  A plain attribute (not a module / buffer): the
        samplers point back at this object, an ordinary reference cycle the garbage collector frees together with the
        captured graphs, their private pool and the scratch arena once the diffusion is dropped.  (A WeakKeyDictionary
        keyed by the diffusion never let go: its values referenced their own key.)"""
        cache = self.__dict__.get("_graph_samplers")
        if cache is None:
            from collections import OrderedDict

            cache = self.__dict__["_graph_samplers"] = OrderedDict()
        return cache

    def __getstate__(self):
        # copy.deepcopy / pickle: neither captured graphs nor the mask cache travel; the copy builds its own
        state = super().__getstate__()
        state["_graph_samplers"] = None
        state["_mask_cache"] = None
        return state

    # ---- helpers
    def domain_mask(self, cell_idx: torch.Tensor, V: int):
        """(uint8 [V] mask, n_cells) for a flat in-domain cell index list.  Cached for the tensor OBJECT it was
        built from (held alive by the cache, so its address cannot be handed to another geometry's index list)
        at the version it had then (an in-place edit rebuilds the mask)."""
        c = self._mask_cache
        if c is None or c[0] is not cell_idx or c[1] != cell_idx._version or c[2] != V:
            c = self._mask_cache = (cell_idx, cell_idx._version, V, ops.cell_mask(cell_idx, V), int(cell_idx.numel()))
        return c[3], c[4]

    @property
    def loss_fn(self):
        return F.l1_loss if self.loss_type == "l1" else F.mse_loss

    # ---- closed-form pieces (torch; used by the general / learned-variance paths)
    def predict_start_from_noise(self, x_t, t, noise):
        return (broadcast_right(self.sqrt_recip_alphas_cumprod[t], x_t) * x_t
                - broadcast_right(self.sqrt_recipm1_alphas_cumprod[t], x_t) * noise)

    def predict_noise_from_start(self, x_t, t, x0):
        return ((broadcast_right(self.sqrt_recip_alphas_cumprod[t], x_t) * x_t - x0)
                / broadcast_right(self.sqrt_recipm1_alphas_cumprod[t], x_t))

    def q_posterior(self, x_start, x_t, t):
        mean = (broadcast_right(self.posterior_mean_coef1[t], x_t) * x_start
                + broadcast_right(self.posterior_mean_coef2[t], x_t) * x_t)
        return mean, broadcast_right(self.posterior_log_var[t], x_t)

    def q_sample(self, x_start, t, noise):
        return ops.q_sample(x_start, noise, self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod, t)

    def _outside_keep(self, mask, inside_vals, outside_vals):
        m = mask.view(inside_vals.shape[-3:]).bool()
        return torch.where(m, inside_vals, outside_vals)

    def model_predictions(self, x_t, t, C, cell_idx, clip_x_start=False, **model_kwargs):
        out = self.model(x_t, t, C, **model_kwargs)
        if self.learned_variances:
            pred_noise, vw = out.chunk(2, dim=1)
            log_var = torch.lerp(broadcast_right(self.log_betas[t], vw), broadcast_right(self.posterior_log_var[t], vw),
                                 torch.sigmoid(vw))
        else:
            pred_noise, log_var = out, self.log_betas[t]
        x_start = self.predict_start_from_noise(x_t, t, pred_noise)
        if not self.noise_bcs:
            mask, _ = self.domain_mask(cell_idx, x_t[0, 0].numel())
            x_start = self._outside_keep(mask, x_start, x_t)
        if clip_x_start:
            x_start = torch.clamp(x_start, min=-1.0, max=1.0)
        mean, _ = self.q_posterior(x_start, x_t, t)
        return ModelPrediction(noise=pred_noise, x_start=x_start, mean=mean, log_var=log_var)

    @torch.no_grad()
    def p_sample(self, x_t, t: int, C, cell_idx):
        times = torch.full((x_t.shape[0],), t, dtype=torch.long, device=x_t.device)
        pred = self.model_predictions(x_t, times, C, cell_idx, clip_x_start=self.clip_denoised)
        return pred.mean, pred.log_var

    @torch.no_grad()
    def p_sample_loop(self, x_bcs, C, cell_idx, pbar=False, start_from: int | None = None, noise_fn=None, seed=None,
                      trajectory_ids=None):
        """Ancestral sampling (reference ddpm.py:767-816).

        Default (`noise_fn is None`): the hipGraph-captured reverse step of `sampling.GraphSampler`, replayed T times --
        this is what `DiffusionTrainer.sample`, `tools/eval_ckpt.py` and a `dropin` user get.  Noise comes from the
        counter-based generator, one stream per trajectory; the per-call nonce is drawn from torch's global CPU generator
        (so `torch.manual_seed` / `seed_everything` make the samples reproducible, as they do for the reference's
        `torch.randn_like`) unless `seed` is given (reduced mod 2^31 - 1: the nonce shares the 64-bit stream id with the
        trajectory id, seeds that differ by a multiple of 2^31 - 1 give the same noise); `trajectory_ids` = global ids of the batch's trajectories when a larger
        set is sharded over ranks (samples then do not depend on the sharding).  The sampler and its graph are kept per
        input shape and pointed at the new batch / geometry by copies (`GraphSampler.rebind`); a weight update re-captures.
        With `noise_fn(like)` -- injected noise in the reference's drawing order (x_T; then per step t > 0: z, and z' if
        noise_bcs), the golden tests -- or TDX_GRAPH_SAMPLER=0 the loop runs eagerly, one launch sequence per step."""
        if self.learned_variances:
            return self._general_sample(x_bcs, C, cell_idx, pbar, start_from, noise_fn)
        if noise_fn is None and GRAPH_SAMPLER and x_bcs.is_cuda and hasattr(self.model, "encode_local"):
            return self._graph_sample(x_bcs, C, cell_idx, pbar, start_from, seed, trajectory_ids)
        return self._eager_sample(x_bcs, C, cell_idx, pbar, start_from, noise_fn)

    def _graph_sample(self, x_bcs, C, cell_idx, pbar, start_from, seed, trajectory_ids):
        from ..sampling import GraphSampler

        nonce = int(torch.randint(0, 2**31 - 1, (1,)).item()) if seed is None else int(seed) % (2**31 - 1)
        x_bcs = x_bcs.contiguous().float()
        sig = GraphSampler.signature_of(self, x_bcs, C)
        cache = self.graph_samplers()
        gs = cache.get(sig)
        if gs is None:
            # calls that alternate between a few shapes (a partial last batch, B = 1 next to B = 8, two grids) keep one
            # captured sampler each; all of them capture on ONE stream, hence share one scratch arena
            shared = next(iter(cache.values()))._capture_stream if cache else None
            gs = cache[sig] = GraphSampler(self, x_bcs, C, cell_idx, seed=0, trajectory_ids=trajectory_ids, nonce=nonce,
                                           capture_stream=shared)
            while len(cache) > MAX_GRAPH_SAMPLERS:
                cache.popitem(last=False)
        else:
            cache.move_to_end(sig)
            gs.rebind(x_bcs, C, cell_idx, nonce=nonce,
                      trajectory_ids=list(range(x_bcs.shape[0])) if trajectory_ids is None else trajectory_ids)
        return gs.sample(start_from, pbar=pbar)

    def _general_sample(self, x_bcs, C, cell_idx, pbar, start_from, noise_fn):
        """The reference's loop step by step over `p_sample` (ddpm.py:767-816) for what the fused update kernel does not
        cover: learned variances, where `log_var` is the per-voxel lerp between log beta_t and the posterior log-variance
        (ddpm.py:732-741).  The unmodified reference cannot finish this loop -- ddpm.py:805 hands the 5-D std to
        `broadcast_right`, which asks for reshape(-1, -1, -1, -1, -1) (utils.py:11, RuntimeError; recorded in
        tests/golden/options.npz) -- so this is the arithmetic the loop spells out, with the std used as it is: the
        U-Net forward runs on the HIP kernels, the elementwise rest on torch ops over the same tensors."""
        randn = noise_fn if noise_fn is not None else torch.randn_like
        x_bcs = x_bcs.contiguous().float()
        B = x_bcs.shape[0]
        inside = self.domain_mask(cell_idx, x_bcs[0, 0].numel())[0].view(x_bcs.shape[-3:]).bool()
        times = lambda t: torch.full((B,), t, dtype=torch.long, device=x_bcs.device)
        if start_from is None:
            x_t, T = randn(x_bcs), self.num_timesteps
        else:
            x_t, T = self.q_sample(x_bcs, times(start_from - 1), randn(x_bcs)), start_from
        if not self.noise_bcs:
            x_t = torch.where(inside, x_t, x_bcs)
        steps = reversed(range(T))
        if pbar:
            from tqdm.auto import tqdm

            steps = tqdm(steps, desc="sampling loop time step", total=T, position=1)
        for t in steps:
            mean, log_var = self.p_sample(x_t, t, C, cell_idx)
            if t == 0:
                x_t = mean
                continue
            noise = randn(x_t)
            if not self.noise_bcs:
                noise = torch.where(inside, noise, torch.zeros_like(noise))
            std = (log_var / 2).exp()
            x_t = mean + (std if std.ndim == noise.ndim else broadcast_right(std, noise)) * noise
            if self.noise_bcs:
                x_t = torch.where(inside, x_t, self.q_sample(x_bcs, times(t), randn(x_bcs)))
        return torch.where(inside, x_t, x_bcs)

    def _eager_sample(self, x_bcs, C, cell_idx, pbar, start_from, noise_fn):
        randn = noise_fn if noise_fn is not None else torch.randn_like
        x_bcs = x_bcs.contiguous().float()
        B, Fd = x_bcs.shape[:2]
        V = x_bcs[0, 0].numel()
        dev = x_bcs.device
        mask, _ = self.domain_mask(cell_idx, V)
        ts = torch.arange(self.num_timesteps, dtype=torch.long, device=dev)
        if start_from is None:
            x_t, T = randn(x_bcs), self.num_timesteps
        else:
            x_t = ops.q_sample(x_bcs, randn(x_bcs), self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod,
                               ts[start_from - 1 : start_from])
            T = start_from
        if not self.noise_bcs:
            x_t = self._outside_keep(mask, x_t, x_bcs)
        enc = self.model.encode_local(C) if hasattr(self.model, "encode_local") else None
        kw = {"encoded_local": enc} if enc is not None else {}
        steps = reversed(range(T))
        if pbar:
            from tqdm.auto import tqdm

            steps = tqdm(steps, desc="sampling loop time step", total=T, position=1)
        for t in steps:
            eps = self.model(x_t, ts[t].expand(B), C, **kw)
            z = randn(x_t) if t > 0 else None
            z2 = randn(x_bcs) if (t > 0 and self.noise_bcs) else None
            x_t = ops.p_sample_step(x_t, eps, z, z2, x_bcs, mask, self.step_tables, self.num_timesteps, ts[t : t + 1],
                                    self.noise_bcs, self.clip_denoised)
        return x_t

    def p_losses(self, x_start, t, C, metadata, variables, noise=None):
        x_start = x_start.contiguous().float()
        cell_idx = metadata.cell_idx
        dm = getattr(metadata, "domain_mask", None)
        if dm is not None:
            # (uint8 [V] mask, int64 device scalar n_cells) in buffers the caller owns: training.GraphedTrainingStep
            # replays one captured step for every geometry of a grid size by copying into them
            mask, n_cells = dm
            assert not (self.learned_variances and self.elbo_weight is not None), "the ELBO term gathers by cell_idx"
        else:
            mask, n_cells = self.domain_mask(cell_idx, x_start[0, 0].numel())
        if noise is None:
            noise = torch.randn_like(x_start)
        x_t = ops.q_sample(x_start, noise, self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod, t,
                           mask=mask, keep_bcs=not self.noise_bcs)
        if not (self.learned_variances and self.elbo_weight is not None):
            out = self.model(x_t, t, C)
            pred_noise = out.chunk(2, dim=1)[0].contiguous() if self.learned_variances else out
            return ops.masked_loss(pred_noise, noise, mask, n_cells, l1=self.loss_type == "l1"), t
        # learned variances + ELBO term (reference ddpm.py:853-870), torch ops
        pred = self.model_predictions(x_t, t, C, cell_idx, clip_x_start=self.clip_denoised)
        loss = ops.masked_loss(pred.noise.contiguous(), noise, mask, n_cells, l1=self.loss_type == "l1")
        true_mean, true_log_var = self.q_posterior(x_start, x_t, t)
        model_mean = pred.mean.detach() if self.detach_elbo_mean else pred.mean
        sel = lambda v: v.flatten(-3)[..., cell_idx]
        kl = sel(normal_kl(true_mean, true_log_var, model_mean, pred.log_var))
        ll = sel(normal_log_lk(x_t, model_mean, pred.log_var))
        elbo = torch.where(t == 0, -batch_mean(ll), batch_mean(kl))
        return loss + self.elbo_weight * elbo.mean(), t

    def forward(self, x, *args, **kwargs):
        t = torch.randint(0, self.num_timesteps, (x.shape[0],), device=x.device, dtype=torch.long)
        return self.p_losses(x, t, *args, **kwargs)


# --------------------------------------------------------------------------- rarely used surface
# Symbols of the reference's ddpm.py that its shipped configuration never instantiates
# (SURVEY.md §2b).  They are provided on stock torch ops over NCDHW tensors so that code importing
# them keeps working; none of them is on the accelerated path.


def pad_to_multiple_of(x: torch.Tensor, n: int, *, mode: str):
    """Pad the three trailing dims up to the next multiple of n (reference ddpm.py:51-63; like the
    reference, a dimension that already is a multiple still receives a full extra block)."""
    pads = [n - s % n for s in x.shape[-3:]]
    if min(pads) <= 0:
        return x, (0, 0, 0)
    h, w, d = pads
    return F.pad(x, (0, d, 0, w, 0, h), mode=mode), (h, w, d)


def unpad(x: torch.Tensor, padding):
    if min(padding) <= 0:
        return x
    h, w, d = padding
    return x[..., :-h, :-w, :-d]


def expand_as(x: torch.Tensor, y: torch.Tensor, dim: int):
    """Broadcast y to x's shape everywhere except along `dim` (reference ddpm.py:314-323)."""
    assert x.ndim >= y.ndim
    target = list(x.shape)
    target[dim] = -1
    return y.expand(target)


class LinearAttention(nn.Module):
    """Efficient attention with linear complexity (Shen et al.); reference ddpm.py:200-229."""

    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        self.heads = heads
        hidden = dim_head * heads
        self.to_qkv = nn.Conv3d(dim, 3 * hidden, 1, bias=False)
        self.combine_heads = nn.Sequential(nn.Conv3d(hidden, dim, 1))

    def forward(self, x):
        b, _, X, Y, Z = x.shape
        qkv = self.to_qkv(x).reshape(b, 3, self.heads, -1, X * Y * Z)
        q, k, v = qkv[:, 0].softmax(dim=-2), qkv[:, 1].softmax(dim=-1), qkv[:, 2]
        context = torch.einsum("bhci,bhdi->bhcd", k, v)           # (c, d) summary of keys x values
        out = torch.einsum("bhcd,bhck->bhdk", context, q)          # apply to every query position
        return self.combine_heads(out.reshape(b, -1, X, Y, Z))


class LocalAttention(nn.Module):
    """Windowed self-attention over non-overlapping w^3 blocks (reference ddpm.py:232-283)."""

    def __init__(self, dim: int, window_size: int, heads: int = 4, dim_head: int = 32):
        super().__init__()
        self.dim, self.window_size, self.heads, self.dim_head = dim, window_size, heads, dim_head
        hidden = dim_head * heads
        self.to_qkv = nn.Conv3d(dim, hidden * 3, 1, bias=False)
        self.merge_heads = nn.Sequential(nn.Conv3d(hidden, dim, 1))

    def forward(self, x):
        from .attention import fused_attention

        w = self.window_size
        qkv = self.to_qkv(x)
        padded = any(s % w != 0 for s in qkv.shape[-3:])
        if padded:
            qkv, padding = pad_to_multiple_of(qkv, w, mode="constant")
        b, _, X, Y, Z = qkv.shape
        nx, ny, nz = X // w, Y // w, Z // w
        t = qkv.reshape(b, 3, self.heads, self.dim_head, nx, w, ny, w, nz, w)
        t = t.permute(1, 0, 4, 6, 8, 2, 5, 7, 9, 3).reshape(3, b * nx * ny * nz, self.heads, w**3, self.dim_head)
        out = fused_attention(t[0].contiguous(), t[1].contiguous(), t[2].contiguous())
        out = out.reshape(b, nx, ny, nz, self.heads, w, w, w, self.dim_head)
        out = out.permute(0, 4, 8, 1, 5, 2, 6, 3, 7).reshape(b, self.heads * self.dim_head, X, Y, Z)
        if padded:
            out = unpad(out, padding)
        return self.merge_heads(out)
