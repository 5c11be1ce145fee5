"""Network blocks with the module API of the reference's models/blocks.py, on HIP kernels.

Same class names, constructor arguments, attribute names, parameter / state_dict keys and
``forward`` signatures as the reference (models/blocks.py:36-755), so that
``models.architectures`` / ``utils.trainer_*`` code and reference checkpoints work unchanged.
What differs is the implementation: ``KPConv.forward`` never materialises the
[N,H,3] / [N,H,K,3] / [N,H,K] / [N,H,Ci] tensors of blocks.py:278-363 -- the gather, the
kernel-point influence and the feature aggregate run fused in one HIP kernel
(weasal_amd/csrc/kpconv.hip); the remaining dense contraction [N, K*Ci] x [K*Ci, Co] and the unary
MLPs run on the f32 MFMA (weasal_amd/csrc/gemm.hip) with the bias / residual / LeakyReLU that follow
them in the blocks applied in the GEMM epilogue.

Quirks of the reference that are preserved on purpose:
  * BatchNormBlock is an identity for 2-D inputs when use_bn is set (blocks.py:454-463); the
    BatchNorm1d parameters exist in the state_dict but never run;
  * shadow neighbours (index == number of supports) are the point (1e6,1e6,1e6) with a zero
    feature row (blocks.py:278,357); max_pool includes that zero row (blocks.py:104).
"""
import math
import os

import torch
import torch.nn as nn
from torch.nn.init import kaiming_uniform_
from torch.nn.parameter import Parameter

from . import fused, ops
from .kernel_points import load_kernels

DEFORM_FAST_PATH = os.environ.get("WEASAL_DEFORM_FAST", "1") != "0"      # A/B switch (diagnostics, tests): 0 = generic kernels


# ---------------------------------------------------------------------------------------------
# free functions (blocks.py:36-134)
# ---------------------------------------------------------------------------------------------
def gather(x, idx, method=2):
    """x[idx] for idx of any shape (blocks.py:36-67; the three `method`s there only differ in how
    autograd scatters the gradient back -- all are x[idx])."""
    if method not in (0, 1, 2):
        raise ValueError('Unkown method')
    return x[idx]


def radius_gaussian(sq_r, sig, eps=1e-9):
    """exp(-sq_r / (2 sig^2 + eps))  (blocks.py:70-77)"""
    return torch.exp(-sq_r / (2 * sig ** 2 + eps))


def closest_pool(x, inds):
    """features of the closest (first-column) neighbour, zero row for shadow (blocks.py:80-92)"""
    return ops.closest_pool(x, inds)


def max_pool(x, inds):
    """max over the neighbourhood, the zero shadow row takes part (blocks.py:95-111)"""
    return ops.max_pool(x, inds)


def _layer_lengths(batch, layer_ind):
    """sphere sizes of a level for HOST-side loops: the host copy the pyramid builder kept (PyramidBatch.lengths_host) when
    there is one -- reading the device vector back would wait for everything queued on the stream, once per block"""
    lh = getattr(batch, "lengths_host", None)
    return lh[layer_ind] if lh is not None else batch.lengths[layer_ind]


def global_average(x, batch_lengths):
    """per-cloud mean of the stacked features (blocks.py:114-134)"""
    lengths = [int(v) for v in batch_lengths]
    return torch.stack([chunk.mean(dim=0) for chunk in torch.split(x, lengths, dim=0)])


# ---------------------------------------------------------------------------------------------
# KPConv (blocks.py:144-379)
# ---------------------------------------------------------------------------------------------
_MATMUL_EPILOGUE = ops.matmul_epilogue      # (oracle.kpconv_ref.cpu_reference_mode swaps ops.*: no links then)

class KPConv(nn.Module):

    def __init__(self, kernel_size, p_dim, in_channels, out_channels, KP_extent, radius,
                 fixed_kernel_points='center', KP_influence='linear', aggregation_mode='sum',
                 deformable=False, modulated=False):
        super(KPConv, self).__init__()
        self.K = kernel_size
        self.p_dim = p_dim
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.radius = radius
        self.KP_extent = KP_extent
        self.fixed_kernel_points = fixed_kernel_points
        self.KP_influence = KP_influence
        self.aggregation_mode = aggregation_mode
        self.deformable = deformable
        self.modulated = modulated

        # read by architectures.p2p_fitting_regularizer (architectures.py:31-48)
        self.min_d2 = None
        self.deformed_KP = None
        self.offset_features = None

        self.weights = Parameter(torch.zeros((self.K, in_channels, out_channels), dtype=torch.float32),
                                 requires_grad=True)
        if deformable:
            self.offset_dim = (self.p_dim + 1) * self.K if modulated else self.p_dim * self.K
            self.offset_conv = KPConv(self.K, self.p_dim, self.in_channels, self.offset_dim, KP_extent, radius,
                                      fixed_kernel_points=fixed_kernel_points, KP_influence=KP_influence,
                                      aggregation_mode=aggregation_mode)
            self.offset_bias = Parameter(torch.zeros(self.offset_dim, dtype=torch.float32), requires_grad=True)
        else:
            self.offset_dim = None
            self.offset_conv = None
            self.offset_bias = None

        self.reset_parameters()
        self.kernel_points = self.init_KP()

    def reset_parameters(self):
        kaiming_uniform_(self.weights, a=math.sqrt(5))
        if self.deformable:
            nn.init.zeros_(self.offset_bias)

    def init_KP(self):
        """frozen kernel-point parameter (blocks.py:223-236)"""
        k_points = load_kernels(self.radius, self.K, dimension=self.p_dim, fixed=self.fixed_kernel_points)
        return Parameter(torch.tensor(k_points, dtype=torch.float32), requires_grad=False)

    def forward(self, q_pts, s_pts, neighb_inds, x, _bias=None, _slope=None, _out_f32=False, _rows_sorted=None):
        """`_bias` / `_slope` (used by the blocks of this module only): the BatchNormBlock bias and the
        LeakyReLU that follow the convolution, applied in the epilogue of the contraction GEMM.
        bf16 feature rows (x.dtype bfloat16, BASELINE config 5) run the bf16-row kernels; `_out_f32` keeps the
        output in f32 (the offsets of a deformable convolution: geometry stays f32)."""
        if self.KP_influence not in ops.INFLUENCE:
            raise ValueError('Unknown influence function type (config.KP_influence)')
        if self.aggregation_mode not in ops.AGGREGATION:
            raise ValueError("Unknown convolution mode. Should be 'closest' or 'sum'")

        # index rows known to be sorted by distance (a matrix of the batch's pyramid): the linear-influence kernels stop at
        # the reach of the kernel points
        rows_sorted = (x.is_cuda and ops.rows_cutoff_pays(neighb_inds, self.radius)) if _rows_sorted is None else _rows_sorted
        deformed = None
        modulations = None
        if self.deformable and DEFORM_FAST_PATH and ops.deform_fast_path_ok(x, self.K, self.KP_influence, self.aggregation_mode):
            # BASELINE config 5's mode (linear influence, sum): offsets, modulations and kernel points in one pass
            # (ws_kpconv_deform_prepare), per-query kernel points packed for the gather kernels; the offset bias rides on
            # the epilogue of the offset convolution's contraction
            self.offset_features = self.offset_conv(q_pts, s_pts, neighb_inds, x, _bias=self.offset_bias, _out_f32=True,
                                                    _rows_sorted=rows_sorted)
            kp4, self.deformed_KP, _, rmax = ops.deform_prepare(self.offset_features, self.kernel_points, self.KP_extent,
                                                                self.modulated)
            wf, self.min_d2 = ops.kpconv_gather_def(x, kp4, q_pts, s_pts, neighb_inds, self.KP_extent, kp_rmax=rmax,
                                                    rows_sorted=rows_sorted)
            return ops.matmul_epilogue(wf.reshape(wf.shape[0], -1),
                                       self.weights.reshape(self.K * self.in_channels, self.out_channels),
                                       bias=_bias, slope=_slope, out_f32=_out_f32)
        if self.deformable:
            # offsets from a rigid KPConv on the same neighbourhood (blocks.py:244-267)
            self.offset_features = self.offset_conv(q_pts, s_pts, neighb_inds, x, _out_f32=True) + self.offset_bias
            nkp = self.p_dim * self.K
            if self.modulated:
                unscaled = self.offset_features[:, :nkp].reshape(-1, self.K, self.p_dim)
                modulations = 2 * torch.sigmoid(self.offset_features[:, nkp:])
            else:
                unscaled = self.offset_features.reshape(-1, self.K, self.p_dim)
            self.deformed_KP = unscaled * self.KP_extent + self.kernel_points      # blocks.py:267,288
            deformed = self.deformed_KP

        kw = {"rows_sorted": True} if (rows_sorted and not self.deformable) else {}      # (the oracle's stand-in has no such argument)
        wf, min_d2 = ops.kpconv_gather(x, q_pts, s_pts, neighb_inds, self.kernel_points, self.KP_extent,
                                       influence=self.KP_influence, aggregation=self.aggregation_mode,
                                       deformed_kp=deformed, modulations=modulations,
                                       want_min_d2=self.deformable, **kw)
        if self.deformable:
            self.min_d2 = min_d2                                                     # blocks.py:304
        # dense contraction over (kernel point, input channel): blocks.py:370-374
        if wf.dtype == torch.bfloat16:
            return ops.matmul_epilogue(wf.reshape(wf.shape[0], -1),
                                       self.weights.reshape(self.K * self.in_channels, self.out_channels),
                                       bias=_bias, slope=_slope, out_f32=_out_f32)
        return ops.matmul_epilogue(wf.reshape(wf.shape[0], -1),
                                   self.weights.reshape(self.K * self.in_channels, self.out_channels),
                                   bias=_bias, slope=_slope)

    def __repr__(self):
        return 'KPConv(radius: {:.2f}, in_feat: {:d}, out_feat: {:d})'.format(self.radius, self.in_channels,
                                                                              self.out_channels)


# ---------------------------------------------------------------------------------------------
# blocks (blocks.py:387-755)
# ---------------------------------------------------------------------------------------------
_SIMPLE = ('simple', 'simple_deformable', 'simple_invariant', 'simple_equivariant', 'simple_strided',
           'simple_deformable_strided', 'simple_invariant_strided', 'simple_equivariant_strided')
_RESNETB = ('resnetb', 'resnetb_invariant', 'resnetb_equivariant', 'resnetb_deformable', 'resnetb_strided',
            'resnetb_deformable_strided', 'resnetb_equivariant_strided', 'resnetb_invariant_strided')


def block_decider(block_name, radius, in_dim, out_dim, layer_ind, config):
    if block_name == 'unary':
        return UnaryBlock(in_dim, out_dim, config.use_batch_norm, config.batch_norm_momentum)
    if block_name in _SIMPLE:
        return SimpleBlock(block_name, in_dim, out_dim, radius, layer_ind, config)
    if block_name in _RESNETB:
        return ResnetBottleneckBlock(block_name, in_dim, out_dim, radius, layer_ind, config)
    if block_name in ('max_pool', 'max_pool_wide'):
        return MaxPoolBlock(layer_ind)
    if block_name == 'global_average':
        return GlobalAverageBlock()
    if block_name == 'nearest_upsample':
        return NearestUpsampleBlock(layer_ind)
    raise ValueError('Unknown block name in the architecture definition : ' + block_name)


class BatchNormBlock(nn.Module):
    """BatchNorm1d holder that is an identity on [N,C] inputs when use_bn (blocks.py:453-463),
    a learned bias otherwise (:465)."""

    def __init__(self, in_dim, use_bn, bn_momentum):
        super(BatchNormBlock, self).__init__()
        self.bn_momentum = bn_momentum
        self.use_bn = use_bn
        self.in_dim = in_dim
        if self.use_bn:
            self.batch_norm = nn.BatchNorm1d(in_dim, momentum=bn_momentum)
        else:
            self.bias = Parameter(torch.zeros(in_dim, dtype=torch.float32), requires_grad=True)

    def reset_parameters(self):
        nn.init.zeros_(self.bias)

    def epilogue_bias(self):
        """what this block adds to a 2-D input: nothing with use_bn (identity), the bias otherwise"""
        return None if self.use_bn else self.bias

    def forward(self, x):
        if not self.use_bn:
            return x + self.bias
        if x.dim() < 3:
            return x
        # 3-D inputs never occur on this path; kept for interface parity (blocks.py:458-462)
        y = self.batch_norm(x.unsqueeze(2).transpose(0, 2))
        return y.transpose(0, 2)

    def __repr__(self):
        return 'BatchNormBlock(in_feat: {:d}, momentum: {:.3f}, only_bias: {:s})'.format(
            self.in_dim, self.bn_momentum, str(not self.use_bn))


class UnaryBlock(nn.Module):
    """Linear(no bias) -> BatchNormBlock -> LeakyReLU(0.1) unless no_relu (blocks.py:473-507)"""

    def __init__(self, in_dim, out_dim, use_bn, bn_momentum, no_relu=False):
        super(UnaryBlock, self).__init__()
        self.bn_momentum = bn_momentum
        self.use_bn = use_bn
        self.no_relu = no_relu
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.mlp = nn.Linear(in_dim, out_dim, bias=False)
        self.batch_norm = BatchNormBlock(out_dim, self.use_bn, self.bn_momentum)
        if not no_relu:
            self.leaky_relu = nn.LeakyReLU(0.1)
        self.out_f32 = False     # bf16 feature rows only: keep this block's output in f32 (the logits)

    def forward(self, x, batch=None):
        return self.forward_fused(x, batch=batch)

    def forward_fused(self, x, residual=None, slope_override=None, batch=None):
        """mlp -> batch_norm(identity | + bias) [-> + residual] -> LeakyReLU, in one GEMM.
        slope_override: activation applied although the block itself has no_relu (the residual sum of
        ResnetBottleneckBlock, blocks.py:709)."""
        slope = slope_override if slope_override is not None else (None if self.no_relu else 0.1)
        if x.dtype == torch.bfloat16:
            return ops.matmul_epilogue(x, self.mlp.weight.t(), bias=self.batch_norm.epilogue_bias(), residual=residual,
                                       slope=slope, out_f32=self.out_f32)
        # gate links (fused.GateLink) when the caller has set them up on the batch (KPFCNN's head): float32 kernel path only
        links = None
        linked = batch is not None and residual is None and (getattr(batch, "gate_link_in", None) is not None
                                                             or getattr(batch, "gate_link_out", None) is not None)
        if linked and ops.matmul_epilogue is _MATMUL_EPILOGUE and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and ops.FUSED_EPILOGUE and x.shape[0] >= ops.GEMM_MIN_ROWS:
            from . import fused
            links = fused.linear_links(batch, x, True)
        if links is None:
            return ops.matmul_epilogue(x, self.mlp.weight.t(), bias=self.batch_norm.epilogue_bias(), residual=residual, slope=slope)
        out = ops.matmul_epilogue(x, self.mlp.weight.t(), bias=self.batch_norm.epilogue_bias(), residual=residual,
                                  slope=slope, links=links)
        fused.linear_links_done(batch, links, out, slope is not None)
        return out

    def __repr__(self):
        return 'UnaryBlock(in_feat: {:d}, out_feat: {:d}, BN: {:s}, ReLU: {:s})'.format(
            self.in_dim, self.out_dim, str(self.use_bn), str(not self.no_relu))


def _layer_geometry(block_name, layer_ind, batch):
    """(q_pts, s_pts, neighb_inds) of a block: pooled queries for strided blocks (blocks.py:554-561)"""
    if 'strided' in block_name:
        return batch.points[layer_ind + 1], batch.points[layer_ind], batch.pools[layer_ind]
    return batch.points[layer_ind], batch.points[layer_ind], batch.neighbors[layer_ind]


def _make_kpconv(block_name, in_dim, out_dim, radius, config):
    extent = radius * config.KP_extent / config.conv_radius                       # blocks.py:523
    return KPConv(config.num_kernel_points, config.in_points_dim, in_dim, out_dim, extent, radius,
                  fixed_kernel_points=config.fixed_kernel_points, KP_influence=config.KP_influence,
                  aggregation_mode=config.aggregation_mode, deformable='deform' in block_name,
                  modulated=config.modulated)


class SimpleBlock(nn.Module):
    """KPConv(in -> out//2) -> BN -> LeakyReLU (blocks.py:510-564)"""

    def __init__(self, block_name, in_dim, out_dim, radius, layer_ind, config):
        super(SimpleBlock, self).__init__()
        self.bn_momentum = config.batch_norm_momentum
        self.use_bn = config.use_batch_norm
        self.layer_ind = layer_ind
        self.block_name = block_name
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.KPConv = _make_kpconv(block_name, in_dim, out_dim // 2, radius, config)
        self.batch_norm = BatchNormBlock(out_dim // 2, self.use_bn, self.bn_momentum)
        self.leaky_relu = nn.LeakyReLU(0.1)

    def forward(self, x, batch):
        q_pts, s_pts, inds = _layer_geometry(self.block_name, self.layer_ind, batch)
        if fused.kpblock_eligible(self, x):
            return fused.simple_block(self, x, batch, q_pts, s_pts, inds)      # one C call each way (csrc/blocks.hip)
        return self.KPConv(q_pts, s_pts, inds, x, _bias=self.batch_norm.epilogue_bias(), _slope=0.1)


class SimpleBlock2(nn.Module):
    """SimpleBlock with the full out_dim (blocks.py:567-622; used by the attention blocks)"""

    def __init__(self, block_name, in_dim, out_dim, radius, layer_ind, config):
        super(SimpleBlock2, self).__init__()
        self.bn_momentum = config.batch_norm_momentum
        self.use_bn = config.use_batch_norm
        self.layer_ind = layer_ind
        self.block_name = block_name
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.KPConv = _make_kpconv(block_name, in_dim, out_dim, radius, config)
        self.batch_norm = BatchNormBlock(out_dim, self.use_bn, self.bn_momentum)
        self.leaky_relu = nn.LeakyReLU(0.1)

    def forward(self, x, batch):
        q_pts, s_pts, inds = _layer_geometry(self.block_name, self.layer_ind, batch)
        if fused.kpblock_eligible(self, x):
            return fused.simple_block(self, x, batch, q_pts, s_pts, inds)      # one C call each way (csrc/blocks.hip)
        return self.KPConv(q_pts, s_pts, inds, x, _bias=self.batch_norm.epilogue_bias(), _slope=0.1)


class ResnetBottleneckBlock(nn.Module):
    """unary(in -> out/4) -> KPConv(out/4 -> out/4) -> unary(out/4 -> out) + shortcut
    (max-pooled when strided, projected when in != out), LeakyReLU (blocks.py:624-709)"""

    def __init__(self, block_name, in_dim, out_dim, radius, layer_ind, config):
        super(ResnetBottleneckBlock, self).__init__()
        self.bn_momentum = config.batch_norm_momentum
        self.use_bn = config.use_batch_norm
        self.block_name = block_name
        self.layer_ind = layer_ind
        self.in_dim = in_dim
        self.out_dim = out_dim
        mid = out_dim // 4
        self.unary1 = UnaryBlock(in_dim, mid, self.use_bn, self.bn_momentum) if in_dim != mid else nn.Identity()
        self.KPConv = _make_kpconv(block_name, mid, mid, radius, config)
        self.batch_norm_conv = BatchNormBlock(mid, self.use_bn, self.bn_momentum)
        self.unary2 = UnaryBlock(mid, out_dim, self.use_bn, self.bn_momentum, no_relu=True)
        if in_dim != out_dim:
            self.unary_shortcut = UnaryBlock(in_dim, out_dim, self.use_bn, self.bn_momentum, no_relu=True)
        else:
            self.unary_shortcut = nn.Identity()
        self.leaky_relu = nn.LeakyReLU(0.1)

    def forward(self, features, batch):
        q_pts, s_pts, inds = _layer_geometry(self.block_name, self.layer_ind, batch)
        if fused.kpblock_eligible(self, features):
            return fused.resnetb_block(self, features, batch, q_pts, s_pts, inds)   # one C call each way (csrc/blocks.hip)
        x = self.unary1(features)
        x = self.KPConv(q_pts, s_pts, inds, x, _bias=self.batch_norm_conv.epilogue_bias(), _slope=0.1)
        shortcut = max_pool(features, inds) if 'strided' in self.block_name else features
        # leaky_relu(unary2(x) + unary_shortcut(shortcut)) with the sum and the activation in unary2's GEMM
        return self.unary2.forward_fused(x, residual=self.unary_shortcut(shortcut), slope_override=0.1)


class GlobalAverageBlock(nn.Module):

    def __init__(self):
        super(GlobalAverageBlock, self).__init__()

    def forward(self, x, batch):
        return global_average(x, _layer_lengths(batch, -1))


class NearestUpsampleBlock(nn.Module):

    def __init__(self, layer_ind):
        super(NearestUpsampleBlock, self).__init__()
        self.layer_ind = layer_ind
        self.name_block = 'upsample'

    def forward(self, x, batch):
        return closest_pool(x, batch.upsamples[self.layer_ind - 1])

    def __repr__(self):
        return 'NearestUpsampleBlock(layer: {:d} -> {:d})'.format(self.layer_ind, self.layer_ind - 1)


class MaxPoolBlock(nn.Module):

    def __init__(self, layer_ind):
        super(MaxPoolBlock, self).__init__()
        self.layer_ind = layer_ind

    def forward(self, x, batch):
        return max_pool(x, batch.pools[self.layer_ind + 1])


# ----------------------------------------------------------------------------------------------------------------------
# Attention blocks of the weak-label network KPFCNN_mprm (models/blocks.py:758-1011).
# Per input sphere a dense softmax attention (points x points, or channels x channels) built from UnaryBlock
# projections; same module names / state_dict keys as the reference.  Device agnostic: the reference's `.cuda()`
# calls on fresh tensors (blocks.py:796,863,989) are replaced by "same device as the features", and its repeated
# torch.cat inside the per-sphere loop by one cat at the end (same values).
# ----------------------------------------------------------------------------------------------------------------------
def _per_cloud(lengths):
    """[(start, end)] of the stacked spheres; `lengths` is the host or device length vector of the layer"""
    ls = [int(v) for v in (lengths.tolist() if hasattr(lengths, "tolist") else lengths)]
    out, s = [], 0
    for n in ls:
        out.append((s, s + n))
        s += n
    return out


class spatial_att(nn.Module):
    """point-to-point attention inside each sphere (blocks.py:758-822): merged = simple2(gamma * softmax(QK^T)V + f),
    and the attention output divided by the sphere's point count (input of the point-wise path)"""

    def __init__(self, block_name, in_dim, out_dim, radius, layer_ind, config):
        super(spatial_att, self).__init__()
        self.bn_momentum = config.batch_norm_momentum
        self.use_bn = config.use_batch_norm
        self.layer_ind = layer_ind
        self.block_name = block_name
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.simple1 = SimpleBlock2(block_name, in_dim, out_dim, radius, layer_ind, config)
        self.unary1 = UnaryBlock(out_dim, out_dim // 8, self.use_bn, self.bn_momentum)
        self.unary2 = UnaryBlock(out_dim, out_dim // 8, self.use_bn, self.bn_momentum)
        self.unary3 = UnaryBlock(out_dim, out_dim, self.use_bn, self.bn_momentum)
        self.gamma = nn.Parameter(torch.zeros(1))
        self.softmax = nn.Softmax(dim=-1)
        self.simple2 = SimpleBlock2(block_name, in_dim, out_dim, radius, layer_ind, config)

    def forward(self, features, batch):
        features = self.simple1(features, batch)
        q, k, v = self.unary1(features), self.unary2(features), self.unary3(features)
        outs, outs_n = [], []
        for a, b in _per_cloud(_layer_lengths(batch, self.layer_ind)):
            att = torch.matmul(self.softmax(torch.matmul(q[a:b], k[a:b].T)), v[a:b])
            outs.append(att)
            outs_n.append(att / float(b - a))
        x = torch.cat(outs, 0) if outs else features.new_zeros((0, features.shape[1]))
        xn = torch.cat(outs_n, 0) if outs_n else features.new_zeros((0, features.shape[1]))
        merged = self.simple2(self.gamma * x + features, batch)
        return merged, xn


class channel_att(nn.Module):
    """channel-to-channel attention inside each sphere (blocks.py:824-883)"""

    def __init__(self, block_name, in_dim, out_dim, radius, layer_ind, config):
        super(channel_att, self).__init__()
        self.bn_momentum = config.batch_norm_momentum
        self.use_bn = config.use_batch_norm
        self.layer_ind = layer_ind
        self.block_name = block_name
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.simple1 = SimpleBlock2(block_name, in_dim, out_dim // 8, radius, layer_ind, config)
        self.unary1 = UnaryBlock(out_dim // 8, out_dim // 8, self.use_bn, self.bn_momentum)
        self.unary2 = UnaryBlock(out_dim // 8, out_dim // 8, self.use_bn, self.bn_momentum)
        self.gamma = nn.Parameter(torch.zeros(1))
        self.softmax = nn.Softmax(dim=-1)
        self.simple2 = SimpleBlock2(block_name, out_dim // 8, out_dim, radius, layer_ind, config)

    def forward(self, features, batch):
        features = self.simple1(features, batch)
        x1, x2 = self.unary1(features), self.unary2(features)
        outs = []
        for a, b in _per_cloud(_layer_lengths(batch, self.layer_ind)):
            energy = torch.matmul(x1[a:b].T, x2[a:b])
            energy_new = torch.max(energy, -1, keepdim=True)[0].expand_as(energy) - energy
            outs.append(torch.matmul(features[a:b], self.softmax(energy_new)))
        x = torch.cat(outs, 0) if outs else features.new_zeros((0, features.shape[1]))
        return self.simple2(self.gamma * x + features, batch)


class multi_path_att(nn.Module):
    """no-attention, spatial, channel and point-wise paths, each projected to the classes (blocks.py:885-928)"""

    def __init__(self, block_name, in_dim, out_dim, radius, layer_ind, config):
        super(multi_path_att, self).__init__()
        self.bn_momentum = config.batch_norm_momentum
        self.use_bn = config.use_batch_norm
        self.layer_ind = layer_ind
        self.block_name = block_name
        self.in_dim = in_dim
        self.out_dim = out_dim
        fdim = config.num_classes
        self.sa_f = spatial_att(block_name, in_dim, out_dim, radius, layer_ind, config)
        self.ca_f = channel_att(block_name, in_dim, out_dim, radius, layer_ind, config)
        self.simple1 = SimpleBlock2(block_name, in_dim + out_dim, out_dim, radius, layer_ind, config)
        self.sa_unary = UnaryBlock(out_dim, fdim, self.use_bn, self.bn_momentum)
        self.ca_unary = UnaryBlock(out_dim, fdim, self.use_bn, self.bn_momentum)
        self.no_unary = UnaryBlock(in_dim, fdim, self.use_bn, self.bn_momentum)
        self.pa_unary = UnaryBlock(out_dim, fdim, self.use_bn, self.bn_momentum)

    def forward(self, features, batch):
        sa, sa_x = self.sa_f(features, batch)
        ca = self.ca_f(features, batch)
        pa = self.simple1(torch.cat((features, sa_x), dim=1), batch)
        return (self.sa_unary(sa, batch), self.ca_unary(ca, batch), self.no_unary(features, batch),
                self.pa_unary(pa, batch))


class global_average_block(nn.Module):
    """per-sphere mean of the features (blocks.py:930-955)"""

    def __init__(self, block_name, in_dim, out_dim, layer_ind, config):
        super(global_average_block, self).__init__()
        self.bn_momentum = config.batch_norm_momentum
        self.use_bn = config.use_batch_norm
        self.layer_ind = layer_ind
        self.block_name = block_name
        self.in_dim = in_dim
        self.out_dim = out_dim

    def forward(self, features, batch):
        return global_average(features, _layer_lengths(batch, self.layer_ind))


class ele_att(nn.Module):
    """elevation attention (blocks.py:957-1011): channel attention whose queries / keys come from the height of
    the points (relative and absolute: h, h + centre height of the sphere)"""

    def __init__(self, block_name, in_dim, out_dim, radius, layer_ind, config):
        super(ele_att, self).__init__()
        self.bn_momentum = config.batch_norm_momentum
        self.use_bn = config.use_batch_norm
        self.layer_ind = layer_ind
        self.block_name = block_name
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.unary1 = UnaryBlock(in_dim, out_dim, self.use_bn, self.bn_momentum)
        self.unary2 = UnaryBlock(in_dim, out_dim, self.use_bn, self.bn_momentum)
        self.gamma = nn.Parameter(torch.zeros(1))
        self.softmax = nn.Softmax(dim=-1)
        self.simple2 = SimpleBlock2(block_name, out_dim, out_dim, radius, layer_ind, config)

    def forward(self, features, h, batch):
        # the two projections run once on all spheres (UnaryBlock is row-wise, so this equals the per-sphere calls)
        spans = _per_cloud(_layer_lengths(batch, self.layer_ind))
        if h.shape[0] > 0:
            centre_z = torch.cat([batch.center_pts[ii][-1].to(h.dtype).expand(b - a, 1) for ii, (a, b) in enumerate(spans)], 0)
        else:
            centre_z = h
        ele_f = torch.cat((h, h + centre_z), dim=1)
        query, key = self.unary1(ele_f), self.unary2(ele_f)
        outs = []
        for a, b in spans:
            att = self.softmax(torch.matmul(query[a:b].T, key[a:b]))
            outs.append(torch.matmul(features[a:b], att))
        x = torch.cat(outs, 0) if outs else features.new_zeros((0, features.shape[1]))
        return self.simple2(self.gamma * x + features, batch)
