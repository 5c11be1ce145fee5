"""KP-FCNN segmentation network with the API of the reference's models/architectures.py.

``KPFCNN(config, lbl_values, ign_lbls)``, ``.forward(batch, config)``, ``.loss``, ``.accuracy``
keep the reference's names, wiring and state_dict keys (models/architectures.py:192-403), so
``utils/trainer_PseudoLabel.py`` drives this class unchanged; every KPConv / pooling operator
underneath is a HIP kernel (weasal_amd/blocks.py).  ``p2p_fitting_regularizer`` follows
architectures.py:24-57.  ``contrast_loss`` (architectures.py:405-504) is restated without torch_scatter and
without ``.cuda()`` calls (**parity unpinned**: torch_scatter is absent, so the
reference function has never run here -- it is checked against oracle/contrast_ref.py only).  The weak-label
``KPFCNN_mprm`` (architectures.py:507-807, rank 3) is at the end of this file, pinned by golden g10.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import fused, ops
from .blocks import KPConv, NearestUpsampleBlock, UnaryBlock, block_decider, closest_pool

_LAYER_CHANGE = ('pool', 'strided', 'upsample', 'global')
REGULARIZER_KERNEL = os.environ.get("WEASAL_REG_KERNEL", "1") != "0"      # A/B switch: 0 = the torch-op form below
DROPOUT_KERNEL = os.environ.get("WEASAL_DROPOUT_KERNEL", "1") != "0"       # A/B switch: 0 = nn.Dropout (the framework's kernels)
DROPOUT_FUSED = os.environ.get("WEASAL_DROPOUT_FUSED", "1") != "0"         # A/B switch: 0 = the dropout kernel as a pass of its own
CONTRAST_KERNELS = os.environ.get("WEASAL_CONTRAST_KERNELS", "1") != "0"  # A/B switch: 0 = contrast_loss's head / tail as torch ops


def p2p_fitting_regularizer(net):
    """Deformable-KPConv regulariser: 2 * L1(min_d2 / extent^2) fitting term plus the pairwise
    repulsion of the deformed kernel points closer than `repulse_extent` (architectures.py:24-57)."""
    fitting_loss = 0
    repulsive_loss = 0
    layers = getattr(net, "_deformable_layers", None)
    if layers is None:                      # the module tree is static: scan it once, not every step
        layers = [m for m in net.modules() if isinstance(m, KPConv) and m.deformable]
        net._deformable_layers = layers
    for m in layers:
        if REGULARIZER_KERNEL and m.min_d2.is_cuda and net.K == 15 and ops.kpconv_gather is ops._KPCONV_GATHER_SELF:
            # both terms and their gradients in one kernel each way (ws_p2p_regularizer_fwd / _bwd)
            fr = ops.p2p_regularizer(m.deformed_KP, m.min_d2, m.KP_extent, net.repulse_extent)
            fitting_loss = fitting_loss + fr[0]
            repulsive_loss = repulsive_loss + fr[1]
            continue
        kp_min_d2 = m.min_d2 / (m.KP_extent ** 2)
        fitting_loss = fitting_loss + net.l1(kp_min_d2, torch.zeros_like(kp_min_d2))
        locs = m.deformed_KP / m.KP_extent                                   # [N, K, 3]
        # the reference loops over the K kernel points (:45-51): point i against the DETACHED other K-1 points.
        # Same terms in one pass: dist[n, i, j] = |locs[n, i] - stopgrad(locs[n, j])|, the j == i column masked out.
        diff = locs.unsqueeze(2) - locs.detach().unsqueeze(1)                # [N, K(i), K(j), 3]
        off_diag = ~torch.eye(net.K, dtype=torch.bool, device=locs.device)
        # + 1 on the (masked) diagonal: sqrt(0) would put an infinite derivative times the zero mask into the gradient
        dist = torch.sqrt(torch.sum(diff ** 2, dim=3) + (~off_diag).to(locs.dtype))
        pen = torch.clamp_max(dist - net.repulse_extent, max=0.0) ** 2
        rep = torch.sum(pen * off_diag, dim=2)                               # [N, K]: sum over the other points
        repulsive_loss = repulsive_loss + torch.sum(torch.mean(torch.abs(rep), dim=0)) / net.K
    return net.deform_fitting_power * (2 * fitting_loss + repulsive_loss)


class KPFCNN(nn.Module):
    """Encoder / decoder KP-FCNN built from config.architecture (architectures.py:197-326)."""

    def __init__(self, config, lbl_values, ign_lbls):
        super(KPFCNN, self).__init__()
        arch = list(config.architecture)
        layer = 0
        r = config.first_subsampling_dl * config.conv_radius
        in_dim = config.in_features_dim
        out_dim = config.first_features_dim
        self.K = config.num_kernel_points
        self.C = len(lbl_values) - len(ign_lbls)

        # ---- encoder: every block up to the first upsampling
        self.encoder_blocks = nn.ModuleList()
        self.encoder_skip_dims = []
        self.encoder_skips = []
        for block_i, block in enumerate(arch):
            if 'equivariant' in block and out_dim % 3 != 0:
                raise ValueError('Equivariant block but features dimension is not a factor of 3')
            if any(tag in block for tag in _LAYER_CHANGE):
                self.encoder_skips.append(block_i)
                self.encoder_skip_dims.append(in_dim)
            if 'upsample' in block:
                break
            self.encoder_blocks.append(block_decider(block, r, in_dim, out_dim, layer, config))
            in_dim = out_dim // 2 if 'simple' in block else out_dim
            if 'pool' in block or 'strided' in block:
                layer += 1
                r *= 2
                out_dim *= 2

        # ---- decoder: from the first upsampling on, skip features concatenated after each upsampling
        self.decoder_blocks = nn.ModuleList()
        self.decoder_concats = []
        start_i = next((i for i, b in enumerate(arch) if 'upsample' in b), 0)
        for block_i, block in enumerate(arch[start_i:]):
            if block_i > 0 and 'upsample' in arch[start_i + block_i - 1]:
                in_dim += self.encoder_skip_dims[layer]
                self.decoder_concats.append(block_i)
            self.decoder_blocks.append(block_decider(block, r, in_dim, out_dim, layer, config))
            in_dim = out_dim
            if 'upsample' in block:
                layer -= 1
                r *= 0.5
                out_dim = out_dim // 2

        # heads: Linear + bias + LeakyReLU, on the logits too (architectures.py:299-300)
        self.head_mlp = UnaryBlock(out_dim, config.first_features_dim, False, 0)
        self.head_softmax = UnaryBlock(config.first_features_dim, self.C, False, 0)
        self.dropout = config.dropout
        if config.dropout:
            self.droplayer = nn.Dropout(p=float(config.dropout))

        # ---- losses
        self.valid_labels = np.sort([c for c in lbl_values if c not in ign_lbls])
        if len(config.class_w) > 0:
            class_w = torch.from_numpy(np.array(config.class_w, dtype=np.float32))
            self.criterion = torch.nn.CrossEntropyLoss(weight=class_w, ignore_index=-1)
        else:
            self.criterion = torch.nn.CrossEntropyLoss(ignore_index=-1)
        self.deform_fitting_mode = config.deform_fitting_mode
        self.deform_fitting_power = config.deform_fitting_power
        self.deform_lr_factor = config.deform_lr_factor
        self.repulse_extent = config.repulse_extent
        self.output_loss = 0
        self.reg_loss = 0
        self.l1 = nn.L1Loss()
        # nearest_upsample -> concat(skip) -> unary is evaluated as
        #   up(x @ W_x^T) + skip @ W_s^T      (W = [W_x | W_s], the unary's weight)
        # i.e. the x-part of the 1x1 MLP runs at the coarse resolution and only its (narrower) result
        # is upsampled; identical in exact arithmetic to architectures.py:339-343 (the gather of a row
        # commutes with a per-row linear map), ~2.3x fewer decoder FLOPs, no [N_fine, C_up + C_skip] tensor.
        self.fuse_decoder = True
        # BASELINE config 5: feature rows (activations, their gradients, weighted features) bf16 in HBM, fp32 accumulate,
        # fp32 master weights; the 3-channel input layer and the 9 logits stay f32
        self.feature_dtype = torch.bfloat16 if getattr(config, 'feature_dtype', 'f32') == 'bf16' else torch.float32
        if self.feature_dtype == torch.bfloat16:
            self.head_softmax.out_f32 = True

    def _fused_upsample_unary(self, x, skip, up_block, unary, batch, drop=None):
        if fused.upunary_eligible(x, skip, unary):
            return fused.upunary(x, skip, unary, batch.upsamples[up_block.layer_ind - 1], drop, batch)   # one C call each way
        assert drop is None
        c_up = x.shape[1]
        w = unary.mlp.weight
        y = closest_pool(ops.linear(x, w[:, :c_up]), batch.upsamples[up_block.layer_ind - 1])
        return ops.matmul_epilogue(skip, w[:, c_up:].t(), bias=unary.batch_norm.epilogue_bias(), residual=y,
                                   slope=None if unary.no_relu else 0.1)

    def forward(self, batch, config):
        if hasattr(batch, "activate"):
            batch.activate()     # stream hand-over, scheduling hints and pre-built tables of the batch
        else:
            ops.clear_table_cache()          # transposed tables belong to one batch
        x = batch.features.clone().detach()
        skips = []
        slots = []
        link = None
        for block_i, block_op in enumerate(self.encoder_blocks):
            slot = None
            if block_i in self.encoder_skips:
                skips.append(x)
                slot = fused.SkipSlot()      # (armed by the strided block call if it is one: fused.resnetb_block)
                slots.append(slot)
            batch.skip_slot = slot
            # gate links (fused.GateLink): consecutive block calls hand the activation backward to the consumer's store
            batch.gate_link_in, batch.gate_link_out = link, fused.GateLink()
            link = batch.gate_link_out
            x = block_op(x, batch)
            batch.skip_slot = None
            batch.gate_link_in = batch.gate_link_out = None
            if block_i == 0 and self.feature_dtype == torch.bfloat16 and x.dtype == torch.float32 and x.shape[1] % 32 == 0:
                x = x.to(torch.bfloat16)     # the 3-channel input layer ran in f32; bf16 rows from here on
        nd = len(self.decoder_blocks)
        block_i = 0
        dropped = False
        while block_i < nd:
            block_op = self.decoder_blocks[block_i]
            nxt = self.decoder_blocks[block_i + 1] if block_i + 1 < nd else None
            if (self.fuse_decoder and isinstance(block_op, NearestUpsampleBlock) and isinstance(nxt, UnaryBlock)
                    and (block_i + 1) in self.decoder_concats and block_i not in self.decoder_concats):
                drop = None
                if (block_i + 2 == nd and self.dropout and DROPOUT_KERNEL and DROPOUT_FUSED and self.training and not nxt.no_relu
                        and torch.is_grad_enabled() and x.dtype == torch.float32 and fused.upunary_eligible(x, skips[-1], nxt)):
                    # the droplayer in front of the head (architectures.py:345-346) rides on the last decoder step's epilogue:
                    # the same keep decisions as ops.dropout (same seed draw), no pass of its own in either direction
                    drop = (float(self.dropout), int(torch.randint(0, 1 << 62, (1,)).item()))
                    dropped = True
                batch.gate_link_in, batch.gate_link_out = link, fused.GateLink()
                link = batch.gate_link_out
                x = self._fused_upsample_unary(x, fused.skip_tap(skips.pop(), slots.pop()), block_op, nxt, batch, drop)
                batch.gate_link_in = batch.gate_link_out = None
                block_i += 2
                continue
            if block_i in self.decoder_concats:
                x = torch.cat([x, fused.skip_tap(skips.pop(), slots.pop())], dim=1)
            link = None                          # (an operator-path step: no link across it)
            x = block_op(x, batch)
            block_i += 1
        if self.dropout and not dropped:
            link = None                      # (a dropout pass of its own sits between the decoder and the head)
            if DROPOUT_KERNEL and self.training and x.is_cuda and x.dtype == torch.float32 and x.requires_grad:
                x = ops.dropout(x, float(self.dropout))          # one pass each way, mask recomputed instead of stored
            else:
                x = self.droplayer(x)
        # the head's two unary blocks continue the chain of gate links (fused.GateLink): head_softmax's dX product applies
        # head_mlp's LeakyReLU', head_mlp's the last decoder step's (and its fused dropout)
        batch.gate_link_in, batch.gate_link_out = link, fused.GateLink()
        link = batch.gate_link_out
        x = self.head_mlp(x, batch)
        batch.gate_link_in, batch.gate_link_out = link, None
        x = self.head_softmax(x, batch)
        batch.gate_link_in = None
        return x

    def _label_lut(self, device):
        """label value -> class position table [vmax + 2] (architectures.py:362-365); the last, spare entry is the -1
        that ignored / unlabeled values outside the table take"""
        lut = getattr(self, "_target_lut", None)
        if lut is None or lut.device != device:
            vmax = int(max(int(np.max(self.valid_labels)) if len(self.valid_labels) else 0, 0))
            lut = -torch.ones(vmax + 2, dtype=torch.int64)
            for i, c in enumerate(self.valid_labels):
                if 0 <= int(c) <= vmax:
                    lut[int(c)] = i
            lut = lut.to(device)
            self._target_lut = lut
        return lut

    def _targets(self, labels):
        """labels -> class index in [0, C) or -1 for ignored labels (architectures.py:362-365)"""
        # one table gather instead of the reference's loop of masked assignments (same mapping)
        lut = self._label_lut(labels.device)
        idx = torch.where((labels >= 0) & (labels < lut.shape[0] - 1), labels, torch.full_like(labels, lut.shape[0] - 1))
        return lut[idx]

    def loss(self, outputs, labels):
        """cross entropy over [1, C, N] with ignore_index -1 (label mapping and criterion fused: ops.cross_entropy),
        plus the deformable regulariser"""
        self.output_loss = ops.cross_entropy(outputs, labels, self._label_lut(labels.device), self.criterion.weight)
        if self.deform_fitting_mode == 'point2point':
            self.reg_loss = p2p_fitting_regularizer(self)
        elif self.deform_fitting_mode == 'point2plane':
            raise ValueError('point2plane fitting mode not implemented yet.')
        else:
            raise ValueError('Unknown fitting mode: ' + self.deform_fitting_mode)
        return self.output_loss + self.reg_loss

    def contrast_loss(self, outputs, labels, config, threshold=0.2, slice_draw=None):
        """Supervised contrastive loss of the pseudo-label trainer (architectures.py:405-504,
        trainer_PseudoLabel.py:204-208): every point is compared with a slice of 1000 randomly drawn
        valid points; positives share the (pseudo) label.  The `[N, slc_con]` part (three masks, similarities,
        masked log-softmax, mean over positives; six 1.6 GB matrices in the reference at N = 400 000) is the
        fused HIP operator `ops.contrast_rows`; `torch_scatter.scatter(reduce="mean")` over the pseudo labels
        is an index_add of sums and counts (classes without points drop out with the reference's `> 0`
        filter).  `slice_draw`: optional LongTensor replacing the `torch.randint`
        draw (tests); `threshold` is overwritten by `config.contrast_thd / 100` as in the reference."""
        temperature = 0.1
        base_temperature = 1
        slc_con = 1000
        dev = outputs.device
        N = outputs.shape[0]
        eps = 1e-8
        threshold = config.contrast_thd / 100
        self.pts_loss = 0
        self.pts_loss_self = 0

        if CONTRAST_KERNELS and outputs.is_cuda and outputs.dtype == torch.float32 and outputs.shape[1] <= 16:
            # the whole loss as one node (ops.contrast_loss: 5 launches each way); the draw: slc_con device uniforms, or the
            # explicit positions of the tests (fewer than slc_con of them = the "fewer valid points" branch, :450-454)
            if slice_draw is None:
                draw = torch.rand(slc_con, device=dev)
            else:
                slice_draw = slice_draw.to(dev).to(torch.int64)
                nv_host = slc_con - slice_draw.shape[0]
                draw = slice_draw if nv_host <= 0 else torch.cat((torch.arange(nv_host, device=dev), slice_draw), dim=0)
            loss, self.pts_loss, self.contrast_slice, _ = ops.contrast_loss(outputs, labels, draw, threshold, temperature, eps)
            return loss if base_temperature == 1 else loss / base_temperature

        prob = torch.softmax(outputs, 1)
        pseudo_logits = prob.max(1)[0]
        label_id = labels < 10                                   # > 10 = unlabeled (:430-433)
        certain_label = (pseudo_logits > threshold) | label_id
        pseudo_lbs = torch.argmax(prob, dim=1)
        pseudo_lbs = torch.where(label_id, labels.to(pseudo_lbs.dtype), pseudo_lbs)
        # ---- the slice of slc_con valid points (:437-454) WITHOUT host synchronisation.  The reference lists the valid points
        # (torch.where: the host must learn their number) and draws torch.randint(0, num_valid) positions in that list.  Here
        # the r-th valid point is found on the device: cs = running count of valid points, position r <-> the first i with
        # cs[i] = r + 1 (a binary search per draw); num_valid = cs[-1] stays a device scalar and the draws are
        # floor(u * num_valid) of slc_con device uniforms -- the same distribution as randint.  The branch for fewer than slc_con
        # valid points (:450-454: all of them once, then random repeats) is the same fixed-size expression.
        cs = torch.cumsum(certain_label.to(torch.int64), dim=0)
        num_valid = cs[-1]                                           # device scalar
        j = torch.arange(slc_con, device=dev)
        if slice_draw is None:
            u = torch.rand(slc_con, device=dev)
            r_rand = (u * num_valid.to(torch.float32)).floor().to(torch.int64)
            r = torch.where((num_valid < slc_con) & (j < num_valid), j, r_rand)
        else:                                                        # explicit draw (tests): its length tells the branch
            slice_draw = slice_draw.to(dev).to(torch.int64)
            nv_host = slc_con - slice_draw.shape[0]
            r = slice_draw if nv_host <= 0 else torch.cat((torch.arange(nv_host, device=dev), slice_draw), dim=0)
        r = torch.minimum(r, (num_valid - 1).clamp(min=0).to(torch.int64))
        slc_idx = torch.searchsorted(cs, r + 1).clamp(max=N - 1)

        # [N, slc_con] part (:455-497): masks, temperature-scaled similarities, masked log-softmax and the mean
        # over the positives -- one fused HIP kernel per direction, nothing of size [N, slc_con] is stored
        outputs = nn.functional.normalize(outputs, dim=1)
        pts_loss = ops.contrast_rows(outputs, outputs[slc_idx], slc_idx, certain_label, pseudo_lbs, temperature, eps)
        if base_temperature != 1:
            pts_loss = pts_loss / base_temperature
        # :498-504 without host synchronisation: points with loss <= 0 are dropped, the rest averaged per
        # pseudo label (scatter-mean), classes whose mean is not > 0 (or empty) dropped, then the mean.
        # Pseudo labels are argmax indices or given labels < 10, so max(C, 10) bins cover them.
        keep = (pts_loss > 0).to(pts_loss.dtype)
        n_cls = max(int(outputs.shape[1]), 10)
        sums = torch.zeros(n_cls, dtype=pts_loss.dtype, device=dev).index_add(0, pseudo_lbs, pts_loss * keep)
        cnts = torch.zeros(n_cls, dtype=pts_loss.dtype, device=dev).index_add(0, pseudo_lbs, keep)
        per_class = sums / cnts.clamp(min=1)
        sel = (per_class > 0).to(pts_loss.dtype)
        self.pts_loss = per_class
        loss = (per_class * sel).sum() / sel.sum()
        # no valid point at all: the reference returns 0 before any of the above (:441-443)
        return torch.where(num_valid > 0, loss, torch.zeros((), dtype=loss.dtype, device=dev))

    def accuracy(self, outputs, labels):
        target = self._targets(labels)
        predicted = torch.argmax(outputs.data, dim=1)
        return (predicted == target).sum().item() / target.size(0)


class KPFCNN_mprm(nn.Module):
    """KP-FCNN for weak labels with multi-path region mining and the elevation attention head
    (models/architectures.py:507-807): encoder as in KPFCNN, then `ele_att`,
    `multi_path_att` (no / point-wise / spatial / channel attention, each projected to the classes), per-sphere
    class logits of every path and the shared parameter-free decoder (nearest upsampling) that turns the four
    class-activation maps into point-wise scores; `x` is their element-wise maximum.  Same constructor, module
    names (state_dict keys), return values and loss methods as the reference; no `.cuda()` anywhere -- tensors
    stay on the device of the batch.  Pinned by tests/golden/g10_mprm.npz, generated with the reference's classes."""

    def __init__(self, config, lbl_values, ign_lbls):
        super(KPFCNN_mprm, self).__init__()
        from .blocks import ele_att, global_average_block, multi_path_att
        arch = list(config.architecture)
        layer = 0
        r = config.first_subsampling_dl * config.conv_radius
        in_dim = config.in_features_dim
        out_dim = config.first_features_dim
        self.K = config.num_kernel_points
        self.C = len(lbl_values) - len(ign_lbls)
        self.features_layer = 'encoder_blocks.5.unary_shortcut.mlp'
        self.forward_features = {}
        self.backward_features = {}

        self.encoder_blocks = nn.ModuleList()
        self.encoder_skip_dims = []
        self.encoder_skips = []
        for block_i, block in enumerate(arch):
            if 'equivariant' in block and out_dim % 3 != 0:
                raise ValueError('Equivariant block but features dimension is not a factor of 3')
            if any(tag in block for tag in _LAYER_CHANGE + ('attention',)):
                self.encoder_skips.append(block_i)
                self.encoder_skip_dims.append(in_dim)
            if 'attention' in block or 'upsample' in block:
                break
            self.encoder_blocks.append(block_decider(block, r, in_dim, out_dim, layer, config))
            in_dim = out_dim // 2 if 'simple' in block else out_dim
            if 'pool' in block or 'strided' in block:
                layer += 1
                r *= 2
                out_dim *= 2

        nc = config.num_classes
        self.multi_att = multi_path_att('attention', out_dim, out_dim, r, layer, config)
        self.ele_head = ele_att('ele_attention', 2, out_dim, r, layer, config)
        self.no_ga = global_average_block('ga1', nc, nc, layer, config)
        self.da_ga = global_average_block('ga2', nc, nc, layer, config)
        self.spa_ga = global_average_block('ga3', nc, nc, layer, config)
        self.cha_ga = global_average_block('ga4', nc, nc, layer, config)

        self.decoder_blocks = nn.ModuleList()
        self.decoder_concats = []
        start_i = next((i for i, b in enumerate(arch) if 'upsample' in b), 0)
        for block_i, block in enumerate(arch[start_i:]):
            if block_i > 0 and 'upsample' in arch[start_i + block_i - 1]:
                in_dim += self.encoder_skip_dims[layer]
                self.decoder_concats.append(block_i)
            self.decoder_blocks.append(block_decider(block, r, in_dim, out_dim, layer, config))
            in_dim = out_dim
            if 'upsample' in block:
                layer -= 1
                r *= 0.5
                out_dim = out_dim // 2

        self.valid_labels = np.sort([c for c in lbl_values if c not in ign_lbls])
        if len(config.class_w) > 0:
            class_w = torch.from_numpy(np.array(config.class_w, dtype=np.float32))
            self.criterion = torch.nn.CrossEntropyLoss(weight=class_w, ignore_index=-1)
            self.criterion_multi = torch.nn.BCEWithLogitsLoss(weight=class_w)
        else:
            self.criterion = torch.nn.CrossEntropyLoss(ignore_index=-1)
            self.criterion_multi = torch.nn.BCEWithLogitsLoss()
        self.deform_fitting_mode = config.deform_fitting_mode
        self.deform_fitting_power = config.deform_fitting_power
        self.deform_lr_factor = config.deform_lr_factor
        self.repulse_extent = config.repulse_extent
        self.output_loss = 0
        self.reg_loss = 0
        self.l1 = nn.L1Loss()
        self.register_hooks()

    def register_hooks(self):
        """keep the output (and its gradient) of `features_layer` per device, as the reference does for its
        class-activation visualisations (architectures.py:651-665)"""
        def forward_hook(_module, _inp, out):
            self.forward_features[out.device] = out

        def backward_hook(_module, _grad_in, grad_out):
            self.backward_features[grad_out[0].device] = grad_out[0]
        for name, module in self.named_modules():
            if name == self.features_layer:
                module.register_forward_hook(forward_hook)
                module.register_full_backward_hook(backward_hook)

    def forward(self, batch, config):
        if hasattr(batch, "activate"):
            batch.activate()
        else:
            ops.clear_table_cache()
        x = batch.features.clone().detach()
        ele_down = batch.points[2][:, -1].unsqueeze(-1).clone().detach()     # heights at the attention level (:672)
        for block_i, block_op in enumerate(self.encoder_blocks):
            x = block_op(x, batch)
        x = self.ele_head(x, ele_down, batch)
        spa_att, cha_att, no_att, poi_att = self.multi_att(x, batch)
        cla_logits = [self.no_ga(no_att, batch), self.da_ga(poi_att, batch), self.spa_ga(spa_att, batch),
                      self.cha_ga(cha_att, batch)]
        for block_op in self.decoder_blocks:
            no_att = block_op(no_att, batch)
            poi_att = block_op(poi_att, batch)
            spa_att = block_op(spa_att, batch)
            cha_att = block_op(cha_att, batch)
        x = torch.max(torch.max(torch.max(no_att, poi_att), spa_att), cha_att)
        return x, cla_logits, [no_att, poi_att, spa_att, cha_att]

    def class_logits_loss(self, class_logits, cloud_lb):
        """BCE-with-logits of the four per-sphere class logits against one weak label vector per sphere, plus the
        deformable regulariser (architectures.py:709-733)"""
        self.output_loss1 = self.criterion_multi(class_logits[0], cloud_lb)
        self.output_loss2 = self.criterion_multi(class_logits[1], cloud_lb)
        self.output_loss3 = self.criterion_multi(class_logits[2], cloud_lb)
        self.output_loss4 = self.criterion_multi(class_logits[3], cloud_lb)
        if self.deform_fitting_mode == 'point2point':
            self.reg_loss = p2p_fitting_regularizer(self)
        elif self.deform_fitting_mode == 'point2plane':
            raise ValueError('point2plane fitting mode not implemented yet.')
        else:
            raise ValueError('Unknown fitting mode: ' + self.deform_fitting_mode)
        return self.output_loss1 + self.output_loss2 + self.output_loss3 + self.output_loss4 + self.reg_loss

    def region_mprm_loss(self, cam, regions_all, regions_lb, batch_lengths):
        """overlap-region loss (architectures.py:735-784): the four class-activation maps are averaged over every
        labelled sub-region of every sphere and compared with the sub-region's weak labels"""
        dev = cam[0].device
        cam_all = torch.stack(cam, dim=0)                              # [4, N, C]
        # The reference averages region by region (one index upload and one gather per region, :752-768).  Same means as ONE
        # product: A [R, N] holds 1 / |region| at the region's points (built on the host from the host-side index lists, one
        # upload), averaged[r, k, :] = A[r, :] @ cam_k -- no per-region copies, no device-to-host reads.
        lens_host = np.asarray(batch_lengths.cpu() if isinstance(batch_lengths, torch.Tensor) else batch_lengths).astype(np.int64)
        n_total = int(cam_all.shape[1])
        rows, all_lbs = [], []
        start = 0
        for ri in range(len(regions_all)):
            n = int(lens_host[ri])
            if len(regions_all[ri]) > 0:
                all_lbs.append(np.stack(regions_lb[ri]).astype('float32'))
                for region in regions_all[ri]:
                    idx = np.asarray(region).astype('int64')
                    assert n >= int(idx.max()), 'logits problem'
                    w = np.zeros(n_total, dtype=np.float32)
                    np.add.at(w, idx + start, np.float32(1.0 / len(idx)))
                    rows.append(w)
            start += n
        all_lbs = torch.from_numpy(np.vstack(all_lbs)).to(dev)
        A = torch.from_numpy(np.stack(rows)).to(dev)                   # [R, N]
        k, _, c = cam_all.shape
        averaged = (A @ cam_all.permute(1, 0, 2).reshape(n_total, k * c)).reshape(A.shape[0], k, c)
        self.output_loss = 0
        for ii in range(averaged.shape[1]):
            self.output_loss = self.output_loss + self.criterion_multi(averaged[:, ii, :], all_lbs)
        return self.output_loss

    def accuracy(self, logits, labels):
        if not len(logits.size()) == 2:
            raise ValueError('Wrong logits output dimension: Expected 2, got ' + str(len(logits.size())))
        target = -torch.ones_like(labels)
        for i, c in enumerate(self.valid_labels):
            target[labels == c] = i
        predicted = torch.argmax(logits, dim=1)
        return (predicted == target).sum().item() / target.size(0)
