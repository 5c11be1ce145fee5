"""KP-FCNN segmentation network with the API of the reference's models/architectures.py.

``KPFCNN(config, lbl_values, ign_lbls)``, ``.forward(batch, config)``, ``.loss``, ``.accuracy``
keep the reference's names, wiring and state_dict keys (models/architectures.py:192-403), so
``utils/trainer_PseudoLabel.py`` drives this class unchanged; every KPConv / pooling operator
underneath is a HIP kernel (weasal_amd/blocks.py).  ``p2p_fitting_regularizer`` follows
architectures.py:24-57.  ``contrast_loss`` (architectures.py:405-504) is restated without torch_scatter and
without ``.cuda()`` calls (**parity unpinned**: torch_scatter is absent, so the
reference function has never run here -- it is checked against oracle/contrast_ref.py only).  Not provided: the
weak-label ``KPFCNN_mprm`` (rank 3).
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .blocks import KPConv, NearestUpsampleBlock, UnaryBlock, block_decider, closest_pool

_LAYER_CHANGE = ('pool', 'strided', 'upsample', 'global')


def p2p_fitting_regularizer(net):
    """Deformable-KPConv regulariser: 2 * L1(min_d2 / extent^2) fitting term plus the pairwise
    repulsion of the deformed kernel points closer than `repulse_extent` (architectures.py:24-57)."""
    fitting_loss = 0
    repulsive_loss = 0
    for m in net.modules():
        if not (isinstance(m, KPConv) and m.deformable):
            continue
        kp_min_d2 = m.min_d2 / (m.KP_extent ** 2)
        fitting_loss = fitting_loss + net.l1(kp_min_d2, torch.zeros_like(kp_min_d2))
        locs = m.deformed_KP / m.KP_extent                                   # [N, K, 3]
        for i in range(net.K):
            others = torch.cat([locs[:, :i, :], locs[:, i + 1:, :]], dim=1).detach()
            dist = torch.sqrt(torch.sum((others - locs[:, i:i + 1, :]) ** 2, dim=2))
            rep = torch.sum(torch.clamp_max(dist - net.repulse_extent, max=0.0) ** 2, dim=1)
            repulsive_loss = repulsive_loss + net.l1(rep, torch.zeros_like(rep)) / net.K
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

    def _fused_upsample_unary(self, x, skip, up_block, unary, batch):
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
        for block_i, block_op in enumerate(self.encoder_blocks):
            if block_i in self.encoder_skips:
                skips.append(x)
            x = block_op(x, batch)
        nd = len(self.decoder_blocks)
        block_i = 0
        while block_i < nd:
            block_op = self.decoder_blocks[block_i]
            nxt = self.decoder_blocks[block_i + 1] if block_i + 1 < nd else None
            if (self.fuse_decoder and isinstance(block_op, NearestUpsampleBlock) and isinstance(nxt, UnaryBlock)
                    and (block_i + 1) in self.decoder_concats and block_i not in self.decoder_concats):
                x = self._fused_upsample_unary(x, skips.pop(), block_op, nxt, batch)
                block_i += 2
                continue
            if block_i in self.decoder_concats:
                x = torch.cat([x, skips.pop()], dim=1)
            x = block_op(x, batch)
            block_i += 1
        if self.dropout:
            x = self.droplayer(x)
        x = self.head_mlp(x, batch)
        return self.head_softmax(x, batch)

    def _targets(self, labels):
        """labels -> class index in [0, C) or -1 for ignored labels (architectures.py:362-365)"""
        # one table gather instead of the reference's loop of masked assignments (same mapping)
        lut = getattr(self, "_target_lut", None)
        if lut is None or lut.device != labels.device:
            vmax = int(max(int(np.max(self.valid_labels)) if len(self.valid_labels) else 0, 0))
            lut = -torch.ones(vmax + 2, dtype=torch.int64)
            for i, c in enumerate(self.valid_labels):
                if 0 <= int(c) <= vmax:
                    lut[int(c)] = i
            lut = lut.to(labels.device)
            self._target_lut = lut
        # labels outside the table (ignored / unlabeled values) map to -1 through the last, spare entry
        idx = torch.where((labels >= 0) & (labels < lut.shape[0] - 1), labels, torch.full_like(labels, lut.shape[0] - 1))
        return lut[idx]

    def loss(self, outputs, labels):
        """cross entropy over [1, C, N] with ignore_index -1, plus the deformable regulariser"""
        target = self._targets(labels)
        self.output_loss = self.criterion(outputs.transpose(0, 1).unsqueeze(0), target.unsqueeze(0))
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

        prob = torch.softmax(outputs, 1)
        pseudo_logits = prob.max(1)[0]
        label_id = labels < 10                                   # > 10 = unlabeled (:430-433)
        certain_label = (pseudo_logits > threshold) | label_id
        pseudo_lbs = torch.argmax(prob, dim=1)
        pseudo_lbs = torch.where(label_id, labels.to(pseudo_lbs.dtype), pseudo_lbs)
        all_valid_idx = torch.where(certain_label)[0]
        num_valid = all_valid_idx.shape[0]
        if num_valid < 1:
            print('Skipped loss calculations because there are no valid points in batch')
            return torch.zeros((), dtype=torch.float32, device=dev)
        n_draw = slc_con if num_valid >= slc_con else slc_con - num_valid
        if slice_draw is None:
            slice_draw = torch.randint(0, num_valid, (n_draw,))
        slice_draw = slice_draw.to(dev)
        if num_valid >= slc_con:
            slc_idx_idx = slice_draw
        else:
            slc_idx_idx = torch.cat((torch.arange(num_valid, device=dev), slice_draw), dim=0)
        slc_idx = all_valid_idx[slc_idx_idx]

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
        return (per_class * sel).sum() / sel.sum()

    def accuracy(self, outputs, labels):
        target = self._targets(labels)
        predicted = torch.argmax(outputs.data, dim=1)
        return (predicted == target).sum().item() / target.size(0)
