"""oracle/contrast_ref.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's supervised contrastive loss, KPFCNN.contrast_loss
(models/architectures.py:405-504), in the reference's own formulation: dense float `[N, slc_con]`
masks (mask_slice built by scattering zeros at the (i, j) with i == slc_idx[j], :455-461; mask_certain
:468; mask_positive :472), softmax statistics :478-492, per-point loss :494-498, and
`torch_scatter.scatter(src, index, reduce="mean")` restated from its published semantics (out[c] = mean of
src where index == c for c in 0..index.max(), 0 where a class has no element) as a loop over classes, :499-504.

PARITY UNPINNED: torch_scatter is not installed in this image, so the reference function itself has never
been executed here and there is no golden vector for it; the reference's repository holds no fixture for this
loss either.  This file pins weasal_amd.architectures.KPFCNN.contrast_loss to a second, independent
formulation only.  The random slice (`torch.randint`, :448/:452) is an explicit argument.
"""
import torch


def contrast_loss_ref(outputs, labels, contrast_thd, slice_draw):
    """outputs [N,C] float32 (requires_grad allowed), labels [N] int64, slice_draw LongTensor of the draw
    (slc_con entries when num_valid >= slc_con, else slc_con - num_valid) -> scalar loss tensor"""
    temperature, base_temperature, slc_con, eps = 0.1, 1, 1000, 1e-8
    N = outputs.shape[0]
    threshold = contrast_thd / 100
    prob = torch.nn.Softmax(1)(outputs)
    pseudo_logits = prob.max(1)[0]
    label_id = labels < 10
    certain_label = ((pseudo_logits > threshold).int() + label_id.int()) > 0
    pseudo_lbs = torch.argmax(prob, dim=1)
    pseudo_lbs[label_id] = labels[label_id]
    all_valid_idx = torch.where(certain_label)[0]
    num_valid = all_valid_idx.shape[0]
    if num_valid < 1:
        return torch.tensor(0).float()
    if num_valid >= slc_con:
        slc_idx = all_valid_idx[slice_draw]
    else:
        slc_idx = all_valid_idx[torch.cat((torch.arange(num_valid), slice_draw), dim=0)]
    mask_slice = torch.ones(N, slc_con)
    for j in range(slc_con):
        mask_slice[int(slc_idx[j]), j] = 0
    mask_certain = (certain_label[slc_idx].unsqueeze(0) == certain_label.unsqueeze(-1)).float()
    mask_positive = (pseudo_lbs[slc_idx].unsqueeze(0) == pseudo_lbs.unsqueeze(-1)).float()
    pos_mask = mask_positive * mask_slice * mask_certain
    outputs = torch.nn.functional.normalize(outputs, dim=1)
    mul = torch.div(torch.matmul(outputs, outputs[slc_idx].T), temperature)
    logits_max, _ = torch.max(mul, dim=1, keepdim=True)
    logits = mul - logits_max.detach()
    exp_logits = torch.exp(logits) * (mask_slice * mask_certain)
    log_prob = (logits - torch.log(exp_logits.sum(1, keepdim=True) + eps)) * (mask_slice * mask_certain)
    mean_log_prob_pos = (pos_mask * log_prob).sum(1) / (pos_mask.sum(1) + 1e-12)
    pts_loss = -(temperature / base_temperature) * mean_log_prob_pos
    cal_slc = pts_loss > 0
    pts_loss = pts_loss[cal_slc]
    cls = pseudo_lbs[cal_slc]
    if pts_loss.numel() == 0:
        return pts_loss.mean()
    per_class = []
    for c in range(int(cls.max()) + 1):           # scatter(..., reduce="mean"): zeros for empty classes
        sel = cls == c
        per_class.append(pts_loss[sel].mean() if bool(sel.any()) else torch.zeros(()))
    per_class = torch.stack(per_class)
    return per_class[per_class > 0].mean()
