"""One training step with the reference's recipe (utils/trainer_PseudoLabel.py:80-87,199-219),
plus the data-parallel gradient exchange the reference does not have.

  zero_grad -> net(batch, config) -> net.loss -> backward -> [all-reduce grads / world] ->
  clip_grad_value_(grad_clip_norm) -> SGD(momentum, weight_decay; 'offset' params at
  lr * deform_lr_factor).step()

The reference's per-step torch.cuda.empty_cache() + synchronize() (:221-222) serialise the GPU
and are deliberately not reproduced.
"""
import torch


def make_optimizer(net, config):
    """SGD with a second parameter group for the deformable offsets (trainer_PseudoLabel.py:80-87)"""
    deform_params = [v for k, v in net.named_parameters() if 'offset' in k]
    other_params = [v for k, v in net.named_parameters() if 'offset' not in k]
    deform_lr = config.learning_rate * config.deform_lr_factor
    return torch.optim.SGD([{'params': other_params}, {'params': deform_params, 'lr': deform_lr}],
                           lr=config.learning_rate, momentum=config.momentum,
                           weight_decay=config.weight_decay)


def train_step(net, optimizer, batch, config, grad_sync=None, epoch=None):
    """-> (loss tensor, logits).  grad_sync: optional callable(net) run between backward and clip
    (weasal_amd.dp.GradSync: one flat RCCL all-reduce).  epoch: when given, the supervised contrastive
    loss joins from `config.contrast_start` on (trainer_PseudoLabel.py:204-208)."""
    optimizer.zero_grad(set_to_none=False)
    outputs = net(batch, config)
    loss = net.loss(outputs, batch.labels)
    if epoch is not None and epoch >= getattr(config, 'contrast_start', 1 << 30):
        loss = loss + net.contrast_loss(outputs, batch.labels, config)
    loss.backward()
    if grad_sync is not None:
        grad_sync(net)
    if config.grad_clip_norm > 0:
        torch.nn.utils.clip_grad_value_(net.parameters(), config.grad_clip_norm)
    optimizer.step()
    return loss, outputs
