"""Counterpart of the reference's training harness body (processor/recognition.py:152-176, 249-296) for the drop-in
Models: optimizer construction, step-LR schedule and one training iteration.  The reference's own
`REC_Processor` runs unchanged on these Models; this module exists because `processor/` needs
h5py / tensorboard / prettytable to import and so that bench.py and the parity tests share one definition of a step.
"""
import numpy as np
import torch
import torch.nn.functional as F


def make_optimizer(model, optimizer='SGD', base_lr=0.1, nesterov=True, weight_decay=1e-4):
    """recognition.py:152-166"""
    if optimizer == 'SGD':
        params = list(model.parameters())
        # same update rule, one multi-tensor kernel per step instead of ~11 (weight decay, momentum, nesterov, update)
        fused = bool(params) and all(p.is_cuda for p in params)
        return torch.optim.SGD(params, lr=base_lr, momentum=0.9, nesterov=nesterov, weight_decay=weight_decay,
                               fused=fused)
    if optimizer == 'Adam':
        return torch.optim.Adam(model.parameters(), lr=base_lr, weight_decay=weight_decay)
    raise ValueError()


def adjust_lr(optimizer, base_lr, epoch, step):
    """recognition.py:168-176: lr = base_lr * 0.1 ** #(epoch >= step)."""
    lr = base_lr * (0.1 ** int(np.sum(epoch >= np.array(step)))) if step else base_lr
    for group in optimizer.param_groups:
        group['lr'] = lr
    return lr


def train_step(model, optimizer, data, label, grad_sync=None):
    """recognition.py:258-289: forward, CrossEntropy, zero_grad, backward, (gradient all-reduce,) step.
    Returns the loss tensor (no host sync here; the reference's `.item()` at :292 is the caller's choice)."""
    data = data.float()
    label = label.long()
    output = model(data)
    loss = F.cross_entropy(output, label)
    optimizer.zero_grad()
    loss.backward()
    if grad_sync is not None:
        grad_sync()
    optimizer.step()
    return loss.detach()


def weights_init(m):
    """recognition.py:31-44 (applied by REC_Processor.load_model via model.apply)."""
    classname = m.__class__.__name__
    if classname.find('Conv1d') != -1 or type(m) is torch.nn.Conv2d:
        m.weight.data.normal_(0.0, 0.02)
        if m.bias is not None:
            m.bias.data.fill_(0)
    elif classname.find('BatchNorm') != -1:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)
