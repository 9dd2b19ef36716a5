"""Training losses of the reference harness (src/run/kitti_seg.py:184-202, src/models/utils/lovasz_losses.py)
for device tensors, without the per-class host round trips of the reference loop.

The Lovasz-softmax term is evaluated for all classes at once: one segmented descending sort of the (C, P) error
matrix (device radix sort), two row-wise prefix sums, and a masked mean over the classes present -- no ``.sum() == 0``
sync per class (the reference does 20 of them on KITTI).
"""
import torch
import torch.nn.functional as F


def lovasz_softmax_flat(probas, labels, classes="present"):
    """ref lovasz_losses.py:174-202.  probas (P, C), labels (P,) in [0, C)."""
    if classes not in ("present", "all"):
        raise NotImplementedError("explicit class lists are not used by the reference harness")
    if probas.numel() == 0:
        return probas * 0.0
    p, c = probas.shape
    fg = F.one_hot(labels, c).to(probas.dtype).t().contiguous()            # (C, P) foreground masks
    err = (fg - probas.t()).abs()                                           # (C, P)
    err_sorted, order = torch.sort(err, dim=1, descending=True)
    fg_sorted = torch.gather(fg, 1, order)
    total = fg.sum(dim=1, keepdim=True)
    inter = total - fg_sorted.cumsum(1)
    union = total + (1.0 - fg_sorted).cumsum(1)
    jac = 1.0 - inter / union
    grad = torch.cat([jac[:, :1], jac[:, 1:] - jac[:, :-1]], dim=1)         # lovasz_grad (:19-31), constant w.r.t. probas
    per_class = (err_sorted * grad.detach()).sum(dim=1)
    if classes == "all":
        return per_class.mean()
    present = (total[:, 0] > 0).to(probas.dtype)
    return (per_class * present).sum() / present.sum()


def seg_loss_kitti(pred, gt, ignore=0, use_lovasz=False, class_weights=None):
    """ref kitti_seg.py:184-202: returns (loss, per-point NLL)."""
    logp = F.log_softmax(pred, dim=-1)
    if class_weights is None:
        per_point = F.nll_loss(logp, gt, reduction="none", ignore_index=ignore)
    else:
        assert ignore == 0
        w = torch.cat([torch.zeros(1, dtype=class_weights.dtype, device=class_weights.device), class_weights], dim=0)
        per_point = F.nll_loss(logp, gt, reduction="none", weight=w.to(pred.device))
    loss = per_point.mean()
    if use_lovasz:
        keep = gt != ignore
        loss = loss + 2 * lovasz_softmax_flat(F.softmax(pred, dim=-1)[keep], gt[keep]).mean()
    return loss, per_point
