"""Torch-CPU restatement of the reference's per-pixel losses (test infrastructure).

  cross_entropy  utils/losses.py:24-31   nn.CrossEntropyLoss(ignore_index, reduction='mean')
  dice           utils/losses.py:33-50   softmax, one-hot (ignore -> target.min(), :40-42), smooth=1,
                                         reduction over the whole batch
  focal          utils/losses.py:52-65   per-pixel CE (0 at ignored), (1-exp(-ce))^gamma * ce, mean over ALL pixels
  lovasz_softmax utils/losses.py:79-89 -> utils/lovasz_losses.py:153-218 with classes='present',
                                         per_image=False; lovasz_grad :19-31
Written from the formulas in SURVEY.md App. B/F (closed forms), not from the reference's code.
"""
import torch
import torch.nn.functional as F


def cross_entropy(logits, target, ignore_index=255):
    logp = F.log_softmax(logits, dim=1)
    valid = target != ignore_index
    picked = logp.gather(1, target.clamp(0, logits.shape[1] - 1).unsqueeze(1)).squeeze(1)
    return -(picked * valid).sum() / valid.sum()


def dice(logits, target, ignore_index=255, smooth=1.0):
    t = target.clone()
    tmin, tmax = int(t.min()), int(t.max())
    if ignore_index not in range(tmin, tmax) and bool((t == ignore_index).any()):
        t[t == ignore_index] = tmin
    onehot = F.one_hot(t, logits.shape[1]).permute(0, 3, 1, 2).to(logits.dtype)
    p = F.softmax(logits, dim=1)
    inter = (p * onehot).sum()
    return 1 - (2.0 * inter + smooth) / (p.sum() + onehot.sum() + smooth)


def focal(logits, target, ignore_index=255, gamma=2.0):
    logp = F.log_softmax(logits, dim=1)
    valid = target != ignore_index
    ce = -(logp.gather(1, target.clamp(0, logits.shape[1] - 1).unsqueeze(1)).squeeze(1)) * valid
    pt = torch.exp(-ce)
    return (((1 - pt) ** gamma) * ce).mean()


def lovasz_grad(gt_sorted):
    """Gradient of the Lovasz extension w.r.t. sorted errors: first difference of the Jaccard index."""
    g = gt_sorted.float()
    total = g.sum()
    inter = total - g.cumsum(0)
    union = total + (1 - g).cumsum(0)
    jac = 1.0 - inter / union
    if g.numel() > 1:
        jac[1:] = jac[1:] - jac[:-1]
    return jac


def lovasz_softmax(logits, target, ignore_index=255):
    C = logits.shape[1]
    p = F.softmax(logits, dim=1).permute(0, 2, 3, 1).reshape(-1, C)
    t = target.reshape(-1)
    keep = t != ignore_index
    p, t = p[keep], t[keep]
    if p.numel() == 0:
        return p.sum() * 0.0
    losses = []
    for c in range(C):
        fg = (t == c).float()
        if fg.sum() == 0:
            continue
        err = (fg - p[:, c]).abs()
        err_sorted, perm = torch.sort(err, 0, descending=True)
        losses.append(torch.dot(err_sorted, lovasz_grad(fg[perm])))
    return sum(losses) / len(losses)


def eval_metrics(logits, target, num_class):
    """utils/metrics.py:42-67: (correct, labeled, inter[C], union[C]) with 1-based labels, argmax = first max."""
    pred = logits.argmax(1) + 1
    tgt = target + 1
    labeled = (tgt > 0) & (tgt <= num_class)
    correct = int(((pred == tgt) & labeled).sum())
    pred = pred * labeled
    inter = pred * (pred == tgt)
    area_inter = torch.histc(inter.float(), bins=num_class, min=1, max=num_class)
    area_pred = torch.histc(pred.float(), bins=num_class, min=1, max=num_class)
    area_lab = torch.histc((tgt * labeled).float(), bins=num_class, min=1, max=num_class)
    return correct, int(labeled.sum()), area_inter, area_pred + area_lab - area_inter
