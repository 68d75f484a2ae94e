"""Loss + accuracy epilogue of the training loops, one pass over the logits on MI355X.

What the launch scripts do after `model(...)` returns the (B, 1+K) logits:

  main_nce.py:314-316    loss = CrossEntropyLoss(output, target); calc_topk_accuracy(output, target, (1,5))
  main_nce.py:318-324    loss = -(log_softmax(output)*mask).sum(1)/mask.sum(1); calc_mask_accuracy
  main_coclr.py:343-346  multi_nce_loss = -log((softmax(logits)*mask).sum(1)).mean(), with
  main_coclr.py:384-392  column 0 dropped for rows that have other positives, calc_mask_accuracy
                         and calc_topk_accuracy against column 0          (utils/utils.py:52-85)

That is 8-12 ATen ops reading the logits 4-5 times, and three to five `.item()` host syncs before
`loss.backward()` may start.  Here the loss forward is `coclr_nce_loss_fwd` (csrc/loss.hip): one
launch computes the loss AND all the hit counts and leaves them as device scalars; the accuracy
helpers below return those scalars (0-dim device tensors, as the reference's do) without another
pass when they are asked about the logits / target / mask the loss has just seen.

Same names and call signatures as the reference helpers, so the caller change is the import.
"""
import torch
import torch.nn as nn

from . import ops

MODE_CE, MODE_MULTI, MODE_UBER = 0, 1, 2


def _as_u8(mask):
    if mask.dtype == torch.bool:
        return mask.contiguous().view(torch.uint8)
    if mask.dtype == torch.uint8:
        return mask.contiguous()
    return (mask != 0).view(torch.uint8)


def _forward_stats(logits, mask_u8, target, mode, drop_self, ks):
    B = logits.shape[0]
    rowstats = torch.empty(B, 8, dtype=torch.float32, device=logits.device)
    flags = torch.empty(B, dtype=torch.uint8, device=logits.device)
    scalars = torch.empty(5, dtype=torch.float32, device=logits.device)
    ops.nce_loss_fwd(logits, mask_u8, target, rowstats, flags, scalars, mode, drop_self, ks[0],
                     ks[1])
    return rowstats, flags, scalars


class _NceLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, mask_u8, target, mode, drop_self, ks, owner):
        logits = logits.contiguous()
        rowstats, flags, scalars = _forward_stats(logits, mask_u8, target, mode, drop_self, ks)
        ctx.save_for_backward(logits, rowstats, flags)
        ctx.mask, ctx.target, ctx.mode = mask_u8, target, mode
        owner["scalars"] = scalars
        return scalars[0]

    @staticmethod
    def backward(ctx, dloss):
        logits, rowstats, flags = ctx.saved_tensors
        dlogits = torch.empty_like(logits)
        ops.nce_loss_bwd(logits, ctx.mask, ctx.target, rowstats, flags,
                         dloss.to(torch.float32).contiguous(), dlogits, ctx.mode)
        return dlogits, None, None, None, None, None, None


def _loss(logits, mask, target, mode, drop_self=False, ks=(1, 5)):
    if logits.dim() != 2 or logits.dtype != torch.float32:
        raise ValueError("coclr_amd: NCE loss expects fp32 logits of shape (B, 1+K)")
    mask_u8 = _as_u8(mask) if mask is not None else None
    if target is not None:
        target = target.to(device=logits.device, dtype=torch.long).contiguous()
    owner = {}
    loss = _NceLossFn.apply(logits, mask_u8, target, mode, bool(drop_self), tuple(ks), owner)
    # remember what these statistics describe, so that the accuracy helpers can reuse them
    logits._coclr_nce_stats = {
        "scalars": owner["scalars"], "mode": mode, "ks": tuple(ks),
        "target": target, "mask_ptr": None if mask is None else mask.data_ptr(),
        "target_src": None, "version": logits._version}
    return loss


def cross_entropy(logits, target):
    """nn.CrossEntropyLoss()(logits, target), mean reduction (main_nce.py:201,315)."""
    loss = _loss(logits, None, target, MODE_CE)
    logits._coclr_nce_stats["target_src"] = target
    return loss


class CrossEntropyLoss(nn.Module):
    """Stand-in for the `criterion` of main_nce.py:201 / main_coclr.py:214 (mean reduction, no
    class weights): same call, one launch, hit counts for calc_topk_accuracy kept on the device."""

    def forward(self, logits, target):
        return cross_entropy(logits, target)


def multi_nce_loss(logits, mask, drop_self=False):
    """main_coclr.py:343-346.  drop_self=True is the branch of main_coclr.py:384-389
    (`mask_clone[mask_sum != 1, 0] = 0; multi_nce_loss(output, mask_clone)`) without cloning the
    mask: rows that have other positives leave column 0 out of their loss, and the accuracy
    statistics still refer to the ORIGINAL mask, as the reference's calc_mask_accuracy call does."""
    return _loss(logits, mask, None, MODE_MULTI, drop_self=drop_self)


def ubernce_loss(logits, mask):
    """main_nce.py:322-323: -(log_softmax(logits)*mask).sum(1)/mask.sum(1), batch mean."""
    return _loss(logits, mask, None, MODE_UBER)


def _cached(output, ks):
    st = getattr(output, "_coclr_nce_stats", None)
    if st is None or st["ks"] != tuple(ks) or st["version"] != output._version:
        return None
    return st


def calc_topk_accuracy(output, target, topk=(1,)):
    """utils/utils.py:52-69 with device-scalar results: fraction of rows whose target column is
    among the row's k largest logits, for each k (one or two values of k per call)."""
    ks = tuple(topk) if len(topk) == 2 else (topk[0], topk[0])
    if len(topk) > 2:
        raise NotImplementedError("coclr_amd: calc_topk_accuracy takes one or two values of k")
    st = _cached(output, ks)
    if st is not None and st["mode"] == MODE_CE and (st["target_src"] is target or
                                                     st["target"] is target):
        sc = st["scalars"]
    else:
        tgt = target.to(device=output.device, dtype=torch.long).contiguous()
        _, _, sc = _forward_stats(output.detach().contiguous(), None, tgt, MODE_CE, False, ks)
    return [sc[1], sc[2]][:len(topk)]


def calc_mask_accuracy(output, target_mask, topk=(1,)):
    """utils/utils.py:71-85: fraction of rows with at least one positive among the k largest logits."""
    ks = tuple(topk) if len(topk) == 2 else (topk[0], topk[0])
    if len(topk) > 2:
        raise NotImplementedError("coclr_amd: calc_mask_accuracy takes one or two values of k")
    st = _cached(output, ks)
    if st is not None and st["mode"] != MODE_CE and st["mask_ptr"] == target_mask.data_ptr():
        sc = st["scalars"]
    else:
        _, _, sc = _forward_stats(output.detach().contiguous(), _as_u8(target_mask), None,
                                  MODE_MULTI, False, ks)
    return [sc[1], sc[2]][:len(topk)]


def calc_self_accuracy(output, topk=(1, 5)):
    """calc_topk_accuracy(output, zeros(B)) of main_coclr.py:392: hits of column 0 alone; free when
    any of the losses above has just run on `output`."""
    ks = tuple(topk) if len(topk) == 2 else (topk[0], topk[0])
    st = _cached(output, ks)
    if st is not None:
        sc = st["scalars"]
    else:
        zeros = torch.zeros(output.shape[0], dtype=torch.long, device=output.device)
        _, _, sc = _forward_stats(output.detach().contiguous(), None, zeros, MODE_CE, False, ks)
    return [sc[3], sc[4]][:len(topk)]
