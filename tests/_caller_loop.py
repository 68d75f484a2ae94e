"""The call sequence of the reference's launch scripts, restated (test infrastructure).

`run_nce` / `run_coclr` make, in the same order, the calls `main_worker` + `train_one_epoch` of
main_nce.py:124-353 / main_coclr.py:140-409 make on the model, DistributedDataParallel, the optimiser,
the criterion, the accuracy helpers and the meters' `.item()` reads.  The GPU box has no /root/reference, so the
scripts themselves cannot be imported there; tests/test_dropin_scripts.py proves (in the build
container, on the CPU double) that this restatement and the unmodified scripts produce IDENTICAL
logits, targets and losses iteration by iteration, and tests/test_gpu_dropin.py then runs it on the
HIP kernels against the fixture recorded from the reference's own scripts and model.
"""
import random

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _tr(x, transforms_mean_std, num_seq, seq_len, img_dim):
    # T.Normalize(mean, std, channel=1) then main_nce.py:299-302
    mean, std = transforms_mean_std
    shape = [1, -1, 1, 1, 1]
    x = (x - torch.as_tensor(mean, device=x.device).view(shape)) / \
        torch.as_tensor(std, device=x.device).view(shape)
    B = x.size(0)
    return x.view(B, 3, num_seq, seq_len, img_dim, img_dim).transpose(1, 2).contiguous()


def _loader(dataset, batch_size, pin):
    """get_dataloader (main_nce.py:413-423): DistributedSampler(shuffle=True) + drop_last batches of
    FastDataLoader (utils/utils.py:247-260), whose iterator is created once, at construction -- that
    draws the loader's base seed from the global RNG, which the model's shuffle permutation
    (model/pretrain.py:112) is drawn from afterwards."""
    sampler = torch.utils.data.distributed.DistributedSampler(dataset, shuffle=True)
    loader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=False, num_workers=0,
                                         pin_memory=pin, sampler=sampler, drop_last=True)
    it = iter(loader)
    sampler.set_epoch(0)
    return (next(it) for _ in range(len(loader)))


NORM = ([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])


def _setup(model, seed, gpu, lr, wd, ddp_kwargs):
    if gpu is not None:
        torch.cuda.set_device(gpu)
        model.cuda(gpu)
    model = torch.nn.parallel.DistributedDataParallel(model, **ddp_kwargs)
    params = [{'params': p} for _, p in model.named_parameters()]
    optimizer = torch.optim.Adam(params, lr=lr, weight_decay=wd)
    criterion = nn.CrossEntropyLoss()
    if gpu is not None:
        criterion = criterion.cuda(gpu)
    return model, optimizer, criterion


def run_nce(product, dataset, *, net="s3d", moco_k=32, batch_size=4, seq_len=16, img_dim=64, seed=0,
            lr=1e-3, wd=1e-5, gpu=None, calc_topk_accuracy=None):
    """main_nce.py with --model infonce, one epoch."""
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    model = product.InfoNCE(net, 128, moco_k, 0.999, 0.07)
    model, optimizer, criterion = _setup(model, seed, gpu, lr, wd,
                                         dict(device_ids=[gpu]) if gpu is not None else {})
    rec = {"outputs": [], "targets": [], "losses": []}
    batches = _loader(dataset, batch_size, gpu is not None)
    np.random.seed(0)
    random.seed(0)
    for g in optimizer.param_groups:
        g['lr'] = lr
    model.train()
    for input_seq, label in batches:
        B = input_seq.size(0)
        if gpu is not None:
            input_seq = input_seq.cuda(non_blocking=True)
        input_seq = _tr(input_seq, NORM, 2, seq_len, img_dim)
        output, target = model(input_seq)
        loss = criterion(output, target)
        top1, top5 = calc_topk_accuracy(output, target, (1, 5))
        rec["outputs"].append(output.detach().cpu().clone())
        rec["targets"].append(target.detach().cpu().clone())
        top1.item(), top5.item()
        rec["losses"].append(loss.item())
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
    rec["model"] = model.module
    rec["optimizer"] = optimizer
    return rec


def _load_pretrained_pair(model_without_ddp, pretrain):
    """main_coclr.py:251-300: the FIRST checkpoint's encoder_q initialises encoder_q AND encoder_k, the
    SECOND checkpoint's encoder_q becomes the frozen sampler; queues are never loaded; then
    utils.neq_load_customized: update the model's own state dict and load it back."""
    second = torch.load(pretrain[1], map_location=torch.device('cpu'), weights_only=False)['state_dict']
    second = {k.replace('encoder_q.', 'sampler.'): v for k, v in second.items() if 'encoder_q.' in k}
    second = {k: v for k, v in second.items() if 'queue' not in k}
    first = torch.load(pretrain[0], map_location=torch.device('cpu'), weights_only=False)['state_dict']
    first = {k: v for k, v in first.items() if 'queue' not in k}
    both = {}
    for k, v in first.items():
        if 'encoder_q.' in k:
            both[k] = v
            both[k.replace('encoder_q.', 'encoder_k.')] = v
    state_dict = {**both, **second}
    state_dict.pop('queue_label', None)
    model_dict = model_without_ddp.state_dict()
    model_dict.update({k: v for k, v in state_dict.items() if k in model_dict})
    model_without_ddp.load_state_dict(model_dict)


def run_coclr(product, dataset, *, net="s3d", moco_k=8, topk=2, batch_size=4, seq_len=8, img_dim=64,
              seed=0, lr=1e-3, wd=1e-5, gpu=None, calc_topk_accuracy=None, calc_mask_accuracy=None,
              pretrain=None):
    """main_coclr.py, one epoch, --pretrain <rgb checkpoint> <flow checkpoint>."""
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    model = product.CoCLR(net, 128, moco_k, 0.999, 0.07, topk=topk, reverse=False)
    model, optimizer, criterion = _setup(model, seed, gpu, lr, wd,
                                         dict(device_ids=[gpu]) if gpu is not None else {})
    rec = {"outputs": [], "targets": [], "losses": []}
    batches = _loader(dataset, batch_size, gpu is not None)
    if pretrain is not None:
        _load_pretrained_pair(model.module, pretrain)
    np.random.seed(0)
    random.seed(0)
    for g in optimizer.param_groups:
        g['lr'] = lr
    model.train()
    model.module.sampler.eval()
    for input_seq, vname, _ in batches:
        B = input_seq[0].size(0)
        if gpu is not None:
            input_seq = [i.cuda(non_blocking=True) for i in input_seq]
            vname = vname.cuda(non_blocking=True)
        input_seq = [_tr(i, NORM, 2, seq_len, img_dim) for i in input_seq]
        output, mask = model(*input_seq, vname)
        mask_sum = mask.sum(1)
        if random.random() < 0.9:
            mask_clone = mask.clone()
            mask_clone[mask_sum != 1, 0] = 0
            loss = - torch.log((F.softmax(output, dim=1) * mask_clone).sum(1)).mean()
        else:
            loss = - torch.log((F.softmax(output, dim=1) * mask).sum(1)).mean()
        top1, top5 = calc_mask_accuracy(output, mask, (1, 5))
        zeros = torch.zeros(B, dtype=torch.long)
        top1_self, top5_self = calc_topk_accuracy(output, zeros.cuda() if gpu is not None else zeros, (1, 5))
        rec["outputs"].append(output.detach().cpu().clone())
        rec["targets"].append(mask.detach().cpu().clone())
        rec["losses"].append(loss.item())
        top1.item(), top5.item(), top1_self.item(), top5_self.item()
        if model.module.queue_is_full:
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
        del loss
    rec["model"] = model.module
    rec["optimizer"] = optimizer
    return rec
