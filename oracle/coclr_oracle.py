"""CPU ORACLE -- test infrastructure, NOT part of the product path.

A plain PyTorch (fp32, CPU, ATen ops + autograd) restatement of the reference's
hot path: the S3D / ResNet2d3d-50 backbones, the projection head and one
InfoNCE / UberNCE / CoCLR training-step forward, written as pure functions over
a state dict that uses the reference's key names.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file;
coclr_amd/ never does.

Pinning: tests/golden/*.pt were produced by importing the *unmodified*
reference from /root/reference (oracle/make_golden.py, committed) and
tests/test_oracle_golden.py checks this restatement against every one of them,
so parity is pinned to outputs of the reference itself (the reference has no
tests or golden vectors of its own -- SURVEY.md section 4).

Every function cites the reference lines it restates (paths relative to the
reference root).
"""
import torch
import torch.nn.functional as F

BN_MOMENTUM = 0.1   # PyTorch defaults, backbone/s3dg.py:4,16
BN_EPS = 1e-5

# backbone/s3dg.py:163-192 -- (name, in_planes, [b0, b1a, b1b, b2a, b2b, b3b])
S3D_INCEPTIONS = [
    ("Mixed_3b", 192, [64, 96, 128, 16, 32, 32]),
    ("Mixed_3c", 256, [128, 128, 192, 32, 96, 64]),
    ("Mixed_4b", 480, [192, 96, 208, 16, 48, 64]),
    ("Mixed_4c", 512, [160, 112, 224, 24, 64, 64]),
    ("Mixed_4d", 512, [128, 128, 256, 24, 64, 64]),
    ("Mixed_4e", 512, [112, 144, 288, 32, 64, 64]),
    ("Mixed_4f", 528, [256, 160, 320, 32, 128, 128]),
    ("Mixed_5b", 832, [256, 160, 320, 32, 128, 128]),
    ("Mixed_5c", 832, [384, 192, 384, 48, 128, 128]),
]


def _bn(sd, pre, x, training):
    """nn.BatchNorm3d forward incl. running-stat update (backbone/s3dg.py:16,26)."""
    if training:
        sd[pre + ".num_batches_tracked"] += 1
    return F.batch_norm(x, sd[pre + ".running_mean"], sd[pre + ".running_var"],
                        sd[pre + ".weight"], sd[pre + ".bias"], training, BN_MOMENTUM, BN_EPS)


def basic_conv3d(sd, pre, x, training, stride=1, padding=0):
    """BasicConv3d.forward: conv -> bn -> relu (backbone/s3dg.py:24-28)."""
    x = F.conv3d(x, sd[pre + ".conv.weight"], None, stride, padding)
    return F.relu(_bn(sd, pre + ".bn", x, training))


def st_conv3d(sd, pre, x, training, stride, padding):
    """STConv3d.forward: (1,k,k) conv/bn/relu then (k,1,1) conv/bn/relu; an int stride
    applies to time and space, a tuple is (t, ., s) (backbone/s3dg.py:33-42,58-65)."""
    ts, ss = (stride[0], stride[-1]) if isinstance(stride, tuple) else (stride, stride)
    x = F.conv3d(x, sd[pre + ".conv1.weight"], None, (1, ss, ss), (0, padding, padding))
    x = F.relu(_bn(sd, pre + ".bn1", x, training))
    x = F.conv3d(x, sd[pre + ".conv2.weight"], None, (ts, 1, 1), (padding, 0, 0))
    return F.relu(_bn(sd, pre + ".bn2", x, training))


def self_gating(sd, pre, x):
    """SelfGating.forward (backbone/s3dg.py:73-78)."""
    w = torch.sigmoid(F.linear(x.mean(dim=[2, 3, 4]), sd[pre + ".fc.weight"], sd[pre + ".fc.bias"]))
    return w[:, :, None, None, None] * x


def sep_inception(sd, pre, x, training, gating):
    """SepInception.forward (backbone/s3dg.py:119-132)."""
    x0 = basic_conv3d(sd, pre + ".branch0.0", x, training)
    x1 = st_conv3d(sd, pre + ".branch1.1", basic_conv3d(sd, pre + ".branch1.0", x, training),
                   training, 1, 1)
    x2 = st_conv3d(sd, pre + ".branch2.1", basic_conv3d(sd, pre + ".branch2.0", x, training),
                   training, 1, 1)
    x3 = basic_conv3d(sd, pre + ".branch3.1", F.max_pool3d(x, (3, 3, 3), 1, 1), training)
    if gating:
        x0 = self_gating(sd, pre + ".gating_b0", x0)
        x1 = self_gating(sd, pre + ".gating_b1", x1)
        x2 = self_gating(sd, pre + ".gating_b2", x2)
        x3 = self_gating(sd, pre + ".gating_b3", x3)
    return torch.cat((x0, x1, x2, x3), 1)


def s3d_forward(sd, pre, x, training, gating=False):
    """S3D.forward (backbone/s3dg.py:143-192,211-217)."""
    x = st_conv3d(sd, pre + "Conv_1a", x, training, 2, 3)
    x = F.max_pool3d(x, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    x = basic_conv3d(sd, pre + "Conv_2b", x, training)
    x = st_conv3d(sd, pre + "Conv_2c", x, training, 1, 1)
    x = F.max_pool3d(x, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    for name, _, _ in S3D_INCEPTIONS:
        if name == "Mixed_4b":
            x = F.max_pool3d(x, (3, 3, 3), (2, 2, 2), (1, 1, 1))
        if name == "Mixed_5b":
            x = F.max_pool3d(x, (2, 2, 2), (2, 2, 2), (0, 0, 0))
        x = sep_inception(sd, pre + name, x, training, gating)
    return x


def _bottleneck(sd, pre, x, training, three_d, stride, has_down, final_relu):
    """Bottleneck2d/3d.forward (backbone/resnet_2d3d.py:67-86,110-129)."""
    out = F.conv3d(x, sd[pre + ".conv1.weight"], None, 1, (1, 0, 0) if three_d else 0)
    out = F.relu(_bn(sd, pre + ".bn1", out, training))
    out = F.conv3d(out, sd[pre + ".conv2.weight"], None, (1, stride, stride), (0, 1, 1))
    out = F.relu(_bn(sd, pre + ".bn2", out, training))
    out = F.conv3d(out, sd[pre + ".conv3.weight"], None, 1, 0)
    out = _bn(sd, pre + ".bn3", out, training)
    res = x
    if has_down:
        res = F.conv3d(x, sd[pre + ".downsample.0.weight"], None, (1, stride, stride), 0)
        res = _bn(sd, pre + ".downsample.1", res, training)
    out = out + res
    return F.relu(out) if final_relu else out


def r2d3d50_forward(sd, pre, x, training):
    """ResNet2d3d.forward for r2d3d50 = [2d,2d,3d,3d] x [3,4,6,3]
    (backbone/resnet_2d3d.py:138-149,191-202,206-210)."""
    x = F.conv3d(x, sd[pre + "conv1.weight"], None, (2, 2, 2), (2, 3, 3))
    x = F.relu(_bn(sd, pre + "bn1", x, training))
    x = F.max_pool3d(x, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    for li, (nblocks, three_d, stride) in enumerate(
            [(3, False, 1), (4, False, 2), (6, True, 2), (3, True, 2)], start=1):
        for b in range(nblocks):
            last = li == 4 and b == nblocks - 1
            x = _bottleneck(sd, "%slayer%d.%d" % (pre, li, b), x, training, three_d,
                            stride if b == 0 else 1, b == 0, not last)
    return F.relu(x)


def backbone_forward(network, sd, pre, x, training):
    """select_backbone dispatch (backbone/select_backbone.py:4-16)."""
    if network == "s3d":
        return s3d_forward(sd, pre, x, training, False)
    if network == "s3dg":
        return s3d_forward(sd, pre, x, training, True)
    if network == "r50":
        return r2d3d50_forward(sd, pre, x, training)
    raise NotImplementedError


def encoder_forward(network, sd, pre, x, training):
    """backbone -> AdaptiveAvgPool3d(1) -> conv1x1x1+b -> ReLU -> conv1x1x1+b
    (model/pretrain.py:49-54), returns (B, dim, 1, 1, 1)."""
    f = backbone_forward(network, sd, pre + "0.", x, training)
    f = F.adaptive_avg_pool3d(f, (1, 1, 1))
    f = F.relu(F.conv3d(f, sd[pre + "2.weight"], sd[pre + "2.bias"]))
    return F.conv3d(f, sd[pre + "4.weight"], sd[pre + "4.bias"])


def momentum_update(sd, m):
    """_momentum_update_key_encoder over parameters only (model/pretrain.py:76-80)."""
    for k in list(sd.keys()):
        if k.startswith("encoder_k.") and not _is_buffer(k):
            q = sd["encoder_q." + k[len("encoder_k."):]]
            sd[k] = sd[k] * m + q.detach() * (1. - m)


def _is_buffer(key):
    return key.endswith(("running_mean", "running_var", "num_batches_tracked")) or \
        key.split(".")[0].startswith("queue")


def nce_step(sd, kind, network, blocks, extra, dim, K, m, T, perm, topk=5, reverse=False,
             queue_full=None, training=True, sampler_training=False, world=1):
    """One forward of InfoNCE / UberNCE / CoCLR for a `world`-rank data-parallel job
    emulated in one process (model/pretrain.py:145-190, 230-278, 344-418).

    sd      : state dict (reference key names). encoder_q.* entries that require grad
              take part in autograd; buffers are updated in place like the module does.
    blocks  : per-rank list of inputs. InfoNCE/UberNCE: block (B,2,C,T,H,W);
              CoCLR: (block1, block2).
    extra   : per-rank list of k_label (UberNCE) / k_vsource (CoCLR) or None.
    perm    : the randperm(B*world) of _batch_shuffle_ddp (pretrain.py:112).
    Returns per-rank (logits, target) and mutates sd (queues, ptr, BN buffers, encoder_k).
    BN running statistics follow DDP(broadcast_buffers=True): every rank starts from
    rank 0's buffers and rank 0's updates survive.
    """
    def split(b):
        if kind == "coclr":
            b1, b2 = b
            x1, f1, x2, f2 = b1[:, 0], b1[:, 1], b2[:, 0], b2[:, 1]
            if reverse:
                x1, f1, x2, f2 = f1, x1, f2, x2
            return x1, x2, f2
        return b[:, 0], b[:, 1], None

    parts = [split(b) for b in blocks]
    B = parts[0][0].shape[0]

    def with_rank_buffers(rank, fn):
        """Run fn on a copy of the BN buffers unless rank == 0."""
        if rank == 0:
            return fn(sd)
        local = dict(sd)
        for k in sd:
            if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
                local[k] = sd[k].clone()
        return fn(local)

    # queries (pretrain.py:153-155)
    qs = []
    for r in range(world):
        q = with_rank_buffers(r, lambda d, r=r: encoder_forward(network, d, "encoder_q.",
                                                                parts[r][0], training))
        qs.append(F.normalize(q, dim=1).view(B, dim))
    in_train_mode = qs[0].requires_grad

    with torch.no_grad():
        if in_train_mode:
            momentum_update(sd, m)
        # shuffle BN (pretrain.py:98-124)
        x_gather = torch.cat([p[1] for p in parts], 0)
        idx_shuffle = perm
        idx_unshuffle = torch.argsort(idx_shuffle)
        ks_shuf = []
        for r in range(world):
            idx_this = idx_shuffle.view(world, -1)[r]
            k = with_rank_buffers(r, lambda d, i=idx_this: encoder_forward(
                network, d, "encoder_k.", x_gather[i], training))
            ks_shuf.append(F.normalize(k, dim=1).view(B, dim))
        k_gather = torch.cat(ks_shuf, 0)
        ks = [k_gather[idx_unshuffle.view(world, -1)[r]] for r in range(world)]
        kfs = None
        if kind == "coclr":
            kfs = []
            for r in range(world):
                kf = with_rank_buffers(r, lambda d, r=r: encoder_forward(
                    network, d, "sampler.", parts[r][2], sampler_training))
                kfs.append(F.normalize(kf, dim=1).view(B, dim))

    outs = []
    queue = sd["queue"].clone().detach()
    for r in range(world):
        l_pos = torch.einsum('nc,nc->n', [qs[r], ks[r]]).unsqueeze(-1)
        l_neg = torch.einsum('nc,ck->nk', [qs[r], queue])
        logits = torch.cat([l_pos, l_neg], dim=1) / T
        if kind == "infonce":
            target = torch.zeros(B, dtype=torch.long)
        elif kind == "ubernce":
            mask = extra[r].unsqueeze(1) == sd["queue_label"].unsqueeze(0)
            target = torch.cat([torch.ones(B, 1, dtype=torch.bool), mask], 1)
        else:
            mask_source = extra[r].unsqueeze(1) == sd["queue_vname"].unsqueeze(0)
            mask = mask_source.clone()
            full = bool(torch.all(sd["queue_label"] != -1)) if queue_full is None else queue_full
            if full and topk != 0:
                sim = kfs[r].matmul(sd["queue_second"])
                sim[mask_source] = -float("inf")
                _, idx = torch.topk(sim, topk, dim=1)
                onehot = torch.zeros_like(sim)
                onehot.scatter_(1, idx, 1)
                mask[onehot.bool()] = True
            target = torch.cat([torch.ones(B, 1, dtype=torch.bool), mask], 1)
        outs.append((logits, target))

    # dequeue / enqueue (pretrain.py:82-96, 207-224, 321-341)
    if in_train_mode:
        with torch.no_grad():
            keys = torch.cat(ks, 0)
            bs = keys.shape[0]
            ptr = int(sd["queue_ptr"])
            assert K % bs == 0
            sd["queue"][:, ptr:ptr + bs] = keys.T
            if kind == "ubernce":
                sd["queue_label"][ptr:ptr + bs] = torch.cat(extra, 0)
            if kind == "coclr":
                sd["queue_second"][:, ptr:ptr + bs] = torch.cat(kfs, 0).T
                vn = torch.cat(extra, 0)
                sd["queue_vname"][ptr:ptr + bs] = vn
                sd["queue_label"][ptr:ptr + bs] = torch.ones_like(vn)
            sd["queue_ptr"][0] = (ptr + bs) % K
    return outs


def multi_nce_loss(logits, mask):
    """main_coclr.py:343-346 (also the UberNCE loss form used at main_nce.py:320-321
    differs: see ubernce_loss)."""
    mask_sum = mask.sum(1)
    loss = - torch.log((F.softmax(logits, dim=1) * mask).sum(1))
    return loss.mean()


def ubernce_loss(logits, mask):
    """main_nce.py:320-321."""
    loss = - (F.log_softmax(logits, dim=1) * mask).sum(1) / mask.sum(1)
    return loss.mean()


def training_state(sd, requires_grad_prefix="encoder_q."):
    """Detach-clone a state dict (alias structure preserved); float tensors under
    `requires_grad_prefix` that are parameters (not BN buffers) become autograd leaves."""
    out, memo = {}, {}
    for k, v in sd.items():
        # S3D registers every stage twice (Conv_1a.* and block1.0.*, backbone/s3dg.py:145-150):
        # keys that alias one tensor must keep aliasing one clone
        ident = (v.data_ptr(), tuple(v.shape), v.dtype) if v.numel() else None
        if ident is not None and ident in memo:
            out[k] = memo[ident]
            continue
        v = v.detach().clone()
        if k.startswith(requires_grad_prefix) and v.is_floating_point() and not _is_buffer(k):
            v.requires_grad_(True)
        out[k] = v
        if ident is not None:
            memo[ident] = v
    return out


# ------------------------------------------------------------------------------------
# Neighbours of the model in the training / evaluation loops.  Pinned by
# tests/golden/next_*.pt, generated by oracle/make_golden_next.py from the reference's own
# functions where they are importable (utils/utils.py, utils/transforms.py, model/classifier.py,
# main_coclr.py::multi_nce_loss) and from torch.optim.Adam itself.
# ------------------------------------------------------------------------------------

def masked_nce_loss_drop_self(logits, mask):
    """main_coclr.py:384-389, the branch taken 90 % of the time: positives other than the clip's
    own key exist -> leave column 0 out of that row's loss."""
    mask_clone = mask.clone()
    mask_clone[mask.sum(1) != 1, 0] = 0
    return multi_nce_loss(logits, mask_clone)


def calc_topk_accuracy(output, target, topk=(1,)):
    """utils/utils.py:52-69."""
    maxk = max(topk)
    batch_size = target.size(0)
    _, pred = output.topk(maxk, 1, True, True)
    pred = pred.t()
    correct = pred.eq(target.view(1, -1).expand_as(pred))
    return [correct[:k].reshape(-1).float().sum(0) * (1 / batch_size) for k in topk]


def calc_mask_accuracy(output, target_mask, topk=(1,)):
    """utils/utils.py:71-85."""
    maxk = max(topk)
    _, pred = output.topk(maxk, 1, True, True)
    zeros = torch.zeros_like(target_mask).long()
    pred_mask = torch.zeros_like(target_mask).long()
    res = []
    for k in range(maxk):
        onehot = zeros.scatter(1, pred[:, k].unsqueeze(1), 1)
        pred_mask = onehot + pred_mask
        if k + 1 in topk:
            res.append(((pred_mask * target_mask).sum(1) >= 1).float().mean(0))
    return res


def tr(frames, num_seq, seq_len, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """Input staging: ToTensor's /255 when the frames are uint8 (utils/transforms.py:49-51 /
    torchvision ToTensor), T.Normalize(mean, std, channel=1) (utils/transforms.py:57-63 via
    main_nce.py:207-209) and the view/transpose/contiguous of main_nce.py:299-302."""
    x = frames.to(torch.float32) / 255 if frames.dtype == torch.uint8 else frames
    shape = [1] * x.dim()
    shape[1] = -1
    x = (x - torch.as_tensor(mean).reshape(shape)) / torch.as_tensor(std).reshape(shape)
    B = x.size(0)
    return x.view(B, 3, num_seq, seq_len, x.shape[-2], x.shape[-1]).transpose(1, 2).contiguous()


def adam_step(params, grads, exp_avgs, exp_avg_sqs, steps, lr, beta1, beta2, eps, weight_decay):
    """torch.optim.Adam (amsgrad=False, maximize=False) as main_nce.py:200,331 uses it -- the
    arithmetic lives in the un-vendored dependency (PyTorch, torch/optim/adam.py
    `_single_tensor_adam`); restated here in float64 as the tolerance anchor and pinned against
    torch.optim.Adam itself by tests/test_oracle_golden.py.  In place; `steps` is a list of ints."""
    for i, (p, g, m, v) in enumerate(zip(params, grads, exp_avgs, exp_avg_sqs)):
        steps[i] += 1
        t = steps[i]
        if weight_decay != 0:
            g = g + weight_decay * p
        m.mul_(beta1).add_(g, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        bc1, bc2 = 1 - beta1 ** t, 1 - beta2 ** t
        denom = v.sqrt() / (bc2 ** 0.5) + eps
        p.addcdiv_(m, denom, value=-lr / bc1)


def linear_classifier_forward(sd, network, block, training, use_l2_norm=False, use_final_bn=False,
                              fc_key="final_fc.1"):
    """LinearClassifier.forward without dropout noise (model/classifier.py:47-61): backbone ->
    global average pool -> [L2 normalise] -> [BatchNorm1d] -> Linear.  fc_key is 'final_fc.1'
    with the Dropout module in front (classifier.py:39-41), 'final_fc.0' without (:43-44)."""
    feat = backbone_forward(network, sd, "backbone.", block, training)
    feat = F.adaptive_avg_pool3d(feat, (1, 1, 1)).view(block.shape[0], -1)
    if use_l2_norm:
        feat = F.normalize(feat, p=2, dim=1)
    x = feat
    if use_final_bn:
        if training:
            sd["final_bn.num_batches_tracked"] += 1
        x = F.batch_norm(x, sd["final_bn.running_mean"], sd["final_bn.running_var"],
                         sd["final_bn.weight"], sd["final_bn.bias"], training, BN_MOMENTUM, BN_EPS)
    logit = F.linear(x, sd[fc_key + ".weight"], sd[fc_key + ".bias"])
    return logit, feat


def nn_retrieval(test_feature, test_label, train_feature, train_label, ks=(1, 5, 10, 20, 50)):
    """eval/main_classifier.py:686-706 (inline code of the retrieval test, not a function there):
    centre, L2-normalise, dot product, k-NN label match.  Returns (accuracies, sim)."""
    test_feature = test_feature - test_feature.mean(dim=0, keepdim=True)
    train_feature = train_feature - train_feature.mean(dim=0, keepdim=True)
    test_feature = F.normalize(test_feature, p=2, dim=1)
    train_feature = F.normalize(train_feature, p=2, dim=1)
    sim = test_feature.matmul(train_feature.t())
    acc = []
    for k in ks:
        topkval, topkidx = torch.topk(sim, k, dim=1)
        acc.append(torch.any(train_label[topkidx] == test_label.unsqueeze(1), dim=1).float().mean().item())
    return acc, sim
