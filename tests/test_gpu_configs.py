"""One training step of BASELINE.json configs 3, 4 and 5 at the benchmarked per-GPU sizes (B=32
clips of 3x32x128x128) on the GPU against the CPU oracle run on this host's cores with the same
state, inputs and permutation -- the single-rank content of those configurations (their 8-rank
exchange is covered by tests/test_host_cpu.py over gloo and tests/test_gpu_multirank.py):

  config 3  S3D InfoNCE, moco-k=16384                          (model/pretrain.py:145-190)
  config 4  S3D CoCLR two-stream, moco-k=2048, topk=5, queue full: frozen sampler, cross-modal
            top-k mining, three queues                         (model/pretrain.py:344-418)
  config 5  ResNet2d3d-50 InfoNCE, moco-k=16384                (backbone/resnet_2d3d.py:133-202)

Bars: the north star's 1e-3 on logits / enqueued keys / BatchNorm running statistics, exact labels,
pointer, untouched queue columns and (CoCLR) positive mask.
"""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

B = 32
CLIP = (3, 32, 128, 128)


def _oracle_threads():
    return max(1, min(32, os.cpu_count() or 1))


def _run_oracle(fn):
    threads = torch.get_num_threads()
    torch.set_num_threads(_oracle_threads())
    try:
        return fn()
    finally:
        torch.set_num_threads(threads)


def _check_common(model, ref_sd, logits, ref_logits, K, keys):
    from _cases import check_close
    check_close(logits, ref_logits, 1e-3, "logits")
    sd = model.state_dict()
    assert int(sd["queue_ptr"]) == int(ref_sd["queue_ptr"]) == B
    check_close(sd["queue"][:, :B], ref_sd["queue"][:, :B], 1e-3, "enqueued keys")
    assert torch.equal(sd["queue"][:, B:].cpu(), ref_sd["queue"][:, B:])      # untouched columns
    for k in keys:
        check_close(sd[k], ref_sd[k], 1e-3, k)
    return sd


def test_config3_infonce_k16384_step_matches_oracle():
    import model.pretrain as product
    from oracle import coclr_oracle as orc
    K = 16384
    torch.manual_seed(0)
    model = product.InfoNCE('s3d', 128, K, 0.999, 0.07)
    ref_sd = orc.training_state(model.state_dict())
    model = model.cuda().train()
    torch.manual_seed(11)
    block = torch.randn(B, 2, *CLIP)
    torch.manual_seed(12)
    perm = torch.randperm(B)
    torch.manual_seed(12)
    logits, labels = model(block.cuda())
    loss = F.cross_entropy(logits, labels)
    loss.backward()
    torch.cuda.synchronize()

    def ref():
        (rl, rt), = orc.nce_step(ref_sd, "infonce", "s3d", [block], None, 128, K, 0.999, 0.07, perm)
        rloss = F.cross_entropy(rl, rt)
        rloss.backward()
        return rl, rt, rloss
    ref_logits, ref_labels, ref_loss = _run_oracle(ref)
    assert logits.shape == (B, 1 + K)
    assert torch.equal(labels.cpu(), ref_labels)
    assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 1e-3 * max(1.0, float(ref_logits.abs().max()))
    _check_common(model, ref_sd, logits, ref_logits, K,
                  ("encoder_q.0.Conv_2c.bn1.running_var", "encoder_k.0.Mixed_4f.branch1.1.bn2.running_mean",
                   "encoder_k.2.weight"))
    from _cases import check_close
    check_close(model.encoder_q[4].weight.grad, ref_sd["encoder_q.4.weight"].grad, 5e-3,
                "head weight gradient")


def test_config4_coclr_two_stream_step_matches_oracle():
    """Queue pre-filled (`queue_label != -1` everywhere, random sources): the mining branch of
    model/pretrain.py:403-413 is live.  The mask must equal the oracle's except where the oracle's
    own top-k decision is a near tie in fp32 (none expected; the assertion message says if so)."""
    import model.pretrain as product
    from oracle import coclr_oracle as orc
    from _cases import check_close
    K, topk, n_sources = 2048, 5, 300
    torch.manual_seed(0)
    model = product.CoCLR('s3d', 128, K, 0.999, 0.07, topk=topk)
    g = torch.Generator().manual_seed(5)
    model.queue_label.fill_(1)
    model.queue_vname.copy_(torch.randint(0, n_sources, (K,), generator=g))
    ref_sd = orc.training_state(model.state_dict())
    queue_second_before = ref_sd["queue_second"].clone()
    vname_before = ref_sd["queue_vname"].clone()
    model = model.cuda().train()
    model.sampler.eval()                                  # main_coclr.py:363
    torch.manual_seed(21)
    block1 = torch.randn(B, 2, *CLIP)
    block2 = torch.randn(B, 2, *CLIP)
    vsrc = torch.randint(0, n_sources, (B,))
    torch.manual_seed(22)
    perm = torch.randperm(B)
    torch.manual_seed(22)
    logits, mask = model(block1.cuda(), block2.cuda(), vsrc.cuda())
    assert model.queue_is_full is True
    loss = (- torch.log((F.softmax(logits, dim=1) * mask).sum(1))).mean()
    loss.backward()
    torch.cuda.synchronize()

    def ref():
        (rl, rm), = orc.nce_step(ref_sd, "coclr", "s3d", [(block1, block2)], [vsrc], 128, K, 0.999,
                                 0.07, perm, topk=topk)
        rloss = (- torch.log((F.softmax(rl, dim=1) * rm).sum(1))).mean()
        rloss.backward()
        return rl, rm, rloss
    ref_logits, ref_mask, ref_loss = _run_oracle(ref)
    sd = _check_common(model, ref_sd, logits, ref_logits, K,
                       ("encoder_q.0.Conv_1a.bn2.running_mean", "encoder_k.0.Mixed_3c.branch2.1.bn1.running_var"))
    # frozen sampler in eval mode: its statistics must not move
    assert int(sd["sampler.0.Conv_2b.bn.num_batches_tracked"]) == 0
    # the three side queues (pretrain.py:321-341)
    check_close(sd["queue_second"][:, :B], ref_sd["queue_second"][:, :B], 1e-3, "queue_second keys")
    assert torch.equal(sd["queue_second"][:, B:].cpu(), ref_sd["queue_second"][:, B:])
    assert torch.equal(sd["queue_vname"].cpu(), ref_sd["queue_vname"])
    assert torch.equal(sd["queue_label"].cpu(), ref_sd["queue_label"])
    # positive mask: siblings + cross-modal top-k
    got = mask.cpu()
    assert got.dtype == torch.bool and got.shape == (B, 1 + K)
    assert bool(got[:, 0].all())
    if not torch.equal(got, ref_mask):
        sim = ref_sd["queue_second"][:, :B].t().matmul(queue_second_before)   # oracle's kf . queue_second
        sim[vsrc[:, None] == vname_before[None, :]] = -float("inf")
        kth = torch.topk(sim, topk + 1, dim=1).values
        gap = float((kth[:, topk - 1] - kth[:, topk]).min())
        diff = (got != ref_mask).nonzero()
        raise AssertionError("positive mask differs in %d places (smallest k-th / (k+1)-th similarity "
                             "gap of the oracle: %.2e): %s" % (len(diff), gap, diff[:8].tolist()))
    assert int(got.sum(1).min()) >= 1 + topk
    assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 2e-3 * max(1.0, abs(float(ref_loss.detach())))
    check_close(model.encoder_q[4].weight.grad, ref_sd["encoder_q.4.weight"].grad, 5e-3,
                "head weight gradient")


def test_config5_r50_k16384_step_matches_oracle():
    """ResNet2d3d-50 at the benchmarked clip size and batch (B = 32 per GPU, BASELINE config 5)."""
    import model.pretrain as product
    from oracle import coclr_oracle as orc
    from _cases import check_close
    K, Bs = 16384, B
    torch.manual_seed(0)
    model = product.InfoNCE('r50', 128, K, 0.999, 0.07)
    ref_sd = orc.training_state(model.state_dict())
    model = model.cuda().train()
    torch.manual_seed(31)
    block = torch.randn(Bs, 2, *CLIP)
    torch.manual_seed(32)
    perm = torch.randperm(Bs)
    torch.manual_seed(32)
    logits, labels = model(block.cuda())
    loss = F.cross_entropy(logits, labels)
    loss.backward()
    torch.cuda.synchronize()

    def ref():
        (rl, rt), = orc.nce_step(ref_sd, "infonce", "r50", [block], None, 128, K, 0.999, 0.07, perm)
        rloss = F.cross_entropy(rl, rt)
        rloss.backward()
        return rl, rt, rloss
    ref_logits, ref_labels, ref_loss = _run_oracle(ref)
    check_close(logits, ref_logits, 1e-3, "logits")
    assert torch.equal(labels.cpu(), ref_labels)
    sd = model.state_dict()
    assert int(sd["queue_ptr"]) == int(ref_sd["queue_ptr"]) == Bs
    check_close(sd["queue"][:, :Bs], ref_sd["queue"][:, :Bs], 1e-3, "enqueued keys")
    assert torch.equal(sd["queue"][:, Bs:].cpu(), ref_sd["queue"][:, Bs:])
    for k in ("encoder_q.0.bn1.running_var", "encoder_q.0.layer4.2.bn3.running_mean",
              "encoder_k.0.layer2.0.downsample.1.running_var", "encoder_k.0.layer3.5.conv1.weight"):
        check_close(sd[k], ref_sd[k], 1e-3, k)
    assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 1e-3 * max(1.0, float(ref_logits.abs().max()))
    check_close(model.encoder_q[4].weight.grad, ref_sd["encoder_q.4.weight"].grad, 5e-3,
                "head weight gradient")
