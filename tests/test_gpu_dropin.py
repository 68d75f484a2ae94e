"""The launch scripts' call sequence on the HIP kernels (tests/_caller_loop.py: proven identical to the
UNMODIFIED main_nce.py / main_coclr.py by tests/test_dropin_scripts.py in the build container; this
box has no /root/reference) against the fixture recorded from the reference's own scripts AND model
(oracle/make_golden_dropin.py): `InfoNCE(...)`, `.cuda(gpu)`, `DistributedDataParallel(model,
device_ids=[gpu])` over a 1-rank RCCL group, Adam over one param group per tensor, `nn.CrossEntropyLoss`,
DistributedSampler-ordered batches, the accuracy helpers, the `.item()` reads, checkpointable state."""
import io

import pytest
import torch

from _cases import check_close, load_golden

pytestmark = pytest.mark.gpu


def _ensure_pg():
    import os
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29611")
        dist.init_process_group("nccl", rank=0, world_size=1)


@pytest.mark.parametrize("name", ["dropin_main_nce", "dropin_main_coclr"])
def test_caller_sequence_on_gpu_matches_reference_scripts(name, tmp_path):
    import dropin_harness as H
    import _caller_loop
    import model.pretrain as product
    import coclr_amd.optim as native
    from oracle import coclr_oracle as orc
    _ensure_pg()
    gold = load_golden(name)
    ds = H.SyntheticClips(**gold["dataset"])
    two_stream = gold["script"] == "main_coclr"
    if two_stream:
        # main_coclr.py --pretrain <rgb> <flow>: well-conditioned encoders (dropin_harness.write_pretrained)
        H.write_pretrained_pair(str(tmp_path), use_reference_model=False, product=product)
        rec = _caller_loop.run_coclr(product, ds, gpu=0, calc_topk_accuracy=orc.calc_topk_accuracy,
                                     calc_mask_accuracy=orc.calc_mask_accuracy,
                                     pretrain=(str(tmp_path / "rgb.pth.tar"), str(tmp_path / "flow.pth.tar")))
    else:
        rec = _caller_loop.run_nce(product, ds, gpu=0, calc_topk_accuracy=orc.calc_topk_accuracy)
    assert isinstance(rec["optimizer"], native.Adam) and rec["optimizer"]._plan is not None
    n = len(gold["outputs"])
    assert len(rec["outputs"]) == n
    first_update = 3 if two_stream else 1               # CoCLR trains once its queue is full
    for i in range(n):
        tgt = rec["targets"][i]
        tgt = tgt.nonzero() if tgt.dtype == torch.bool else tgt
        if i < first_update:
            # forward passes at the reference's initial weights: the north star's 1e-3
            check_close(rec["outputs"][i], gold["outputs"][i], 1e-3, "logits of iteration %d" % i)
            assert torch.equal(tgt, gold["targets"][i]), "targets of iteration %d" % i
            assert abs(rec["losses"][i] - gold["losses"][i]) <= 5e-3 * max(1.0, abs(gold["losses"][i]))
        else:
            # after an Adam step at initialisation no build follows the reference's trajectory tightly
            # (tests/test_host_cpu.py explains): this bar is SHAPE / BALL PARK ONLY -- the parity of an
            # iteration that starts from updated weights is held at 1e-3 against the oracle continued from the
            # product's own state (tests/test_gpu_model.py::_replay, step 2)
            check_close(rec["outputs"][i], gold["outputs"][i], 0.25,
                        "logits of iteration %d (ball-park bar only, see the comment above)" % i)
    # what the script would checkpoint (main_nce.py:278-290): reference keys, loadable by torch.save/load
    model, opt = rec["model"], rec["optimizer"]
    sd = model.state_dict()
    assert list(sd.keys()) == gold["state_keys"]
    for k in ("queue_ptr", "queue_vname", "queue_label"):
        if k in gold["state"]:
            assert torch.equal(sd[k].cpu(), gold["state"][k]), k
    osd = opt.state_dict()
    assert len(osd["param_groups"]) == gold["optimizer_groups"]
    assert len(osd["state"]) == gold["optimizer_state_entries"]
    buf = io.BytesIO()
    torch.save({"state_dict": sd, "optimizer": osd}, buf)       # the flat buffers must be saveable
    buf.seek(0)
    back = torch.load(buf, map_location="cpu", weights_only=False)
    assert torch.equal(back["state_dict"]["queue"], sd["queue"].cpu())
