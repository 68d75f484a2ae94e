"""The reference's launch scripts main_nce.py / main_coclr.py executed UNMODIFIED (imported from
/root/reference) on this repository's shadow packages `model/` and `backbone/`: north_star's "drop in
unchanged", run rather than argued.  Their real `main_worker()` is driven: constructor call,
`.cuda(gpu)`, `DistributedDataParallel(model, device_ids=[gpu])`, Adam over one param group per
tensor, `nn.CrossEntropyLoss`, FastDataLoader + DistributedSampler, `train_one_epoch`, the accuracy
helpers of utils/utils.py, `.item()` meters, checkpoint save (tests/dropin_harness.py lists what the
harness has to supply in an image without torchvision / tensorboardX / lmdb / a GPU).  Kernels are
the ATen double here (CPU tier); the fixture is the same scripts on the reference's OWN model
(oracle/make_golden_dropin.py).  Skipped where /root/reference does not exist (the GPU box)."""
import pytest
import torch

import dropin_harness as H
import fake_backend
from _cases import check_close, load_golden

pytestmark = pytest.mark.skipif(not H.reference_available(), reason="needs /root/reference")


@pytest.fixture
def fake(monkeypatch):
    fake_backend.install(monkeypatch)


@pytest.fixture(autouse=True)
def _leave_a_core_free():
    """The scripts run in-process next to their own helper threads (DDP's reducer, the autograd thread, the
    reference's plotter thread): with one ATen intra-op thread per core every OpenMP barrier spins until the
    scheduler lets the displaced thread back in, and these four tests took anything between 1 and 22 minutes on
    the 8-core container.  Half the cores for ATen keeps them at their fast end."""
    n = torch.get_num_threads()
    torch.set_num_threads(max(1, min(n, 4)))
    yield
    torch.set_num_threads(n)


def _compare(gold, rec, two_stream):
    n = len(gold["outputs"])
    assert len(rec["outputs"]) == len(rec["losses"]) == n
    for i in range(n):
        tgt = rec["targets"][i]
        tgt = tgt.nonzero() if tgt.dtype == torch.bool else tgt
        # iteration 0 is the forward at initialisation: tight.  From the first optimiser step on
        # Adam at initialisation turns fp32 round-off into +-lr weight changes (tests/test_host_cpu.py),
        # so the reference's own trajectory can only be followed loosely by ANY other build.
        first_update = 3 if two_stream else 1           # CoCLR trains once its queue is full
        if i < first_update:
            check_close(rec["outputs"][i], gold["outputs"][i], 1e-3, "logits of iteration %d" % i)
            assert torch.equal(tgt, gold["targets"][i]), "targets of iteration %d" % i
            assert abs(rec["losses"][i] - gold["losses"][i]) <= 5e-3 * max(1.0, abs(gold["losses"][i]))
        else:
            assert rec["outputs"][i].shape == gold["outputs"][i].shape
            check_close(rec["outputs"][i], gold["outputs"][i], 0.25, "logits of iteration %d" % i)
    ck = rec["checkpoint"]
    # the checkpoint the script wrote is the reference's checkpoint: same keys, same order
    assert list(ck["state_dict"].keys()) == gold["state_keys"]
    assert ck["epoch"] == gold["epoch"] and ck["iteration"] == gold["iteration"]
    assert len(ck["optimizer"]["param_groups"]) == gold["optimizer_groups"]
    assert len(ck["optimizer"]["state"]) == gold["optimizer_state_entries"]
    sd = ck["state_dict"]
    for k in ("queue_ptr", "queue_vname", "queue_label"):
        if k in gold["state"]:
            assert torch.equal(sd[k], gold["state"][k]), k


@pytest.mark.parametrize("name", ["dropin_main_nce", "dropin_main_coclr"])
def test_unmodified_script_runs_on_shadow_modules(fake, tmp_path, name, capsys):
    gold = load_golden(name)
    ds = H.SyntheticClips(**gold["dataset"])
    seen = {}

    def before_train(model):
        import coclr_amd.model.pretrain as impl
        import coclr_amd.optim as native
        seen["model_cls"] = type(model.module)
        seen["is_product"] = isinstance(model.module, impl.InfoNCE)
        seen["adam_cls"] = torch.optim.Adam
        seen["native"] = native.Adam
        seen["bucket_view"] = model.gradient_as_bucket_view
    if gold["script"] == "main_coclr":
        H.write_pretrained_pair(str(tmp_path), use_reference_model=False)
    rec = H.run_script(gold["script"], [a.format(tmp=str(tmp_path)) for a in gold["argv"]], ds,
                       use_reference_model=False, cpu=True, workdir=str(tmp_path), port=29643,
                       before_train=before_train)
    assert seen["is_product"], "the script did not pick up the shadow model package"
    assert issubclass(seen["adam_cls"], seen["native"]), "torch.optim.Adam was not resolved to the native subclass"
    assert seen["bucket_view"] is True
    _compare(gold, rec, gold["script"] == "main_coclr")
    out = capsys.readouterr().out
    assert "Training from ep 0 to ep 1 finished" in out


@pytest.mark.parametrize("name", ["dropin_main_nce", "dropin_main_coclr"])
def test_restated_caller_loop_is_the_script(fake, tmp_path, name):
    """tests/_caller_loop.py (what the GPU tier and bench.py's `value_unmodified_caller` leg run, the
    GPU box having no /root/reference) against the unmodified script on the same backend: identical
    logits, targets and losses, iteration by iteration."""
    import model.pretrain as product
    import _caller_loop
    gold = load_golden(name)
    ds = H.SyntheticClips(**gold["dataset"])
    if gold["script"] == "main_coclr":
        H.write_pretrained_pair(str(tmp_path), use_reference_model=False)
    rec = H.run_script(gold["script"], [a.format(tmp=str(tmp_path)) for a in gold["argv"]], ds,
                       use_reference_model=False, cpu=True, workdir=str(tmp_path), port=29644)
    from oracle import coclr_oracle as orc
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29645", rank=0, world_size=1)
    try:
        with H.script_environment(False, True):        # same lenient Tensor.view as the script saw
            if gold["script"] == "main_nce":
                mine = _caller_loop.run_nce(product, ds, calc_topk_accuracy=orc.calc_topk_accuracy)
            else:
                mine = _caller_loop.run_coclr(product, ds, calc_topk_accuracy=orc.calc_topk_accuracy,
                                              calc_mask_accuracy=orc.calc_mask_accuracy,
                                              pretrain=(str(tmp_path / "rgb.pth.tar"),
                                                        str(tmp_path / "flow.pth.tar")))
    finally:
        dist.destroy_process_group()
    assert len(mine["outputs"]) == len(rec["outputs"])
    for i, (a, b) in enumerate(zip(mine["outputs"], rec["outputs"])):
        assert torch.equal(a, b), "logits of iteration %d differ" % i
        assert torch.equal(mine["targets"][i], rec["targets"][i])
        assert mine["losses"][i] == rec["losses"][i]
    sd = mine["model"].state_dict()
    for k, v in rec["checkpoint"]["state_dict"].items():
        assert torch.equal(sd[k], v), k
