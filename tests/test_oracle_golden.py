"""Pin the CPU oracle (oracle/coclr_oracle.py) to the reference itself: replay every
fixture in tests/golden/ (recorded from the unmodified reference by
oracle/make_golden.py) through the restatement.  Same ATen CPU kernels on both sides,
so agreement is expected at fp32-roundoff level; asserted at 2e-5 relative (1e-4 on
gradients).  Also pins the product's constructors: same seed -> bit-identical
initial state dict as the reference (checksums recorded in the fixture)."""
import pytest
import torch

from _cases import (assert_checksums, build_model, case_inputs, check_close, checksum_table, load_golden, loss_fn,
                    sample)
from oracle import coclr_oracle as orc

SINGLE = ["infonce_s3d_conditioned", "infonce_s3d_small", "ubernce_s3d_small", "coclr_s3d_small",
          "coclr_s3d_small_reverse_cold", "infonce_r50_small", "infonce_s3dg_small"]


def _replay_oracle(gold_per_rank, tol=2e-5):
    import model.pretrain as product
    cfg = gold_per_rank[0]["cfg"]
    world = len(gold_per_rank)
    kind, B, K = cfg["kind"], cfg["B"], cfg["K"]
    model = build_model(cfg, product)
    sd0 = model.state_dict()
    keys = gold_per_rank[0]["init_checksums"]["keys"]
    assert_checksums(checksum_table(sd0, keys), gold_per_rank[0]["init_checksums"]["vals"])
    sd = orc.training_state(sd0)
    # the caller's optimiser (main_nce.py:190-200): Adam, one param group per tensor
    leaves = [sd[k] for k, _ in model.named_parameters() if sd[k].requires_grad]
    adam = torch.optim.Adam([{"params": p} for p in leaves], lr=1e-3, weight_decay=1e-5)
    for step in range(cfg["steps"]):
        recs = [g["steps"][step] for g in gold_per_rank]
        # Step 0 is bit-reproducible.  From step 1 on, the multi-rank fixtures were
        # recorded with a different intra-op thread count (ranks share the host cores);
        # the ~1e-7 re-association noise of step 0's backward goes through an Adam step
        # (update ~ lr*sign(g)) and BatchNorm over 2 samples per rank, which makes early-
        # layer gradients reproducible only to ~1e-2 even between two runs of the
        # reference itself.  Logits / loss / state are held to the north-star 1e-3 there
        # (observed 2-4e-4 on an 8-thread host against the 4-thread recording).
        loose = world > 1 and step > 0
        ltol, gtol = (1e-3, 5e-2) if loose else (tol, 1e-4)
        blocks, extra = case_inputs(cfg, step, world)
        per_rank_blocks, per_rank_extra = [], []
        for r in range(world):
            sl = slice(r * B, (r + 1) * B)
            per_rank_blocks.append(blocks[0][sl] if kind != "coclr"
                                   else (blocks[0][sl], blocks[1][sl]))
            per_rank_extra.append(extra[sl] if extra is not None else None)
        for k, v in sd.items():
            if v.requires_grad:
                v.grad = None
        outs = orc.nce_step(sd, kind, cfg["network"], per_rank_blocks, per_rank_extra, cfg["dim"],
                            K, cfg["m"], cfg["T"], recs[0]["perm"], topk=cfg.get("topk", 5),
                            reverse=cfg.get("reverse", False), world=world)
        losses = [loss_fn(kind, o, t) for o, t in outs]
        # DDP averages gradients over ranks
        (sum(losses) / world).backward()
        for r, rec in enumerate(recs):
            out, tgt = outs[r]
            check_close(out, rec["logits"], ltol, "logits rank %d" % r)
            if kind == "infonce":
                assert torch.equal(tgt, rec["target"])
            else:
                assert torch.equal(tgt.nonzero(), rec["target"])
            check_close(losses[r], rec["loss"], 1e-3 if loose else 1e-4, "loss")
            for k, ref in rec["grads"].items():
                check_close(sample(sd[k].grad), ref, gtol, "grad " + k)
        grads = {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}
        gk = recs[0]["grad_checksums"]["keys"]
        got = checksum_table({k: grads[k] for k in gk}, gk)
        ref = recs[0]["grad_checksums"]["vals"]
        assert float(((got[:, 1] - ref[:, 1]).abs() / (ref[:, 1].abs() + 1e-12)).max()) < \
            (5e-2 if loose else 1e-3)
        adam.step()
        rec = recs[0]
        assert int(sd["queue_ptr"]) == int(rec["queue_ptr"])
        bw = B * world
        ptr0 = (int(sd["queue_ptr"]) - bw) % K
        check_close(sd["queue"][:, ptr0:ptr0 + bw], rec["queue_cols"], ltol, "queue cols")
        if "queue_second_cols" in rec:
            check_close(sd["queue_second"][:, ptr0:ptr0 + bw], rec["queue_second_cols"], ltol,
                        "queue_second cols")
        for k in ("queue_label", "queue_vname"):
            if k + "_cols" in rec:
                assert torch.equal(sd[k][ptr0:ptr0 + bw], rec[k + "_cols"])
        for k, ref in rec["buffers"].items():
            if ref.is_floating_point():
                check_close(sd[k], ref, 1e-3 if loose else 1e-4, "buffer " + k)
            else:
                assert torch.equal(sd[k], ref), k
        for k, ref in rec["params_after"].items():
            check_close(sd[k], ref, 1e-3 if loose else 1e-4, "param " + k)


@pytest.mark.parametrize("name", SINGLE)
def test_oracle_matches_reference_single_rank(name):
    _replay_oracle([load_golden(name)])


@pytest.mark.parametrize("name", ["infonce_s3d_small_world2", "coclr_s3d_small_world2"])
def test_oracle_matches_reference_two_ranks(name):
    _replay_oracle([load_golden(name + "_rank0"), load_golden(name + "_rank1")])


def test_oracle_matches_reference_config1():
    """BASELINE.json configs[0] (B=4, K=2048, 3x32x128x128)."""
    _replay_oracle([load_golden("infonce_s3d_config1")])
