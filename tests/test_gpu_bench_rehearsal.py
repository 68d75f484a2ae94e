"""bench.py at N = 2 exactly as the driver launches it (torch.distributed.run, one process per rank) with
both ranks on ONE MI355X over gloo (tests/bench_rehearse_gpu.py): the first-contact evidence block must come
out complete and the replicas must be BIT-identical after a full step with the real kernels."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shuffle", ["routed", "allgather"])
def test_bench_n2_rehearsal_on_one_gpu(shuffle):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29761" if shuffle == "routed" else "29762",
           os.path.join(root, "tests", "bench_rehearse_gpu.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--batch", "8", "--moco-k", "2048"]
    env = dict(os.environ, COCLR_SHUFFLE=shuffle, COCLR_QUIET="1")
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 16 and "REHEARSAL" in rec["data"]
    assert rec["value"] > 0 and rec["config"]["final_loss"] == rec["config"]["final_loss"]
    mg = rec["multi_gpu"]
    assert mg["shuffle_mode"] == shuffle and mg["split_stages"] is True
    assert mg["cross_rank"]["replicas_identical"] is True, mg["cross_rank"]
    assert mg["cross_rank"]["logits_finite_on_every_rank"] is True
    names = " | ".join(c["collective"] for c in mg["collectives"])
    for what in (("all_to_all_single",) if shuffle == "routed" else ()) + (
            "all_gather_into_tensor", "ddp bucket 0 all_reduce", "broadcast of the flat float32 buffer"):
        assert what in names, (what, names)
    print("N=2 rehearsal (%s): %.1f clips/s, collectives %.2f ms/step serialised"
          % (shuffle, rec["value"], mg["collectives_ms_per_step_serialised"]))
