"""Rehearsal of bench.py's launch contract on the host: run as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P tests/bench_dryrun.py --gpus N --steps K --warmup W [...]
with the kernel entry points replaced by the ATen double of tests/fake_backend.py and tiny clips.
It exercises what the driver's multi-GPU run depends on and a 1-GPU box cannot show: RANK /
WORLD_SIZE handling, the gloo/RCCL group and DDP wrap, K % (B*world), the barrier + max-over-ranks
timing, and that rank 0 -- and only rank 0 -- prints ONE JSON line.  Not a measurement."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import fake_backend  # noqa: E402


class _MP:
    @staticmethod
    def setattr(obj, name, val):
        setattr(obj, name, val)


if __name__ == "__main__":
    import torch
    torch.set_num_threads(1)
    fake_backend.install(_MP)
    import bench
    sys.argv += ["--dry-run-host"]
    for flag, val in (("--seq-len", "8"), ("--img-dim", "32"), ("--batch", "2")):
        if flag not in sys.argv:
            sys.argv += [flag, val]
    bench.main()
