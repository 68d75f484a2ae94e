"""Rehearsal of bench.py's N > 1 path with the REAL kernels on ONE MI355X: run as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port P tests/bench_rehearse_gpu.py --gpus 2 --steps K --warmup W [...]
Every rank lives on cuda:0 and the process group is gloo (RCCL refuses two ranks on one device).  gloo
carries device tensors for broadcast / all_reduce / all_gather only: `all_gather_into_tensor` and
`all_to_all_single` of DEVICE tensors are staged through the host underneath torch.distributed's API (as in
tests/test_gpu_multirank.py), so bench.py and the product's exchange code run unmodified.  What this shows
that the host dry run cannot: the checked step's cross-rank digests with real kernels, streams, captured
graphs, the single-launch Adam and the bucket hook under DDP.  Not the fabric, not a measurement."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

if __name__ == "__main__":
    import torch
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # Two PROCESSES share this GPU: with each one's weight-gradient stream at high priority (the product's default)
    # rank A's weight gradients starve rank B's data-gradient chain and vice versa -- steps of a second instead of
    # 40 ms.  One process per GPU never meets that; the rehearsal runs the streams at equal priority.
    os.environ.setdefault("COCLR_WGRAD_PRIORITY", "0")
    gather_native, a2a_native = dist.all_gather_into_tensor, dist.all_to_all_single

    def all_gather_into_tensor(out, tensor, *a, **kw):
        if not tensor.is_cuda:
            return gather_native(out, tensor, *a, **kw)
        host = torch.empty(out.shape, dtype=out.dtype)
        gather_native(host, tensor.contiguous().cpu(), *a, **kw)
        out.copy_(host)

    def all_to_all_single(out, tensor, output_split_sizes=None, input_split_sizes=None, *a, **kw):
        if not tensor.is_cuda:
            return a2a_native(out, tensor, output_split_sizes, input_split_sizes, *a, **kw)
        host = torch.empty(out.shape, dtype=out.dtype)
        a2a_native(host, tensor.contiguous().cpu(), output_split_sizes, input_split_sizes, *a, **kw)
        out.copy_(host)

    dist.all_gather_into_tensor = all_gather_into_tensor
    dist.all_to_all_single = all_to_all_single
    import bench
    sys.argv += ["--one-gpu-rehearsal"]
    bench.main()
