"""Same-host A/B of bench.py's CPU baseline ("kind": "port", oracle/coclr_oracle.py) against the reference's
OWN module (`model.pretrain.InfoNCE` imported unmodified from /root/reference, full
step = forward + nn.CrossEntropyLoss + backward + Adam over one param group per tensor, main_nce.py:190-200).

Runs only where /root/reference exists (the build container; needs no GPU):
    python tools/cpu_port_vs_reference.py [threads=8] [rounds=3] > profiles/r05_cpu_port_vs_reference.txt
Both sides: B=4, K=2048, 3x32x128x128 fp32, same seeds, same inputs, same thread count, one warm-up step then
timed steps ALTERNATING reference / port so that drift of the host hits both alike.
"""
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    torch.set_num_threads(threads)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=0, world_size=1)
    B, K = 4, 2048

    # ---- the reference, untouched (harness of SURVEY.md 8c) ------------------------------------------
    sys.path.insert(0, REF)
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    import model.pretrain as ref_pretrain
    assert ref_pretrain.__file__.startswith(REF), ref_pretrain.__file__
    torch.manual_seed(0)
    ref = ref_pretrain.InfoNCE('s3d', 128, K, 0.999, 0.07)
    ref.train()
    ref_opt = torch.optim.Adam([{"params": p} for _, p in ref.named_parameters()], lr=1e-3,
                               weight_decay=1e-5)
    ref_ce = nn.CrossEntropyLoss()

    # ---- the port (what bench.py::cpu_baseline times) ---------------------------------------------------
    sys.path.insert(0, ROOT)
    from oracle import coclr_oracle as orc
    torch.manual_seed(0)
    proto = ref_pretrain.InfoNCE('s3d', 128, K, 0.999, 0.07)        # same init, same state-dict keys
    sd = orc.training_state(proto.state_dict())
    leaves = [sd[k] for k, _ in proto.named_parameters() if sd[k].requires_grad]
    port_opt = torch.optim.Adam([{"params": p} for p in leaves], lr=1e-3, weight_decay=1e-5)

    def inputs(step):
        g = torch.Generator().manual_seed(100 + step)
        return torch.randn(B, 2, 3, 32, 128, 128, generator=g), torch.randperm(B, generator=g)

    def ref_step(step):
        block, _ = inputs(step)
        t0 = time.perf_counter()
        out, tgt = ref(block)
        loss = ref_ce(out, tgt)
        ref_opt.zero_grad()
        loss.backward()
        ref_opt.step()
        return time.perf_counter() - t0, float(loss)

    def port_step(step):
        block, perm = inputs(step)
        t0 = time.perf_counter()
        port_opt.zero_grad()
        (logits, labels), = orc.nce_step(sd, "infonce", "s3d", [block], None, 128, K, 0.999, 0.07, perm)
        loss = F.cross_entropy(logits, labels)
        loss.backward()
        port_opt.step()
        return time.perf_counter() - t0, float(loss)

    print("host: %d hardware threads, %d intra-op threads; torch %s; B=%d K=%d 3x32x128x128 fp32"
          % (os.cpu_count() or 0, torch.get_num_threads(), torch.__version__, B, K))
    ref_step(0)
    port_step(0)                                           # warm-up, untimed
    tr, tp = [], []
    for r in range(rounds):
        a, la = ref_step(1 + r)
        b, lb = port_step(1 + r)
        tr.append(a)
        tp.append(b)
        print("round %d: reference %.3f s/step (loss %.5f)   port %.3f s/step (loss %.5f)" % (r, a, la, b, lb))
    mr, mp = sum(tr) / len(tr), sum(tp) / len(tp)
    print("mean: reference %.3f s/step = %.3f clips/s   port %.3f s/step = %.3f clips/s   port/reference "
          "time ratio %.4f" % (mr, B / mr, mp, B / mp, mp / mr))
    print("within 3 %%: %s" % (abs(mp / mr - 1.0) <= 0.03))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
