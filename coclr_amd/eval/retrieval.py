"""Nearest-neighbour video retrieval on MI355X (eval/main_classifier.py:686-706).

The reference centres and L2-normalises the (n_test, C) and (n_train, C) feature matrices,
multiplies them (`sim = test @ train.T`), and for every k in (1,5,10,20,50) runs `torch.topk` over
the full similarity matrix and checks whether any of the k nearest training clips carries the test
clip's label.  Here: two centring launches, two normalisations, ONE fp32-MFMA GEMM and ONE
selection kernel that walks the top-50 of every row in order and records all five answers (the same
GEMM + per-row top-k pattern as CoCLR's cross-modal mining, model/pretrain.py:405-410).
"""
import torch

from .. import ops


def nn_retrieval(test_feature, test_label, train_feature, train_label, ks=(1, 5, 10, 20, 50)):
    """Returns (accuracies: fp32 device tensor (len(ks),), sim: (n_test, n_train) fp32,
    topidx: int32 (n_test, max(ks)) nearest training clips in order).  Labels are int64 vectors;
    features fp32 (n, C) on the device.  `ks` must be ascending."""
    ks = [int(k) for k in ks]
    if ks != sorted(set(ks)) or ks[0] < 1:
        raise ValueError("coclr_amd: ks must be ascending, distinct and >= 1")
    nt, C_ = test_feature.shape
    ntr = train_feature.shape[0]
    if train_feature.shape[1] != C_ or ks[-1] > ntr:
        raise ValueError("coclr_amd: feature widths differ or k exceeds the training set")
    dev = test_feature.device

    def prep(f):
        f = f.contiguous()
        n = f.shape[0]
        ws = torch.empty(ops.colstats_workspace(n, C_), dtype=torch.float32, device=dev)
        c = torch.empty_like(f)
        ops.center_rows(f, c, ws)                        # f - f.mean(0)          (ref :690-691)
        out = torch.empty_like(f)
        inv = torch.empty(n, dtype=torch.float32, device=dev)
        ops.l2norm_fwd(c, out, inv)                      # F.normalize(p=2, dim=1) (ref :694-695)
        return out

    te, tr = prep(test_feature), prep(train_feature)
    sim = torch.empty(nt, ntr, dtype=torch.float32, device=dev)
    # sim[m][n] = sum_c te[m][c] * tr[n][c]                                     (ref :698)
    ops.gemm(te, C_, 1, tr, 1, C_, sim, ntr, None, nt, ntr, C_)
    ks_t = torch.tensor(ks, dtype=torch.int32).to(dev)
    hits = torch.empty(nt, len(ks), dtype=torch.float32, device=dev)
    topidx = torch.empty(nt, ks[-1], dtype=torch.int32, device=dev)
    ops.retrieval_hits(sim, train_label.to(device=dev, dtype=torch.long).contiguous(),
                       test_label.to(device=dev, dtype=torch.long).contiguous(), ks_t, hits, topidx)
    acc = torch.empty(len(ks), dtype=torch.float32, device=dev)
    ops.colsum(hits, acc)
    ops.plane_scale(acc.view(1, len(ks), 1, 1, 1), torch.full((1, len(ks)), 1.0 / nt, device=dev), None,
                    acc.view(1, len(ks), 1, 1, 1))
    return acc, sim, topidx
