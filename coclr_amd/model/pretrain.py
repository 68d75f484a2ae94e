"""InfoNCE / UberNCE / CoCLR heads (MoCo for video) on the gfx950 kernel library.

API, buffer names and step semantics follow the reference model/pretrain.py
(InfoNCE :28-190, UberNCE :193-278, CoCLR :281-418, concat_all_gather :14-25);
the launch scripts main_nce.py / main_coclr.py drive these classes unchanged.

Differences are all below the API:
  * encoders run on coclr_amd.engine (hand-written HIP conv/BN/pool kernels)
  * the clip pair is consumed as strided views, never `.contiguous()`-copied
    (ref :149-150), and the shuffle-BN row gather (ref :124) is folded into the
    first conv of the key encoder as a sample-index indirection
  * momentum update = one multi-tensor launch (ref :79-80 is ~700 launches)
  * logits = one fused MFMA kernel writing [l_pos | l_neg]/T (ref :175-182)
  * the queue pointer stays on the device (ref :89 syncs the host every step)
  * keys cross ranks once: the un-shuffle gather (ref :134) and the enqueue
    gather (ref :85) are the same all_gather_into_tensor over RCCL
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.distributed as dist

from .. import engine as _engine
from .. import ops
from ..parallel import collective as _coll
from .. import optim as _optim
from ..backbone.select_backbone import select_backbone

_CHUNK = 32768   # elements per workgroup of the momentum kernel
# shuffle-BN exchange at world > 1:
#   "pull"      each rank reads its clips out of its peers' hipIpc-mapped staging buffers with one HIP
#               kernel (coclr_pull_rows): the hand-written exchange, no collective on the data path
#   "routed"    RCCL all_to_all_single of exactly the clips each rank will encode
#   "allgather" the reference's own scheme (all-gather, keep B of B*world)
#   "auto"      (default) the first forward on device tensors maps the peers, runs the routed exchange AND the
#               pull on the same permutation, compares what arrived bit for bit on every rank, and stays on
#               "pull" if all agree -- "routed" otherwise (mapping refused, any difference): the kernel earns
#               its place on the data path at run time instead of behind an environment variable
_SHUFFLE_MODE = os.environ.get("COCLR_SHUFFLE", "auto")
_SHUFFLE_INFO = {"requested": _SHUFFLE_MODE}       # how the mode in force was arrived at (bench.py prints it)
# DDP(broadcast_buffers=True) re-sends rank 0's QUEUES with every forward (main_nce.py:172).  They are the
# one part of the buffers that cannot differ between ranks once they have been made equal: every rank enqueues
# the same gathered keys at the same pointer (ref :82-96).  So they travel with the FIRST broadcast after
# construction / load_state_dict / a storage move -- when ranks may hold different random queues -- and not
# again: 0.5 MB of BatchNorm statistics per forward instead of 8.9 MB at K = 16384 (17 MB for CoCLR).
# COCLR_SYNC_QUEUES=1 re-sends them with every forward, as the reference does.
_SYNC_QUEUES = os.environ.get("COCLR_SYNC_QUEUES", "0") != "0"
_OVERLAP_KEYS = os.environ.get("COCLR_OVERLAP_KEYS", "1") != "0"
_GRAPHS = os.environ.get("COCLR_GRAPHS", "1") != "0"


def _par_issued():
    """Collectives this rank has entered so far (coclr_amd.parallel.collective counts them)."""
    from .. import parallel as _par
    return _par.LAST[2]


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def route_plan(perm, B, world, rank):
    """Routing of the shuffle-BN exchange for one rank.  `perm` is the global permutation
    (B*world clip ids, clip g lives on rank g // B at local index g % B); rank r encodes
    the clips perm[r*B:(r+1)*B] in that order.  Returns
      send_order  local indices of this rank's clips laid out by destination rank (for each
                  destination, in the order that destination wants them),
      in_splits   clips sent to each rank,  out_splits  clips received from each rank,
      pos         pos[i] = row of the receive buffer (rows grouped by source rank, all_to_all
                  order) that holds the i-th clip this rank has to encode.
    Vectorised (numpy): this runs on the host in front of every training forward."""
    perm = np.asarray(perm, dtype=np.int64)
    src = perm // B                                   # owner of the clip encoded at position i
    mine = np.nonzero(src == rank)[0]                 # positions whose clip lives here, ascending
    send_order = perm[mine] % B                       # = grouped by destination (i // B), then by i
    in_splits = np.bincount(mine // B, minlength=world)
    wsrc = src[rank * B:(rank + 1) * B]
    out_splits = np.bincount(wsrc, minlength=world)
    order = np.argsort(wsrc, kind="stable")           # receive buffer: by source, arrival order kept
    pos = np.empty(B, dtype=np.int64)
    pos[order] = np.arange(B)
    return send_order.tolist(), in_splits.tolist(), out_splits.tolist(), pos.tolist()


@torch.no_grad()
def concat_all_gather(tensor):
    """All-gather along dim 0 (no gradient), one RCCL all_gather_into_tensor
    instead of the list API + torch.cat (ref :14-25)."""
    world, _ = _world()
    if world == 1:
        return tensor
    tensor = tensor.contiguous()
    out = torch.empty((world * tensor.shape[0],) + tuple(tensor.shape[1:]), dtype=tensor.dtype,
                      device=tensor.device)
    _coll("all_gather_into_tensor %s (pretrain.concat_all_gather; ref :14-25)" % (tuple(tensor.shape),),
          lambda: dist.all_gather_into_tensor(out, tensor), out.numel() * out.element_size(),
          tensor.device)
    return out


# ---------------------------------------------------------------------------------
# head modules (indices 1..4 of the encoder nn.Sequential, ref :49-54)
# ---------------------------------------------------------------------------------

class _AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.shape = x.shape
        y = torch.empty(x.shape[0], x.shape[1], 1, 1, 1, dtype=x.dtype, device=x.device)
        ops.global_avgpool_fwd(x, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        dx = torch.empty(ctx.shape, dtype=dy.dtype, device=dy.device)
        ops.global_avgpool_bwd(dy.contiguous(), dx)
        return dx


class GlobalAvgPool3d(nn.AdaptiveAvgPool3d):
    """nn.AdaptiveAvgPool3d((1,1,1)) on the HIP kernel."""

    def forward(self, x):
        if tuple(self.output_size) != (1, 1, 1):
            raise NotImplementedError("coclr_amd: only global average pooling is implemented")
        return _AvgPoolFn.apply(x)


def _fc_splits(M, N, K):
    """split-K so that a skinny product (M = batch) still fills the chip: a 32 x 1024 x 1024 head layer is 8
    output tiles.  K slices of at least 64 (profiles/r05_head_probe.txt: 16 splits of a K = 1024 product beat
    32 -- the fold kernel reads every partial), at most ~256 workgroups."""
    tiles = ((N + 127) // 128) * ((M + 31) // 32)
    return max(1, min(K // 64, 256 // tiles))


def _gemm_auto(a, sam, sak, b, sbk, sbn, c, ldc, bias, M, N, K):
    """ops.gemm with split-K sized so a skinny product (M = batch) still fills the chip:
    a 32 x 1024 x 1024 head layer is 8 output tiles -- split K until ~256 workgroups."""
    splits = _fc_splits(M, N, K)
    ws = None
    if splits > 1:
        ws = torch.empty(ops.gemm_workspace(M, N, K, splits), dtype=c.dtype, device=c.device)
    ops.gemm(a, sam, sak, b, sbk, sbn, c, ldc, bias, M, N, K, splits=splits, workspace=ws)


class _PointwiseFn(torch.autograd.Function):
    """y[n][co] = sum_ci x[n][ci] w[co][ci] + b[co] on pooled (N,C,1,1,1) features."""

    @staticmethod
    def forward(ctx, x, w, b):
        N, Cin = x.shape[0], x.shape[1]
        Cout = w.shape[0]
        x2 = x.reshape(N, Cin).contiguous()
        w2 = w.reshape(Cout, Cin)
        y = torch.empty(N, Cout, dtype=x.dtype, device=x.device)
        _gemm_auto(x2, Cin, 1, w2, 1, Cin, y, Cout, b, N, Cout, Cin)
        ctx.save_for_backward(x2, w2)
        ctx.has_bias = b is not None
        return y.view(N, Cout, 1, 1, 1)

    @staticmethod
    def backward(ctx, dy):
        x2, w2 = ctx.saved_tensors
        N, Cin = x2.shape
        Cout = w2.shape[0]
        dy2 = dy.reshape(N, Cout).contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x2)
            _gemm_auto(dy2, Cout, 1, w2, Cin, 1, dx, Cin, None, N, Cin, Cout)
            dx = dx.view(N, Cin, 1, 1, 1)
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w2)
            # dw[co][ci] = sum_n dy[n][co] x[n][ci]
            _gemm_auto(dy2, 1, Cout, x2, Cin, 1, dw, Cin, None, Cout, Cin, N)
            dw = dw.view(Cout, Cin, 1, 1, 1)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(Cout, dtype=dy.dtype, device=dy.device)
            ops.colsum(dy2, db)
        return dx, dw, db


class PointwiseConv3d(nn.Conv3d):
    """nn.Conv3d(cin, cout, kernel_size=1, bias=True) of the projection head (ref :52,54);
    constructed through nn.Conv3d so init and RNG use are identical."""

    def forward(self, x):
        if x.dim() != 5 or x.shape[2] * x.shape[3] * x.shape[4] != 1:
            raise NotImplementedError(
                "coclr_amd: projection-head conv expects globally pooled (N,C,1,1,1) input")
        return _PointwiseFn.apply(x, self.weight, self.bias)


class _ReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        ops.relu_fwd(x, y)
        ctx.save_for_backward(y)
        if _engine.DECISION_PROBE is not None and x.requires_grad:
            _engine.DECISION_PROBE("relu", "head", y > 0)       # test instrumentation, see engine.py
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dx = torch.empty_like(y)
        ops.relu_bwd(dy.contiguous(), y, dx)
        return dx


class HeadReLU(nn.ReLU):
    def forward(self, x):
        return _ReluFn.apply(x)


class _L2NormFn(torch.autograd.Function):
    """F.normalize(x, dim=1) on (B, D) rows (ref :154)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        inv = torch.empty(x.shape[0], dtype=x.dtype, device=x.device)
        ops.l2norm_fwd(x, y, inv)
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        dx = torch.empty_like(y)
        ops.l2norm_bwd(dy.contiguous(), y, inv, dx)
        return dx


class _NceLogitsFn(torch.autograd.Function):
    """logits = cat([<q,k>, q @ queue], 1) / T with gradient to q only (ref :175-182)."""

    @staticmethod
    def forward(ctx, q, k, queue, T):
        B, D = q.shape
        K = queue.shape[1]
        logits = torch.empty(B, 1 + K, dtype=q.dtype, device=q.device)
        ops.nce_logits_fwd(q, k, queue, logits, T)
        if ctx.needs_input_grad[0]:
            # the enqueue that follows overwrites columns of `queue` in place
            ctx.save_for_backward(k, queue.clone())
        ctx.T = T
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        k, queue = ctx.saved_tensors
        B, D = k.shape
        K = queue.shape[1]
        splits = _logits_bwd_splits(K)
        ws = torch.empty(max(1, ops.gemm_workspace(B, D, K, splits)), dtype=k.dtype,
                         device=k.device)
        dq = torch.empty(B, D, dtype=k.dtype, device=k.device)
        ops.nce_logits_bwd(dlogits.contiguous(), k, queue, dq, ws, ctx.T, splits)
        return dq, None, None, None


# ---------------------------------------------------------------------------------
# The head as the training step runs it
# ---------------------------------------------------------------------------------
# Module by module (above) the projection head and the logits are 8 launches forward and 21 backward (17
# kernels + DistributedDataParallel's four copies of the head's gradients into its buckets): ~0.3 ms of
# 7-20 us kernels with the chip otherwise idle, at the seam between the forward and the backward pass.  The
# step runs the SAME arithmetic through coclr_gemm_fused (csrc/nce.hip): the kernel that folds a product's
# split-K partials also applies the row-level op that follows it, and a weight-gradient product forms the
# bias gradient as its row sums:
#   forward   avg-pool | fc1 (+ bias + ReLU) | fc2 (+ bias + F.normalize) | logits            6 launches
#   backward  dlogits . queue^T (+ l_pos term + F.normalize backward) | . W2 (+ ReLU backward) |
#             dW2 + db2 | . W1 | avg-pool backward | dW1 + db1                                9 launches
# Weight and bias gradients are written straight into DistributedDataParallel's bucket views when it has
# published them (engine.grad_out_for).  Same fold order, same row arithmetic: bit-identical to the module-
# by-module path (tests/test_gpu_model.py), which COCLR_FUSED_HEAD=0 keeps and which also serves encoders
# whose head is not the reference's Sequential.  (A first form folded the partials INSIDE the product's
# launch -- last workgroup of a tile -- and was 2-8x slower per product on MI355X: the device-scope fence
# every workgroup needs writes back and invalidates L2 across the eight XCDs, and one workgroup folds a
# whole tile; profiles/r05_head_probe_lastblock.txt.)
FUSED_HEAD = os.environ.get("COCLR_FUSED_HEAD", "1") != "0"


def _head_parts(encoder, dim):
    """(fc1, fc2) when `encoder` is the reference's nn.Sequential(backbone, AdaptiveAvgPool3d((1,1,1)),
    Conv3d 1x1x1, ReLU, Conv3d 1x1x1 -> dim) built from this module's classes; else None."""
    ent = encoder.__dict__.get("_coclr_head_parts")
    if ent is None:
        ok = isinstance(encoder, nn.Sequential) and len(encoder) == 5 and \
            isinstance(encoder[1], GlobalAvgPool3d) and tuple(encoder[1].output_size) == (1, 1, 1) and \
            isinstance(encoder[2], PointwiseConv3d) and isinstance(encoder[3], HeadReLU) and \
            isinstance(encoder[4], PointwiseConv3d) and \
            all(tuple(c.kernel_size) == (1, 1, 1) and c.bias is not None and c.groups == 1
                for c in (encoder[2], encoder[4])) and \
            encoder[2].out_channels == encoder[4].in_channels and encoder[4].out_channels == dim and dim <= 128
        ent = encoder.__dict__["_coclr_head_parts"] = (encoder[2], encoder[4]) if ok else False
    return ent or None


def _logits_bwd_splits(K):
    """split count of dlogits . queue^T (measured alone on the chip, profiles/r05_head_probe.txt: K = 2048 is
    fastest at 32 splits, 21.8 us; K = 16384 at 64, 48.8 us against 60.5 at 128: the fold reads every partial)"""
    return max(1, min(K // 64, 64))


def _head_forward(feat, w1, b1, w2, b2, eps=1e-12):
    """feature map (N, Cf, T, H, W) -> (q, inv_norm, h0, h1): pooled features, hidden activations and the
    L2-normalised projection (ref :49-54,153-154), five launches."""
    feat = feat.contiguous()
    N_, Cf = feat.shape[0], feat.shape[1]
    Ch, D = w1.shape[0], w2.shape[0]
    dev = feat.device
    h0 = torch.empty(N_, Cf, dtype=feat.dtype, device=dev)
    ops.global_avgpool_fwd(feat, h0.view(N_, Cf, 1, 1, 1))
    h1 = torch.empty(N_, Ch, dtype=feat.dtype, device=dev)
    s1 = _fc_splits(N_, Ch, Cf)
    ws = torch.empty(ops.gemm_fused_workspace(N_, Ch, Cf, s1), dtype=feat.dtype, device=dev)
    ops.gemm_fused(h0, Cf, 1, w1.reshape(Ch, Cf), 1, Cf, h1, Ch, b1, N_, Ch, Cf, relu=True, splits=s1,
                   workspace=ws, mode=0)
    q = torch.empty(N_, D, dtype=feat.dtype, device=dev)
    inv = torch.empty(N_, dtype=feat.dtype, device=dev)
    s2 = _fc_splits(N_, D, Ch)
    ws2 = torch.empty(ops.gemm_fused_workspace(N_, D, Ch, s2), dtype=feat.dtype, device=dev)
    ops.gemm_fused(h1, Ch, 1, w2.reshape(D, Ch), 1, Ch, q, D, b2, N_, D, Ch, splits=s2, workspace=ws2,
                   mode=2, out2=inv, f=eps)
    return q, inv, h0, h1


class _QueryHeadFn(torch.autograd.Function):
    """feature map of the query encoder -> logits [<q,k> | q . queue] / T, with the gradient to the feature
    map and the head's parameters (ref :49-54,153-155,175-182 and their autograd backward)."""

    @staticmethod
    def forward(ctx, feat, w1, b1, w2, b2, k, queue, T, join):
        q, inv, h0, h1 = _head_forward(feat, w1, b1, w2, b2)
        if _engine.DECISION_PROBE is not None:
            _engine.DECISION_PROBE("relu", "head", (h1 > 0).view(h1.shape[0], h1.shape[1], 1, 1, 1))
        join()                                    # the keys come from the key stream
        B, D = q.shape
        K = queue.shape[1]
        logits = torch.empty(B, 1 + K, dtype=q.dtype, device=q.device)
        ops.nce_logits_fwd(q, k, queue, logits, T)
        # the enqueue that follows overwrites columns of `queue` in place
        ctx.save_for_backward(h0, h1, q, inv, k, queue.clone(), w1, w2)
        ctx.T, ctx.fshape = T, tuple(feat.shape)
        ctx.params = (w1, b1, w2, b2)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        h0, h1, q, inv, k, queue, w1, w2 = ctx.saved_tensors
        w1p, b1p, w2p, b2p = ctx.params
        B, D = q.shape
        K = queue.shape[1]
        Cf, Ch = h0.shape[1], h1.shape[1]
        dev, dt = q.device, q.dtype
        dl = dlogits.contiguous()
        inv_T = 1.0 / ctx.T
        # d(un-normalised projection): dlogits[:, 1:] . queue^T / T + the l_pos term, through F.normalize
        sp = _logits_bwd_splits(K)
        ws = torch.empty(ops.gemm_fused_workspace(B, D, K, sp), dtype=dt, device=dev)
        df = torch.empty(B, D, dtype=dt, device=dev)
        ops.gemm_fused(dl[:, 1:], 1 + K, 1, queue, 1, K, df, D, None, B, D, K, alpha=inv_T, splits=sp,
                       workspace=ws, mode=3, ep_a=dl, lda=1 + K, ep_b=k, ep_y=q, inv_norm=inv, f=inv_T)
        need = ctx.needs_input_grad
        dw1 = db1 = dw2 = db2 = dx = None
        # fc2: d(hidden) with the ReLU's backward, weight gradient with the bias gradient as its row sums
        sp = _fc_splits(B, Ch, D)
        ws = torch.empty(ops.gemm_fused_workspace(B, Ch, D, sp), dtype=dt, device=dev)
        dh1 = torch.empty(B, Ch, dtype=dt, device=dev)
        ops.gemm_fused(df, D, 1, w2.reshape(D, Ch), Ch, 1, dh1, Ch, None, B, Ch, D, splits=sp, workspace=ws,
                       mode=1, ep_a=h1, lda=Ch)
        if need[3] or need[4]:
            dw2 = _engine.grad_out_for(w2p)
            db2 = _engine.grad_out_for(b2p)
            ops.gemm_fused(df, 1, D, h1, Ch, 1, dw2, Ch, None, D, Ch, B, rowsum=db2)
        # fc1: d(feature map) through the average pool, weight gradient + bias gradient
        if need[0]:
            # (the fold kernel can spread the values over their planes itself -- mode 4 -- but every one of
            # the S threads of a plane then reads all the partials: 28.9 us against 15.7 + 7.9 for fold +
            # average-pool backward as two launches, profiles/r05_head_probe.txt)
            sp = _fc_splits(B, Cf, Ch)
            ws = torch.empty(ops.gemm_fused_workspace(B, Cf, Ch, sp), dtype=dt, device=dev)
            dh0 = torch.empty(B, Cf, 1, 1, 1, dtype=dt, device=dev)
            ops.gemm_fused(dh1, Ch, 1, w1.reshape(Ch, Cf), Cf, 1, dh0, Cf, None, B, Cf, Ch, splits=sp,
                           workspace=ws)
            dx = torch.empty(ctx.fshape, dtype=dt, device=dev)
            ops.global_avgpool_bwd(dh0, dx)
        if need[1] or need[2]:
            dw1 = _engine.grad_out_for(w1p)
            db1 = _engine.grad_out_for(b1p)
            ops.gemm_fused(dh1, 1, Ch, h0, Cf, 1, dw1, Cf, None, Ch, Cf, B, rowsum=db1)
        return dx, dw1, db1, dw2, db2, None, None, None, None


def _make_encoder(network, dim):
    backbone, param = select_backbone(network)
    fs = param["feature_size"]
    enc = nn.Sequential(backbone,
                        GlobalAvgPool3d((1, 1, 1)),
                        PointwiseConv3d(fs, fs, kernel_size=1, bias=True),
                        HeadReLU(),
                        PointwiseConv3d(fs, dim, kernel_size=1, bias=True))
    return enc, param


class InfoNCE(nn.Module):
    """MoCo for video (ref :28-190)."""

    def __init__(self, network='s3d', dim=128, K=2048, m=0.999, T=0.07):
        super().__init__()
        self.dim = dim
        self.K = K
        self.m = m
        self.T = T

        # construction order fixes the global-RNG stream: q backbone+head, k backbone+head, queue
        self.encoder_q, self.param = _make_encoder(network, dim)
        self.encoder_k, _ = _make_encoder(network, dim)
        for param_q, param_k in zip(self.encoder_q.parameters(), self.encoder_k.parameters()):
            param_k.data.copy_(param_q.data)
            param_k.requires_grad = False

        self.register_buffer("queue", torch.randn(dim, K))
        self.queue = nn.functional.normalize(self.queue, dim=0)
        self.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))
        self._momentum_table = None
        self.__dict__["_ptr_checked_for"] = None    # global batch the loaded pointer was validated for
        _optim.register_momentum_model(self)        # for coclr_amd.optim.Adam(fold_momentum=True)

    def _load_from_state_dict(self, *args, **kwargs):
        # a resumed queue_ptr must be re-validated against the batch it will be used with
        self.__dict__["_ptr_checked_for"] = None
        self.__dict__["_sync_full_next"] = True     # loaded queues may differ between ranks until broadcast
        return super()._load_from_state_dict(*args, **kwargs)

    def _check_queue_ptr(self, batch_size):
        """The reference fails on `queue[:, ptr:ptr+B] = keys.T` (ref :93) when a resumed pointer
        does not fit the current global batch; the device-side pointer never reaches the host in
        the steady state, so it is read ONCE per (construction | load_state_dict, batch size)."""
        if self.__dict__.get("_ptr_checked_for") == batch_size:
            return
        ptr = int(self.queue_ptr)
        if ptr % batch_size != 0 or ptr + batch_size > self.K or ptr < 0:
            raise RuntimeError(
                "coclr_amd: queue_ptr=%d does not fit a global batch of %d in a queue of %d "
                "(checkpoint written with another batch size / world size?)" % (ptr, batch_size, self.K))
        self.__dict__["_ptr_checked_for"] = batch_size

    # -- buffers: one flat allocation, one broadcast --------------------------------
    @property
    def _ddp_params_and_buffers_to_ignore(self):
        """DistributedDataParallel skips the buffers named here (it reads this attribute at
        wrap time).  The model keeps ALL its buffers (BN statistics of three encoders, the
        queues, the pointer) as views of one flat allocation per dtype and broadcasts those from
        rank 0 itself at the start of every forward -- the semantics of the reference's
        DDP(broadcast_buffers=True) (main_nce.py:172) as TWO
        collectives instead of ~470 tensor copies into and out of a coalescing buffer."""
        return [n for n, _ in self.named_buffers()]

    def _flatten_buffers(self):
        """One allocation PER DTYPE (fp32: BN statistics and the queues; int64: counters, pointer,
        name / label queues).  Not one for everything: torch.save refuses tensors of different dtypes
        that view one storage, and the launch scripts checkpoint `state_dict()` as it is
        (main_nce.py:278-290).  The model's own queue buffers sit at the END of their allocation, so that
        the steady-state broadcast (see _SYNC_QUEUES) is a prefix."""
        seen, entries = set(), []
        for mod in self.modules():
            for key, b in mod._buffers.items():
                if b is not None and id(b) not in seen:
                    seen.add(id(b))
                    entries.append((mod, key, b))
        flats, steady = [], []
        for dtype in sorted({b.dtype for _, _, b in entries}, key=str):
            mine = [(mod, key, b) for mod, key, b in entries if b.dtype == dtype]
            mine.sort(key=lambda e: e[0] is self and e[1].startswith("queue"))      # stable: queues last
            esz = mine[0][2].element_size()
            pad = max(1, 16 // esz)
            offs, total, prefix = [], 0, None
            for mod, key, b in mine:
                if prefix is None and mod is self and key.startswith("queue"):
                    prefix = total
                offs.append(total)
                total += (b.numel() + pad - 1) // pad * pad
            flat = torch.empty(total, dtype=dtype, device=self.queue.device)
            with torch.no_grad():
                for (mod, key, b), off in zip(mine, offs):
                    v = flat[off:off + b.numel()].view(b.shape)
                    v.copy_(b)
                    mod._buffers[key] = v
            flats.append(flat)
            # only the big floating-point allocation is worth a second message size
            steady.append(total if (prefix is None or not dtype.is_floating_point) else prefix)
        self.__dict__["_flat_buffers"] = flats
        self.__dict__["_flat_steady"] = steady
        self.__dict__["_sync_full_next"] = True

    def _sync_buffers(self):
        dev = self.queue.device
        if dev.type == "cuda" and dev.index != torch.cuda.current_device():
            # every kernel is enqueued on the current device's current stream (ops._stream)
            raise RuntimeError("coclr_amd: model lives on %s but the current device is cuda:%d; "
                               "call torch.cuda.set_device first" % (dev, torch.cuda.current_device()))
        flats = self.__dict__.get("_flat_buffers")
        if flats is not None:
            ptrs = {f.untyped_storage().data_ptr() for f in flats}
        if flats is None or flats[0].device != self.queue.device or \
                self.queue.untyped_storage().data_ptr() not in ptrs or \
                self.queue_ptr.untyped_storage().data_ptr() not in ptrs:
            self._flatten_buffers()       # first use, or .cuda()/.to() re-created the buffers
            flats = self.__dict__["_flat_buffers"]
        world, _ = _world()
        if world > 1:
            # new_group is a collective: every rank creates the host-side channel HERE, at its first
            # forward, whatever shuffle scheme / train-or-eval path it takes afterwards
            grp = self._host_group()
            full = _SYNC_QUEUES
            if not full:
                # The message size must be the SAME on every rank, and a rank's own flag is local knowledge:
                # load_state_dict on rank 0 alone is legal under the reference (DDP re-sends rank 0's buffers with
                # every forward), so is a re-flatten after `.to()` on some ranks.  The ranks therefore agree on
                # the host first -- eight bytes over the gloo side channel, MAX over the flags: if ANY rank's
                # queues may be stale, everybody takes part in the whole-allocation broadcast.  No device sync.
                # (In-place edits of the queues that bypass load_state_dict must set `_sync_full_next = True`
                # themselves, or run with COCLR_SYNC_QUEUES=1.)
                flag = torch.tensor([1 if self.__dict__.get("_sync_full_next", True) else 0])
                _coll("host agreement on the buffer broadcast size over gloo (pretrain._sync_buffers)",
                      lambda: dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=grp), 8)
                full = bool(int(flag[0]))
            self.__dict__["_sync_full_next"] = False
            with torch.no_grad():
                for flat, n in zip(flats, self.__dict__["_flat_steady"]):
                    part = flat if (full or n >= flat.numel()) else flat[:n]
                    if part.numel() == 0:
                        continue
                    _coll("broadcast of the flat %s buffer allocation%s (pretrain._sync_buffers; DDP "
                          "broadcast_buffers, main_nce.py:172)"
                          % (str(flat.dtype).replace("torch.", ""), "" if part is flat else " without the queues"),
                          lambda part=part: dist.broadcast(part, src=0),
                          part.numel() * part.element_size(), flat.device)

    # -- momentum encoder ---------------------------------------------------------
    def _build_momentum_table(self):
        rows = []
        pq_list, pk_list = list(self.encoder_q.parameters()), list(self.encoder_k.parameters())
        for pq, pk in zip(pq_list, pk_list):
            if not (pq.is_contiguous() and pk.is_contiguous()):
                raise RuntimeError("coclr_amd: encoder parameters must be contiguous")
            n = pk.numel()
            for off in range(0, n, _CHUNK):
                rows.append((pk.data_ptr() + 4 * off, pq.data_ptr() + 4 * off,
                             min(_CHUNK, n - off)))
        dev = pk_list[0].device
        table = torch.tensor(rows, dtype=torch.int64).to(dev)
        self._momentum_table = (self._momentum_sig(pq_list, pk_list), table, len(rows), pq_list,
                                pk_list)

    @staticmethod
    def _momentum_sig(pq_list, pk_list):
        # storage moves (.cuda(), .to()) re-create every parameter's data at once
        return (pq_list[0].data_ptr(), pq_list[-1].data_ptr(), pk_list[0].data_ptr(),
                pk_list[-1].data_ptr(), len(pq_list))

    @torch.no_grad()
    def _momentum_update_key_encoder(self):
        '''p_k <- p_k * m + p_q * (1 - m) for every parameter pair, one launch.'''
        mt = self._momentum_table
        if mt is None or mt[0] != self._momentum_sig(mt[3], mt[4]):
            self._build_momentum_table()
            mt = self._momentum_table
        ops.momentum_update(mt[1], mt[2], float(self.m), float(1. - self.m), pairs=(mt[4], mt[3]))

    def _momentum_pre(self, in_train_mode):
        """The momentum update to run in front of the key encoder, or None: outside training
        (ref :157-160), or when coclr_amd.optim.Adam(fold_momentum=True) has already applied this
        forward's update at the end of its step (same arithmetic, same operands)."""
        if not in_train_mode:
            return None
        folded = self.__dict__.pop("_momentum_folded", None)
        if folded is None:
            return self._momentum_update_key_encoder
        if folded != float(self.m):
            raise RuntimeError(
                "coclr_amd: the momentum coefficient changed (%.6f -> %.6f) after an optimizer "
                "step that had folded the key-encoder update; use fold_momentum=False with a "
                "per-iteration momentum schedule" % (folded, float(self.m)))
        return None

    # -- queue ---------------------------------------------------------------------
    @torch.no_grad()
    def _enqueue_gathered(self, keys_all, extra=()):
        """keys_all: (B*world, dim) in global batch order."""
        batch_size = keys_all.shape[0]
        assert self.K % batch_size == 0  # for simplicity
        self._check_queue_ptr(batch_size)
        ops.queue_enqueue(self.queue, keys_all.contiguous(), self.queue_ptr)
        for qbuf, vals, const in extra:
            ops.queue_fill_i64(qbuf, vals, const, batch_size, self.queue_ptr)
        ops.queue_advance(self.queue_ptr, batch_size, self.K)

    @torch.no_grad()
    def _dequeue_and_enqueue(self, keys):
        self._enqueue_gathered(concat_all_gather(keys))

    # -- shuffle BN ------------------------------------------------------------------
    @torch.no_grad()
    def _shuffle_indices(self, batch_size_this, device):
        """Common permutation of the global batch: CPU randperm (same RNG stream as the
        reference, ref :112) broadcast from rank 0.  At world size 1 permutation and inverse are
        formed on the host and reach the device through a pinned double buffer with ONE
        non-blocking copy: a pageable `.to(device)` would make the host wait, at the start of every
        forward, for everything the previous step still has queued."""
        world, rank = _world()
        batch_size_all = batch_size_this * world
        perm = torch.randperm(batch_size_all)
        if world > 1 or device.type != "cuda":
            idx_shuffle = perm.to(device)
            if world > 1:
                _coll("broadcast of the permutation (pretrain._shuffle_indices; ref :115)",
                      lambda: dist.broadcast(idx_shuffle, src=0), 8 * batch_size_all, device)
            idx_unshuffle = torch.argsort(idx_shuffle)
            return idx_shuffle.view(world, -1)[rank], idx_unshuffle
        st = self.__dict__.get("_perm_staging")
        if st is None or st["n"] != batch_size_all or st["dev"][0].device != device:
            st = self.__dict__["_perm_staging"] = {
                "n": batch_size_all, "flip": 0, "events": [None, None],
                "host": [torch.empty(2 * batch_size_all, dtype=torch.int64).pin_memory()
                         for _ in range(2)],
                "dev": [torch.empty(2 * batch_size_all, dtype=torch.int64, device=device)
                        for _ in range(2)]}
        f = st["flip"]
        st["flip"] = 1 - f
        if st["events"][f] is not None:
            st["events"][f].synchronize()            # the copy issued two steps ago
        h = st["host"][f].numpy()
        pn = perm.numpy()
        h[:batch_size_all] = pn
        h[batch_size_all:] = np.argsort(pn, kind="stable")
        st["dev"][f].copy_(st["host"][f], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        st["events"][f] = ev
        return st["dev"][f][:batch_size_all], st["dev"][f][batch_size_all:]

    @torch.no_grad()
    def _batch_shuffle_ddp(self, x):
        '''Same contract as the reference (:98-124): returns (x_shuffled, idx_unshuffle).'''
        x_gather = concat_all_gather(x)
        idx_this, idx_unshuffle = self._shuffle_indices(x.shape[0], x.device)
        out = torch.empty((idx_this.shape[0],) + tuple(x_gather.shape[1:]), dtype=x.dtype,
                          device=x.device)
        ops.gather_rows(x_gather.contiguous(), idx_this.contiguous(), out)
        return out, idx_unshuffle

    @torch.no_grad()
    def _batch_unshuffle_ddp(self, x, idx_unshuffle):
        '''Same contract as the reference (:126-143).'''
        world, rank = _world()
        x_gather = concat_all_gather(x).contiguous()
        idx_this = idx_unshuffle.view(world, -1)[rank].contiguous()
        out = torch.empty((idx_this.shape[0],) + tuple(x_gather.shape[1:]), dtype=x.dtype,
                          device=x.device)
        ops.gather_rows(x_gather, idx_this, out)
        return out

    # -- key path on its own HIP stream --------------------------------------------------
    def _key_stream(self, x):
        """The key encoder (no gradient) is independent of the query encoder's forward: run it
        on a second stream so its launches fill the CUs the query encoder's small late-stage
        kernels leave idle (and vice versa).  Ordering: the side stream first waits for
        everything already queued (the previous optimiser step writes the query weights the
        momentum update reads); the caller joins before the logits."""
        if not (_OVERLAP_KEYS and x.is_cuda):
            return None
        st = self.__dict__.get("_side_stream")
        if st is None or st.device != x.device:
            st = torch.cuda.Stream(device=x.device,
                                   priority=int(os.environ.get("COCLR_KEY_PRIORITY", "0")))
            self.__dict__["_side_stream"] = st
        st.wait_stream(torch.cuda.current_stream(x.device))
        return st

    def _query_requires_grad(self, x):
        """Whether the query features will carry autograd history (the reference's
        `in_train_mode = q.requires_grad`), known before the encoder runs."""
        if not torch.is_grad_enabled():
            return False
        ps = self.encoder_q.__dict__.get("_coclr_plist")
        if ps is None:
            ps = self.encoder_q.__dict__["_coclr_plist"] = list(self.encoder_q.parameters())
        return x.requires_grad or any(p.requires_grad for p in ps)

    @staticmethod
    def _join(st):
        if st is not None:
            torch.cuda.current_stream(st.device).wait_stream(st)

    # -- hipGraph replay of the gradient-free encoders --------------------------------------
    def _encode_graphed(self, encoder, src, n_index, pre=None):
        """encoder(src[n_index]) -> L2-normalised keys, without autograd, replayed from a
        captured hipGraph after one eager warm-up call.

        A no-grad encoder pass is ~330 kernel launches whose arguments do not change from step
        to step (weights, BN buffers and packed operands live at fixed addresses): capturing it
        removes ~4 ms of host launch work per step and lets the key encoder start at the same
        time as the query encoder instead of after the host has finished enqueueing that one.
        Inputs enter through a static buffer filled by the shuffle gather itself; `pre` (the
        momentum update) is captured in front of the encoder."""
        dev = src.device
        if not (_GRAPHS and src.is_cuda):
            if pre is not None:
                pre()
            return self._encode(encoder, src, n_index=n_index)
        params = encoder.__dict__.get("_coclr_plist")
        if params is None:
            params = encoder.__dict__["_coclr_plist"] = list(encoder.parameters())
        flats = self.__dict__.get("_flat_buffers")
        # One cache entry per (encoder, BN mode, with/without the momentum prologue): toggling
        # train()/eval() or no_grad switches between captured graphs instead of re-capturing.
        key = (id(encoder), encoder.training, encoder[0].training, pre is not None)
        # Everything a replay bakes in: shapes, every address the captured kernels read or write
        # (encoder parameters, the flat buffer allocation, and -- for the momentum prologue -- the
        # query parameters and m itself, which the kernel receives as constants).
        sig = (tuple(src.shape[1:]), int(n_index.shape[0]), src.dtype, dev, params[0].data_ptr(),
               params[-1].data_ptr(), () if flats is None else tuple(f.data_ptr() for f in flats))
        if pre is not None:
            qps = self.encoder_q.__dict__.get("_coclr_plist")
            if qps is None:
                qps = self.encoder_q.__dict__["_coclr_plist"] = list(self.encoder_q.parameters())
            sig += (float(self.m), qps[0].data_ptr(), qps[-1].data_ptr())
        bns = encoder.__dict__.get("_coclr_bns")
        if bns is None:
            bns = encoder.__dict__["_coclr_bns"] = [
                mod for mod in encoder.modules() if isinstance(mod, nn.modules.batchnorm._BatchNorm)]
        sig += (hash(tuple((mod.momentum, mod.eps) for mod in bns)),)   # kernel constants too
        store = self.__dict__.setdefault("_graphs", {})
        ent = store.get(key)
        if ent is None or ent["sig"] != sig:
            ent = store[key] = {"sig": sig, "calls": 0}
        if ent["calls"] < 1:                       # first call eager: one-time attribute calls etc.
            ent["calls"] += 1
            if pre is not None:
                pre()
            return self._encode(encoder, src, n_index=n_index)
        if ent.get("disabled"):
            if pre is not None:
                pre()
            return self._encode(encoder, src, n_index=n_index)
        if "graph" not in ent:
            static_x = torch.empty((n_index.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype,
                                   device=dev)
            ops.gather_rows(src, n_index, static_x)
            torch.cuda.current_stream(dev).synchronize()
            g = torch.cuda.CUDAGraph()
            # Capture on the stream the eager call ran on (the key stream) when there is one: the
            # packed-operand buffers and the batch re-layout plan recorded by that call are per
            # stream, so the graph then holds ONE re-layout launch instead of 77.
            cur = torch.cuda.current_stream(dev)
            cap = cur if cur != torch.cuda.default_stream(dev) else torch.cuda.Stream(device=dev)
            # thread-local capture: RCCL's watchdog thread keeps polling events meanwhile
            try:
                with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
                    if pre is not None:
                        pre()
                    out = self._encode(encoder, static_x)
            except RuntimeError as e:
                # nothing of a failed capture has executed: this (encoder, mode) stays on the eager
                # path (same kernels, ~4 ms more host work per step) instead of ending the job
                import warnings
                warnings.warn("coclr_amd: hipGraph capture of the key encoder failed (%s); running "
                              "it eagerly" % e)
                ent["disabled"] = True
                torch.cuda.synchronize(dev)
                if pre is not None:
                    pre()
                return self._encode(encoder, src, n_index=n_index)
            ent.update(graph=g, x=static_x, out=out)
        else:
            ops.gather_rows(src, n_index, ent["x"])
        ent["graph"].replay()
        return ent["out"].clone()

    # -- encoders --------------------------------------------------------------------
    def _encode(self, encoder, x, n_index=None):
        """encoder(x) -> L2-normalised (B, dim); x may be a strided clip view."""
        feat = encoder[0](x, n_index=n_index) if n_index is not None else encoder[0](x)
        parts = _head_parts(encoder, self.dim) if FUSED_HEAD else None
        if parts is not None and not (torch.is_grad_enabled() and (
                feat.requires_grad or any(p.requires_grad for p in encoder[2].parameters()) or
                any(p.requires_grad for p in encoder[4].parameters()))):
            fc1, fc2 = parts
            return _head_forward(feat, fc1.weight, fc1.bias, fc2.weight, fc2.bias)[0]
        for mod in list(encoder)[1:]:
            feat = mod(feat)
        return _L2NormFn.apply(feat.view(feat.shape[0], self.dim))

    # -- shuffle-BN exchange: every clip crosses the fabric once ---------------------------
    def _host_group(self):
        """gloo side channel for host-resident metadata (the 8*B-entry permutation): lets every
        rank learn rank 0's permutation on the HOST without a device->host sync."""
        g = InfoNCE.__dict__.get("_HOST_GROUP")
        if g is None:
            if os.environ.get("MASTER_ADDR", "") in ("127.0.0.1", "localhost", "::1"):
                # single-node rendezvous: gloo would otherwise look up the host NAME, which a
                # container need not be able to resolve
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            g = dist.new_group(backend="gloo") if dist.get_backend() != "gloo" else dist.group.WORLD
            InfoNCE._HOST_GROUP = g
        return g

    def _host_perm(self, BW):
        """The step's permutation of the global batch, identical on every rank, on the HOST: CPU randperm
        (same RNG use as the reference, :112) broadcast from rank 0 over the gloo side channel (ref :115)."""
        perm = torch.randperm(BW)
        _coll("host broadcast of the permutation over gloo (pretrain._host_perm; ref :115)",
              lambda: dist.broadcast(perm, src=0, group=self._host_group()), 8 * BW)
        return perm

    @torch.no_grad()
    def _routed_shuffle(self, x2, perm=None):
        """Shuffle-BN input exchange (ref :98-124) as a routed all-to-all.

        The reference all-gathers every rank's key clips (201 MB out, 1.4 GB in per rank at
        world 8) and then keeps B of the B*world it received.  The permutation is known on the
        host, so each rank sends a clip only to the rank that will encode it: B clips leave and
        B clips arrive per rank, whatever the world size.
        Returns (recv buffer (B, C, T, H, W), n_index: row of the buffer holding the i-th clip
        of this rank's shuffled mini-batch, idx_unshuffle).
        Host side: numpy routing, ONE pinned staging buffer and ONE host-to-device copy for the
        three index vectors (send order, receive positions, un-shuffle)."""
        world, rank = _world()
        B = x2.shape[0]
        BW = B * world
        if perm is None:
            perm = self._host_perm(BW)
        pn = perm.numpy()
        src = pn // B
        mine = np.nonzero(src == rank)[0]
        wsrc = src[rank * B:(rank + 1) * B]
        in_splits = np.bincount(mine // B, minlength=world).tolist()
        out_splits = np.bincount(wsrc, minlength=world).tolist()
        dev = x2.device
        st = self.__dict__.get("_route_staging")
        if st is None or st[0].shape[0] != 2 * B + BW or st[1].device != dev:
            host = torch.empty(2 * B + BW, dtype=torch.int64)
            if dev.type == "cuda":
                host = host.pin_memory()
            st = self.__dict__["_route_staging"] = [host, torch.empty(2 * B + BW, dtype=torch.int64,
                                                                      device=dev), None]
        host, devbuf, ev = st
        if ev is not None:
            ev.synchronize()                              # last step's copy has read the buffer
        h = host.numpy()
        h[:B] = pn[mine] % B                              # send order: by destination, then position
        h[B:2 * B][np.argsort(wsrc, kind="stable")] = np.arange(B)   # receive positions
        h[2 * B:] = np.argsort(pn, kind="stable")         # un-shuffle (a permutation: any sort)
        devbuf.copy_(host, non_blocking=True)
        if dev.type == "cuda":
            ev = torch.cuda.Event()
            ev.record()
            st[2] = ev
        order_t, n_index, idx_unshuffle = devbuf[:B], devbuf[B:2 * B], devbuf[2 * B:]
        sendbuf = torch.empty((B,) + tuple(x2.shape[1:]), dtype=x2.dtype, device=dev)
        ops.gather_rows(x2, order_t, sendbuf)
        recvbuf = torch.empty_like(sendbuf)
        _coll("all_to_all_single of %d key clips (pretrain._routed_shuffle; ref :106,124)" % B,
              lambda: dist.all_to_all_single(recvbuf, sendbuf, output_split_sizes=out_splits,
                                             input_split_sizes=in_splits),
              sendbuf.numel() * sendbuf.element_size(), dev)
        # the index vectors are views of a buffer the next step overwrites: the key path consumes
        # them (shuffle gather, un-shuffle) before this forward returns, on this stream
        return recvbuf, n_index, idx_unshuffle

    # -- shuffle-BN exchange as a peer row pull -----------------------------------------------
    def _peer_stage(self, x2):
        """Two staging buffers for this rank's key clips (double-buffered by step parity), mapped into
        every peer process through hipIpc (torch's CUDA-IPC tensor sharing carries the handles over
        the host channel).  Built once per input shape; a collective (every rank calls it)."""
        world, rank = _world()
        key = (tuple(x2.shape), x2.dtype, x2.device)
        st = self.__dict__.get("_peer_stage_state")
        if st is not None and st["key"] == key:
            return st
        from torch.multiprocessing.reductions import reduce_tensor
        local = [torch.empty(x2.shape, dtype=x2.dtype, device=x2.device) for _ in range(2)]
        if x2.is_cuda:
            torch.cuda.current_stream(x2.device).synchronize()
        objs = [None] * world
        err, peers, handles = None, [], None
        try:
            handles = [reduce_tensor(t) for t in local]
        except Exception as e:              # hipIpcGetMemHandle refused: still take part in the gather
            err = e
        _coll("all_gather_object of the hipIpc handles over gloo (pretrain._peer_stage)",
              lambda: dist.all_gather_object(objs, handles, group=self._host_group()))
        try:
            if err is None and any(o is None for o in objs):
                raise RuntimeError("a peer could not export its staging buffers")
            for r, lst in enumerate(objs if err is None else []):
                if r == rank:
                    peers.append(local)
                else:
                    mapped = [fn(*args) for fn, args in lst]
                    for t in mapped:
                        if t.device != x2.device:
                            t.reshape(-1)[:1].to(x2.device)      # makes torch enable peer access to it
                    peers.append(mapped)
        except Exception as e:              # hipIpc across devices / processes refused on this system
            err = e
        # the outcome must be the same on every rank (a rank pulling while another routes would hang):
        # agree on the host channel, and fall back TOGETHER
        ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32)
        _coll("host all_reduce(MIN) 'every rank mapped its peers' over gloo (pretrain._peer_stage)",
              lambda: dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self._host_group()))
        if int(ok) == 0:
            global _SHUFFLE_MODE
            import warnings
            why = "%s on rank %d" % (err if err is not None else "a peer could not map the staging buffers", rank)
            warnings.warn("coclr_amd: the peer row pull is not available here (%s); every rank uses the "
                          "routed all-to-all" % why)
            _SHUFFLE_MODE = "routed"
            _SHUFFLE_INFO.update(selected="routed", why="peer mapping refused: " + why[:200])
            self.__dict__["_peer_stage_state"] = None
            return None
        B = x2.shape[0]
        st = {"key": key, "local": local, "peers": peers, "step": 0,
              "base": np.array([[t.data_ptr() for t in pr] for pr in peers], dtype=np.int64),
              "row_bytes": x2[0].numel() * x2.element_size(),
              "ident": torch.arange(B, device=x2.device),
              "sync": torch.zeros(1, device=x2.device),
              "host": None, "dev": None, "event": None}
        self.__dict__["_peer_stage_state"] = st
        return st

    @torch.no_grad()
    def _pull_shuffle(self, x2, perm=None):
        """Shuffle-BN input exchange (ref :98-124) as a row pull: every rank parks its B key clips
        in a buffer its peers have mapped, and one HIP kernel (`coclr_pull_rows`) fetches the B clips
        this rank has to encode straight from their owners -- B clips cross the fabric per rank, no
        collective on the data path.  A one-element all-reduce is the stream-ordered barrier between
        "everybody has parked" and "pull"; the staging buffers alternate by step, and the key
        all-gather of the same step orders a peer's pull before this rank's next-but-one overwrite.
        Returns (recv buffer, identity n_index, idx_unshuffle)."""
        world, rank = _world()
        B = x2.shape[0]
        BW = B * world
        st = self._peer_stage(x2)
        if st is None:
            return None
        par = st["step"] & 1
        st["step"] += 1
        if perm is None:
            perm = self._host_perm(BW)
        pn = perm.numpy()
        dev = x2.device
        if st["host"] is None:
            st["host"] = torch.empty(B + BW, dtype=torch.int64).pin_memory()
            st["dev"] = torch.empty(B + BW, dtype=torch.int64, device=dev)
        if st["event"] is not None:
            st["event"].synchronize()
        h = st["host"].numpy()
        wanted = pn[rank * B:(rank + 1) * B]
        h[:B] = st["base"][wanted // B, par] + (wanted % B) * st["row_bytes"]
        h[B:] = np.argsort(pn, kind="stable")
        st["dev"].copy_(st["host"], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        st["event"] = ev
        ops.gather_rows(x2, st["ident"], st["local"][par])       # park my clips (x2 is a strided view)
        _coll("one-element all_reduce = 'everybody has parked' (pretrain._pull_shuffle)",
              lambda: dist.all_reduce(st["sync"]), 4, dev)          # stream-ordered barrier
        recv = torch.empty_like(st["local"][par])
        ops.pull_rows(st["dev"][:B], recv, keep=st["peers"])
        return recv, st["ident"], st["dev"][B:]

    @torch.no_grad()
    def _auto_shuffle(self, x2):
        """COCLR_SHUFFLE=auto, first exchange: routed AND pull on one permutation, compared bit for bit.
        Returns the routed result (the step proceeds on it) and leaves _SHUFFLE_MODE decided: "pull" when
        every rank received exactly the same clips both ways, "routed" otherwise.  One host synchronisation,
        once per process."""
        global _SHUFFLE_MODE
        world, rank = _world()
        if not x2.is_cuda:
            _SHUFFLE_MODE = "routed"
            _SHUFFLE_INFO.update(selected="routed", why="host tensors: nothing to map")
            return self._routed_shuffle(x2)
        if self._peer_stage(x2) is None:              # collective; on refusal every rank is on "routed" now
            return self._routed_shuffle(x2)
        perm = self._host_perm(x2.shape[0] * world)
        routed = self._routed_shuffle(x2, perm=perm)
        same, why = 0, "the pull raised"
        st = self.__dict__.get("_peer_stage_state")
        issued = _par_issued()
        try:
            pulled = self._pull_shuffle(x2, perm=perm)
            if pulled is not None:
                same = int(torch.equal(routed[0].index_select(0, routed[1]), pulled[0])
                           and torch.equal(routed[2], pulled[2]))
                why = "the clips pulled differ from the clips routed" if not same else ""
        except Exception as e:            # the verdict below must still be reached by every rank
            why = "the pull raised %s" % (str(e)[:160],)
            # ... and so must the DEVICE collective sequence stay aligned: the peers enqueued the pull's
            # one-element all-reduce; if this rank raised in front of it, it posts the matching call now
            # (otherwise the next all-to-all here would pair with that stale all-reduce over there)
            if st is not None and _par_issued() == issued:
                try:
                    _coll("one-element all_reduce matching the peers' 'everybody has parked' (pretrain._auto_shuffle)",
                          lambda: dist.all_reduce(st["sync"]), 4, x2.device)
                except Exception:
                    pass
        ok = torch.tensor([same], dtype=torch.int32)
        _coll("host all_reduce(MIN) 'pull == routed on every rank' over gloo (pretrain._auto_shuffle)",
              lambda: dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self._host_group()))
        if int(ok) == 1:
            _SHUFFLE_MODE = "pull"
            _SHUFFLE_INFO.update(selected="pull", why="first exchange: pulled clips bit-identical to the routed "
                                                      "all-to-all's on every rank")
        else:
            import warnings
            _SHUFFLE_MODE = "routed"
            _SHUFFLE_INFO.update(selected="routed", why=why or "a peer's pull differed or raised")
            warnings.warn("coclr_amd: peer row pull rejected (%s); every rank stays on the routed all-to-all"
                          % _SHUFFLE_INFO["why"])
        return routed

    @torch.no_grad()
    def _encode_keys(self, x2, pre=None):
        """Key path: [pre = momentum update] -> shuffle -> encoder_k -> normalise -> un-shuffle.
        Returns (k for this rank's samples, keys of the whole global batch in order)."""
        world, rank = _world()
        B = x2.shape[0]
        pulled = None
        if world > 1 and _SHUFFLE_MODE == "auto":
            pulled = self._auto_shuffle(x2)
        elif world > 1 and _SHUFFLE_MODE == "pull":
            pulled = self._pull_shuffle(x2)
        if pulled is not None:
            src, n_index, idx_unshuffle = pulled
        elif world > 1 and _SHUFFLE_MODE != "allgather":
            src, n_index, idx_unshuffle = self._routed_shuffle(x2)
        elif world > 1:
            # the reference's own scheme (all-gather, keep B of B*world): COCLR_SHUFFLE=allgather
            n_index, idx_unshuffle = self._shuffle_indices(B, x2.device)
            src = concat_all_gather(x2)
        else:
            n_index, idx_unshuffle = self._shuffle_indices(B, x2.device)
            src = x2
        k_shuf = self._encode_graphed(self.encoder_k, src, n_index.contiguous(), pre=pre)
        k_all_shuf = concat_all_gather(k_shuf)
        k_all = torch.empty_like(k_all_shuf)
        ops.gather_rows(k_all_shuf.contiguous(), idx_unshuffle.contiguous(), k_all)
        return k_all[rank * B:(rank + 1) * B], k_all

    def _query_logits(self, x1, k, side, in_train_mode):
        """q = normalize(encoder_q(x1)); logits = [<q,k> | q . queue] / T (ref :153-155,175-182)."""
        parts = _head_parts(self.encoder_q, self.dim) if FUSED_HEAD else None
        if parts is None or not in_train_mode:
            q = self._encode(self.encoder_q, x1)
            assert q.requires_grad == in_train_mode
            self._join(side)
            return _NceLogitsFn.apply(q, k.contiguous(), self.queue, float(self.T))
        fc1, fc2 = parts
        feat = self.encoder_q[0](x1)
        logits = _QueryHeadFn.apply(feat, fc1.weight, fc1.bias, fc2.weight, fc2.bias, k.contiguous(),
                                    self.queue, float(self.T), lambda: self._join(side))
        assert logits.requires_grad
        return logits

    def _split_pair(self, block):
        (B, N, *_) = block.shape  # [B,N,C,T,H,W]
        assert N == 2
        return block[:, 0], block[:, 1]

    def forward(self, block):
        '''Output: logits, targets'''
        self._sync_buffers()
        x1, x2 = self._split_pair(block)
        B = x1.shape[0]

        side = self._key_stream(x1)
        in_train_mode = self._query_requires_grad(x1)    # == q.requires_grad (ref :157)
        # key path first: it is one graph replay on the side stream, so both encoders start
        # together instead of the key encoder waiting for the host to enqueue the query one
        with torch.no_grad(), torch.cuda.stream(side):
            k, k_all = self._encode_keys(
                x2, pre=self._momentum_pre(in_train_mode))
        logits = self._query_logits(x1, k, side, in_train_mode)
        labels = torch.zeros(B, dtype=torch.long, device=logits.device)

        if in_train_mode:
            self._enqueue_gathered(k_all)
        return logits, labels


def _bool_mask(mask_u8):
    return mask_u8.view(torch.bool)


class UberNCE(InfoNCE):
    '''Supervised InfoNCE: positives are queue entries with the same label (ref :193-278).'''

    def __init__(self, network='s3d', dim=128, K=2048, m=0.999, T=0.07):
        super().__init__(network, dim, K, m, T)
        self.register_buffer("queue_label", torch.ones(K, dtype=torch.long) * -1)

    @torch.no_grad()
    def _dequeue_and_enqueue(self, keys, labels):
        labels_all = concat_all_gather(labels).contiguous()
        self._enqueue_gathered(concat_all_gather(keys), extra=[(self.queue_label, labels_all, 0)])

    def forward(self, block, k_label):
        '''Output: logits, binary mask for positive pairs'''
        self._sync_buffers()
        x1, x2 = self._split_pair(block)
        B = x1.shape[0]

        side = self._key_stream(x1)
        in_train_mode = self._query_requires_grad(x1)    # == q.requires_grad (ref :157)
        # key path first: it is one graph replay on the side stream, so both encoders start
        # together instead of the key encoder waiting for the host to enqueue the query one
        with torch.no_grad(), torch.cuda.stream(side):
            k, k_all = self._encode_keys(
                x2, pre=self._momentum_pre(in_train_mode))
        logits = self._query_logits(x1, k, side, in_train_mode)

        # mask[:,0] = True, mask[:,1+j] = (k_label == queue_label[j])   (ref :267-269)
        k_label = k_label.to(device=logits.device, dtype=torch.long).contiguous()
        mask = torch.empty(B, 1 + self.K, dtype=torch.uint8, device=logits.device)
        ops.positive_mask(None, k_label, self.queue_label, mask, 0)

        if in_train_mode:
            labels_all = concat_all_gather(k_label).contiguous()
            self._enqueue_gathered(k_all, extra=[(self.queue_label, labels_all, 0)])
        return logits, _bool_mask(mask)


class CoCLR(InfoNCE):
    '''CoCLR: positives mined in the queue of the other modality (ref :281-418).'''

    def __init__(self, network='s3d', dim=128, K=2048, m=0.999, T=0.07, topk=5, reverse=False):
        super().__init__(network, dim, K, m, T)
        self.topk = topk

        # frozen encoder of the second view
        self.sampler, _ = _make_encoder(network, dim)
        for param_s in self.sampler.parameters():
            param_s.requires_grad = False

        self.register_buffer("queue_second", torch.randn(dim, K))
        self.queue_second = nn.functional.normalize(self.queue_second, dim=0)
        self.register_buffer("queue_vname", torch.ones(K, dtype=torch.long) * -1)
        self.register_buffer("queue_label", torch.ones(K, dtype=torch.long) * -1)

        self.queue_is_full = False
        self.reverse = reverse

    @torch.no_grad()
    def _enqueue_coclr(self, keys_all, keys_second_all, vnames_all):
        batch_size = keys_all.shape[0]
        assert self.K % batch_size == 0  # for simplicity
        self._check_queue_ptr(batch_size)
        ops.queue_enqueue(self.queue, keys_all.contiguous(), self.queue_ptr)
        ops.queue_enqueue(self.queue_second, keys_second_all.contiguous(), self.queue_ptr)
        ops.queue_fill_i64(self.queue_vname, vnames_all.contiguous(), 0, batch_size,
                           self.queue_ptr)
        ops.queue_fill_i64(self.queue_label, None, 1, batch_size, self.queue_ptr)
        ops.queue_advance(self.queue_ptr, batch_size, self.K)

    @torch.no_grad()
    def _dequeue_and_enqueue(self, keys, keys_second, vnames):
        self._enqueue_coclr(concat_all_gather(keys), concat_all_gather(keys_second),
                            concat_all_gather(vnames))

    def forward(self, block1, block2, k_vsource):
        '''Output: logits, targets'''
        self._sync_buffers()
        x1, f1 = self._split_pair(block1)
        x2, f2 = self._split_pair(block2)
        if self.reverse:
            x1, f1 = f1, x1
            x2, f2 = f2, x2
        B = x1.shape[0]

        side = self._key_stream(x1)
        in_train_mode = self._query_requires_grad(x1)    # == q.requires_grad (ref :157)
        with torch.no_grad(), torch.cuda.stream(side):
            k, k_all = self._encode_keys(
                x2, pre=self._momentum_pre(in_train_mode))
            # second view: frozen sampler (eval-mode BN in the reference's training loop),
            # not shuffled
            ident = self.__dict__.get("_ident_idx")
            if ident is None or ident.shape[0] != B or ident.device != f2.device:
                ident = self.__dict__["_ident_idx"] = torch.arange(B, device=f2.device)
            kf = self._encode_graphed(self.sampler, f2, ident)
        logits = self._query_logits(x1, k, side, in_train_mode)

        k_vsource = k_vsource.to(device=logits.device, dtype=torch.long).contiguous()
        if not self.queue_is_full:
            # one host sync per step until the queue has wrapped once (ref :400-402); a
            # plain bool afterwards so the steady state never blocks on the device
            self.queue_is_full = bool(torch.all(self.queue_label != -1))
            if self.queue_is_full:
                print('\n===== queue is full now =====')

        mask = torch.empty(B, 1 + self.K, dtype=torch.uint8, device=logits.device)
        if self.queue_is_full and (self.topk != 0):
            # cross-modal similarity against the second queue, top-k mined per row with
            # same-source (sibling) entries excluded (ref :405-410)
            if self.dim == 128 and 0 < int(self.topk) <= 16:
                # similarity tiles on the MFMA pipe with a running top-k: one launch, no (B, K) tensor
                ws = self.__dict__.get("_mine_ws")
                if ws is None or ws[3] != (B, self.K, int(self.topk), kf.device):
                    ws = self.__dict__["_mine_ws"] = ops.mine_workspace(
                        B, self.K, int(self.topk), kf.device) + ((B, self.K, int(self.topk), kf.device),)
                ops.mine_positives(kf.contiguous(), self.queue_second, k_vsource, self.queue_vname,
                                   mask, int(self.topk), ws[:3])
            else:
                sim = torch.empty(B, self.K, dtype=kf.dtype, device=kf.device)
                ops.gemm(kf, self.dim, 1, self.queue_second, self.K, 1, sim, self.K, None, B, self.K,
                         self.dim)
                ops.positive_mask(sim, k_vsource, self.queue_vname, mask, int(self.topk))
        else:
            ops.positive_mask(None, k_vsource, self.queue_vname, mask, 0)

        if in_train_mode:
            self._enqueue_coclr(k_all, concat_all_gather(kf), concat_all_gather(k_vsource))
        return logits, _bool_mask(mask).detach()
