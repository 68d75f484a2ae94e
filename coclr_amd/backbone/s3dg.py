"""S3D / S3D-G separable-3D-conv backbone on the gfx950 kernel engine.

Same module tree, parameter/buffer names, init distributions and RNG
consumption order as the reference backbone/s3dg.py (BasicConv3d :8-28,
STConv3d :30-65, SelfGating :68-78, SepInception :81-132, S3D :135-217), so
reference checkpoints load with strict=True -- including the alias keys that
come from registering every stage both by name and inside `blockN`.

The nn.Conv3d / nn.BatchNorm3d children are parameter containers only; their
ATen forward is never called.  Each module implements `_emit(run, val)`, which
enqueues HIP kernels through coclr_amd.engine; `forward` wraps a whole
(sub)network in a single autograd node.
"""
import os

import torch
import torch.nn as nn

from .. import engine

# One autograd node per stage lets DistributedDataParallel start reducing the late stages' gradients
# while the early stages are still in backward; with a single rank there is nothing to overlap and
# the per-stage joins of the weight-gradient stream only cost time (889 -> 899 clips/s as one node).
# "auto" (default): split iff the process group has more than one rank; "1" / "0" force it.
_SPLIT_MODE = os.environ.get("COCLR_SPLIT_STAGES", "auto")


def _split_stages():
    if _SPLIT_MODE in ("0", "1"):
        return _SPLIT_MODE == "1"
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class _Emitter(nn.Module):
    """nn.Module whose forward is an engine run over `_emit`."""

    def forward(self, x, **kw):
        return engine.run_module(self, x, **kw)


def _conv_bn(cin, cout, kernel, stride, padding):
    """(Conv3d, BatchNorm3d) initialised like s3dg.py:20-22 / :51-56."""
    conv = nn.Conv3d(cin, cout, kernel_size=kernel, stride=stride, padding=padding, bias=False)
    bn = nn.BatchNorm3d(cout)          # PyTorch defaults: eps 1e-5, momentum 0.1 (s3dg.py:5,16)
    return conv, bn


def _init_conv_bn(conv, bn):
    conv.weight.data.normal_(mean=0, std=0.01)
    bn.weight.data.fill_(1)
    bn.bias.data.zero_()


class BasicConv3d(_Emitter):
    """conv(bias=False) -> BN -> ReLU."""

    def __init__(self, in_planes, out_planes, kernel_size, stride, padding=0):
        super().__init__()
        self.conv, self.bn = _conv_bn(in_planes, out_planes, kernel_size, stride, padding)
        self.relu = nn.ReLU(inplace=True)
        _init_conv_bn(self.conv, self.bn)

    def _emit(self, run, x, out=None, n_index=None):
        return engine.conv_bn_act(run, x, self.conv, self.bn, relu=True, out=out, n_index=n_index)

    def _emit_gen(self, run, x, out=None):
        return (yield from engine.conv_bn_act_gen(run, x, self.conv, self.bn, relu=True, out=out))


class STConv3d(_Emitter):
    """Separable conv: (1,k,k) spatial then (k,1,1) temporal, each with BN+ReLU.
    A tuple stride means (t_stride, ..., spatial_stride) as in s3dg.py:33-37."""

    def __init__(self, in_planes, out_planes, kernel_size, stride, padding=0):
        super().__init__()
        if isinstance(stride, tuple):
            t_stride, s_stride = stride[0], stride[-1]
        else:
            t_stride = s_stride = stride
        self.conv1 = nn.Conv3d(in_planes, out_planes, kernel_size=(1, kernel_size, kernel_size),
                               stride=(1, s_stride, s_stride), padding=(0, padding, padding),
                               bias=False)
        self.conv2 = nn.Conv3d(out_planes, out_planes, kernel_size=(kernel_size, 1, 1),
                               stride=(t_stride, 1, 1), padding=(padding, 0, 0), bias=False)
        self.bn1 = nn.BatchNorm3d(out_planes)
        self.bn2 = nn.BatchNorm3d(out_planes)
        self.relu = nn.ReLU(inplace=True)
        # RNG order of the reference: conv1 weights, then conv2 weights
        self.conv1.weight.data.normal_(mean=0, std=0.01)
        self.conv2.weight.data.normal_(mean=0, std=0.01)
        for bn in (self.bn1, self.bn2):
            bn.weight.data.fill_(1)
            bn.bias.data.zero_()

    def _emit(self, run, x, out=None, n_index=None):
        mid = engine.conv_bn_act(run, x, self.conv1, self.bn1, relu=True, n_index=n_index)
        return engine.conv_bn_act(run, mid, self.conv2, self.bn2, relu=True, out=out)

    def _emit_gen(self, run, x, out=None):
        """The same two units as a coroutine of launch requests (engine.drive_pair runs two of them in
        lockstep: the branch1 / branch2 tails of an inception block)."""
        mid = yield from engine.conv_bn_act_gen(run, x, self.conv1, self.bn1, relu=True)
        return (yield from engine.conv_bn_act_gen(run, mid, self.conv2, self.bn2, relu=True, out=out))


class SelfGating(_Emitter):
    def __init__(self, input_dim):
        super().__init__()
        self.fc = nn.Linear(input_dim, input_dim)

    def _emit(self, run, x, out=None):
        return engine.self_gating(run, x, self.fc, out=out)


class _Pool(nn.MaxPool3d):
    """nn.MaxPool3d node of the layer program."""

    def forward(self, x):
        return engine.run_module(self, x)

    def _emit(self, run, x):
        return engine.max_pool(run, x, self.kernel_size, self.stride, self.padding)


class _Chain(nn.Sequential):
    """nn.Sequential whose members are emitted into the caller's run."""

    def forward(self, x, **kw):
        return engine.run_module(self, x, **kw)

    def _emit(self, run, x, out=None, n_index=None):
        mods = list(self)
        for i, m in enumerate(mods):
            kw = {}
            if i == 0 and n_index is not None:
                kw["n_index"] = n_index
            if i == len(mods) - 1 and out is not None:
                kw["out"] = out
            x = m._emit(run, x, **kw)
        return x


class SepInception(_Emitter):
    """Four-branch separable inception block; out_planes = [b0, b1a, b1b, b2a, b2b, b3b]
    and the output channels are b0 | b1b | b2b | b3b in that order (s3dg.py:88-130)."""

    def __init__(self, in_planes, out_planes, gating=False):
        super().__init__()
        assert isinstance(out_planes, list) and len(out_planes) == 6
        b0, b1a, b1b, b2a, b2b, b3b = out_planes
        self.branch0 = _Chain(BasicConv3d(in_planes, b0, kernel_size=1, stride=1))
        self.branch1 = _Chain(BasicConv3d(in_planes, b1a, kernel_size=1, stride=1),
                              STConv3d(b1a, b1b, kernel_size=3, stride=1, padding=1))
        self.branch2 = _Chain(BasicConv3d(in_planes, b2a, kernel_size=1, stride=1),
                              STConv3d(b2a, b2b, kernel_size=3, stride=1, padding=1))
        self.branch3 = _Chain(_Pool(kernel_size=(3, 3, 3), stride=1, padding=1),
                              BasicConv3d(in_planes, b3b, kernel_size=1, stride=1))
        self._widths = (b0, b1b, b2b, b3b)
        self.out_channels = sum(self._widths)
        self.gating = gating
        if gating:
            self.gating_b0 = SelfGating(b0)
            self.gating_b1 = SelfGating(b1b)
            self.gating_b2 = SelfGating(b2b)
            self.gating_b3 = SelfGating(b3b)

    def _emit(self, run, x):
        N, _, T, H, W = x.shape
        odim = (T, H, W)          # every branch preserves the extent
        run.lanes_on = engine.lanes_for(run, odim)
        block = run.empty(N, self.out_channels, *odim)
        dst, c0 = [], 0
        for width in self._widths:
            dst.append(engine.Val(block, c0, width))
            c0 += width
        # the three 1x1x1 heads read the same x: one convolution over concatenated channels
        b0, b1a, b2a = self.branch0[0], self.branch1[0], self.branch2[0]
        head_units = [(b0.conv, b0.bn, None if self.gating else dst[0]), (b1a.conv, b1a.bn, None),
                      (b2a.conv, b2a.bn, None)]
        tails = (None, self.branch1[1], self.branch2[1])
        if engine.PAIR_UNITS and not self.gating and not run.lanes_on:
            # Sibling units in lockstep, one launch per step of the pair (engine.drive_pair):
            #   * the fused heads and the pool branch's 1x1x1 convolution (both pointwise, both on the block's
            #     map: x and the pooled x), then their four BatchNorm units;
            #   * the separable tails of branch 1 and branch 2 (same stencils): convolutions, BatchNorm
            #     passes and, in backward, BatchNorm backward passes and data gradients.
            pooled = self.branch3[0]._emit(run, x)
            heads, _ = engine.drive_pair(run, engine.pointwise_group_gen(run, x, head_units),
                                         self.branch3[1]._emit_gen(run, pooled, out=dst[3]))
            engine.drive_pair(run, tails[1]._emit_gen(run, heads[1], out=dst[1]),
                              tails[2]._emit_gen(run, heads[2], out=dst[2]))
            return engine.Val(block)
        heads = engine.pointwise_group(run, x, head_units)
        if self.gating:
            getattr(self, "gating_b0")._emit(run, heads[0], out=dst[0])
        # the separable tails of branch 1 / 2 and the pool branch are independent of each other:
        # one lane (HIP stream) each, joined before the block output is consumed
        for i in (1, 2, 3):
            with run.lane(i):
                if i == 3:
                    y = self.branch3._emit(run, x, out=None if self.gating else dst[3])
                else:
                    y = tails[i]._emit(run, heads[i], out=None if self.gating else dst[i])
                if self.gating:
                    getattr(self, "gating_b%d" % i)._emit(run, y, out=dst[i])
        run.join_lanes()
        return engine.Val(block)


# stage table: (attribute name, constructor) in registration order; `blockN`
# aliases are built from these exactly like s3dg.py:147-192.
_INCEPTION = {
    "Mixed_3b": (192, [64, 96, 128, 16, 32, 32]),
    "Mixed_3c": (256, [128, 128, 192, 32, 96, 64]),
    "Mixed_4b": (480, [192, 96, 208, 16, 48, 64]),
    "Mixed_4c": (512, [160, 112, 224, 24, 64, 64]),
    "Mixed_4d": (512, [128, 128, 256, 24, 64, 64]),
    "Mixed_4e": (512, [112, 144, 288, 32, 64, 64]),
    "Mixed_4f": (528, [256, 160, 320, 32, 128, 128]),
    "Mixed_5b": (832, [256, 160, 320, 32, 128, 128]),
    "Mixed_5c": (832, [384, 192, 384, 48, 128, 128]),
}


class S3D(_Emitter):
    def __init__(self, input_channel=3, gating=False, slow=False):
        super().__init__()
        self.gating = gating
        self.slow = slow

        stem_stride = (1, 2, 2) if slow else 2
        self.Conv_1a = STConv3d(input_channel, 64, kernel_size=7, stride=stem_stride, padding=3)
        self.block1 = _Chain(self.Conv_1a)

        self.MaxPool_2a = _Pool(kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))
        self.Conv_2b = BasicConv3d(64, 64, kernel_size=1, stride=1)
        self.Conv_2c = STConv3d(64, 192, kernel_size=3, stride=1, padding=1)
        self.block2 = _Chain(self.MaxPool_2a, self.Conv_2b, self.Conv_2c)

        self.MaxPool_3a = _Pool(kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))
        self._add_inceptions("Mixed_3b", "Mixed_3c")
        self.block3 = _Chain(self.MaxPool_3a, self.Mixed_3b, self.Mixed_3c)

        self.MaxPool_4a = _Pool(kernel_size=(3, 3, 3), stride=(2, 2, 2), padding=(1, 1, 1))
        self._add_inceptions("Mixed_4b", "Mixed_4c", "Mixed_4d", "Mixed_4e", "Mixed_4f")
        self.block4 = _Chain(self.MaxPool_4a, self.Mixed_4b, self.Mixed_4c, self.Mixed_4d,
                             self.Mixed_4e, self.Mixed_4f)

        self.MaxPool_5a = _Pool(kernel_size=(2, 2, 2), stride=(2, 2, 2), padding=(0, 0, 0))
        self._add_inceptions("Mixed_5b", "Mixed_5c")
        self.block5 = _Chain(self.MaxPool_5a, self.Mixed_5b, self.Mixed_5c)

    def _add_inceptions(self, *names):
        for name in names:
            cin, widths = _INCEPTION[name]
            setattr(self, name, SepInception(in_planes=cin, out_planes=list(widths),
                                             gating=self.gating))

    def _emit(self, run, x, n_index=None):
        x = self.block1._emit(run, x, n_index=n_index)
        for blk in (self.block2, self.block3, self.block4, self.block5):
            x = blk._emit(run, x)
        return x

    def _stage_groups(self):
        """Autograd-node partition used when the backbone runs as several nodes: the reference's
        block1..block5 with every stage's LEADING max-pool moved to the end of the stage before it.
        The pool that follows Conv_1a / Conv_2c then sits in the same engine run as the unit it
        consumes and applies that unit's BatchNorm + ReLU while it reads (engine.max_pool, lazy apply)
        -- a node boundary between them would force the 1 GB / 0.4 GB apply pass back in.  Plain
        objects, not registered modules: the state dict keeps the reference's keys."""
        groups = self.__dict__.get("_coclr_groups")
        if groups is None:
            b2, b3, b4, b5 = list(self.block2), list(self.block3), list(self.block4), list(self.block5)
            groups = self.__dict__["_coclr_groups"] = [
                _Group(list(self.block1) + b2[:1]), _Group(b2[1:] + b3[:1]), _Group(b3[1:] + b4[:1]),
                _Group(b4[1:] + b5[:1]), _Group(b5[1:])]
            # every node but the one that runs LAST in backward (stage 1) may leave the weight-gradient
            # stream un-joined when its gradients live in DDP's buckets (engine.Run.defer_side)
            for grp in groups[1:]:
                grp.__dict__["_coclr_defer_join"] = True
        return groups

    def _late_split(self):
        """(everything up to MaxPool_4a, Mixed_4b..Mixed_5c): the second part is the launch-bound one
        that engine.GRAPH_LATE replays from a hipGraph."""
        pair = self.__dict__.get("_coclr_late_split")
        if pair is None:
            b4 = list(self.block4)
            front = _Group(list(self.block1) + list(self.block2) + list(self.block3) + b4[:1])
            late = _Group(b4[1:] + list(self.block5))
            late.__dict__["_coclr_graph_late"] = True
            pair = self.__dict__["_coclr_late_split"] = (front, late)
        return pair

    def forward(self, x, n_index=None):
        """One autograd node per stage, like the reference's forward (backbone/s3dg.py:211-217).  With
        gradients enabled this lets the gradients of the late stages -- 216 of the 231 backbone tensors
        live in block3-5 -- reach DistributedDataParallel while the early stages are still in
        backward: its bucket all-reduce then overlaps the weight-gradient stream instead of forming a
        tail after the whole backward."""
        if not (torch.is_grad_enabled() and _split_stages()):
            if engine.GRAPH_LATE and torch.is_grad_enabled():
                front, late = self._late_split()
                x = engine.run_module(front, x, n_index=n_index) if n_index is not None \
                    else engine.run_module(front, x)
                return engine.run_module(late, x)
            return engine.run_module(self, x, n_index=n_index) if n_index is not None \
                else engine.run_module(self, x)
        groups = self._stage_groups()
        x = engine.run_module(groups[0], x, n_index=n_index) if n_index is not None \
            else engine.run_module(groups[0], x)
        for grp in groups[1:]:
            x = engine.run_module(grp, x)
        return x


class _Group:
    """A run of backbone modules executed as ONE engine run / autograd node (see S3D._stage_groups).
    Not an nn.Module: nothing is registered, `parameters()` is what engine.run_module needs."""

    def __init__(self, mods):
        self.mods = mods

    def parameters(self):
        for m in self.mods:
            yield from m.parameters()

    def modules(self):
        for m in self.mods:
            yield from m.modules()

    def _emit(self, run, x, n_index=None):
        for i, m in enumerate(self.mods):
            x = m._emit(run, x, n_index=n_index) if (i == 0 and n_index is not None) else m._emit(run, x)
        return x
