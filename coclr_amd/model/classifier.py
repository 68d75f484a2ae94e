"""LinearClassifier (linear probe / fine-tuning head) on the gfx950 kernel library.

API, attribute and state-dict names follow the reference model/classifier.py:10-69
(`backbone`, `final_bn`, `final_fc.{0|1}`; forward returns `(logit, feat3d)`); the evaluation
scripts (eval/main_classifier.py:66-72,180,253) construct and load it unchanged.  Below the API the
backbone runs on coclr_amd.engine (eval-mode passes fold BatchNorm + ReLU into the convolution
epilogue: one pass per layer) and the head is the same GEMM / pooling / L2-norm / BatchNorm kernels
as the contrastive head -- no ATen op computes anything.
"""
import torch
import torch.nn as nn

from .. import ops
from ..backbone.select_backbone import select_backbone
from .pretrain import _AvgPoolFn, _L2NormFn, _PointwiseFn


class _Bn1dFn(torch.autograd.Function):
    """nn.BatchNorm1d on (N, C) features through the BatchNorm3d kernels (one position per plane)."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn):
        N, C_ = x.shape
        x = x.contiguous()
        small = torch.empty(4, C_, dtype=x.dtype, device=x.device)
        mean, invstd, scale, shift = small[0], small[1], small[2], small[3]
        training = bn.training or bn.running_mean is None
        if training:
            if bn.momentum is None:
                raise NotImplementedError("coclr_amd: cumulative-average BatchNorm momentum")
            if N < 2:
                raise ValueError("Expected more than 1 value per channel when training")
            stats = torch.empty(2 * C_, dtype=x.dtype, device=x.device)
            ws = torch.empty(ops.colstats_workspace(N, C_), dtype=x.dtype, device=x.device)
            ops.bn1d_stats(x, stats, ws)
            ops.bn_finalize(stats, C_, 1, N, weight, bias, bn.running_mean, bn.running_var,
                            bn.num_batches_tracked, float(bn.momentum), float(bn.eps), mean, invstd,
                            scale, shift)
        else:
            ops.bn_eval_affine(weight, bias, bn.running_mean, bn.running_var, float(bn.eps), C_,
                               mean, invstd, scale, shift)
        z = torch.empty_like(x)
        ops.bn_act_apply(x.view(N, C_, 1, 1, 1), scale, shift, None, z.view(N, C_, 1, 1, 1), False)
        ctx.save_for_backward(x, small)
        ctx.training = training
        return z

    @staticmethod
    def backward(ctx, dz):
        x, small = ctx.saved_tensors
        N, C_ = x.shape
        mean, invstd, scale, shift = small[0], small[1], small[2], small[3]
        dz = dz.contiguous()
        dy = torch.empty_like(x)
        dgb = torch.empty(2, C_, dtype=x.dtype, device=x.device)
        sums = torch.empty(ops.bn_backward_workspace(N, C_), dtype=torch.float64, device=x.device)
        v5 = (N, C_, 1, 1, 1)
        ops.bn_act_backward(dz.view(v5), x.view(v5), None, scale, shift, mean, invstd, sums,
                            dy.view(v5), None, dgb[0], dgb[1], False, ctx.training)
        return dy, dgb[0], dgb[1], None


class FeatureBatchNorm1d(nn.BatchNorm1d):
    def forward(self, x):
        if x.dim() != 2:
            raise NotImplementedError("coclr_amd: final_bn expects (N, C) features")
        return _Bn1dFn.apply(x, self.weight, self.bias, self)


class _ScaleFn(torch.autograd.Function):
    """y = x * gain elementwise on (N, C) (dropout with a precomputed keep/(1-p) gain)."""

    @staticmethod
    def forward(ctx, x, gain):
        N, C_ = x.shape
        x = x.contiguous()
        y = torch.empty_like(x)
        ops.plane_scale(x.view(N, C_, 1, 1, 1), gain, None, y.view(N, C_, 1, 1, 1))
        ctx.save_for_backward(gain)
        return y

    @staticmethod
    def backward(ctx, dy):
        (gain,) = ctx.saved_tensors
        N, C_ = dy.shape
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        ops.plane_scale(dy.view(N, C_, 1, 1, 1), gain, None, dx.view(N, C_, 1, 1, 1))
        return dx, None


class FeatureDropout(nn.Dropout):
    """nn.Dropout on (N, C) features: torch's generator draws the keep mask (random numbers are
    plumbing, not arithmetic), the scaling runs on the HIP kernel.  Identity in eval mode."""

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        if self.p >= 1.0:
            gain = torch.zeros_like(x)
        else:
            gain = torch.empty_like(x).bernoulli_(1.0 - self.p).div_(1.0 - self.p)
        return _ScaleFn.apply(x, gain.contiguous())


class FeatureLinear(nn.Linear):
    """nn.Linear on (N, C) features as the fp32-MFMA GEMM of the projection head."""

    def forward(self, x):
        N = x.shape[0]
        w = self.weight
        y = _PointwiseFn.apply(x.reshape(N, w.shape[1], 1, 1, 1), w.view(w.shape[0], w.shape[1], 1, 1, 1),
                               self.bias)
        return y.view(N, w.shape[0])


class LinearClassifier(nn.Module):
    def __init__(self, num_class=101,
                 network='resnet50',
                 dropout=0.5,
                 use_dropout=True,
                 use_l2_norm=False,
                 use_final_bn=False):
        super(LinearClassifier, self).__init__()
        self.network = network
        self.num_class = num_class
        self.dropout = dropout
        self.use_dropout = use_dropout
        self.use_l2_norm = use_l2_norm
        self.use_final_bn = use_final_bn

        message = 'Classifier to %d classes with %s backbone;' % (num_class, network)
        if use_dropout: message += ' + dropout %f' % dropout
        if use_l2_norm: message += ' + L2Norm'
        if use_final_bn: message += ' + final BN'
        print(message)

        self.backbone, self.param = select_backbone(network)
        fs = self.param['feature_size']

        if use_final_bn:
            self.final_bn = FeatureBatchNorm1d(fs)
            self.final_bn.weight.data.fill_(1)
            self.final_bn.bias.data.zero_()

        if use_dropout:
            self.final_fc = nn.Sequential(FeatureDropout(dropout), FeatureLinear(fs, self.num_class))
        else:
            self.final_fc = nn.Sequential(FeatureLinear(fs, self.num_class))
        self._initialize_weights(self.final_fc)
        from .. import optim as _optim
        _optim.register_model(self)          # its parameters may take the single-launch Adam step

    def forward(self, block):
        (B, C, T, H, W) = block.shape
        feat3d = self.backbone(block)
        feat3d = _AvgPoolFn.apply(feat3d)                     # [B,C,1,1,1]
        feat3d = feat3d.view(B, self.param['feature_size'])   # [B,C]

        if self.use_l2_norm:
            feat3d = _L2NormFn.apply(feat3d)

        if self.use_final_bn:
            logit = self.final_fc(self.final_bn(feat3d))
        else:
            logit = self.final_fc(feat3d)

        return logit, feat3d

    def _initialize_weights(self, module):
        for name, param in module.named_parameters():
            if 'bias' in name:
                nn.init.constant_(param, 0.0)
            elif 'weight' in name:
                nn.init.normal_(param, mean=0.0, std=0.01)
