"""ResNet-50 with 2D (1x3x3) early stages and 3D ((3,1,1)+(1,3,3)) late stages on
the gfx950 kernel engine.

Module tree / state-dict names / init follow the reference
backbone/resnet_2d3d.py: Bottleneck3d :46-86, Bottleneck2d :89-129,
ResNet2d3d :133-202 (kaiming-normal fan_out conv init :150-156, last block of
layer4 without ReLU :184 but `F.relu` on the way out :202), r2d3d50 :206-210.
`r3d50` is kept for API parity; in the reference it dies with a NameError
(`BasicBlock2d`, :163) -- here it builds.
"""
import torch.nn as nn

from .. import engine
from .s3dg import _Chain, _Emitter, _Pool

__all__ = ["ResNet2d3d", "r2d3d50", "r3d50"]


class _Bottleneck(_Emitter):
    expansion = 4
    first_kernel = (1, 1, 1)      # conv1 stencil; the 3D flavour uses (3,1,1)

    def __init__(self, inplanes, planes, stride=1, downsample=None, use_final_relu=True):
        super().__init__()
        self.use_final_relu = use_final_relu
        kt = self.first_kernel[0]
        self.conv1 = nn.Conv3d(inplanes, planes, kernel_size=self.first_kernel if kt > 1 else 1,
                               padding=(kt // 2, 0, 0) if kt > 1 else 0, bias=False)
        self.bn1 = nn.BatchNorm3d(planes)
        self.conv2 = nn.Conv3d(planes, planes, kernel_size=(1, 3, 3), stride=(1, stride, stride),
                               padding=(0, 1, 1), bias=False)
        self.bn2 = nn.BatchNorm3d(planes)
        self.conv3 = nn.Conv3d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm3d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def _emit(self, run, x, force_relu=False):
        h = engine.conv_bn_act(run, x, self.conv1, self.bn1, relu=True)
        h = engine.conv_bn_act(run, h, self.conv2, self.bn2, relu=True)
        res = x if self.downsample is None else self.downsample._emit(run, x)
        # out = relu?(bn3(conv3(h)) + residual), fused into the BN apply pass
        return engine.conv_bn_act(run, h, self.conv3, self.bn3,
                                  relu=self.use_final_relu or force_relu, residual=res)


class Bottleneck3d(_Bottleneck):
    first_kernel = (3, 1, 1)


class Bottleneck2d(_Bottleneck):
    first_kernel = (1, 1, 1)


class _Downsample(_Chain):
    """Sequential(conv1x1x1(stride), BN) on the shortcut (resnet_2d3d.py:172-175)."""

    def _emit(self, run, x):
        return engine.conv_bn_act(run, x, self[0], self[1], relu=False)


class _Stage(_Chain):
    def _emit(self, run, x, force_relu_last=False):
        blocks = list(self)
        for i, b in enumerate(blocks):
            x = b._emit(run, x, force_relu=force_relu_last and i == len(blocks) - 1)
        return x


class ResNet2d3d(_Emitter):
    def __init__(self, block, layers, input_channel=3):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv3d(input_channel, 64, kernel_size=(5, 7, 7), stride=(2, 2, 2),
                               padding=(2, 3, 3), bias=False)
        self.bn1 = nn.BatchNorm3d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = _Pool(kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))
        if not isinstance(block, list):
            block = [block] * 4
        self.layer1 = self._make_layer(block[0], 64, layers[0])
        self.layer2 = self._make_layer(block[1], 128, layers[1], stride=(1, 2, 2))
        self.layer3 = self._make_layer(block[2], 256, layers[2], stride=(1, 2, 2))
        self.layer4 = self._make_layer(block[3], 512, layers[3], stride=(1, 2, 2), is_final=True)
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                m.weight = nn.init.kaiming_normal_(m.weight, mode="fan_out")
            elif isinstance(m, nn.BatchNorm3d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1, is_final=False):
        shortcut = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            if isinstance(stride, tuple):
                conv_stride, stride = stride, stride[-1]
            elif block is Bottleneck2d:
                conv_stride = (1, stride, stride)
            else:
                conv_stride = stride
            shortcut = _Downsample(
                nn.Conv3d(self.inplanes, planes * block.expansion, kernel_size=1,
                          stride=conv_stride, bias=False),
                nn.BatchNorm3d(planes * block.expansion))
        members = [block(self.inplanes, planes, stride, shortcut)]
        self.inplanes = planes * block.expansion
        if is_final:
            members += [block(self.inplanes, planes) for _ in range(1, blocks - 1)]
            members.append(block(self.inplanes, planes, use_final_relu=False))
        else:
            members += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return _Stage(*members)

    def _emit(self, run, x, n_index=None):
        x = engine.conv_bn_act(run, x, self.conv1, self.bn1, relu=True, n_index=n_index)
        x = self.maxpool._emit(run, x)
        x = self.layer1._emit(run, x)
        x = self.layer2._emit(run, x)
        x = self.layer3._emit(run, x)
        # the reference applies F.relu to the un-rectified layer4 output; fold it
        # into the last block's epilogue
        return self.layer4._emit(run, x, force_relu_last=True)


def r2d3d50(**kwargs):
    return ResNet2d3d([Bottleneck2d, Bottleneck2d, Bottleneck3d, Bottleneck3d], [3, 4, 6, 3],
                      **kwargs)


def r3d50(**kwargs):
    return ResNet2d3d([Bottleneck3d, Bottleneck3d, Bottleneck3d, Bottleneck3d], [3, 4, 6, 3],
                      **kwargs)
