"""Backbone factory with the reference's contract (backbone/select_backbone.py:4-16):
`select_backbone(network, first_channel=3) -> (nn.Module, {'feature_size': int})`."""
from .s3dg import S3D
from .resnet_2d3d import r2d3d50

_FEATURE_SIZE = {"s3d": 1024, "s3dg": 1024, "r50": 2048}


def select_backbone(network, first_channel=3):
    if network not in _FEATURE_SIZE:
        raise NotImplementedError
    param = {"feature_size": _FEATURE_SIZE[network]}
    if network == "r50":
        model = r2d3d50(input_channel=first_channel)
    else:
        model = S3D(input_channel=first_channel, gating=(network == "s3dg"))
    return model, param
