"""Drop-in for the reference's `backbone.resnet_2d3d` import path."""
from coclr_amd.backbone.resnet_2d3d import (Bottleneck2d, Bottleneck3d, ResNet2d3d,  # noqa: F401
                                            r2d3d50, r3d50)
