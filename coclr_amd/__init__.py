"""MI355X-native CoCLR training hot path (S3D / ResNet2d3d backbones + MoCo-style
InfoNCE / UberNCE / CoCLR heads) on hand-written gfx950 HIP kernels.

Public surface mirrors the reference:
    from model.pretrain import InfoNCE, UberNCE, CoCLR
    from backbone.select_backbone import select_backbone
(the top-level `model/` and `backbone/` packages re-export from here).
"""
__version__ = "0.1.0"
