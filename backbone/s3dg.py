"""Drop-in for the reference's `backbone.s3dg` import path."""
from coclr_amd.backbone.s3dg import (BasicConv3d, STConv3d, SelfGating, SepInception,  # noqa: F401
                                     S3D)
