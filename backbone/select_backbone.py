"""Drop-in for the reference's `backbone.select_backbone` import path
(model/pretrain.py:10, model/classifier.py:7)."""
from coclr_amd.backbone.select_backbone import select_backbone  # noqa: F401
