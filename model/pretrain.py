"""Drop-in for the reference's `model.pretrain` import path (main_nce.py:34,
main_coclr.py:34): re-exports the MI355X-native implementations."""
from coclr_amd.model.pretrain import InfoNCE, UberNCE, CoCLR, concat_all_gather  # noqa: F401
