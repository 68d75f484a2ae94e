"""Drop-in for the reference's `model.pretrain` import path (main_nce.py:34,
main_coclr.py:34): re-exports the MI355X-native implementations."""
from coclr_amd.model.pretrain import InfoNCE, UberNCE, CoCLR, concat_all_gather  # noqa: F401

# main_nce.py:200 / main_coclr.py:213 construct `optim.Adam` over one param group per tensor:
# resolve it to the single-launch subclass (COCLR_PATCH_ADAM=0 leaves torch.optim untouched).
from coclr_amd import optim as _optim  # noqa: E402
_optim.install()

# main_nce.py:172 / main_coclr.py:184 wrap the model in DistributedDataParallel with torch's defaults:
# for these three classes gradients become views of DDP's buckets unless the caller says otherwise
# (COCLR_PATCH_DDP=0 keeps torch's default) -- see coclr_amd/parallel.py.
from coclr_amd import parallel as _parallel  # noqa: E402
_parallel.install((InfoNCE, UberNCE, CoCLR))
