"""DistributedDataParallel default for this hot path.

The launch scripts wrap the model as `DistributedDataParallel(model, device_ids=[gpu])`
(main_nce.py:172, main_coclr.py:184).  With torch's default `gradient_as_bucket_view=False` DDP copies
every gradient INTO its bucket when it becomes ready and BACK out after the all-reduce: 235 + 235
tiny launches per step on the critical stream, ~1 ms of a 37 ms step on MI355X
(profiles/r02_serial_kernel_stats.csv: `__amd_rocclr_copyBuffer`).  With
`gradient_as_bucket_view=True` the gradients ARE views of the buckets and the copy-back disappears.
The flag only constrains callers that `detach_()` gradients; the launch scripts never touch
`.grad` (they call `zero_grad()`, `backward()`, `step()`), so for modules of THIS package the shim
makes True the default when the caller did not say otherwise.  `COCLR_PATCH_DDP=0` opts out.
"""
import inspect
import os
import sys

import torch

_DDP = torch.nn.parallel.DistributedDataParallel


def _positional_index():
    """Index of gradient_as_bucket_view among the positional arguments after `module` (9 on torch
    >= 2.4, which has init_sync in front of it; 8 before)."""
    names = [n for n in inspect.signature(_DDP.__init__).parameters if n not in ("self", "module")]
    if "gradient_as_bucket_view" not in names:      # somebody wrapped __init__ with (*args, **kwargs)
        return 9
    return names.index("gradient_as_bucket_view")


_POS = _positional_index()
_installed = [False]


def install(module_types):
    """Make `gradient_as_bucket_view=True` the default for DDP wrappers around `module_types`."""
    if os.environ.get("COCLR_PATCH_DDP", "1") == "0" or _installed[0]:
        return False
    orig = _DDP.__init__

    def __init__(self, module, *args, **kwargs):
        if isinstance(module, module_types) and "gradient_as_bucket_view" not in kwargs and \
                len(args) <= _POS:
            kwargs["gradient_as_bucket_view"] = True
        return orig(self, module, *args, **kwargs)

    __init__.__wrapped__ = orig
    __init__.__doc__ = orig.__doc__
    _DDP.__init__ = __init__
    _installed[0] = True
    if os.environ.get("COCLR_QUIET", "0") != "1":
        print("coclr_amd: DistributedDataParallel defaults to gradient_as_bucket_view=True for "
              "InfoNCE/UberNCE/CoCLR modules (COCLR_PATCH_DDP=0 opts out)", file=sys.stderr)
    return True
