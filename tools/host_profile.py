"""Host-side cost of one training step with every kernel call stubbed out (CPU, no GPU):
what the Python engine + ctypes marshalling costs per step.  python tools/host_profile.py [cprofile]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coclr_amd import _lib, ops

real = _lib.load()
calls = [0]
class Stub:
    def __getattr__(self, name):
        fn = getattr(real, name)
        if name in ("coclr_conv3d_ntiles", "coclr_conv3d_wgrad_workspace", "coclr_conv_packed_size",
                    "coclr_gemm_workspace", "coclr_bn_backward_workspace", "coclr_abi_version"):
            return fn
        def stub(*a):
            calls[0] += 1
            # still pay the ctypes marshalling: call a cheap real entry point with the same arity? no --
            return 0
        stub.argtypes = fn.argtypes
        return stub
stub = Stub()
_lib.load = lambda: stub
ops._lib.load = lambda: stub
ops._HANDLE = stub
ops._p = lambda t, dtype=torch.float32: None if t is None else t.data_ptr()
ops._stream = lambda: 0

from model.pretrain import InfoNCE
import torch.nn.functional as F
torch.manual_seed(0)
B = 32
model = InfoNCE('s3d', 128, 2048, 0.999, 0.07).train()
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3)
block = torch.empty(B, 2, 3, 32, 128, 128)     # never read: kernels are stubs

def step():
    logits, labels = model(block)
    loss = logits.sum() * 0
    opt.zero_grad(set_to_none=True)
    loss.backward()

for _ in range(2):
    step()
n = 5
calls[0] = 0
t0 = time.perf_counter()
if len(sys.argv) > 1:
    pr = cProfile.Profile(); pr.enable()
for _ in range(n):
    step()
if len(sys.argv) > 1:
    pr.disable()
dt = (time.perf_counter() - t0) / n
print("host time per step (fwd q + fwd k + bwd, kernels stubbed): %.2f ms, %d library calls/step -> %.1f us/call"
      % (dt * 1e3, calls[0] // n, dt * 1e6 / (calls[0] / n)))
if len(sys.argv) > 1:
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
