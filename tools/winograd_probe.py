"""CPU numerics probe for the 'what comes next' item of DESIGN.md: Winograd F(2,3) along T for
the (3,1,1) temporal convs and F(2x2,3x3) for the (1,3,3) spatial convs, evaluated in fp32 and
compared with a float64 direct convolution -- how much of the 1e-3 parity budget would they use?"""
import torch, torch.nn.functional as F
torch.manual_seed(0)

def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())

# ---- temporal F(2,3): Conv_2c.conv2-like, 192 -> 192, T=16 ------------------------------------
N, C, T, H, W = 2, 192, 16, 8, 8
x = torch.randn(N, C, T, H, W)
w = torch.randn(C, C, 3, 1, 1) * (1.5 / (3 * C) ** 0.5)
ref = F.conv3d(x.double(), w.double(), None, 1, (1, 0, 0))
direct = F.conv3d(x, w, None, 1, (1, 0, 0))
xp = F.pad(x, (0, 0, 0, 0, 1, 1))                       # frames -1 .. T
d0, d1, d2, d3 = (xp[:, :, k:k + T:2] for k in range(4))    # per output pair
g = w[:, :, :, 0, 0]
G = [g[:, :, 0], (g[:, :, 0] + g[:, :, 1] + g[:, :, 2]) * 0.5, (g[:, :, 0] - g[:, :, 1] + g[:, :, 2]) * 0.5, g[:, :, 2]]
D = [d0 - d2, d1 + d2, d2 - d1, d1 - d3]
m = [torch.einsum("oc,ncthw->nothw", Gi, Di) for Gi, Di in zip(G, D)]
y = torch.empty_like(direct)
y[:, :, 0::2] = m[0] + m[1] + m[2]
y[:, :, 1::2] = m[1] - m[2] - m[3]
print("(3,1,1) 192->192: fp32 direct err %.2e, fp32 Winograd F(2,3) err %.2e (vs float64 direct)" % (rel(direct, ref), rel(y, ref)))

# ---- spatial F(2x2,3x3): Conv_2c.conv1-like, 64 -> 192 ----------------------------------------
N, Ci, Co, H, W = 2, 64, 192, 16, 16
x = torch.randn(N, Ci, H, W)
w = torch.randn(Co, Ci, 3, 3) * (1.5 / (9 * Ci) ** 0.5)
ref = F.conv2d(x.double(), w.double(), None, 1, 1)
direct = F.conv2d(x, w, None, 1, 1)
Bt = torch.tensor([[1., 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
Gm = torch.tensor([[1., 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
At = torch.tensor([[1., 1, 1, 0], [0, 1, -1, -1]])
xp = F.pad(x, (1, 1, 1, 1))
tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)              # N, Ci, H/2, W/2, 4, 4
V = torch.einsum("ij,nchwjk,lk->nchwil", Bt, tiles, Bt)
U = torch.einsum("ij,ocjk,lk->ocil", Gm, w, Gm)
M = torch.einsum("ocil,nchwil->nohwil", U, V)
Y = torch.einsum("ij,nohwjk,lk->nohwil", At, M, At)     # N, Co, H/2, W/2, 2, 2
y = Y.permute(0, 1, 2, 4, 3, 5).reshape(N, Co, H, W)
print("(1,3,3) 64->192:  fp32 direct err %.2e, fp32 Winograd F(2x2,3x3) err %.2e" % (rel(direct, ref), rel(y, ref)))
