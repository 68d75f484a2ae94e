import torch, torch.nn.functional as F
torch.manual_seed(0)
N, Ci, Co, H, W = 2, 3, 4, 6, 8
x = torch.randn(N, Ci, H, W, dtype=torch.float64)
dy = torch.randn(N, Co, H, W, dtype=torch.float64)
w = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
F.conv2d(x, w, padding=1).backward(dy)
ref = w.grad
xp = F.pad(x, (1, 1, 1, 1))
out = torch.zeros(2, Co, Ci, 3, 3, dtype=torch.float64)   # two xi-half partial slices
for xh in (0, 1):
    acc = torch.zeros(2, 4, Co, Ci, dtype=torch.float64)
    for n in range(N):
        for bh in range(H // 2):
            for bw in range(W // 2):
                Y = dy[n, :, 2*bh:2*bh+2, 2*bw:2*bw+2]                  # Co x 2 x 2
                p, q, r, s = Y[:, 0, 0], Y[:, 0, 1], Y[:, 1, 0], Y[:, 1, 1]
                D = xp[n, :, 2*bh + xh:2*bh + xh + 3, 2*bw:2*bw+4]        # Ci x 3 x 4 (patch rows xh..xh+2)
                if xh == 0:
                    ra = [(p, q), (p + r, q + s)]
                    rb = [D[:, 0] - D[:, 2], D[:, 1] + D[:, 2]]
                else:
                    ra = [(p - r, q - s), (r, s)]
                    rb = [D[:, 1] - D[:, 0], D[:, 0] - D[:, 2]]
                for ii in range(2):
                    a0, a1 = ra[ii]
                    A4 = [a0, a0 + a1, a0 - a1, a1]
                    R = rb[ii]
                    B4 = [R[:, 0] - R[:, 2], R[:, 1] + R[:, 2], R[:, 2] - R[:, 1], R[:, 1] - R[:, 3]]
                    for j in range(4):
                        acc[ii, j] += A4[j][:, None] * B4[j][None, :]
    t = torch.zeros(2, 3, Co, Ci, dtype=torch.float64)
    for ii in range(2):
        u0, u1, u2, u3 = acc[ii]
        hs = 0.5 * (u1 + u2)
        t[ii, 0] = u0 + hs; t[ii, 1] = 0.5 * (u1 - u2); t[ii, 2] = hs - u3
    for kw in range(3):
        h1 = 0.5 * (t[1, kw] if xh == 0 else t[0, kw])
        if xh == 0:
            out[xh, :, :, 0, kw] = t[0, kw] + h1; out[xh, :, :, 1, kw] = h1; out[xh, :, :, 2, kw] = h1
        else:
            out[xh, :, :, 0, kw] = h1; out[xh, :, :, 1, kw] = -h1; out[xh, :, :, 2, kw] = h1 - t[1, kw]
got = out.sum(0)
print("max abs err", float((got - ref).abs().max()), "ref max", float(ref.abs().max()))
