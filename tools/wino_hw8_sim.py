"""Lane-level simulation (numpy, CPU) of the index arithmetic of conv_wino_hw8_kernel: packed
operand layout, LDS stage layout, per-lane operand addresses, v_mfma_f32_16x16x4_f32 operand /
result placement (A[i][k]: lane 16k+i, B[k][j]: lane 16k+j, D[i][j]: lane 16*(i//4)+j register i%4)
and the epilogue's row / block mapping, for one box.  It restates the formulas of the kernel by
hand, so it checks the DESIGN of the mapping, not the compiled code."""
import numpy as np
import torch
import torch.nn.functional as F

rng = np.random.default_rng(0)
Cin, Cout, H, W = 8, 64, 16, 16
BM, CC = 64, 8
x = rng.standard_normal((Cin, H, W)).astype(np.float64)
w = (rng.standard_normal((Cout, Cin, 3, 3)) * 0.1).astype(np.float64)
ref = F.conv2d(torch.from_numpy(x)[None], torch.from_numpy(w), padding=1)[0].numpy()

G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
# pack: P[r][c][slot], slot = (((xi>>2) + (c>>2)) & 3)*4 + (xi & 3)
P = np.zeros((Cin, Cout, 16))
for co in range(Cout):
    for ci in range(Cin):
        U = G @ w[co, ci] @ G.T
        for xi in range(16):
            P[ci, co, ((((xi >> 2) + (co >> 2)) & 3) << 2) + (xi & 3)] = U[xi >> 2, xi & 3]
# LDS stage: weights [c][m][16] verbatim, then window [c][planeS]
lTW = lTH = 3
WH = WW = ((1 << lTH) - 1) * 2 + 4
plane1 = WH * WW
planeS = (plane1 + 63) // 64 * 64
W_FLOATS = 16 * CC * BM
stage = np.zeros(W_FLOATS + CC * planeS)
stage[:W_FLOATS] = P.reshape(-1)
for c in range(CC):
    for wh in range(WH):
        for ww in range(WW):
            ih, iw = wh - 1, ww - 1
            if 0 <= ih < H and 0 <= iw < W:
                stage[W_FLOATS + c * planeS + wh * WW + ww] = x[c, ih, iw]

y = np.zeros((Cout, H, W))
for wave in range(8):
    wm, wn = wave >> 2, wave & 3
    acc = np.zeros((2, 16, 64, 4))           # [s][xi][lane][reg]
    for q in range(2):
        for s_ in range(2):
            A = np.zeros((16, 64)); B = np.zeros((16, 64))
            for lane in range(64):
                ln, lk = lane & 15, lane >> 4
                abase = (lk * BM + wm * 32 + ln) * 16
                arot = (ln >> 2) & 3
                for g in range(4):
                    for k in range(4):
                        A[4 * g + k, lane] = stage[abase + s_ * 256 + 4 * q * BM * 16 + ((g + arot) & 3) * 4 + k]
                p = wn * 16 + ln
                ptw, pth = p & 7, (p >> 3) & 7
                lanebase = W_FLOATS + (pth * 2) * WW + ptw * 2 + lk * planeS
                d = np.array([[stage[lanebase + rr * WW + 4 * q * planeS + cc] for cc in range(4)] for rr in range(4)])
                t = np.stack([d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]])
                V = np.stack([t[:, 0] - t[:, 2], t[:, 1] + t[:, 2], t[:, 2] - t[:, 1], t[:, 1] - t[:, 3]], 1)
                B[:, lane] = V.reshape(-1)
            for xi in range(16):
                Am = np.zeros((16, 4)); Bm = np.zeros((4, 16))
                for lane in range(64):
                    Am[lane & 15, lane >> 4] = A[xi, lane]
                    Bm[lane >> 4, lane & 15] = B[xi, lane]
                D = Am @ Bm
                for i in range(16):
                    for j in range(16):
                        acc[s_, xi, 16 * (i // 4) + j, i % 4] += D[i, j]
    for lane in range(64):
        ln, lk = lane & 15, lane >> 4
        p = wn * 16 + ln
        ptw, pth = p & 7, (p >> 3) & 7
        for s_ in range(2):
            for i in range(4):
                row = wm * 32 + s_ * 16 + 4 * lk + i
                M = acc[s_, :, lane, i].reshape(4, 4)
                r0 = M[0] + M[1] + M[2]; r1 = M[1] - M[2] - M[3]
                y[row, 2 * pth, 2 * ptw] = r0[0] + r0[1] + r0[2]
                y[row, 2 * pth, 2 * ptw + 1] = r0[1] - r0[2] - r0[3]
                y[row, 2 * pth + 1, 2 * ptw] = r1[0] + r1[1] + r1[2]
                y[row, 2 * pth + 1, 2 * ptw + 1] = r1[1] - r1[2] - r1[3]
err = np.abs(y - ref).max() / np.abs(ref).max()
print("max rel err of the simulated mapping vs conv2d:", err)
assert err < 1e-12
