"""Host-side check of the arithmetic conv3x3_wino4_kernel implements (F(4x4,3x3), optional bilinear x2 folded into the input
transform of the second source): the exact per-thread formulas of the kernel, in numpy, against torch conv2d on the
concatenated / up-sampled input.  Development aid; tests/test_host_cpu.py runs it as a known-answer test of the transform tables."""
import numpy as np
import torch
import torch.nn.functional as F

BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)


def col_pass(t):
    """v[nu] = sum_c BT[nu][c] t[c] with the kernel's shared sub-expressions (14 operations)."""
    a = t[4] - 4 * t[2]
    b = t[3] - 4 * t[1]
    c = t[4] - t[2]
    d = 2 * (t[3] - t[1])
    return [4 * t[0] - 5 * t[2] + t[4], a + b, a - b, c + d, c - d, 4 * t[1] - 5 * t[3] + t[5]]


def out_pass(z):
    """y[b] = sum_nu AT[b][nu] z[nu] (also the row pass of the output transform)."""
    s12, d12, s34, d34 = z[1] + z[2], z[1] - z[2], z[3] + z[4], z[3] - z[4]
    return [z[0] + s12 + s34, d12 + 2 * d34, s12 + 4 * s34, d12 + 8 * d34 + z[5]]


def lowres_rows(xi, y_tile0, H):
    """Folded coefficients C[xi][m] on the 4 low-res rows m (rows y_tile0/2 - 1 .. + 2, clamped by the loader) of the six up-sampled
    rows y_tile0 - 1 .. y_tile0 + 4; up-sampled rows outside [0, H) are the conv's zero padding."""
    U = np.zeros((6, 4))
    for r in range(6):
        Y = y_tile0 - 1 + r
        if Y < 0 or Y >= H:
            continue
        m0 = (r + 1) // 2 - (1 if (r % 2 == 1) else 0)  # r = 0 -> rows (0, 1), 1 -> (0, 1), 2 -> (1, 2), 3 -> (1, 2), 4 -> (2, 3), 5 -> (2, 3)
        m0 = r // 2
        w0, w1 = (0.75, 0.25) if r % 2 == 0 else (0.25, 0.75)  # r even <-> Y odd (Y = y_tile0 - 1 + r, y_tile0 even)
        U[r, m0] += w0
        U[r, m0 + 1] += w1
    return BT[xi] @ U


def conv_wino4(x0, x1_lr, w, bias, relu=True):
    """x0 (C0, H, W) full resolution, x1_lr (C1, H/2, W/2) or None, w (Cout, C0 + C1, 3, 3)."""
    C0, H, W = x0.shape
    assert H % 4 == 0 and W % 4 == 0
    Cout = w.shape[0]
    U = np.einsum("ik,ockl,jl->ocij", G, w.astype(np.float64), G)  # (Cout, Cin, 6, 6)
    xp = np.pad(x0.astype(np.float64), ((0, 0), (1, 1), (1, 1)))
    out = np.zeros((Cout, H, W))
    Hl, Wl = H // 2, W // 2
    for ty in range(H // 4):
        for tx in range(W // 4):
            M = np.zeros((Cout, 6, 6))
            # source 0: plain patch
            d = xp[:, 4 * ty : 4 * ty + 6, 4 * tx : 4 * tx + 6]
            V = np.zeros((d.shape[0], 6, 6))
            for xi in range(6):
                t = [sum(BT[xi][r] * d[:, r, c] for r in range(6) if BT[xi][r] != 0) for c in range(6)]
                v = col_pass(t)
                for nu in range(6):
                    V[:, xi, nu] = v[nu]
            M += np.einsum("ocij,cij->oij", U[:, :C0], V)
            if x1_lr is not None:
                C1 = x1_lr.shape[0]
                rows = np.clip(np.arange(2 * ty - 1, 2 * ty + 3), 0, Hl - 1)
                cols = np.clip(np.arange(2 * tx - 1, 2 * tx + 3), 0, Wl - 1)
                l = x1_lr.astype(np.float64)[:, rows][:, :, cols]  # (C1, 4, 4), index-clamped like the loader
                V1 = np.zeros((C1, 6, 6))
                zx0 = 0.0 if 4 * tx - 1 < 0 else 1.0
                zx5 = 0.0 if 4 * tx + 4 >= W else 1.0
                for xi in range(6):
                    C = lowres_rows(xi, 4 * ty, H)
                    tl = [sum(C[m] * l[:, m, c] for m in range(4)) for c in range(4)]  # row pass on the 4 low-res columns
                    th = [zx0 * (0.75 * tl[0] + 0.25 * tl[1]), 0.25 * tl[0] + 0.75 * tl[1], 0.75 * tl[1] + 0.25 * tl[2], 0.25 * tl[1] + 0.75 * tl[2],
                          0.75 * tl[2] + 0.25 * tl[3], zx5 * (0.25 * tl[2] + 0.75 * tl[3])]
                    v = col_pass(th)
                    for nu in range(6):
                        V1[:, xi, nu] = v[nu]
                M += np.einsum("ocij,cij->oij", U[:, C0:], V1)
            Z = [out_pass([M[:, xi, nu] for xi in range(6)]) for nu in range(6)]  # Z[nu][a]
            for a in range(4):
                y = out_pass([Z[nu][a] for nu in range(6)])
                for b in range(4):
                    out[:, 4 * ty + a, 4 * tx + b] = y[b] + bias
    return np.maximum(out, 0) if relu else out


def check(seed=0):
    rng = np.random.RandomState(seed)
    C0, C1, Cout, H, W = 5, 7, 6, 8, 12
    x0 = rng.randn(C0, H, W).astype(np.float32)
    x1 = rng.randn(C1, H // 2, W // 2).astype(np.float32)
    w = (rng.randn(Cout, C0 + C1, 3, 3) * 0.2).astype(np.float32)
    b = rng.randn(Cout).astype(np.float32)
    up = F.interpolate(torch.from_numpy(x1)[None], scale_factor=2, mode="bilinear", align_corners=False)[0]
    ref = F.relu(F.conv2d(torch.cat([torch.from_numpy(x0), up])[None].double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=1))[0].numpy()
    got = conv_wino4(x0, x1, w, b)
    err = np.abs(got - ref).max()
    ref0 = F.relu(F.conv2d(torch.from_numpy(x0)[None].double(), torch.from_numpy(w[:, :C0]).double(), torch.from_numpy(b).double(), padding=1))[0].numpy()
    err0 = np.abs(conv_wino4(x0, None, w[:, :C0], b) - ref0).max()
    return err, err0


if __name__ == "__main__":
    print(check())
