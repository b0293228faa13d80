"""Probe (CPU only): would Winograd F(4x4, 3x3) in fp32 keep the parity bars the F(2x2, 3x3) kernels are held to?

F(4x4,3x3) needs 36 multiplications per 4x4 output tile and channel pair = 2.25 per output (F(2x2): 4, direct: 9), and its
transformed input V is 2.25x the layer input (F(2x2): 4x) — but its transform matrices are no longer made of 0, +-1, +-1/2: they
amplify rounding errors.  The amplification depends on the interpolation points; two sets are emulated here in torch fp32 on the
CPU, forward / data-gradient dataflow (B^T d B, G g G^T, contraction over channels, A^T m A) and filter-gradient dataflow
(G^T [sum_tiles (A dy A^T) (.) (B^T x B)] G), against fp64 direct convolutions:

  lavin   : 0, +-1, +-2, inf        (Lavin & Gray 2016; what wincnn prints by default)
  half    : 0, +-1, +-1/2, inf      (Barabasz et al. 2018, "Error analysis and improving the accuracy of Winograd convolution")

    python tools/probes/winograd_f4_numerics.py
"""
import sys
from fractions import Fraction as Fr

import torch
import torch.nn.functional as F


def toom_cook(points, m=4, r=3):
    """(A^T [m x a], G [a x r], B^T [a x a]) for F(m, r) on the finite `points` + infinity, as float64 tensors.
    Linear convolution by evaluation/interpolation s = V^-1 [(V_r g) . (V_m d)]; the correlation algorithm is its transpose:
    y = V_m^T [(V_r g) . (V^-T x)].  The Lagrange denominators are moved from B^T into G (B^T then has small integers / halves)."""
    a = m + r - 1
    pts = [Fr(p) for p in points]
    assert len(pts) == a - 1
    V = [[p ** j for j in range(a)] for p in pts] + [[Fr(0)] * (a - 1) + [Fr(1)]]
    # exact inverse by Gauss-Jordan over the rationals
    n = a
    M = [row[:] + [Fr(int(i == j)) for j in range(n)] for i, row in enumerate(V)]
    for c in range(n):
        piv = next(i for i in range(c, n) if M[i][c] != 0)
        M[c], M[piv] = M[piv], M[c]
        pv = M[c][c]
        M[c] = [v / pv for v in M[c]]
        for i in range(n):
            if i != c and M[i][c] != 0:
                f = M[i][c]
                M[i] = [vi - f * vc for vi, vc in zip(M[i], M[c])]
    Vinv = [row[n:] for row in M]
    Nj = []
    for j, p in enumerate(pts):
        d = Fr(1)
        for l, q in enumerate(pts):
            if l != j:
                d *= (p - q)
        Nj.append(d)
    Nj.append(Fr(1))
    # evaluation of a degree-(m-1) / degree-(r-1) polynomial: at infinity that is ITS leading coefficient
    Vm = [V[j][:m] for j in range(a - 1)] + [[Fr(0)] * (m - 1) + [Fr(1)]]
    Vr = [V[j][:r] for j in range(a - 1)] + [[Fr(0)] * (r - 1) + [Fr(1)]]
    AT = [[Vm[j][i] for j in range(a)] for i in range(m)]
    G = [[Vr[j][k] / Nj[j] for k in range(r)] for j in range(a)]
    BT = [[Vinv[i][j] * Nj[j] for i in range(a)] for j in range(a)]       # (V^-T diag(N))[j][i] = Vinv[i][j] * N_j
    f = lambda Mx: torch.tensor([[float(v) for v in row] for row in Mx], dtype=torch.float64)
    return f(AT), f(G), f(BT)


def wino_fwd(x, w, AT, G, BT, m):
    """3x3 stride-1 pad-1 convolution as F(m x m, 3x3) in x.dtype."""
    a = m + 2
    N, C, H, W = x.shape
    K = w.shape[0]
    th, tw = -(-H // m), -(-W // m)
    xp = F.pad(x, (1, m * tw + 1 - W, 1, m * th + 1 - H))
    d = xp.unfold(2, a, m).unfold(3, a, m)                                   # [N, C, th, tw, a, a]
    V = torch.einsum("ij,nctujk,lk->nctuil", BT, d, BT)
    U = torch.einsum("ij,kcjl,ml->kcim", G, w, G)
    Mm = torch.einsum("kcil,nctuil->nktuil", U, V)
    Y = torch.einsum("ij,nktujl,ml->nktuim", AT, Mm, AT)
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, K, m * th, m * tw)[:, :, :H, :W]


def wino_wgrad(x, dy, AT, G, BT, m):
    """dW[k,c,3,3] = G^T [sum_tiles (A dy A^T) (.) (B^T x B)] G."""
    a = m + 2
    N, C, H, W = x.shape
    K = dy.shape[1]
    th, tw = -(-H // m), -(-W // m)
    xp = F.pad(x, (1, m * tw + 1 - W, 1, m * th + 1 - H))
    d = xp.unfold(2, a, m).unfold(3, a, m)
    V = torch.einsum("ij,nctujk,lk->nctuil", BT, d, BT)                      # [N,C,th,tw,a,a]
    dyp = F.pad(dy, (0, m * tw - W, 0, m * th - H)).reshape(N, K, th, m, tw, m).permute(0, 1, 2, 4, 3, 5)
    E = torch.einsum("ji,nktujl,lm->nktuim", AT, dyp, AT)                    # A dy A^T: [N,K,th,tw,a,a]
    S = torch.einsum("nktuil,nctuil->kcil", E, V)
    return torch.einsum("ji,kcjl,lm->kcim", G, S, G)


def rel_max(a, ref):
    return ((a.double() - ref).abs().max() / ref.abs().max()).item()


def main():
    sets = {"F(2x2)": ((0, 1, -1), 2), "F(4x4) lavin": ((0, 1, -1, 2, -2), 4), "F(4x4) half": ((0, 1, -1, Fr(1, 2), Fr(-1, 2)), 4)}
    mats = {k: (toom_cook(p, m), m) for k, (p, m) in sets.items()}
    for k, ((AT, G, BT), m) in mats.items():
        print(k, "max|A^T| %.3g max|G| %.3g max|B^T| %.3g" % (AT.abs().max(), G.abs().max(), BT.abs().max()))
    g = torch.Generator().manual_seed(0)
    layers = (("C256->K256 32x32", 2, 256, 256, 32), ("C512->K512 16x16 (d4 sub-grid)", 2, 512, 512, 16), ("C2048->K512 32x32", 1, 2048, 512, 32),
              ("C1024->K256 33x33", 1, 1024, 256, 33))
    print("forward: max|err| / max|ref| against fp64 direct")
    print("    %-34s %11s " % ("layer", "direct fp32") + " ".join("%13s" % k for k in mats))
    for name, N, C, K, H in layers:
        x = torch.relu(torch.randn(N, C, H, H, generator=g))
        w = torch.randn(K, C, 3, 3, generator=g) * (2.0 / (C * 9)) ** 0.5
        ref = F.conv2d(x.double(), w.double(), None, 1, 1)
        row = [rel_max(F.conv2d(x, w, None, 1, 1), ref)]
        for k, ((AT, G, BT), m) in mats.items():
            row.append(rel_max(wino_fwd(x, w, AT.float(), G.float(), BT.float(), m), ref))
        print("    %-34s %11.3e " % (name, row[0]) + " ".join("%13.3e" % v for v in row[1:]))
    print("filter gradient: max|err| / max|ref| against fp64 direct")
    for name, N, C, K, H in layers:
        x = torch.relu(torch.randn(N, C, H, H, generator=g))
        dy = torch.randn(N, K, H, H, generator=g) * 1e-3
        xd = x.double().requires_grad_(False)
        wd = torch.zeros(K, C, 3, 3, dtype=torch.float64, requires_grad=True)
        (F.conv2d(xd, wd, None, 1, 1) * dy.double()).sum().backward()
        ref = wd.grad
        w32 = torch.zeros(K, C, 3, 3, requires_grad=True)
        (F.conv2d(x, w32, None, 1, 1) * dy).sum().backward()
        row = [rel_max(w32.grad, ref)]
        for k, ((AT, G, BT), m) in mats.items():
            row.append(rel_max(wino_wgrad(x, dy, AT.float(), G.float(), BT.float(), m), ref))
        print("    %-34s %11.3e " % (name, row[0]) + " ".join("%13.3e" % v for v in row[1:]))


if __name__ == "__main__":
    torch.manual_seed(0)
    main()
