"""numpy emulation of one TILEQ elimination (cuda-matrix-inversion_amd/csrc/tileq_impl.hpp): blocks of four FIXED pivot rows, the
pivot COLUMN of each searched along its row over the columns not used yet, W_new = C_masked + A' B' with A' = -(pivot columns) and
I_4 in the pivot rows, B' = D^-1 W[P, :] with D^-1 in the pivot columns; the permutation folded into the output addresses:
inverse[rowaddr[i]][coladdr[j]] = F[i][j], rowaddr[i] = j_i, coladdr[j_i] = i. Checks the algebra against numpy.linalg.inv."""
import numpy as np
def tileq_raw(Win):
    n = Win.shape[0]
    W = Win.copy()
    used = np.zeros(n, bool)
    rowaddr = np.zeros(n, int)
    for kb in range(n // 4):
        P = np.arange(4*kb, 4*kb+4)
        a = W[P, :].copy()
        J = []
        for k in range(4):
            key = np.where(used, 0, np.abs(a[k]))
            p = int(np.argmax(key)); used[p] = True; J.append(p)
            u = a[:, p].copy(); rp = 1.0 / u[k]
            nk = a[k] * rp
            for i in range(4):
                if i != k:
                    a[i] = a[i] - u[i] * nk
                    a[i, p] = -u[i] * rp
            a[k] = nk; a[k, p] = rp
        Bp = a
        C = W[:, J].copy()
        Ap = -C; Ap[P, :] = np.eye(4)
        Wm = W.copy(); Wm[P, :] = 0; Wm[:, J] = 0
        W = Wm + Ap @ Bp
        rowaddr[P] = J
    return W, rowaddr
rng = np.random.default_rng(1)
n = 8
Win = rng.random((n, n))
F, ra = tileq_raw(Win)
Wi = np.linalg.inv(Win)
# find relation: try X[ra[i], :] = F[i, :]
X = np.zeros_like(F); X[ra, :] = F
print(np.abs(X - Wi).max())
X2 = np.zeros_like(F); X2[:, :] = F[ra, :]
print(np.abs(X2 - Wi).max())
# unpivoted check
def nat(Win):
    n = Win.shape[0]; W = Win.copy()
    for k in range(n):
        piv = W[k,k]; col = W[:,k].copy(); row = W[k,:].copy()/piv
        W = W - np.outer(col, row); W[k,:] = row; W[:,k] = -col/piv; W[k,k] = 1/piv
    return W
D = Win + n*np.eye(n)
print(np.abs(nat(D) - np.linalg.inv(D)).max())
F, ra = tileq_raw(D)
print(ra, np.abs(F - np.linalg.inv(D)).max())
for n in (8, 16, 64, 128):
    Win = rng.random((n, n))
    F, ra = tileq_raw(Win)
    ca = np.zeros(n, int); ca[ra] = np.arange(n)
    X = np.zeros_like(F)
    for i in range(n):
        for c in range(n):
            X[ra[i], ca[c]] = F[i, c]
    print(n, np.abs(X - np.linalg.inv(Win)).max())
