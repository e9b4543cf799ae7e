"""numpy restatement of the correspondence analysis KPopTwist runs in R (src/KPopTwist:93-116, library `ca`).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: R is not available here, so this restates the published
algorithm of ca::ca / ca::cacoord (Nenadic & Greenacre) as the wrapper uses it:
    stuff   <- per-column normalised counts               (:93-94)
    ca(stuff): P = N/sum(N); r, c masses; S = D_r^-1/2 (P - r c') D_c^-1/2; S = U diag(sv) V'
    twisted <- cacoord(cols=TRUE)  = principal column coordinates  D_c^-1/2 V diag(sv)     (:98-100)
    inertia <- sv^2 / sum(sv^2)                                                       (:105)
    twister <- t(cacoord(rows=TRUE) / sv) = standard row coordinates' = (D_r^-1/2 U)'   (:110-116)
with nd = min(I, J) - 1 dimensions.  The sign of every dimension is arbitrary (LAPACK).
"""
import numpy as np


def ca(counts, normalize=True):
    """counts: I x J (k-mers x spectra).  -> twisted (J x nd), inertia (nd), twister (nd x I)."""
    N = np.asarray(counts, dtype=np.float64)
    I, J = N.shape
    if normalize:
        N = N / N.sum(axis=0, keepdims=True)
    P = N / N.sum()
    r = P.sum(axis=1)
    c = P.sum(axis=0)
    with np.errstate(divide="ignore", invalid="ignore"):
        S = (P - np.outer(r, c)) / np.sqrt(np.outer(r, c))
    S[~np.isfinite(S)] = 0.0  # k-mers seen in no spectrum carry no mass
    U, sv, Vt = np.linalg.svd(S, full_matrices=False)
    nd = min(I, J) - 1
    sv, U, V = sv[:nd], U[:, :nd], Vt.T[:, :nd]
    with np.errstate(divide="ignore", invalid="ignore"):
        rowstd = U / np.sqrt(r)[:, None]
    rowstd[~np.isfinite(rowstd)] = 0.0
    twisted = V / np.sqrt(c)[:, None] * sv
    inertia = sv ** 2 / np.sum(sv ** 2)
    return twisted, inertia, rowstd.T.copy()


def align_signs(a, ref, axis):
    """Flip the sign of each dimension of `a` (dimension index along `axis`) to match `ref`."""
    a = np.array(a, dtype=np.float64, copy=True)
    dot = np.sum(a * ref, axis=1 - axis)
    s = np.where(dot < 0, -1.0, 1.0)
    return a * (s[None, :] if axis == 1 else s[:, None])
