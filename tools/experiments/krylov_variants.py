"""CPU experiment (round 3): preconditioned operator applications needed by BiCGStab, BiCGStab(2), IDR(s) and GMRES on the
P1 weak-Dirichlet sphere system with the shipped preconditioner (K_box^-1 on u, Jacobi on p; right preconditioning),
rtol 1e-8.  One "application" = one A M^-1 product = what costs 200 us on the GPU.  usage: krylov_variants.py [n ...]"""
import os, sys, warnings
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cpu_backend import assemble_local
from precond_variants import lattice


def bicgstab(op, b, rtol, maxit=4000):
    x = np.zeros_like(b); r = b.copy(); rh = r.copy(); rho = alpha = om = 1.0
    v = np.zeros_like(b); p = np.zeros_like(b); bn = np.linalg.norm(b); napp = 0
    for it in range(1, maxit + 1):
        rho1 = rh @ r; beta = (rho1 / rho) * (alpha / om); rho = rho1
        p = r + beta * (p - om * v); v = op(p); napp += 1
        alpha = rho / (rh @ v); s = r - alpha * v
        t = op(s); napp += 1
        om = (t @ s) / (t @ t); x += alpha * p + om * s; r = s - om * t
        if np.linalg.norm(r) <= rtol * bn: break
    return x, napp


def idrs(op, b, s, rtol, maxit=8000, seed=1):
    """IDR(s) with bi-orthogonalisation (van Gijzen & Sonneveld 2011, Algorithm 2), unpreconditioned form on `op`."""
    n = b.size; rng = np.random.default_rng(seed)
    P = np.linalg.qr(rng.standard_normal((n, s)))[0].T
    x = np.zeros(n); r = b.copy(); bn = np.linalg.norm(b); napp = 0
    G = np.zeros((n, s)); U = np.zeros((n, s)); M = np.eye(s); om = 1.0
    while np.linalg.norm(r) > rtol * bn and napp < maxit:
        f = P @ r
        for k in range(s):
            c = np.linalg.solve(M[k:, k:], f[k:])
            v = r - G[:, k:] @ c
            U[:, k] = U[:, k:] @ c + om * v
            G[:, k] = op(U[:, k]); napp += 1
            for i in range(k):
                al = (P[i] @ G[:, k]) / M[i, i]
                G[:, k] -= al * G[:, i]; U[:, k] -= al * U[:, i]
            M[k:, k] = P[k:] @ G[:, k]
            beta = f[k] / M[k, k]
            r = r - beta * G[:, k]; x = x + beta * U[:, k]
            if np.linalg.norm(r) <= rtol * bn: return x, napp
            if k + 1 < s: f[k + 1:] -= beta * M[k + 1:, k]
        t = op(r); napp += 1
        # "maintaining the convergence" choice of omega
        tt = t @ t; tr = t @ r; rho = abs(tr) / (np.sqrt(tt) * np.linalg.norm(r)); om = tr / tt
        if rho < 0.7: om *= 0.7 / rho
        x = x + om * r; r = r - om * t
    return x, napp


def bicgstab2(op, b, rtol, maxit=4000):
    """BiCGStab(2) (Sleijpen & Fokkema)."""
    x = np.zeros_like(b); r0 = b.copy(); rt = r0.copy(); bn = np.linalg.norm(b)
    rho0, alpha, omega = 1.0, 0.0, 1.0; u = np.zeros_like(b); napp = 0
    while np.linalg.norm(r0) > rtol * bn and napp < maxit:
        rho0 = -omega * rho0
        # even BiCG step
        rho1 = rt @ r0; beta = alpha * rho1 / rho0; rho0 = rho1
        u = r0 - beta * u; v = op(u); napp += 1
        gamma = v @ rt; alpha = rho0 / gamma
        r0 = r0 - alpha * v; s = op(r0); napp += 1
        x = x + alpha * u
        # odd BiCG step
        rho1 = rt @ s; beta = alpha * rho1 / rho0; rho0 = rho1
        v = s - beta * v; w = op(v); napp += 1
        gamma = w @ rt; alpha = rho0 / gamma
        u = r0 - beta * u; r0 = r0 - alpha * v; s = s - alpha * w
        t = op(s); napp += 1
        # GCR(2) part
        om1 = r0 @ s; mu = s @ s; nu = s @ t; tau = t @ t; om2 = r0 @ t
        tau = tau - nu * nu / mu; om2 = (om2 - nu * om1 / mu) / tau; om1 = (om1 - nu * om2) / mu
        x = x + om1 * r0 + om2 * s + alpha * u
        r0 = r0 - om1 * s - om2 * t
        u = u - om1 * v - om2 * w; omega = om2
    return x, napp


def main(n, rtol=1e-8):
    x, topo, cv, A, b, act = assemble_local(n, 1, 0, n, sphere=True)
    nv = topo.nv; idx = np.flatnonzero(act)
    Aa = A[idx][:, idx].tocsr(); ba = b[idx]; nu = int((idx < nv).sum())
    K, pos = lattice(n, x, idx[:nu]); Klu = spla.splu(K); dpp = Aa.diagonal()[nu:]
    def Minv(r):
        g = np.zeros(K.shape[0]); g[pos] = r[:nu]
        return np.concatenate([Klu.solve(g)[pos], r[nu:] / dpp])
    op = lambda v: Aa @ Minv(v)
    res = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, fn in [("BiCGStab", lambda: bicgstab(op, ba, rtol)), ("BiCGStab(2)", lambda: bicgstab2(op, ba, rtol)),
                         ("IDR(1)", lambda: idrs(op, ba, 1, rtol)), ("IDR(2)", lambda: idrs(op, ba, 2, rtol)),
                         ("IDR(4)", lambda: idrs(op, ba, 4, rtol)), ("IDR(8)", lambda: idrs(op, ba, 8, rtol))]:
            y, napp = fn()
            xx = Minv(y)
            res[name] = (napp, np.linalg.norm(Aa @ xx - ba) / np.linalg.norm(ba))
        cnt = [0]
        def cb(_): cnt[0] += 1
        lin = spla.LinearOperator(Aa.shape, matvec=op)
        yg, info = spla.gmres(lin, ba, rtol=rtol, restart=400, maxiter=1, callback=cb, callback_type="pr_norm")
        res["GMRES(full)"] = (cnt[0], np.linalg.norm(Aa @ Minv(yg) - ba) / np.linalg.norm(ba))
    print(f"n={n}: {idx.size} DoFs; applications of A M^-1 to rtol {rtol:g}: " + "; ".join(f"{k} {v[0]} ({v[1]:.0e})" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    for n in [int(a) for a in sys.argv[1:]] or [24, 32, 48]:
        main(n)
