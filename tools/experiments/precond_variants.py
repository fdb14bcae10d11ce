"""CPU experiment (numpy/scipy oracle, small n): iteration counts of right-preconditioned BiCGStab on the weak-Dirichlet
phi-FEM system for variants of the block preconditioner.  Decides whether a better p-block / coupled preconditioner is
worth a GPU implementation.  Not part of the product."""
import sys, os, warnings
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from cpu_backend import assemble_local


def bicgstab(A, b, M, rtol=1e-8, maxit=2000):
    x = np.zeros_like(b); r = b.copy(); rh = r.copy(); rho = alpha = om = 1.0
    v = np.zeros_like(b); p = np.zeros_like(b); bn = np.linalg.norm(b)
    for it in range(1, maxit + 1):
        rho1 = rh @ r
        beta = (rho1 / rho) * (alpha / om); rho = rho1
        p = r + beta * (p - om * v)
        ph = M(p); v = A @ ph
        alpha = rho / (rh @ v)
        s = r - alpha * v
        sh = M(s); t = A @ sh
        om = (t @ s) / (t @ t)
        x += alpha * ph + om * sh
        r = s - om * t
        if np.linalg.norm(r) <= rtol * bn:
            return x, it
    return x, maxit


def lattice(n, x, uidx, margin=4):
    nv1 = n + 1
    ijk = np.stack([uidx % nv1, (uidx // nv1) % nv1, uidx // (nv1 * nv1)], 1)
    lo = ijk.min(0) - margin; hi = ijk.max(0) + margin
    m = hi - lo + 1
    h = 3.0 / n
    def T(k): return sp.diags([-np.ones(k - 1), 2 * np.ones(k), -np.ones(k - 1)], [-1, 0, 1])
    I = [sp.identity(k) for k in m]
    K = h * (sp.kron(I[2], sp.kron(I[1], T(m[0]))) + sp.kron(I[2], sp.kron(T(m[1]), I[0])) + sp.kron(T(m[2]), sp.kron(I[1], I[0])))
    pos = (ijk[:, 0] - lo[0]) + m[0] * ((ijk[:, 1] - lo[1]) + m[1] * (ijk[:, 2] - lo[2]))
    return K.tocsc(), pos


def main(n):
    x, topo, cv, A, b, act = assemble_local(n, 1, 0, n, sphere=True)
    nv = topo.nv
    idx = np.flatnonzero(act)
    Aa = A[idx][:, idx].tocsr(); ba = b[idx]
    isu = idx < nv
    nu = int(isu.sum()); npp = idx.size - nu
    print(f"n={n} nu={nu} np={npp}")
    Auu = Aa[:nu][:, :nu].tocsc(); Aup = Aa[:nu][:, nu:].tocsc(); Apu = Aa[nu:][:, :nu].tocsc(); App = Aa[nu:][:, nu:].tocsc()
    K, pos = lattice(n, x, idx[:nu])
    Klu = spla.splu(K); Auulu = spla.splu(Auu); Applu = spla.splu(App)
    dpp = App.diagonal(); d = Aa.diagonal()
    def Ku(r):
        g = np.zeros(K.shape[0]); g[pos] = r
        return Klu.solve(g)[pos]
    S_exact = None
    variants = {
        "jacobi": lambda r: r / d,
        "Kbox | diag(App)": lambda r: np.concatenate([Ku(r[:nu]), r[nu:] / dpp]),
        "Kbox | App^-1": lambda r: np.concatenate([Ku(r[:nu]), Applu.solve(r[nu:])]),
        "Auu^-1 | diag(App)": lambda r: np.concatenate([Auulu.solve(r[:nu]), r[nu:] / dpp]),
        "Auu^-1 | App^-1": lambda r: np.concatenate([Auulu.solve(r[:nu]), Applu.solve(r[nu:])]),
    }
    def tri_lower(usolve, psolve):
        def M(r):
            zu = usolve(r[:nu]); zp = psolve(r[nu:] - Apu @ zu)
            return np.concatenate([zu, zp])
        return M
    def tri_upper(usolve, psolve):
        def M(r):
            zp = psolve(r[nu:]); zu = usolve(r[:nu] - Aup @ zp)
            return np.concatenate([zu, zp])
        return M
    variants["lower-tri Kbox, diag"] = tri_lower(Ku, lambda r: r / dpp)
    variants["lower-tri Kbox, App^-1"] = tri_lower(Ku, Applu.solve)
    variants["upper-tri Kbox, diag"] = tri_upper(Ku, lambda r: r / dpp)
    variants["upper-tri Kbox, App^-1"] = tri_upper(Ku, Applu.solve)
    # Schur on u: S_u = Auu - Aup App^-1 Apu (exact, small n) -- floor for "eliminate p" strategies
    Su = (Auu - Aup @ sp.csc_matrix(Applu.solve(Apu.toarray()))).tocsc() if nu < 40000 else None
    if Su is not None:
        Sulu = spla.splu(Su)
        variants["upper-tri Su^-1, App^-1 (exact Schur)"] = tri_upper(Sulu.solve, Applu.solve)
        # symmetric part / how close is Su to K?
    for name, M in variants.items():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            xs, it = bicgstab(Aa, ba, M)
        print(f"  {name:45s} it={it:4d}  res={np.linalg.norm(Aa @ xs - ba) / np.linalg.norm(ba):.1e}")


def band_variants(n):
    x, topo, cv, A, b, act = assemble_local(n, 1, 0, n, sphere=True)
    nv = topo.nv
    idx = np.flatnonzero(act)
    Aa = A[idx][:, idx].tocsr(); ba = b[idx]
    nu = int((idx < nv).sum())
    Auu = Aa[:nu][:, :nu].tocsr()
    K, pos = lattice(n, x, idx[:nu])
    Klu = spla.splu(K)
    Kaa = K.tocsr()[pos][:, pos]
    diff = abs(Auu - Kaa)
    rowdiff = np.asarray(diff.max(axis=1).todense()).ravel()
    band = np.flatnonzero(rowdiff > 1e-9 * abs(Auu).max())
    print(f"n={n} nu={nu} band rows={band.size} ({100.0 * band.size / nu:.1f} %)")
    d = Aa.diagonal(); dpp = d[nu:]
    Abb = Auu[band][:, band].tocsc(); Abblu = spla.splu(Abb)
    Aband = Auu[band]           # band rows, all u columns
    def Ku(r):
        g = np.zeros(K.shape[0]); g[pos] = r
        return Klu.solve(g)[pos]
    def mk(post, pre=None):
        def M(r):
            ru = r[:nu].copy(); z = np.zeros(nu)
            if pre is not None:
                z[band] = pre(ru[band]); ru = r[:nu] - Auu @ z
            z = z + Ku(ru)
            if post is not None:
                rb = r[:nu][band] - Aband @ z
                z[band] += post(rb)
            return np.concatenate([z, r[nu:] / dpp])
        return M
    db = Auu.diagonal()[band]
    def gs(k):
        L = sp.tril(Abb).tocsr()
        def f(rb):
            zb = np.zeros_like(rb)
            for _ in range(k):
                zb = zb + spla.spsolve_triangular(L, rb - Abb @ zb, lower=True)
            return zb
        return f
    def jac(k, om):
        def f(rb):
            zb = np.zeros_like(rb)
            for _ in range(k):
                zb = zb + om * (rb - Abb @ zb) / db
            return zb
        return f
    variants = {
        "Kbox": mk(None),
        "Kbox, post Jacobi(1, 1.0)": mk(jac(1, 1.0)),
        "Kbox, post Jacobi(1, 0.7)": mk(jac(1, 0.7)),
        "Kbox, post Jacobi(3, 0.7)": mk(jac(3, 0.7)),
        "Kbox, post GS(1)": mk(gs(1)),
        "Kbox, post GS(3)": mk(gs(3)),
        "Kbox, post Abb^-1": mk(Abblu.solve),
        "pre Abb^-1, Kbox": mk(None, Abblu.solve),
        "pre Abb^-1, Kbox, post Abb^-1": mk(Abblu.solve, Abblu.solve),
        "pre Jacobi(1,.7), Kbox, post Jacobi(1,.7)": mk(jac(1, 0.7), jac(1, 0.7)),
    }
    for name, M in variants.items():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            xs, it = bicgstab(Aa, ba, M)
        print(f"  {name:45s} it={it:4d}  res={np.linalg.norm(Aa @ xs - ba) / np.linalg.norm(ba):.1e}")


def patch_variants(n):
    x, topo, cv, A, b, act = assemble_local(n, 1, 0, n, sphere=True)
    nv = topo.nv
    idx = np.flatnonzero(act)
    Aa = A[idx][:, idx].tocsr(); ba = b[idx]
    nu = int((idx < nv).sum())
    Auu = Aa[:nu][:, :nu].tocsr()
    K, pos = lattice(n, x, idx[:nu])
    Klu = spla.splu(K)
    Kaa = K.tocsr()[pos][:, pos]
    diff = abs(Auu - Kaa)
    rowdiff = np.asarray(diff.max(axis=1).todense()).ravel()
    band = np.flatnonzero(rowdiff > 1e-9 * abs(Auu).max())
    print(f"n={n} nu={nu} band rows={band.size} ({100.0 * band.size / nu:.1f} %)")
    d = Aa.diagonal(); dpp = d[nu:]
    Aband = Auu[band]
    Abb = Auu[band][:, band].tocsr()
    nv1 = n + 1
    uid = idx[:nu][band]
    ijk = np.stack([uid % nv1, (uid // nv1) % nv1, uid // (nv1 * nv1)], 1)
    def Ku(r):
        g = np.zeros(K.shape[0]); g[pos] = r
        return Klu.solve(g)[pos]
    def patches(c, shift=0):
        key = ((ijk + shift) // c)
        kk = key[:, 0] + 1000 * (key[:, 1] + 1000 * key[:, 2])
        order = np.argsort(kk, kind="stable"); ks = kk[order]
        cuts = np.flatnonzero(np.diff(ks)) + 1
        return np.split(order, cuts)
    def block_jacobi(c, shift=0):
        P = patches(c, shift)
        Ad = Abb.toarray() if Abb.shape[0] < 12000 else None
        blocks = [np.linalg.inv(Ad[np.ix_(p, p)] if Ad is not None else Abb[p][:, p].toarray()) for p in P]
        perm = np.concatenate(P)
        Binv = sp.block_diag(blocks, format="csr")
        sizes = [p.size for p in P]
        def f(rb):
            zb = np.empty_like(rb)
            zb[perm] = Binv @ rb[perm]
            return zb
        return f, (len(P), max(sizes), sum(s * s for s in sizes))
    def mk(post, sweeps=1):
        def M(r):
            z = Ku(r[:nu])
            for _ in range(sweeps):
                rb = r[:nu][band] - Aband @ z
                z[band] += post(rb)
            return np.concatenate([z, r[nu:] / dpp])
        return M
    def two(f1, f2):
        def M(r):
            z = Ku(r[:nu])
            for f in (f1, f2):
                rb = r[:nu][band] - Aband @ z
                z[band] += f(rb)
            return np.concatenate([z, r[nu:] / dpp])
        return M
    for c in (2, 3, 4, 6, 8):
        f, st = block_jacobi(c)
        for sw in (1, 2):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                xs, it = bicgstab(Aa, ba, mk(f, sw))
            print(f"  Kbox, post block-Jacobi c={c} sweeps={sw} (patches {st[0]}, max {st[1]}, inv entries {st[2]:.2e}): it={it}", flush=True)
        f2, _ = block_jacobi(c, c // 2)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            xs, it = bicgstab(Aa, ba, two(f, f2))
        print(f"  Kbox, post block-Jacobi c={c} then shifted c/2: it={it}")


def x0_variants(n):
    """does x0 = M^-1 b (one extra application + SpMV) save iterations?"""
    x, topo, cv, A, b, act = assemble_local(n, 1, 0, n, sphere=True)
    nv = topo.nv
    idx = np.flatnonzero(act)
    Aa = A[idx][:, idx].tocsr(); ba = b[idx]
    nu = int((idx < nv).sum())
    K, pos = lattice(n, x, idx[:nu])
    Klu = spla.splu(K)
    dpp = Aa.diagonal()[nu:]
    def M(r):
        g = np.zeros(K.shape[0]); g[pos] = r[:nu]
        return np.concatenate([Klu.solve(g)[pos], r[nu:] / dpp])
    xs, it0 = bicgstab(Aa, ba, M)
    x0 = M(ba)
    r0 = ba - Aa @ x0
    # same absolute target: rtol relative to |b|
    xs1, it1 = bicgstab(Aa, r0, M, rtol=1e-8 * np.linalg.norm(ba) / np.linalg.norm(r0))
    print(f"n={n}: x0 = 0: {it0} iterations; x0 = M^-1 b: |r0|/|b| = {np.linalg.norm(r0) / np.linalg.norm(ba):.3f}, {it1} iterations (+ ~0.4 for the start)")


def fgmres(A, b, M, rtol=1e-8, maxit=300):
    """flexible GMRES without restart (experiment sizes): returns x, iterations"""
    n = b.size
    bn = np.linalg.norm(b)
    V = [b / bn]; Z = []; H = np.zeros((maxit + 1, maxit))
    g = np.zeros(maxit + 1); g[0] = bn
    for k in range(maxit):
        z = M(V[k]); Z.append(z)
        w = A @ z
        for i in range(k + 1):
            H[i, k] = V[i] @ w; w = w - H[i, k] * V[i]
        H[k + 1, k] = np.linalg.norm(w); V.append(w / H[k + 1, k])
        y, res, _, _ = np.linalg.lstsq(H[:k + 2, :k + 1], g[:k + 2], rcond=None)
        r = np.linalg.norm(H[:k + 2, :k + 1] @ y - g[:k + 2])
        if r <= rtol * bn:
            x = sum(yi * zi for yi, zi in zip(y, Z))
            return x, k + 1
    x = sum(yi * zi for yi, zi in zip(y, Z))
    return x, maxit


def inner_variants(n):
    """Kbox followed by a few inner Krylov steps on the band block (flexible outer GMRES)."""
    x, topo, cv, A, b, act = assemble_local(n, 1, 0, n, sphere=True)
    nv = topo.nv
    idx = np.flatnonzero(act)
    Aa = A[idx][:, idx].tocsr(); ba = b[idx]
    nu = int((idx < nv).sum())
    Auu = Aa[:nu][:, :nu].tocsr()
    K, pos = lattice(n, x, idx[:nu])
    Klu = spla.splu(K)
    Kaa = K.tocsr()[pos][:, pos]
    rowdiff = np.asarray(abs(Auu - Kaa).max(axis=1).todense()).ravel()
    band = np.flatnonzero(rowdiff > 1e-9 * abs(Auu).max())
    Abb = Auu[band][:, band].tocsr(); Aband = Auu[band]
    db = Abb.diagonal()
    dpp = Aa.diagonal()[nu:]
    Abblu = spla.splu(Abb.tocsc())
    print(f"n={n} nu={nu} band={band.size}")
    def Ku(r):
        g = np.zeros(K.shape[0]); g[pos] = r
        return Klu.solve(g)[pos]
    def mk(post):
        def M(r):
            z = Ku(r[:nu])
            if post is not None:
                rb = r[:nu][band] - Aband @ z
                z[band] += post(rb)
            return np.concatenate([z, r[nu:] / dpp])
        return M
    def inner(k):
        def f(rb):
            zb, info = spla.gmres(Abb, rb, M=sp.diags(1.0 / db), restart=k, maxiter=1, rtol=1e-14)
            return zb
        return f
    for name, M in [("Kbox", mk(None)), ("Kbox + exact band", mk(Abblu.solve))] + [(f"Kbox + {k} inner GMRES(Jacobi) steps on the band", mk(inner(k))) for k in (3, 5, 10, 20)]:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            xs, it = fgmres(Aa, ba, M)
        print(f"  FGMRES, {name:50s} it={it:4d} res={np.linalg.norm(Aa @ xs - ba) / np.linalg.norm(ba):.1e}", flush=True)


def poly_variants(n):
    """Kbox followed by a FIXED polynomial in the Jacobi-scaled band block (the GMRES polynomial of one start vector):
    a linear preconditioner, so the outer BiCGStab stays valid."""
    x, topo, cv, A, b, act = assemble_local(n, 1, 0, n, sphere=True)
    nv = topo.nv
    idx = np.flatnonzero(act)
    Aa = A[idx][:, idx].tocsr(); ba = b[idx]
    nu = int((idx < nv).sum())
    Auu = Aa[:nu][:, :nu].tocsr()
    K, pos = lattice(n, x, idx[:nu])
    Klu = spla.splu(K)
    Kaa = K.tocsr()[pos][:, pos]
    rowdiff = np.asarray(abs(Auu - Kaa).max(axis=1).todense()).ravel()
    band = np.flatnonzero(rowdiff > 1e-9 * abs(Auu).max())
    Abb = Auu[band][:, band].tocsr(); Aband = Auu[band]
    db = Abb.diagonal()
    B = Abb @ sp.diags(1.0 / db)          # right-scaled band block: z_b = D^-1 q(B) r_b
    dpp = Aa.diagonal()[nu:]
    print(f"n={n} nu={nu} band={band.size}")
    def Ku(r):
        g = np.zeros(K.shape[0]); g[pos] = r
        return Klu.solve(g)[pos]
    rng = np.random.default_rng(5)
    def gmres_poly(k, v0):
        # power basis with per-step normalisation; c minimises |v0 - B Kc|
        cols = [v0 / np.linalg.norm(v0)]; scal = [1.0 / np.linalg.norm(v0)]
        for j in range(1, k):
            w = B @ cols[-1]; s = 1.0 / np.linalg.norm(w)
            cols.append(w * s); scal.append(scal[-1] * s)
        Km = np.stack(cols, 1)
        c, *_ = np.linalg.lstsq(B @ Km, v0, rcond=None)
        coef = c * np.array(scal)          # q(B) v = sum_j coef_j B^j v  (for ANY v)
        return coef
    def apply_poly(coef, r):
        # Horner: q(B) r = coef_0 r + B (coef_1 r + B (...))
        acc = coef[-1] * r
        for cj in coef[-2::-1]:
            acc = cj * r + B @ acc
        return acc / db
    def mk(post):
        def M(r):
            z = Ku(r[:nu])
            if post is not None:
                rb = r[:nu][band] - Aband @ z
                z[band] += post(rb)
            return np.concatenate([z, r[nu:] / dpp])
        return M
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        xs, it = bicgstab(Aa, ba, mk(None))
    print(f"  BiCGStab, Kbox: {it}")
    rb0 = ba[:nu][band] - Aband @ Ku(ba[:nu])
    for k in (3, 5, 8, 12):
        for label, v0 in (("random start", rng.standard_normal(band.size)), ("first band residual", rb0)):
            coef = gmres_poly(k, v0)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                xs, it = bicgstab(Aa, ba, mk(lambda rb: apply_poly(coef, rb)), maxit=400)
                xf, itf = fgmres(Aa, ba, mk(lambda rb: apply_poly(coef, rb)))
            print(f"  Kbox + degree-{k - 1} GMRES polynomial ({label}): BiCGStab {it}  (FGMRES {itf})  res={np.linalg.norm(Aa @ xs - ba) / np.linalg.norm(ba):.1e}", flush=True)


def _main():
    for n in [int(a) for a in sys.argv[2:]] or [24, 32]:
        {"band": band_variants, "patch": patch_variants, "x0": x0_variants, "inner": inner_variants, "poly": poly_variants}.get(sys.argv[1], main)(n)


if __name__ == "__main__":
    _main()
