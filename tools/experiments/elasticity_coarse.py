"""CPU experiment (round 3): two-level preconditioner for the 5-field interface-elasticity system (oracle matrices):
vertex-block Jacobi  +  a Galerkin coarse correction on trilinear functions of a coarse lattice (spacing H = ratio * h),
one set per displacement block (u_in[a], u_out[a]) restricted to the active, unconstrained DoFs of that block.
  B      vertex-block Jacobi (shipped)
  B+C    additive:        z = B^-1 r + R Ac^-1 R^T r
  C*B    multiplicative:  z = zc + B^-1 (r - A zc),  zc = R Ac^-1 R^T r
usage: elasticity_coarse.py n ratio [n ratio ...]"""
import os, sys, time, warnings
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from elasticity_lattice import problem
from precond_variants import bicgstab


def hat(t):
    return np.clip(1.0 - np.abs(t), 0.0, None)


def main(n, ratio, with_yp=False):
    t0 = time.time()
    A, b, act, nv, touched, bcv = problem(n, 1e-3)
    idx = np.flatnonzero(act)
    Aa = A[idx][:, idx].tocsr(); ba = b[idx]
    blk, vert = idx // nv, idx % nv
    isbc = np.zeros(nv, dtype=bool); isbc[bcv] = True
    n1 = n + 1
    i, j, k = vert % n1, (vert // n1) % n1, vert // (n1 * n1)
    nc = n // ratio                        # coarse cells per axis
    m1 = nc + 1
    # trilinear interpolation: fine vertex (i, j, k) <- coarse nodes
    cols, rows, vals = [], [], []
    ncoarse = 0
    nblocks = 27 if with_yp else 6
    for bq in range(nblocks):
        sel = np.flatnonzero((blk == bq) & ~((bq < 3) & isbc[vert]))
        if sel.size == 0:
            continue
        ti, tj, tk = i[sel] / ratio, j[sel] / ratio, k[sel] / ratio
        for di in (0, 1):
            for dj in (0, 1):
                for dk in (0, 1):
                    ci = np.minimum(np.floor(ti).astype(int) + di, nc); cj = np.minimum(np.floor(tj).astype(int) + dj, nc); ck = np.minimum(np.floor(tk).astype(int) + dk, nc)
                    w = hat(ti - ci) * hat(tj - cj) * hat(tk - ck)
                    keep = w > 1e-14
                    rows.append(sel[keep]); cols.append(ncoarse + ci[keep] + m1 * (cj[keep] + m1 * ck[keep])); vals.append(w[keep])
        ncoarse += m1 ** 3
    R = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(idx.size, ncoarse))
    R.sum_duplicates()
    used = np.flatnonzero(np.asarray(abs(R).sum(axis=0)).ravel() > 0)
    R = R[:, used].tocsr()
    Ac = (R.T @ Aa @ R).tocsc()
    if os.environ.get("PROBE", "0") == "1":
        # what the device does: 27 colours per block, a row takes the entry of the one same-colour node within one
        # coarse cell of its own node (entries two cells away are lumped onto it)
        co = Ac.tocoo()
        node = used % (m1 ** 3); cb = used // (m1 ** 3)
        cx, cy, cz = node % m1, (node // m1) % m1, node // (m1 * m1)
        def tgt(ci, cjp):      # per axis: the coordinate within +-1 of ci with the colour of cjp
            out = np.full(ci.shape, -1)
            for dl in (-1, 0, 1):
                q = ci + dl
                ok = (q >= 0) & (q < m1) & (q % 3 == cjp % 3)
                out = np.where(ok, q, out)
            return out
        tx, ty, tz = tgt(cx[co.row], cx[co.col]), tgt(cy[co.row], cy[co.col]), tgt(cz[co.row], cz[co.col])
        okk = (tx >= 0) & (ty >= 0) & (tz >= 0)
        full = cb[co.col] * m1 ** 3 + tx + m1 * (ty + m1 * tz)
        inv = -np.ones(ncoarse, dtype=np.int64); inv[used] = np.arange(used.size)
        J = np.where(okk, inv[np.where(okk, full, 0)], -1)
        keep = J >= 0
        moved = np.abs(co.data[keep & (J != co.col)]).sum() / np.abs(co.data).sum()
        print(f"   probing: {100 * moved:.2f} % of |Ac| lumped onto a neighbouring column, {np.abs(co.data[~keep]).sum() / np.abs(co.data).sum():.2e} dropped")
        Ac = sp.csc_matrix((co.data[keep], (co.row[keep], J[keep])), shape=Ac.shape)
    Aclu = spla.splu(Ac)

    order = np.argsort(vert, kind="stable")
    cuts = np.flatnonzero(np.diff(vert[order])) + 1
    groups = np.split(order, cuts)
    Binv = sp.block_diag([np.linalg.inv(Aa[g][:, g].toarray()) for g in groups], format="csr")
    perm = np.concatenate(groups)

    def MB(r):
        z = np.empty_like(r); z[perm] = Binv @ r[perm]; return z

    def C(r):
        return R @ Aclu.solve(R.T @ r)

    def MBC(r):
        return MB(r) + C(r)

    def MCB(r):
        zc = C(r)
        return zc + MB(r - Aa @ zc)

    def MBCB(r):     # symmetric multiplicative: B, C, B
        z = MB(r)
        z = z + C(r - Aa @ z)
        return z + MB(r - Aa @ z)

    print(f"n={n} H={ratio}h: {idx.size} DoFs, coarse {used.size}, set-up {time.time() - t0:.0f} s", flush=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, M in (("B", MB), ("B+C", MBC), ("C*B", MCB)):
            t0 = time.time(); xs, it = bicgstab(Aa, ba, M, rtol=1e-8, maxit=3000)
            print(f"   {name:8s} {it:5d} it  res {np.linalg.norm(Aa @ xs - ba) / np.linalg.norm(ba):.1e}  {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]] or [16, 4, 24, 4]
    for q in range(0, len(a), 2):
        main(a[q], a[q + 1])
