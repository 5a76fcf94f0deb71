"""The two elliptic systems of the hot path assembled INDEPENDENTLY of the solvers, as scipy sparse matrices (missing 4).

FBoxLib -- the reference's ml_cc_solve / ml_nd_solve (src/mac_multigrid.f90:53, src/hg_multigrid.f90:95) -- is not in the reference tree, so the
HIP solvers and the oracle share their author.  What this module adds is a second, unrelated derivation of the SAME discrete systems from the
statements of SURVEY.md Appendix C, with no code in common with oracle/ or varden_amd/:
  * cell-centred (MAC, Appendix C.1): (alpha - div beta grad) phi = rh, face by face -- an interior face couples its two cells with beta/h^2,
    a Neumann face carries nothing, a homogeneous Dirichlet face adds 2 beta/h^2 to the diagonal (linear closure), periodic faces wrap
    (src/macproject.f90:185-196, 376-394, 611-612; src/define_bc_tower.f90:297-334);
  * nodal (HG, Appendix C.2): the Q1 finite-element stiffness matrix with cell-constant sigma assembled ELEMENT BY ELEMENT from the tensor
    products of the 1-D stiffness and mass matrices, scaled by 1/(hx hy hz); walls are natural boundaries (no elements outside), outflow nodes
    are Dirichlet, periodic nodes are identified (src/hg_multigrid.f90:68-80, src/hgproject.f90:52, 434-513);
  * the two-level COMPOSITE forms of both (our definitions, oracle/vo_amr.c and vo_hgproject.c headers): for the cells the finite-volume
    equations on fine + uncovered coarse cells with the quadratic coarse-fine ghost interpolation and the fine fluxes through the interface; for the
    nodes the plain Galerkin system P^T K P of the conforming space with hanging (slave) interface nodes -- which has the oracle's row-combined
    system's SOLUTION but not its rows.
The tests compare A x with the oracle's operator (vo_cc_apply / vo_nd_apply) to round-off and a sparse direct solve with the multigrid / FAC
solutions of the oracle (CPU) and of the HIP library (GPU) to the solvers' tolerances."""
import itertools

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

PER, INT, DIR, NEU = -1, 0, 1, 2


def _coo():
    return [], [], []


def _csr(acc, shape):
    r, c, v = acc
    return sp.csr_matrix((np.concatenate([np.asarray(x, dtype=float).ravel() for x in v]),
                          (np.concatenate([np.asarray(x).ravel() for x in r]), np.concatenate([np.asarray(x).ravel() for x in c]))), shape=shape)


# ------------------------------------------------------------------------------------------------------------------------------------
# cell-centred, one level
# ------------------------------------------------------------------------------------------------------------------------------------
def cc_matrix(n, dx, beta, ellbc, alpha=None):
    """n: cells per direction; beta[d]: array of shape n + e_d (face coefficients); ellbc[d][side]; alpha: cell array or None"""
    n = tuple(int(x) for x in n)
    N = n[0] * n[1] * n[2]
    idx = np.arange(N).reshape(n, order="F")
    acc = _coo()

    def add(r, c, v):
        acc[0].append(r); acc[1].append(c); acc[2].append(np.broadcast_to(v, np.shape(r)))
    for d in range(3):
        h2 = 1.0 / (dx[d] * dx[d])
        nd = n[d]
        take = lambda a, s: np.take(a, s, axis=d)        # noqa: E731
        if nd > 1:                                        # interior faces 1 .. nd-1: -b (phi_hi - phi_lo)/h^2 leaves lo, enters hi
            lo, hi, c = take(idx, range(0, nd - 1)), take(idx, range(1, nd)), take(beta[d], range(1, nd)) * h2
            add(lo, lo, c); add(hi, hi, c); add(lo, hi, -c); add(hi, lo, -c)
        first, last = take(idx, [0]), take(idx, [nd - 1])
        b0, b1 = take(beta[d], [0]) * h2, take(beta[d], [nd]) * h2
        if ellbc[d][0] == PER:                            # each cell sees the coefficient of ITS OWN face (face 0 / face nd: equal for periodic data)
            add(first, first, b0); add(first, last, -b0); add(last, last, b1); add(last, first, -b1)
        else:
            if ellbc[d][0] == DIR:
                add(first, first, 2.0 * b0)               # phi = 0 ON the face, linear closure: gradient (phi_i - 0)/(h/2)
            if ellbc[d][1] == DIR:
                add(last, last, 2.0 * b1)
    if alpha is not None:
        add(idx, idx, np.asarray(alpha))
    return _csr(acc, (N, N))


def solve_maybe_singular(A, b, null=None):
    """sparse direct solve; with `null` (a vector spanning the null space: the constants of an all-Neumann / periodic problem) the bordered system
    [[A, e], [e^T, 0]] is solved instead -- the zero-mean solution; returns (x, lagrange multiplier)"""
    if null is None:
        return spla.spsolve(A.tocsc(), b), 0.0
    e = sp.csr_matrix(np.asarray(null, dtype=float).reshape(-1, 1))
    B = sp.bmat([[A, e], [e.T, None]], format="csc")
    x = spla.spsolve(B, np.concatenate([b, [0.0]]))
    return x[:-1], x[-1]


# ------------------------------------------------------------------------------------------------------------------------------------
# nodal, one level: Q1 elements
# ------------------------------------------------------------------------------------------------------------------------------------
def q1_element_matrices(h):
    """8 x 8 stiffness of the brick hx x hy x hz with unit sigma (corner a + 2 b + 4 c at offset (a, b, c)), and the 8 x 3 matrix of
    integrals of dN_corner/dx_d over the brick"""
    K1 = [np.array([[1.0, -1.0], [-1.0, 1.0]]) / h[d] for d in range(3)]
    M1 = [h[d] / 6.0 * np.array([[2.0, 1.0], [1.0, 2.0]]) for d in range(3)]
    Ke = np.kron(M1[2], np.kron(M1[1], K1[0])) + np.kron(M1[2], np.kron(K1[1], M1[0])) + np.kron(K1[2], np.kron(M1[1], M1[0]))
    G = np.zeros((8, 3))
    for c, b, a in itertools.product(range(2), repeat=3):
        o = (a, b, c)
        for d in range(3):
            t1, t2 = (d + 1) % 3, (d + 2) % 3
            G[a + 2 * b + 4 * c, d] = (1.0 if o[d] else -1.0) * h[t1] * h[t2] / 4.0
    return Ke, G


class NodalLevel:
    """nodes 0..n[d] of a box of n cells; periodic directions identify node n[d] with node 0.  K: stiffness (energy form, unscaled);
    w(u): the load vector  int u . grad N  of a cell-constant vector field"""

    def __init__(self, n, dx, per=(0, 0, 0)):
        self.n, self.dx, self.per = tuple(int(x) for x in n), tuple(float(x) for x in dx), tuple(int(x) for x in per)
        self.nn = tuple(self.n[d] + (0 if self.per[d] else 1) for d in range(3))       # unique nodes
        self.N = self.nn[0] * self.nn[1] * self.nn[2]
        self.Ke, self.G = q1_element_matrices(self.dx)
        self.vol = self.dx[0] * self.dx[1] * self.dx[2]
        ci = np.meshgrid(*[np.arange(self.n[d]) for d in range(3)], indexing="ij")
        self.corner = []
        for c, b, a in itertools.product(range(2), repeat=3):
            q = [(ci[0] + a), (ci[1] + b), (ci[2] + c)]
            for d in range(3):
                if self.per[d]:
                    q[d] = q[d] % self.n[d]
            self.corner.append((a + 2 * b + 4 * c, q[0] + self.nn[0] * (q[1] + self.nn[1] * q[2])))
        self.corner.sort()

    def node_index(self, i, j, k):
        q = [i, j, k]
        for d in range(3):
            if self.per[d]:
                q[d] = q[d] % self.n[d]
        return q[0] + self.nn[0] * (q[1] + self.nn[1] * q[2])

    def stiffness(self, sigma):
        acc = _coo()
        for a, na in self.corner:
            for b, nb in self.corner:
                acc[0].append(na); acc[1].append(nb); acc[2].append(sigma * self.Ke[a, b])
        return _csr(acc, (self.N, self.N))

    def load(self, u):
        """u: (nx, ny, nz, 3) cell values"""
        w = np.zeros(self.N)
        for a, na in self.corner:
            for d in range(3):
                np.add.at(w, na.ravel(), (u[..., d] * self.G[a, d]).ravel())
        return w

    def to_grid(self, x):
        """unique-node vector -> array on nodes 0..n[d] (aliases filled)"""
        g = np.asarray(x).reshape(self.nn, order="F")
        for d in range(3):
            if self.per[d]:
                g = np.concatenate([g, np.take(g, [0], axis=d)], axis=d)
        return g

    def from_grid(self, g):
        sl = tuple(slice(0, self.nn[d]) for d in range(3))
        return np.asarray(g)[sl].ravel(order="F")

    def dirichlet_mask(self, ellbc):
        m = np.zeros(self.nn, dtype=bool)
        for d in range(3):
            if ellbc[d][0] == DIR:
                sl = [slice(None)] * 3; sl[d] = 0; m[tuple(sl)] = True
            if ellbc[d][1] == DIR:
                sl = [slice(None)] * 3; sl[d] = self.nn[d] - 1; m[tuple(sl)] = True
        return m.ravel(order="F")


# ------------------------------------------------------------------------------------------------------------------------------------
# cell-centred, two levels (composite): unknowns = all coarse cells (the covered ones tied to the mean of their children) + fine cells
# ------------------------------------------------------------------------------------------------------------------------------------
class CompositeCC:
    """coarse level: one box [0, nc)^3 = the domain; fine level: the box flo..fhi, or -- `boxes` -- a UNION of boxes inside it (fine indices, even-aligned,
    properly nested; flo..fhi is then the bounding box and every fine array a level array over it).
    beta_c[d], beta_f[d]: face arrays of the two levels (fine: shape nf + e_d);  ellbc: the domain's [d][side] (no periodic sides here)"""

    def __init__(self, nc, dxc, flo, fhi, beta_c, beta_f, ellbc, boxes=None):
        self.nc = (nc,) * 3 if np.isscalar(nc) else tuple(nc)
        self.dxc = tuple(float(x) for x in dxc)
        self.dxf = tuple(0.5 * x for x in self.dxc)
        self.flo, self.fhi = tuple(flo), tuple(fhi)
        self.nf = tuple(fhi[d] - flo[d] + 1 for d in range(3))
        self.bc, self.bf, self.ellbc = beta_c, beta_f, ellbc
        self.mask = np.zeros(self.nf, dtype=bool)
        for lo, hi in (boxes if boxes is not None else [(self.flo, self.fhi)]):
            self.mask[tuple(slice(lo[d] - self.flo[d], hi[d] - self.flo[d] + 1) for d in range(3))] = True
        self.fnum = np.full(self.nf, -1, dtype=np.int64)       # unknown number of a fine cell (x fastest), -1 outside the union
        self.fnum[self.mask] = 0
        order = np.flatnonzero(self.mask.ravel(order="F"))
        flat = self.fnum.ravel(order="F").copy(); flat[order] = np.arange(order.size); self.fnum = flat.reshape(self.nf, order="F")
        self.Nc = int(np.prod(self.nc)); self.Nf = int(order.size)
        self.A = None

    # -- index helpers
    def cidx(self, Q):
        return Q[0] + self.nc[0] * (Q[1] + self.nc[1] * Q[2])

    def fidx(self, q):
        n = self.fnum[q[0] - self.flo[0], q[1] - self.flo[1], q[2] - self.flo[2]]
        assert n >= 0, "a fine cell outside the union was asked for"
        return self.Nc + int(n)

    def covered(self, Q):
        return self.in_fine((2 * Q[0], 2 * Q[1], 2 * Q[2]))

    def in_fine(self, q):
        return all(self.flo[d] <= q[d] <= self.fhi[d] for d in range(3)) and bool(self.mask[q[0] - self.flo[0], q[1] - self.flo[1], q[2] - self.flo[2]])

    def coarse_val(self, Q):
        """phi of coarse cell Q as a linear form; outside the domain the closure ghost: Neumann = the cell inside, Dirichlet = minus it"""
        Q = list(Q); w = 1.0
        for d in range(3):
            if Q[d] < 0:
                Q[d] = 0; w *= 1.0 if self.ellbc[d][0] == NEU else -1.0
            elif Q[d] >= self.nc[d]:
                Q[d] = self.nc[d] - 1; w *= 1.0 if self.ellbc[d][1] == NEU else -1.0
        return {self.cidx(Q): w}

    @staticmethod
    def axpy(acc, a, lin):
        for k, v in lin.items():
            acc[k] = acc.get(k, 0.0) + a * v

    def fine_val(self, q, d):
        """phi at fine position q reached from the fine cell next to it along d: a fine cell, a closure ghost at the domain boundary, or the
        coarse-fine ghost: (8/15) pcs + (2/3) f1 - (1/5) f2 with pcs the coarse parent moved to the ghost cell's transverse position by central
        differences of the coarse field (+-1/8 per transverse direction) and f1, f2 the two fine cells inside"""
        if self.in_fine(q):
            return {self.fidx(q): 1.0}
        side = 0 if q[d] < self.flo[d] else 1
        inside = list(q); inside[d] += 1 if side == 0 else -1
        nfd = 2 * self.nc[d]
        if q[d] < 0 or q[d] >= nfd:                        # domain boundary
            return {self.fidx(inside): 1.0 if self.ellbc[d][side] == NEU else -1.0}
        P = [x // 2 for x in q]
        out = {}
        pcs = dict(self.coarse_val(P))
        for t in range(3):
            if t == d:
                continue
            sg = 0.125 if (q[t] - 2 * P[t]) else -0.125
            Pp, Pm = list(P), list(P); Pp[t] += 1; Pm[t] -= 1
            self.axpy(pcs, sg, self.coarse_val(Pp)); self.axpy(pcs, -sg, self.coarse_val(Pm))
        self.axpy(out, 8.0 / 15.0, pcs)
        f2 = list(inside); f2[d] += 1 if side == 0 else -1
        self.axpy(out, 2.0 / 3.0, {self.fidx(inside): 1.0}); self.axpy(out, -0.2, {self.fidx(f2): 1.0})
        return out

    def bf_at(self, d, q):
        return self.bf[d][q[0] - self.flo[0], q[1] - self.flo[1], q[2] - self.flo[2]]

    def assemble(self):
        rows = []
        # coarse cells
        for K in range(self.nc[2]):
            for J in range(self.nc[1]):
                for I in range(self.nc[0]):
                    Q = (I, J, K); row = {}
                    if self.covered(Q):                     # tied to the mean of its eight children
                        row[self.cidx(Q)] = 1.0
                        for c, b, a in itertools.product(range(2), repeat=3):
                            self.axpy(row, -0.125, {self.fidx((2 * I + a, 2 * J + b, 2 * K + c)): 1.0})
                        rows.append(row); continue
                    me = {self.cidx(Q): 1.0}
                    for d in range(3):
                        hc2 = 1.0 / (self.dxc[d] ** 2)
                        for side in (0, 1):
                            Nb = list(Q); Nb[d] += 1 if side else -1
                            face = list(Q); face[d] += side          # face index of that side of the cell
                            if 0 <= Nb[d] < self.nc[d] and self.covered(Nb):
                                # interface: minus the mean of the four fine gradient fluxes beta (phi_out - phi_in)/h_f, divided by h_c; out = this side
                                t1, t2 = (d + 1) % 3, (d + 2) % 3
                                for b, a in itertools.product(range(2), repeat=2):
                                    qin = [0, 0, 0]; qin[d] = 2 * Nb[d] + (0 if side else 1); qin[t1] = 2 * Q[t1] + a; qin[t2] = 2 * Q[t2] + b
                                    qout = list(qin); qout[d] += -1 if side else 1
                                    fface = list(qin) if side else list(qout)         # the fine face between them (index of its hi cell)
                                    bfv = self.bf_at(d, fface)
                                    coef = 0.25 * bfv / (self.dxf[d] * self.dxc[d])
                                    self.axpy(row, coef, self.fine_val(qout, d)); self.axpy(row, -coef, {self.fidx(qin): 1.0})
                            else:
                                bcv = self.bc[d][tuple(face)]
                                self.axpy(row, bcv * hc2, me); self.axpy(row, -bcv * hc2, self.coarse_val(Nb))
                    rows.append(row)
        # fine cells (x fastest, the cells of the union only: the order of fnum)
        for k in range(self.flo[2], self.fhi[2] + 1):
            for j in range(self.flo[1], self.fhi[1] + 1):
                for i in range(self.flo[0], self.fhi[0] + 1):
                    if not self.in_fine((i, j, k)):
                        continue
                    q = (i, j, k); row = {}; me = {self.fidx(q): 1.0}
                    for d in range(3):
                        hf2 = 1.0 / (self.dxf[d] ** 2)
                        for side in (0, 1):
                            nb = list(q); nb[d] += 1 if side else -1
                            face = list(q); face[d] += side
                            bfv = self.bf_at(d, face)
                            self.axpy(row, bfv * hf2, me); self.axpy(row, -bfv * hf2, self.fine_val(nb, d))
                    rows.append(row)
        acc = _coo()
        for r, row in enumerate(rows):
            ks = list(row.keys())
            acc[0].append(np.full(len(ks), r)); acc[1].append(np.array(ks)); acc[2].append(np.array([row[k] for k in ks]))
        N = self.Nc + self.Nf
        self.A = _csr(acc, (N, N))
        return self.A

    def fine_vector(self, a):
        """the cells of the union out of a fine level array, in unknown order"""
        return np.asarray(a).ravel(order="F")[np.flatnonzero(self.mask.ravel(order="F"))]

    def rhs(self, rh_c, rh_f):
        b = np.concatenate([np.asarray(rh_c).ravel(order="F"), self.fine_vector(rh_f)])
        for K in range(self.nc[2]):
            for J in range(self.nc[1]):
                for I in range(self.nc[0]):
                    if self.covered((I, J, K)):
                        b[self.cidx((I, J, K))] = 0.0       # the tie rows
        return b

    def vector(self, phi_c, phi_f):
        return np.concatenate([np.asarray(phi_c).ravel(order="F"), self.fine_vector(phi_f)])

    def split(self, x):
        f = np.full(int(np.prod(self.nf)), np.nan)
        f[np.flatnonzero(self.mask.ravel(order="F"))] = x[self.Nc:]
        return x[:self.Nc].reshape(self.nc, order="F"), f.reshape(self.nf, order="F")


# ------------------------------------------------------------------------------------------------------------------------------------
# nodal, two levels: the conforming Galerkin system with slave interface nodes
# ------------------------------------------------------------------------------------------------------------------------------------
class CompositeND:
    """unknowns: every coarse node that is not strictly inside the fine region + every fine node strictly inside it; fine nodes ON the
    interface (not on the domain boundary) are trilinear slaves of the coarse nodes.  K = P^T diag(K_c[sigma = 0 under the fine level], K_f) P.
    The fine region is the box flo..fhi or -- `boxes` -- a union of boxes inside it (fine arrays are level arrays over the bounding box; a node
    belongs to the fine level when a cell of the union touches it, and is a slave when a cell of the domain that is NOT of the union touches it too)."""

    def __init__(self, nc, dxc, flo, fhi, boxes=None):
        self.nc = (nc,) * 3 if np.isscalar(nc) else tuple(nc)
        self.C = NodalLevel(self.nc, dxc)
        self.flo, self.fhi = tuple(flo), tuple(fhi)
        self.nf = tuple(fhi[d] - flo[d] + 1 for d in range(3))
        self.F = NodalLevel(self.nf, [0.5 * x for x in dxc])
        nfd = [2 * self.nc[d] for d in range(3)]
        self.fmask = np.zeros(self.nf, dtype=bool)
        for lo, hi in (boxes if boxes is not None else [(self.flo, self.fhi)]):
            self.fmask[tuple(slice(lo[d] - flo[d], hi[d] - flo[d] + 1) for d in range(3))] = True
        # per fine node (of the bounding box): cells of the union / cells of the domain outside the union around it
        own = np.zeros(self.F.nn, dtype=int); other = np.zeros(self.F.nn, dtype=int)
        for c, b, a in itertools.product(range(-1, 1), repeat=3):
            for k in range(self.F.nn[2]):
                kk = k + c
                for j in range(self.F.nn[1]):
                    jj = j + b
                    for i in range(self.F.nn[0]):
                        ii = i + a
                        g = (flo[0] + ii, flo[1] + jj, flo[2] + kk)
                        if not all(0 <= g[d] < nfd[d] for d in range(3)):
                            continue                          # outside the domain: a natural boundary, neither
                        inb = 0 <= ii < self.nf[0] and 0 <= jj < self.nf[1] and 0 <= kk < self.nf[2]
                        if inb and self.fmask[ii, jj, kk]:
                            own[i, j, k] += 1
                        else:
                            other[i, j, k] += 1
        self.slave = (own > 0) & (other > 0)
        funk = (own > 0) & (other == 0)
        # unknowns: coarse nodes whose fine twin is not a fine unknown, then the fine unknowns
        Cn = np.zeros(self.C.nn, dtype=int) - 1
        cnt = 0
        for K in range(self.C.nn[2]):
            for J in range(self.C.nn[1]):
                for I in range(self.C.nn[0]):
                    t = (2 * I - flo[0], 2 * J - flo[1], 2 * K - flo[2])
                    twin = all(0 <= t[d] < self.F.nn[d] for d in range(3)) and funk[t]
                    if not twin:
                        Cn[I, J, K] = cnt; cnt += 1
        self.Cn, self.ncu = Cn, cnt
        Fn = np.zeros(self.F.nn, dtype=int) - 1
        for k in range(self.F.nn[2]):
            for j in range(self.F.nn[1]):
                for i in range(self.F.nn[0]):
                    if funk[i, j, k]:
                        Fn[i, j, k] = cnt; cnt += 1
        self.Fn, self.N = Fn, cnt
        # prolongation P: (all coarse nodes, all fine nodes of the bounding box) <- unknowns
        acc = _coo()
        Ncn = self.C.N
        for K in range(self.C.nn[2]):
            for J in range(self.C.nn[1]):
                for I in range(self.C.nn[0]):
                    if Cn[I, J, K] >= 0:
                        acc[0].append([self.C.node_index(I, J, K)]); acc[1].append([Cn[I, J, K]]); acc[2].append([1.0])
        for k in range(self.F.nn[2]):
            for j in range(self.F.nn[1]):
                for i in range(self.F.nn[0]):
                    r = Ncn + self.F.node_index(i, j, k)
                    if funk[i, j, k]:
                        acc[0].append([r]); acc[1].append([Fn[i, j, k]]); acc[2].append([1.0])
                        continue
                    if not self.slave[i, j, k]:
                        continue                              # no cell of the union touches it: not a node of the level
                    g = (flo[0] + i, flo[1] + j, flo[2] + k)                   # global fine node index
                    base = [x // 2 for x in g]; odd = [x & 1 for x in g]
                    for c, b, a in itertools.product(range(2), repeat=3):
                        if (a and not odd[0]) or (b and not odd[1]) or (c and not odd[2]):
                            continue
                        w = 1.0 / ((1 + odd[0]) * (1 + odd[1]) * (1 + odd[2]))
                        Q = (base[0] + a, base[1] + b, base[2] + c)
                        assert Cn[Q] >= 0, "a slave node's parent is not an unknown"
                        acc[0].append([r]); acc[1].append([Cn[Q]]); acc[2].append([w])
        self.P = _csr(acc, (Ncn + self.F.N, self.N))
        self.fnode = own > 0
        self.cmask = np.ones(self.nc)                      # 0 in covered coarse cells
        cm = self.fmask[::2, ::2, ::2]
        self.cmask[tuple(slice(flo[d] // 2, flo[d] // 2 + cm.shape[d]) for d in range(3))] = np.where(cm, 0.0, 1.0)

    def system(self, sigma_c, sigma_f, u_c, u_f):
        fm = self.fmask.astype(float)
        Kc = self.C.stiffness(sigma_c * self.cmask); Kf = self.F.stiffness(sigma_f * fm)
        Kall = sp.block_diag([Kc, Kf], format="csr")
        w = np.concatenate([self.C.load(u_c * self.cmask[..., None]), self.F.load(u_f * fm[..., None])])
        return (self.P.T @ Kall @ self.P).tocsr(), self.P.T @ w

    def scatter(self, y):
        """solution on all nodes of both levels (slaves interpolated; coarse nodes inside the fine region and nodes of the fine array outside the
        level NaN)"""
        full = self.P @ y
        c = self.C.to_grid(full[:self.C.N]); f = self.F.to_grid(full[self.C.N:])
        c = np.where(self.Cn >= 0, c, np.nan)
        f = np.where(self.fnode, f, np.nan)
        return c, f
