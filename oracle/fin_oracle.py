"""CPU ORACLE for the thermal-fin FOM + ROM hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain NumPy/SciPy restatement of the reference algorithm
(sheroze1123/BayesianInferenceDL); it is the *checker* for the HIP product path and
the timed ``cpu_baseline`` of ``bench.py``.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s cpu_baseline leg may import it.  The product package
``bayesianinferencedl_amd`` never does.

PARITY UNPINNED.  The reference holds no golden vectors, known-answer tests or seeded
outputs for this path (all its "tests" are plots, RNG unseeded), and its
own implementation cannot be imported here (dolfin / mshr / petsc4py / tensorflow are
absent: ordinary ModuleNotFoundError, not a permission denial).  Its mshr mesh is not
reproducible either, so the shipped bases / B_obs.txt cannot be bound to any mesh we can
build.  What pins this oracle instead (tests/test_oracle.py):
  * algebraic / physical invariants (sum F = 1, K 1 = 0, heat in = heat out, rows of S
    sum to 1, symmetry, SPD, mirror symmetry, five->nine map), all <= 1e-12;
  * the reference-style consistency checks restated as assertions (dense LSPG A8 ==
    sparse LSPG A6; snapshot-in-basis => ROM QoI == FOM QoI);
  * an np.longdouble re-computation of the LSPG normal equations;
  * the properties of the reference's data files (data/B_obs.txt rows are non-negative
    and sum to 1, 9 rows; bases are n x 81 CSV) checked when /root/reference is present.

Every function cites the reference file:line it restates (paths relative to the
reference repository root).  Deliberately written with explicit loops and floating-point
``between``/``near`` predicates, independently of the vectorised integer-lattice code in
``bayesianinferencedl_amd/fem.py`` -- the tests compare the two.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg
import scipy.sparse as sp
import scipy.sparse.linalg as spl
from scipy import spatial

DOLFIN_EPS = 3.0e-16
BIOT = 0.1                       # fom/forward_solve.py:112


def near(x, x0, eps=DOLFIN_EPS):
    return x0 - eps <= x <= x0 + eps


def between(x, ab):
    return ab[0] - DOLFIN_EPS <= x <= ab[1] + DOLFIN_EPS


# --------------------------------------------------------------------------------------
# A0  geometry + mesh (fom/thermal_fin.py:4-20).  mshr is replaced by a conforming
# lattice triangulation of pitch 1/m; node numbering: y-major, then x.
# --------------------------------------------------------------------------------------
RECTS = [((2.5, 0.0), (3.5, 4.0)),
         ((0.0, 0.75), (2.5, 1.0)), ((0.0, 1.75), (2.5, 2.0)),
         ((0.0, 2.75), (2.5, 3.0)), ((0.0, 3.75), (2.5, 4.0)),
         ((3.5, 0.75), (6.0, 1.0)), ((3.5, 1.75), (6.0, 2.0)),
         ((3.5, 2.75), (6.0, 3.0)), ((3.5, 3.75), (6.0, 4.0))]


def fin_mesh(m):
    """Returns (coords [n,2] float64, cells [nc,3] int)."""
    assert m % 4 == 0
    squares = set()
    for (xa, ya), (xb, yb) in RECTS:
        for i in range(int(round(xa * m)), int(round(xb * m))):
            for j in range(int(round(ya * m)), int(round(yb * m))):
                squares.add((i, j))
    pts = set()
    for (i, j) in squares:
        pts.update([(i, j), (i + 1, j), (i + 1, j + 1), (i, j + 1)])
    order = sorted(pts, key=lambda p: (p[1], p[0]))
    num = {p: t for t, p in enumerate(order)}
    coords = np.array([[i / m, j / m] for (i, j) in order])
    cells = []
    for (i, j) in sorted(squares, key=lambda p: (p[1], p[0])):
        a, b, c, d = num[(i, j)], num[(i + 1, j)], num[(i + 1, j + 1)], num[(i, j + 1)]
        if i < 3 * m:                       # mirror-symmetric about x = 3
            cells += [(a, b, c), (a, c, d)]
        else:
            cells += [(a, b, d), (b, c, d)]
    return coords, np.array(cells, dtype=np.int64)


# --------------------------------------------------------------------------------------
# sub-domain predicates (fom/forward_solve.py:5-38 == rom/averaged_affine_ROM.py:18-50)
# --------------------------------------------------------------------------------------
def subfin_inside(x, y_b, is_left):
    if is_left:
        return between(x[1], (y_b, y_b + 0.75)) and between(x[0], (0.0, 2.5))
    return between(x[1], (y_b, y_b + 0.75)) and between(x[0], (3.5, 6.0))


def centre_inside(x):
    return between(x[0], (2.5, 3.5))


# fin1..fin9 in the reference's order (rom/averaged_affine_ROM.py:91-99)
SUBFINS = [(0.75, True), (1.75, True), (2.75, True), (3.75, True), None,
           (3.75, False), (2.75, False), (1.75, False), (0.75, False)]


def _entity_inside(pred, pts):
    """DOLFIN SubDomain.mark: all vertices AND the midpoint must be inside."""
    mid = np.mean(pts, axis=0)
    return all(pred(p) for p in pts) and pred(mid)


def mark_cells(coords, cells):
    """rom/averaged_affine_ROM.py:101-112: markers 1..9 applied in order (later wins)."""
    marks = np.zeros(len(cells), dtype=int)
    for c, tri in enumerate(cells):
        pts = coords[tri]
        for idx, sf in enumerate(SUBFINS):
            pred = centre_inside if sf is None else (lambda x, sf=sf: subfin_inside(x, sf[0], sf[1]))
            if _entity_inside(pred, pts):
                marks[c] = idx + 1
    return marks


def exterior_facets(cells):
    cnt = {}
    for tri in cells:
        for a, b in ((tri[0], tri[1]), (tri[1], tri[2]), (tri[2], tri[0])):
            key = (min(a, b), max(a, b))
            cnt[key] = cnt.get(key, 0) + 1
    return [k for k, v in sorted(cnt.items()) if v == 1]


def mark_facets(coords, cells):
    """Facet markers of BOTH reference classes reduced to {robin, root, none}.

    Fin   (fom/forward_solve.py:147-152): exterior ``!near(y,0) && on_boundary`` -> 1,
          bottom ``near(y,0) && on_boundary`` -> 2.
    ROM   (rom/averaged_affine_ROM.py:116-138): fin1_b..fin9_b -> 1..9 (all carry the same
          Bi*v*w integrand, :154-162), bottom -> 10.
    Both give: robin = exterior facets with no vertex on y = 0, root = both vertices on
    y = 0; the two side-wall facets touching y = 0 get neither (all-vertices rule).
    Returns (robin_fom, root_fom, robin_rom, root_rom) as lists of (a, b)."""
    ext = exterior_facets(cells)
    robin_f, root_f, robin_r, root_r = [], [], [], []
    for (a, b) in ext:
        pts = coords[[a, b]]
        if _entity_inside(lambda x: not near(x[1], 0.0), pts):
            robin_f.append((a, b))
        if _entity_inside(lambda x: near(x[1], 0.0), pts):
            root_f.append((a, b))
        mark = 0
        for idx, sf in enumerate(SUBFINS):
            if sf is None:      # CenterFinBoundary (:48-50)
                pred = lambda x: between(x[0], (2.5, 3.5)) and not near(x[1], 0.0)
            else:
                pred = lambda x, sf=sf: subfin_inside(x, sf[0], sf[1])
            if _entity_inside(pred, pts):
                mark = idx + 1
        if _entity_inside(lambda x: near(x[1], 0.0), pts):
            mark = 10
        if 1 <= mark <= 9:
            robin_r.append((a, b))
        elif mark == 10:
            root_r.append((a, b))
    return robin_f, root_f, robin_r, root_r


# --------------------------------------------------------------------------------------
# P1 element matrices
# --------------------------------------------------------------------------------------
def p1_stiffness(p):
    """area * grad(phi_a) . grad(phi_b) on a triangle with vertices p[0..2]."""
    B = np.array([[p[1][0] - p[0][0], p[2][0] - p[0][0]],
                  [p[1][1] - p[0][1], p[2][1] - p[0][1]]])
    area = 0.5 * abs(np.linalg.det(B))
    G = np.linalg.solve(B.T, np.array([[-1.0, 1.0, 0.0], [-1.0, 0.0, 1.0]]))  # grads as columns
    return area, area * (G.T @ G)


class FinProblem:
    """All operators of the discrete problem, assembled cell by cell (dense-friendly sizes)."""

    def __init__(self, m):
        self.m = m
        self.coords, self.cells = fin_mesh(m)
        self.n = n = len(self.coords)
        self.cell_mark = mark_cells(self.coords, self.cells)
        assert (self.cell_mark > 0).all(), "conforming mesh: every cell in exactly one sub-fin (SURVEY S6)"
        rob_f, root_f, rob_r, root_r = mark_facets(self.coords, self.cells)
        assert rob_f == rob_r and root_f == root_r
        self.robin, self.root = rob_f, root_f

        self.area = np.zeros(len(self.cells))
        self.Kc = np.zeros((len(self.cells), 3, 3))
        for c, tri in enumerate(self.cells):
            self.area[c], self.Kc[c] = p1_stiffness(self.coords[tri])

        # Bi * int_Gamma w v ds  (exact P1 x P1: l/3, l/6)
        M = sp.lil_matrix((n, n))
        for (a, b) in self.robin:
            l = np.linalg.norm(self.coords[a] - self.coords[b])
            M[a, a] += l / 3; M[b, b] += l / 3; M[a, b] += l / 6; M[b, a] += l / 6
        self.BiM = (BIOT * M).tocsr()
        # B = assemble(v * ds(root))  (fom/forward_solve.py:162-163; rom :163,171,178)
        self.B = np.zeros(n)
        for (a, b) in self.root:
            l = np.linalg.norm(self.coords[a] - self.coords[b])
            self.B[a] += l / 2; self.B[b] += l / 2
        # A_i = assemble(grad w . grad v dx(i))   (rom/averaged_affine_ROM.py:215-220)
        self.A_sub = []
        for i in range(1, 10):
            Ai = sp.lil_matrix((n, n))
            for c in np.nonzero(self.cell_mark == i)[0]:
                tri = self.cells[c]
                for a in range(3):
                    for b in range(3):
                        Ai[tri[a], tri[b]] += self.Kc[c][a, b]
            self.A_sub.append(Ai.tocsr())
        # areas (fom :205-213) and the averaging operator (fom :488-511; rom :420-445)
        self.fin_area = np.array([self.area[self.cell_mark == i].sum() for i in range(1, 10)])
        S = np.zeros((9, n))
        for c, tri in enumerate(self.cells):
            for t in tri:
                S[self.cell_mark[c] - 1, t] += self.area[c] / 3.0      # int phi_t over a P1 cell
        self.S = S / self.fin_area[:, None]

    # fom/forward_solve.py:160-161: int k_h grad w . grad v dx + Bi int w v ds, k_h in P1
    # (P1 coefficient x piecewise-constant gradients: exact with the cell mean of k)
    def assemble_fom_loops(self, k_nodal):
        n = self.n
        rows, cols, vals = [], [], []
        for c, tri in enumerate(self.cells):
            kbar = (k_nodal[tri[0]] + k_nodal[tri[1]] + k_nodal[tri[2]]) / 3.0
            for a in range(3):
                for b in range(3):
                    rows.append(tri[a]); cols.append(tri[b]); vals.append(kbar * self.Kc[c][a, b])
        return (sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr() + self.BiM).tocsr()

    def assemble_fom(self, k_nodal):
        """Same operator as assemble_fom_loops with the cell loop vectorised (so that the timed
        cpu_baseline is not dominated by Python loop overhead DOLFIN would not have)."""
        if not hasattr(self, "_coo"):
            rows = np.repeat(self.cells, 3, axis=1).ravel()
            cols = np.tile(self.cells, (1, 3)).ravel()
            self._coo = (rows, cols)
        kbar = np.asarray(k_nodal)[self.cells].sum(1) / 3.0
        vals = (kbar[:, None, None] * self.Kc).ravel()
        return (sp.coo_matrix((vals, self._coo), shape=(self.n, self.n)).tocsr() + self.BiM).tocsr()

    # rom/averaged_affine_ROM.py:154-163
    def assemble_affine(self, theta):
        A = self.BiM.copy()
        for i in range(9):
            A = A + theta[i] * self.A_sub[i]
        return A.tocsr()


# --------------------------------------------------------------------------------------
# A2/A3/A4/A8  fom/forward_solve.py::Fin
# --------------------------------------------------------------------------------------
class FinOracle:
    def __init__(self, prob: FinProblem, external_obs=False, obs_seed=32):
        self.prob = prob
        self.dofs = prob.n
        self.B = prob.B
        if external_obs:
            # fom/forward_solve.py:215-228; the .npy is absent from the reference, its
            # commented recipe is: np.random.seed(32); np.random.choice(boundary dofs, 40)
            self.n_obs = 40
            bnd = np.zeros(prob.n, bool)
            for (a, b) in prob.robin:
                bnd[a] = bnd[b] = True
            rs = np.random.RandomState(obs_seed)
            b_vals = rs.choice(np.nonzero(bnd)[0], self.n_obs)
            self.B_obs = np.zeros((self.n_obs, prob.n))
            self.B_obs[np.arange(self.n_obs), b_vals] = 1
        else:
            self.n_obs = 9
            self.B_obs = self.observation_operator()       # :229-231

    def forward(self, k_nodal):                              # :270-291
        A = self.prob.assemble_fom(np.asarray(k_nodal, float))
        return spl.spsolve(A.tocsc(), self.B)              # DOLFIN default: sparse LU

    def gradient(self, k_nodal, data):                       # :293-322 (adjoint method)
        A = self.prob.assemble_fom(np.asarray(k_nodal, float)).tocsc()
        z = spl.spsolve(A, self.B)
        pred_obs = self.B_obs @ z
        adj_RHS = -(pred_obs - data) @ self.B_obs
        v = spl.spsolve(A, adj_RHS)                          # A_adj == A (symmetric form)
        # grad_j = int phi_j grad z . grad v dx = sum_cells (1/3) v_c^T K_c z_c
        g = np.zeros(self.prob.n)
        for c, tri in enumerate(self.prob.cells):
            val = v[tri] @ self.prob.Kc[c] @ z[tri] / 3.0
            for t in tri:
                g[t] += val
        return g

    def sensitivity(self, k_nodal):                          # :324-342 (one adjoint solve per observation)
        A = self.prob.assemble_fom(np.asarray(k_nodal, float)).tocsc()
        z = spl.spsolve(A, self.B)
        J = np.zeros((self.B_obs.shape[0], self.prob.n))
        for o in range(self.B_obs.shape[0]):
            v = spl.spsolve(A, -np.asarray(self.B_obs[o]).ravel())
            for c, tri in enumerate(self.prob.cells):
                val = v[tri] @ self.prob.Kc[c] @ z[tri] / 3.0
                for t in tri:
                    J[o, t] += val
        return J

    def reg(self, k_nodal, gamma=1e-6):                      # :190  0.5 gamma int |grad k|^2
        k = np.asarray(k_nodal, float)
        return 0.5 * gamma * sum(k[tri] @ self.prob.Kc[c] @ k[tri] for c, tri in enumerate(self.prob.cells))

    def grad_reg(self, k_nodal, gamma=1e-6):                 # :189  gamma int grad k . grad v
        k = np.asarray(k_nodal, float)
        g = np.zeros(self.prob.n)
        for c, tri in enumerate(self.prob.cells):
            g[tri] += gamma * (self.prob.Kc[c] @ k[tri])
        return g

    def qoi_operator(self, w):                               # :408-412
        return self.B_obs @ w

    def subfin_avg_op(self, k_nodal):                        # :466-480
        return self.prob.S @ k_nodal

    def observation_operator(self):                          # :488-511
        return self.prob.S.copy()

    def nine_param_to_function(self, k_s):                   # :61-91, 482-486
        out = np.zeros(self.prob.n)
        k1, k2, k3, k4, k5, k6, k7, k8, k9 = k_s
        for t, x in enumerate(self.prob.coords):
            if between(x[0], (2.5, 3.5)):
                v = k5
            elif x[0] <= 2.5:
                v = (k1 if between(x[1], (0.75, 1.0)) else k2 if between(x[1], (1.75, 2.0))
                     else k3 if between(x[1], (2.75, 3.0)) else k4 if between(x[1], (3.75, 4.0)) else 0.0)
            else:
                v = (k9 if between(x[1], (0.75, 1.0)) else k8 if between(x[1], (1.75, 2.0))
                     else k7 if between(x[1], (2.75, 3.0)) else k6 if between(x[1], (3.75, 4.0)) else 0.0)
            out[t] = v
        return out

    def five_param_to_function(self, k_s):                   # fom/forward_solve_petsc.py:243-260
        k1, k2, k3, k4, k5 = k_s
        out = np.zeros(self.prob.n)
        for t, x in enumerate(self.prob.coords):
            side = (x[0] < 2.5) or (x[0] > 3.5)
            out[t] = (k5 * ((x[0] >= 2.5) and (x[0] <= 3.5))
                      + k1 * ((0.75 <= x[1] <= 1.0) and side) + k2 * ((1.75 <= x[1] <= 2.0) and side)
                      + k3 * ((2.75 <= x[1] <= 3.0) and side) + k4 * ((3.75 <= x[1] <= 4.0) and side))
        return out

    def forward_five_param(self, k_s):                       # :267-268
        return self.forward(self.five_param_to_function(k_s))

    def reduced_forward(self, A, B, C, psi, phi):            # :421-452
        A_r = np.dot(psi.T, np.dot(A, phi))
        B_r = np.dot(psi.T, B)
        C_r = np.dot(C, phi)
        x_r = np.linalg.solve(A_r, B_r)
        y_r = np.dot(C_r, x_r)
        return A_r, B_r, C_r, x_r, y_r

    def r_fwd_no_full(self, k_nodal, phi, C):                # :454-464
        A_m = self.prob.assemble_fom(np.asarray(k_nodal, float)).toarray()
        psi = np.dot(A_m, phi)
        return self.reduced_forward(A_m, self.B, C, psi, phi)


# --------------------------------------------------------------------------------------
# A5/A6/A7  rom/averaged_affine_ROM.py::AffineROMFin
# --------------------------------------------------------------------------------------
class AffineROMOracle:
    def __init__(self, prob: FinProblem, phi, B_obs=None):
        self.prob = prob
        self.phi = np.asarray(phi, float)
        self.n, self.n_r = self.phi.shape                   # :72
        self.B = prob.B                                      # :171,178
        self.dsigma_dk = prob.S.copy()                       # :210
        self.B_obs = prob.S.copy() if B_obs is None else B_obs   # :192-208
        self.n_obs = self.B_obs.shape[0]
        self.B_obs_phi = self.B_obs @ self.phi               # :212
        self.dA_dsigmak_phi = np.stack([A @ self.phi for A in prob.A_sub])   # :215-220
        self.data = None

    def subfin_avg_op(self, k_nodal):                        # :404-418
        return self.prob.S @ k_nodal

    def forward(self, k_nodal):                              # :237-258 ("averaged FOM")
        A = self.prob.assemble_affine(self.subfin_avg_op(k_nodal))
        return spl.spsolve(A.tocsc(), self.B)

    def forward_reduced(self, k_nodal):                      # :260-276
        return self.forward_nine_param_reduced(self.subfin_avg_op(k_nodal))

    def forward_nine_param_reduced(self, k_s, return_parts=False):   # :278-310
        A = self.prob.assemble_affine(np.asarray(k_s, float))        # :289-292
        psi = A @ self.phi                                   # :295  matMult
        A_r = psi.T @ psi                                    # :296  transposeMatMult
        B_r = psi.T @ self.B                                 # :297  multTranspose
        w_r = np.linalg.solve(A_r, B_r)                      # :304
        if return_parts:
            return w_r, A_r, B_r, psi
        return w_r

    def qoi(self, w):                                        # :312-320
        return self.B_obs @ w

    def qoi_reduced(self, w_r):                              # :323-333
        return self.B_obs_phi @ w_r

    def set_data(self, data):                                # :398-399
        self.data = np.asarray(data, float)

    def grad_reduced(self, k_nodal):                         # :335-356
        w_r, A_r, B_r, psi = self.forward_nine_param_reduced(self.subfin_avg_op(k_nodal), True)
        obs = self.B_obs_phi @ w_r
        rhs = self.B_obs_phi.T @ (self.data - obs)
        v_r = np.linalg.solve(A_r.T, rhs)
        psi_v_r = psi @ v_r
        A_phi_w_r = np.dot(self.dA_dsigmak_phi, w_r).T       # [n, 9]
        dJ_dk = (psi_v_r @ A_phi_w_r) @ self.dsigma_dk       # == psi_v_r.T @ (A_phi_w_r @ dsigma_dk)
        J = 0.5 * np.linalg.norm(self.data - obs) ** 2
        return dJ_dk, J


def grad_romml_oracle(rom: "AffineROMOracle", model, k_nodal):
    """rom/averaged_affine_ROM.py:358-396 with an error model exposing predict / vjp (float32 like the Keras model)."""
    k_nodal = np.asarray(k_nodal, float)
    w_r, A_r, B_r, psi = rom.forward_nine_param_reduced(rom.subfin_avg_op(k_nodal), True)
    e_nn = np.asarray(model.predict(k_nodal[None, :])[0], float)
    obs = rom.B_obs_phi @ w_r
    resid = rom.data - (obs + e_nn)
    v_r = np.linalg.solve(A_r.T, rom.B_obs_phi.T @ resid)
    A_phi_w_r = np.dot(rom.dA_dsigmak_phi, w_r).T
    f_x = ((psi @ v_r) @ A_phi_w_r) @ rom.dsigma_dk
    nn = -np.asarray(model.vjp(k_nodal[None, :], resid[None, :])[0], float)
    return f_x + nn, 0.5 * float(resid @ resid)


def lspg_longdouble(prob: FinProblem, phi, theta):
    """Extended-precision (x87 80-bit) LSPG normal equations + Cholesky: tolerance 'truth'."""
    A = prob.assemble_affine(theta).toarray().astype(np.longdouble)
    P = phi.astype(np.longdouble)
    psi = A @ P
    A_r = psi.T @ psi
    B_r = psi.T @ prob.B.astype(np.longdouble)
    r = len(B_r)
    L = np.zeros((r, r), np.longdouble)
    for j in range(r):
        L[j, j] = np.sqrt(A_r[j, j] - L[j, :j] @ L[j, :j])
        for i in range(j + 1, r):
            L[i, j] = (A_r[i, j] - L[i, :j] @ L[j, :j]) / L[j, j]
    y = np.zeros(r, np.longdouble)
    for i in range(r):
        y[i] = (B_r[i] - L[i, :i] @ y[:i]) / L[i, i]
    x = np.zeros(r, np.longdouble)
    for i in range(r - 1, -1, -1):
        x[i] = (y[i] - L[i + 1:, i] @ x[i + 1:]) / L[i, i]
    return x


# --------------------------------------------------------------------------------------
# A9  bayesian_inference/gaussian_field.py:9-31
# --------------------------------------------------------------------------------------
def make_cov_chol(points, kern_type='m52', length=1.6):
    dists = spatial.distance.squareform(spatial.distance.pdist(points))
    if kern_type == 'sq_exp':
        alpha = 1 / (2 * length ** 2)
        cov = np.exp(-alpha * dists ** 2) + np.eye(len(points)) * 1e-5
    elif kern_type == 'm52':
        tmp = np.sqrt(5) * dists / length
        cov = (1 + tmp + tmp * tmp / 3) * np.exp(-tmp)
    else:
        tmp = np.sqrt(3) * dists / length
        cov = (1 + tmp) * np.exp(-tmp)
    return scipy.linalg.cholesky(cov)


# --------------------------------------------------------------------------------------
# A10  deep_learning/generate_fin_dataset.py:62-111 -- the hot loop, one sample at a time
# --------------------------------------------------------------------------------------
def gen_affine_avg_rom_dataset(prob, phi, fields, external_obs=False):
    """``fields`` [S,n]: the nodal conductivity samples (the reference draws
    exp(0.5 chol.T @ randn) unseeded at :87-88; the caller supplies them so both sides of
    a parity test see identical inputs).  Returns (z_s, qoi_errors, qois, qois_r)."""
    solver = FinOracle(prob, external_obs)
    solver_r = AffineROMOracle(prob, phi, B_obs=solver.B_obs)
    S = len(fields)
    qoi_errors = np.zeros((S, solver_r.n_obs)); qois = np.zeros((S, solver_r.n_obs))
    qois_r = np.zeros((S, solver_r.n_obs))
    for i in range(S):
        x = solver.forward(fields[i])                       # :93
        w_r = solver_r.forward_reduced(fields[i])           # :94
        qoi = solver.qoi_operator(x)                        # :96
        qoi_r = solver_r.qoi_reduced(w_r)                   # :97
        qoi_errors[i] = qoi - qoi_r; qois[i] = qoi; qois_r[i] = qoi_r   # :99-100
    return np.asarray(fields), qoi_errors, qois, qois_r


def sample_fields(chol, xi):
    """generate_fin_dataset.py:87-88: nodal_vals = exp(0.5 * chol.T @ norm), row-wise."""
    return np.exp(0.5 * (xi @ chol))


def philox4x32_10(c, k0, k1):
    """Philox4x32-10 (Salmon et al., SC'11; Random123): c = four uint64 arrays holding 32-bit counter words, (k0, k1) the key.
    Pinned by Random123's known-answer vectors in tests/test_oracle.py."""
    M = np.uint64(0xFFFFFFFF)
    c = [np.asarray(w, np.uint64) for w in c]
    k0, k1 = np.uint64(k0), np.uint64(k1)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]; p1 = np.uint64(0xCD9E8D57) * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k0, p1 & M, (p0 >> np.uint64(32)) ^ c[3] ^ k1, p0 & M]
        k0 = (k0 + np.uint64(0x9E3779B9)) & M; k1 = (k1 + np.uint64(0xBB67AE85)) & M
    return c


def philox_normal(seed, first, S, n):
    """NumPy restatement of the library's device draw (csrc/util_kernels.hip::philox_normal_kernel): Philox4x32-10 with
    counter (global sample index lo, hi, pair index, 0) and key (seed lo, hi); words (o0, o1) and (o2, o3) give two 53-bit
    uniforms u1 in (0, 1], u2 in [0, 1); Box-Muller pair -> xi[s, 2 jb], xi[s, 2 jb + 1].  The reference draws
    np.random.randn (deep_learning/generate_fin_dataset.py:87); this is the seeded, shard-independent stand-in."""
    npair = (n + 1) // 2
    g = (np.arange(S, dtype=np.uint64) + np.uint64(first))[:, None] + np.zeros((1, npair), np.uint64)
    c = philox4x32_10([g & np.uint64(0xFFFFFFFF), g >> np.uint64(32),
                       np.broadcast_to(np.arange(npair, dtype=np.uint64), g.shape).copy(), np.zeros_like(g)],
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    a = (c[1] << np.uint64(32)) | c[0]; b = (c[3] << np.uint64(32)) | c[2]
    u1 = ((a >> np.uint64(11)) + np.uint64(1)).astype(np.float64) * 2.0 ** -53
    u2 = (b >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    rad = np.sqrt(-2.0 * np.log(u1)); ang = 6.283185307179586476925286766559 * u2
    xi = np.empty((S, 2 * npair))
    xi[:, 0::2] = rad * np.cos(ang); xi[:, 1::2] = rad * np.sin(ang)
    return xi[:, :n]


def pod_basis(snapshots, r):
    """Orthonormal POD basis of the snapshot rows [num, n] -> [n, r] (SVD; the reference's
    recipe at rom/generate_reduced_basis_nine_param.py:296-318 leaves modes unnormalised,
    SURVEY S2/S8 -- orthonormal columns are used for parity on raw w_r)."""
    U, s, Vt = np.linalg.svd(np.asarray(snapshots), full_matrices=False)
    return Vt[:r].T.copy()
