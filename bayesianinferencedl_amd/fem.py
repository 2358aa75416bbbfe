"""Host-side problem definition for the thermal-fin hot path: conforming
structured triangulation, P1 function-space shim and one-time operator assembly.

This is one-time *setup* code (NumPy); every per-sample operation of the hot
path runs in the HIP library (``csrc/``).  Nothing here imports ``oracle/``.

Reference semantics restated here (all paths relative to the reference repo):

* geometry        fom/thermal_fin.py:7-15   (post [2.5,3.5]x[0,4], 8 fins 2.5x0.25)
* cell markers    rom/averaged_affine_ROM.py:101-112 (fin1..4 left bottom->top,
                  fin5 centre, fin6..9 right TOP->BOTTOM)
* facet markers   fom/forward_solve.py:147-152 ; rom/averaged_affine_ROM.py:116-138
                  (exterior facets not touching y=0 are Robin, facets on y=0 are the
                  root; the two lowest side-wall facets have one vertex on y=0 and so
                  are *neither* -- DOLFIN marks an entity only if all its vertices and
                  its midpoint are inside)
* weak forms      fom/forward_solve.py:160-163 ; rom/averaged_affine_ROM.py:154-163
* sub-fin average fom/forward_solve.py:466-511 ; rom/averaged_affine_ROM.py:404-445

The reference meshes with mshr (not reproducible); this module builds a
*conforming* lattice mesh of pitch 1/m (m a multiple of 4 so that every rectangle
edge is a mesh line), mirror-symmetric about x = 3.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

BIOT = 0.1  # fom/forward_solve.py:112


import contextlib


@contextlib.contextmanager
def deterministic_blas():
    """One BLAS thread for the one-time dense host products (B_obs Phi, Psi_p^T Psi_q, the POD SVD): a threaded GEMM / SVD
    splits its sums differently for different thread counts, and the ranks of a multi-GPU run (OMP_NUM_THREADS=1 each) must
    build the SAME operators as a lone process, bit for bit, for the outputs to be independent of the GPU count."""
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        import os
        import warnings
        if os.environ.get("OMP_NUM_THREADS") != "1":
            warnings.warn("threadpoolctl is not installed and OMP_NUM_THREADS != 1: the one-time dense host products may be summed "
                          "in a thread-count-dependent order, so outputs are no longer guaranteed bit-identical across rank counts "
                          "(set OMP_NUM_THREADS=1 or install threadpoolctl)", RuntimeWarning, stacklevel=3)
        yield
        return
    with threadpool_limits(limits=1):
        yield

# sub-fin bands, reference numbering (rom/averaged_affine_ROM.py:91-99):
#   index 0..3 = fin1..4 (left,  y_b = 0.75, 1.75, 2.75, 3.75)
#   index 4    = fin5    (centre post)
#   index 5..8 = fin6..9 (right, y_b = 3.75, 2.75, 1.75, 0.75)
_LEFT_YB = (0.75, 1.75, 2.75, 3.75)
_RIGHT_YB = (3.75, 2.75, 1.75, 0.75)


def resolution_to_m(resolution: int) -> int:
    """Map the reference's mshr ``resolution`` (fom/thermal_fin.py:17, 40 -> 1446
    DoF) to our lattice divisor m (40 -> m=12 -> 1597 DoF; 67 -> m=20 -> 4101)."""
    return max(4, 4 * int(round(0.075 * resolution)))


class FinMesh:
    """Conforming triangulation of the fin on the lattice of pitch 1/m."""

    def __init__(self, m: int):
        if m % 4 != 0 or m <= 0:
            raise ValueError("lattice divisor m must be a positive multiple of 4")
        self.m = m
        q, h2, x0, x1, xe, ytop = m // 4, m // 2, 5 * m // 2, 7 * m // 2, 6 * m, 4 * m
        # square (i, j) = [i, i+1] x [j, j+1] in lattice units; region id 0..8 per square
        sq_i, sq_j, sq_reg = [], [], []
        for j in range(ytop):
            for i in range(x0, x1):
                sq_i.append(i); sq_j.append(j); sq_reg.append(4)
        for b, yb in enumerate(_LEFT_YB):
            j0 = int(round(yb * m))
            for j in range(j0, j0 + q):
                for i in range(0, x0):
                    sq_i.append(i); sq_j.append(j); sq_reg.append(b)
        for b, yb in enumerate(_RIGHT_YB):
            j0 = int(round(yb * m))
            for j in range(j0, j0 + q):
                for i in range(x1, xe):
                    sq_i.append(i); sq_j.append(j); sq_reg.append(5 + b)
        sq_i = np.asarray(sq_i); sq_j = np.asarray(sq_j); sq_reg = np.asarray(sq_reg)

        # nodes = lattice corners of the squares, numbered y-major then x
        corners = np.concatenate([
            np.stack([sq_i + di, sq_j + dj], 1) for di in (0, 1) for dj in (0, 1)])
        key = corners[:, 1] * (xe + 1) + corners[:, 0]
        ukey = np.unique(key)
        self.lattice = np.stack([ukey % (xe + 1), ukey // (xe + 1)], 1)  # (i, j) ints
        self.coords = self.lattice.astype(np.float64) / m
        self.n = len(ukey)

        def nid(i, j):
            return np.searchsorted(ukey, j * (xe + 1) + i)

        a = nid(sq_i, sq_j); b_ = nid(sq_i + 1, sq_j)
        c = nid(sq_i + 1, sq_j + 1); d = nid(sq_i, sq_j + 1)
        # '/' diagonal left of the symmetry axis x = 3, '\' right of it
        left = sq_i < 3 * m
        t1 = np.where(left[:, None], np.stack([a, b_, c], 1), np.stack([a, b_, d], 1))
        t2 = np.where(left[:, None], np.stack([a, c, d], 1), np.stack([b_, c, d], 1))
        self.cells = np.concatenate([t1, t2]).astype(np.int64)
        self.cell_region = np.concatenate([sq_reg, sq_reg]).astype(np.int64)

        # exterior facets: edges owned by exactly one cell
        e = np.concatenate([self.cells[:, [0, 1]], self.cells[:, [1, 2]], self.cells[:, [2, 0]]])
        e.sort(axis=1)
        ek = e[:, 0] * self.n + e[:, 1]
        uk, cnt = np.unique(ek, return_counts=True)
        bk = uk[cnt == 1]
        self.bfacets = np.stack([bk // self.n, bk % self.n], 1)
        jy = self.lattice[:, 1]
        on_root = jy[self.bfacets] == 0
        self.root_facets = self.bfacets[on_root.all(1)]          # ds(2) / ds(10)
        self.robin_facets = self.bfacets[~on_root.any(1)]        # ds(1) / ds(1..9)

    # -- DOLFIN-like accessors ------------------------------------------------
    def num_vertices(self):
        return self.n

    def num_cells(self):
        return len(self.cells)

    def coordinates(self):
        return self.coords


class _DofMap:
    def __init__(self, n):
        self._n = n

    def dofs(self):
        return np.arange(self._n)


class FunctionSpace:
    """Minimal P1 ``FunctionSpace`` exposing what the hot-path callers touch
    (deep_learning/generate_fin_dataset.py:64,81; bayesian_inference/gaussian_field.py:10-12)."""

    def __init__(self, mesh: FinMesh):
        self._mesh = mesh
        self._ops = None

    def dim(self):
        return self._mesh.n

    def dofmap(self):
        return _DofMap(self._mesh.n)

    def mesh(self):
        return self._mesh

    def tabulate_dof_coordinates(self):
        return self._mesh.coords.copy()

    def operators(self) -> "FinOperators":
        if self._ops is None:
            self._ops = FinOperators(self._mesh)
        return self._ops


class _Vector:
    """The slice of dolfin's GenericVector API the callers use.  Every write invalidates what the owning Function carries about
    its old state (the observables the solve kernel computed for it)."""

    def __init__(self, a, owner=None):
        self._a = a
        self._owner = owner

    def _touch(self):
        if self._owner is not None:
            self._owner._qoi = None

    def set_local(self, values):
        self._a[:] = np.asarray(values, dtype=np.float64).reshape(self._a.shape)
        self._touch()

    def get_local(self):
        return self._a.copy()

    def axpy(self, alpha, other):
        self._a += alpha * (other._a if isinstance(other, _Vector) else np.asarray(other))
        self._touch()

    def __getitem__(self, idx):
        return self._a[idx].copy() if isinstance(idx, slice) else self._a[idx]

    def __setitem__(self, idx, v):
        self._a[idx] = v
        self._touch()

    def __len__(self):
        return len(self._a)

    def __array__(self, dtype=None, copy=None):
        return np.array(self._a, dtype=dtype)


class Function:
    def __init__(self, V: FunctionSpace, values=None):
        self.V = V
        self._a = np.zeros(V.dim())
        self._qoi = None                 # B_obs w as computed by the solve kernel, valid only while the state is untouched
        if values is not None:
            self._a[:] = values

    def vector(self):
        return _Vector(self._a, self)

    def assign(self, other):
        self._a[:] = other._a if isinstance(other, Function) else np.asarray(other)
        self._qoi = None

    def function_space(self):
        return self.V


def as_nodal(k) -> np.ndarray:
    """Function | array -> 1-D float64 nodal array (no copy for Functions)."""
    if isinstance(k, Function):
        return k._a
    return np.ascontiguousarray(k, dtype=np.float64)


def _p1_gradients(xy):
    """xy: [nc,3,2] -> (area [nc], K_c [nc,3,3] = area * grad_a . grad_b)."""
    x, y = xy[..., 0], xy[..., 1]
    b = np.stack([y[:, 1] - y[:, 2], y[:, 2] - y[:, 0], y[:, 0] - y[:, 1]], 1)
    c = np.stack([x[:, 2] - x[:, 1], x[:, 0] - x[:, 2], x[:, 1] - x[:, 0]], 1)
    det = (x[:, 1] - x[:, 0]) * (y[:, 2] - y[:, 0]) - (x[:, 2] - x[:, 0]) * (y[:, 1] - y[:, 0])
    area = 0.5 * np.abs(det)
    K = (b[:, :, None] * b[:, None, :] + c[:, :, None] * c[:, None, :]) / (4.0 * area)[:, None, None]
    return area, K


class FinOperators:
    """All one-time operators of the hot path on ONE shared CSR pattern.

    Attributes
    ----------
    indptr, indices : shared CSR pattern (n x n, symmetric, sorted, diagonal present)
    robin_vals      : Bi * M_Gamma on the pattern          (rom :154-163, fom :160-161)
    sub_vals[9]     : stiffness of sub-domain i on the pattern (rom :215-220 ``dA_dsigmak``)
    F               : load  int_{root} v ds                 (fom :162-163)
    S               : 9 x n sub-fin averaging operator (= observation_operator, fom :488-511)
    areas           : |Omega_i|                              (fom :205-213)
    W_field         : (nnz x n) CSR; A(k).data = robin_vals + W_field @ k for a nodal
                      P1 conductivity field k (1-point rule exact)
    N9              : n x 9 nodal interpolation of nine_param_to_function (fom :61-91)
    """

    def __init__(self, mesh: FinMesh):
        self.mesh = mesh
        n, cells = mesh.n, mesh.cells
        area, K = _p1_gradients(mesh.coords[cells])
        self.cell_area = area

        rows = np.repeat(cells, 3, axis=1).ravel()          # a index repeated over b
        cols = np.tile(cells, (1, 3)).ravel()
        # boundary mass (exact P1xP1: l/3 diag, l/6 off-diag) and load (l/2)
        rf, tf = mesh.robin_facets, mesh.root_facets
        lr = np.linalg.norm(mesh.coords[rf[:, 0]] - mesh.coords[rf[:, 1]], axis=1)
        lt = np.linalg.norm(mesh.coords[tf[:, 0]] - mesh.coords[tf[:, 1]], axis=1)
        mr = np.concatenate([rf[:, 0], rf[:, 1], rf[:, 0], rf[:, 1]])
        mc = np.concatenate([rf[:, 0], rf[:, 1], rf[:, 1], rf[:, 0]])
        mv = BIOT * np.concatenate([lr / 3, lr / 3, lr / 6, lr / 6])

        pat = sp.coo_matrix((np.ones(len(rows) + len(mr)),
                             (np.concatenate([rows, mr]), np.concatenate([cols, mc]))),
                            shape=(n, n)).tocsr()
        pat.sum_duplicates(); pat.sort_indices()
        self.indptr = pat.indptr.astype(np.int32)
        self.indices = pat.indices.astype(np.int32)
        self.nnz = int(pat.nnz)
        self.n = n
        keys = np.repeat(np.arange(n, dtype=np.int64), np.diff(pat.indptr)) * n + pat.indices

        def eidx(r, c):
            return np.searchsorted(keys, np.asarray(r, np.int64) * n + np.asarray(c, np.int64))

        def on_pattern(r, c, v):
            out = np.zeros(self.nnz)
            np.add.at(out, eidx(r, c), v)
            return out

        self.robin_vals = on_pattern(mr, mc, mv)
        self.sub_vals = np.stack([
            on_pattern(rows[np.repeat(mesh.cell_region == i, 9)],
                       cols[np.repeat(mesh.cell_region == i, 9)],
                       K[mesh.cell_region == i].ravel()) for i in range(9)])
        self.F = np.zeros(n)
        np.add.at(self.F, tf[:, 0], lt / 2); np.add.at(self.F, tf[:, 1], lt / 2)

        # sub-fin averaging operator S (9 x n)
        self.areas = np.array([area[mesh.cell_region == i].sum() for i in range(9)])
        S = np.zeros((9, n))
        for t in range(3):
            np.add.at(S, (mesh.cell_region, cells[:, t]), area / 3.0)
        self.S = S / self.areas[:, None]

        # nodal-field operator: entry(a,b) of cell c gets K_c[a,b]/3 * (k_0+k_1+k_2)
        e_cell = eidx(rows, cols)                               # [nc*9]
        wr = np.repeat(e_cell, 3)
        wc = np.repeat(cells, 9, axis=0).ravel()                # each entry sees the 3 nodes
        wv = np.repeat(K.ravel() / 3.0, 3)
        W = sp.coo_matrix((wv, (wr, wc)), shape=(self.nnz, n)).tocsr()
        W.sum_duplicates(); W.sort_indices()
        self.W_field = W

        # nine-parameter nodal interpolation (fom/forward_solve.py:61-91): nodes with
        # 2.5 <= x <= 3.5 take k5; else the band value of their side; right side k9..k6
        li, lj = mesh.lattice[:, 0], mesh.lattice[:, 1]
        m = mesh.m
        N9 = np.zeros((n, 9))
        centre = (li >= 5 * m // 2) & (li <= 7 * m // 2)
        N9[centre, 4] = 1.0
        for b, yb in enumerate(_LEFT_YB):
            j0 = int(round(yb * m))
            band = (lj >= j0) & (lj <= j0 + m // 4)
            N9[band & (li < 5 * m // 2), b] = 1.0
        for b, yb in enumerate(_RIGHT_YB):
            j0 = int(round(yb * m))
            band = (lj >= j0) & (lj <= j0 + m // 4)
            N9[band & (li > 7 * m // 2), 5 + b] = 1.0
        self.N9 = N9
        # five -> nine (fom/forward_solve_petsc.py:243-260): k5 centre, k_i both sides
        E = np.zeros((9, 5))
        for b in range(4):
            E[b, b] = 1.0          # left fin b   (bottom -> top)
            E[8 - b, b] = 1.0      # right fin, same height (fin9 is the lowest)
        E[4, 4] = 1.0
        self.E59 = E

    def band_plan(self):
        """Plan of the frontal band sweep (bandplan.py) for this mesh, or None if the operator does not have the structure
        it assumes.  Entries that are zero in EVERY operator table (the hypotenuse couplings) are left out of the band."""
        if not hasattr(self, "_band_plan"):
            from .bandplan import BandPlan, BandPlanError, nonzero_entries
            try:
                self._band_plan = BandPlan(self.mesh, self.indptr, self.indices,
                                           nonzero_entries(self.robin_vals, self.W_field, list(self.sub_vals)))
            except BandPlanError:
                self._band_plan = None
        return self._band_plan

    # -- convenience ----------------------------------------------------------
    def csr(self, vals):
        return sp.csr_matrix((vals, self.indices, self.indptr), shape=(self.n, self.n))

    def fom_values(self, k_nodal):
        """CSR values of A(k) for a nodal conductivity field (fom :160-161)."""
        return self.robin_vals + self.W_field @ k_nodal

    def affine_values(self, theta9):
        """CSR values of sum_i theta_i A_i + Bi M_Gamma (rom :154-163)."""
        return self.robin_vals + np.asarray(theta9) @ self.sub_vals

    def boundary_dofs(self):
        """DoFs on exterior facets not on y=0 (fom :215-220 ``boundary_indices``)."""
        jy = self.mesh.lattice[:, 1]
        nodes = np.unique(self.mesh.bfacets)
        return nodes[jy[nodes] != 0]
