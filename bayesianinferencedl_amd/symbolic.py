"""One-time symbolic phase of the batched sparse Cholesky used by the FOM solve.

Every conductivity sample shares the sparsity pattern of A(k) (fom/forward_solve.py:160-161
assembles the same form for every k), so the fill-reducing ordering, the structure of
L and the complete elimination *schedule* are computed once on the host.  The device
kernel (csrc/fom_kernels.hip) then executes that schedule with lane = sample: the
control flow is identical for all samples, all loads are coalesced over the batch.

The reference delegates this solve to DOLFIN's default sparse LU
(``solve(self._F == self._a, z)``, fom/forward_solve.py:285-286); A(k) is SPD for k>0,
so Cholesky computes the same w.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def minimum_degree(indptr, indices, n):
    """Plain minimum-degree ordering on the elimination graph (ties: lowest index).

    n <= ~5k here, so explicit adjacency sets are fine (one-time setup)."""
    adj = [set(indices[indptr[i]:indptr[i + 1]].tolist()) - {i} for i in range(n)]
    import heapq
    heap = [(len(adj[i]), i) for i in range(n)]
    heapq.heapify(heap)
    done = np.zeros(n, bool)
    order = []
    while heap:
        d, v = heapq.heappop(heap)
        if done[v] or d != len(adj[v]):
            continue
        done[v] = True
        order.append(v)
        nb = adj[v]
        for u in nb:
            au = adj[u]
            au.discard(v)
            au |= nb
            au.discard(u)
        for u in nb:
            heapq.heappush(heap, (len(adj[u]), u))
        adj[v] = set()
    return np.asarray(order, dtype=np.int64)


def _etree(n, lower_rows):
    """Liu's elimination tree from the strictly-lower row patterns (sorted arrays)."""
    parent = np.full(n, -1, np.int64)
    anc = np.full(n, -1, np.int64)
    for i in range(n):
        for k in lower_rows[i]:
            k = int(k)
            while k != -1 and k < i:
                nxt = anc[k]
                anc[k] = i
                if nxt == -1:
                    parent[k] = i
                k = nxt
    return parent


def _row_structures(n, lower_rows, parent):
    """Strictly-lower row structure of L via the row-subtree walk (ereach)."""
    mark = np.full(n, -1, np.int64)
    rows = []
    for i in range(n):
        mark[i] = i
        out = []
        for k in lower_rows[i]:
            k = int(k)
            while mark[k] != i:
                out.append(k)
                mark[k] = i
                k = int(parent[k])
        rows.append(np.sort(np.asarray(out, dtype=np.int64)))
    return rows


def _ordering(indptr, indices, n, name):
    if name == "md":
        return minimum_degree(indptr, indices, n)
    if name == "natural":
        return np.arange(n, dtype=np.int64)
    A = sp.csr_matrix((np.ones(len(indices)), indices, indptr), shape=(n, n))
    if name == "rcm":
        from scipy.sparse.csgraph import reverse_cuthill_mckee
        return np.asarray(reverse_cuthill_mckee(A, symmetric_mode=True), np.int64)
    if name == "mmd":
        # SuperLU's multiple-minimum-degree on A'+A; only its column permutation is used
        from scipy.sparse.linalg import splu
        B = (A + sp.identity(n) * (2.0 * abs(A).sum(1).max())).tocsc()
        return np.argsort(splu(B, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0).perm_c).astype(np.int64)
    raise ValueError(name)


def _pair_count(indptr, indices, n, perm):
    """Multiply-adds of the Cholesky factorisation under ``perm`` (column counts only)."""
    iperm = np.empty(n, np.int64); iperm[perm] = np.arange(n)
    lower = []
    for i in range(n):
        o = perm[i]
        c = iperm[indices[indptr[o]:indptr[o + 1]]]
        lower.append(np.sort(c[c < i]))
    rows = _row_structures(n, lower, _etree(n, lower))
    cc = np.bincount(np.concatenate(rows), minlength=n) if n else np.zeros(0)
    return int((cc * (cc + 1) // 2).sum())


class CholeskyPlan:
    """Ordering + structure of L + elimination schedule.

    L is stored row-major over the *permuted* matrix: row i holds its off-diagonal
    columns (ascending) followed by the diagonal.  ``e`` indexes that entry array.

    Arrays (all int32 unless noted)
    -------------------------------
    perm[n]                 new -> old dof
    row_ptr[n+1]            entries of row i: row_ptr[i] .. row_ptr[i+1]-1 (last = diagonal)
    ent_col[nnzL]           column of entry e
    pair_ptr[nnzL+1], pair_a, pair_b
                            L_e = (A_e - sum_p L[pair_a[p]] * L[pair_b[p]]) (/ L_jj)
    a_ent[nnzL]             index into the CSR pattern of A (permuted lower part) or -1 (fill)
    col_ptr[n+1], col_ent, col_row
                            strictly-lower entries of column j (for the L^T solve)
    """

    def __init__(self, indptr, indices, n, ordering="auto"):
        indptr = np.asarray(indptr); indices = np.asarray(indices)
        if isinstance(ordering, str) and ordering == "auto":
            # pick whichever candidate ordering needs the fewest multiply-adds
            best = None
            for cand in ("mmd", "md"):
                perm = _ordering(indptr, indices, n, cand)
                cost = _pair_count(indptr, indices, n, perm)
                if best is None or cost < best[0]:
                    best = (cost, perm)
            ordering = best[1]
        if isinstance(ordering, str):
            perm = _ordering(indptr, indices, n, ordering)
        else:
            perm = np.asarray(ordering, np.int64)
        self._build(indptr, indices, n, perm)

    def _build(self, indptr, indices, n, perm):
        self.n = n
        self.perm = perm
        iperm = np.empty(n, np.int64); iperm[perm] = np.arange(n)
        self.iperm = iperm

        # permuted strictly-lower row patterns + where each comes from in the CSR of A
        lower_rows, lower_src, diag_src = [], [], np.empty(n, np.int64)
        for i in range(n):
            o = perm[i]
            sl = slice(indptr[o], indptr[o + 1])
            cols = iperm[indices[sl]]
            src = np.arange(indptr[o], indptr[o + 1])
            dmask = cols == i
            if dmask.sum() != 1:
                raise ValueError("pattern must contain the diagonal exactly once")
            diag_src[i] = src[dmask][0]
            lm = cols < i
            o2 = np.argsort(cols[lm])
            lower_rows.append(cols[lm][o2]); lower_src.append(src[lm][o2])
        parent = _etree(n, lower_rows)
        self.parent = parent
        Lrows = _row_structures(n, lower_rows, parent)

        counts = np.array([len(r) + 1 for r in Lrows])
        row_ptr = np.zeros(n + 1, np.int64); np.cumsum(counts, out=row_ptr[1:])
        nnzL = int(row_ptr[-1])
        ent_col = np.empty(nnzL, np.int64)
        a_ent = np.full(nnzL, -1, np.int64)
        for i in range(n):
            s, e = row_ptr[i], row_ptr[i + 1]
            ent_col[s:e - 1] = Lrows[i]; ent_col[e - 1] = i
            pos = np.searchsorted(Lrows[i], lower_rows[i])
            a_ent[s + pos] = lower_src[i]
            a_ent[e - 1] = diag_src[i]
        self.nnzL = nnzL
        self.row_ptr = row_ptr.astype(np.int32)
        self.ent_col = ent_col.astype(np.int32)
        self.a_ent = a_ent.astype(np.int32)

        # elimination schedule
        pa, pb, pptr = [], [], np.zeros(nnzL + 1, np.int64)
        for i in range(n):
            s, e = row_ptr[i], row_ptr[i + 1]
            ci = Lrows[i]
            for t in range(s, e):
                j = ent_col[t]
                if j == i:
                    ia = np.arange(s, e - 1); ib = ia
                else:
                    cj = Lrows[j]
                    _, xi, xj = np.intersect1d(ci, cj, assume_unique=True, return_indices=True)
                    ia = s + xi; ib = row_ptr[j] + xj
                pa.append(ia); pb.append(ib)
                pptr[t + 1] = pptr[t] + len(ia)
        self.pair_ptr = pptr.astype(np.int32)
        self.pair_a = np.concatenate(pa).astype(np.int32) if pa else np.zeros(0, np.int32)
        self.pair_b = np.concatenate(pb).astype(np.int32) if pb else np.zeros(0, np.int32)
        self.npairs = int(pptr[-1])

        # column view of the strictly-lower part
        ent_row = np.repeat(np.arange(n), counts)
        off = ent_col != ent_row
        ce = np.nonzero(off)[0]
        order = np.lexsort((ent_row[ce], ent_col[ce]))
        ce = ce[order]
        self.col_ent = ce.astype(np.int32)
        self.col_row = ent_row[ce].astype(np.int32)
        cc = np.bincount(ent_col[ce], minlength=n)
        col_ptr = np.zeros(n + 1, np.int64); np.cumsum(cc, out=col_ptr[1:])
        self.col_ptr = col_ptr.astype(np.int32)

    # Cholesky flops as counted in SURVEY 8(d): sum_j c_j^2 with c_j = column count incl. diagonal
    def flops_factor(self):
        cc = np.diff(self.col_ptr).astype(np.int64) + 1
        return int((cc * cc).sum())

    def entry_table(self, c0_csr, W_csr):
        """Re-index a sparse-affine value map ``vals = c0 + W @ x`` (defined on the CSR
        pattern of A) onto the entries of L: returns (c0[nnzL], ptr[nnzL+1], idx, w)."""
        W_csr = sp.csr_matrix(W_csr)
        has = self.a_ent >= 0
        c0 = np.zeros(self.nnzL); c0[has] = np.asarray(c0_csr)[self.a_ent[has]]
        cnt = np.zeros(self.nnzL, np.int64)
        cnt[has] = np.diff(W_csr.indptr)[self.a_ent[has]]
        ptr = np.zeros(self.nnzL + 1, np.int64); np.cumsum(cnt, out=ptr[1:])
        idx = np.empty(ptr[-1], np.int32); w = np.empty(ptr[-1])
        for e in np.nonzero(has)[0]:
            a = self.a_ent[e]
            sl = slice(W_csr.indptr[a], W_csr.indptr[a + 1])
            idx[ptr[e]:ptr[e + 1]] = W_csr.indices[sl]
            w[ptr[e]:ptr[e + 1]] = W_csr.data[sl]
        return c0, ptr.astype(np.int32), idx, w
