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

import os

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


def minimum_fill(indptr, indices, n, seed=0):
    """Greedy minimum-fill (minimum local deficiency) ordering: eliminate the vertex whose elimination adds the fewest
    edges; ties by degree, then by a seeded random key (different seeds give different, equally valid orderings and the
    caller keeps the cheapest).  On the fin lattices it needs 6-8 % fewer multiply-adds than multiple-minimum-degree."""
    rng = np.random.default_rng(seed)
    adj = [set(indices[indptr[i]:indptr[i + 1]].tolist()) - {i} for i in range(n)]

    def deficiency(v):
        nb = list(adj[v])
        f = 0
        for a in range(len(nb)):
            sa = adj[nb[a]]
            for b in range(a + 1, len(nb)):
                if nb[b] not in sa:
                    f += 1
        return f

    import heapq
    noise = rng.random(n)
    key = lambda v: (deficiency(v), len(adj[v]), noise[v])
    cur = [key(v) for v in range(n)]
    heap = [(cur[v], v) for v in range(n)]
    heapq.heapify(heap)
    done = np.zeros(n, bool)
    order = []
    while heap:
        k, v = heapq.heappop(heap)
        if done[v] or k != cur[v]:
            continue
        done[v] = True
        order.append(v)
        nb = list(adj[v])
        touched = set(nb)
        for a in nb:
            adj[a].discard(v)
        for i, a in enumerate(nb):
            for b in nb[i + 1:]:
                if b not in adj[a]:
                    adj[a].add(b); adj[b].add(a)
        for a in nb:
            touched |= adj[a]
        adj[v] = set()
        for u in touched:
            if not done[u]:
                cur[u] = key(u)
                heapq.heappush(heap, (cur[u], u))
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
    if name.startswith("mf"):
        return minimum_fill(indptr, indices, n, seed=int(name[2:] or 0))
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
            for cand in ("mmd", "md") + (("mf0", "mf1", "mf2", "mf3") if n <= 2500 else ("mf0", "mf1")):
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

    def op_streams(self, cache_slots, rhs_perm=None, fwd_chunk=None, fused_asm=None):
        """Op streams of the device interpreter (cached per cache size, chunk size, load pattern and -- when the
        affine assembly is fused into the stream -- value map: fused_asm = entry_table(...) output)."""
        fwd_chunk = CHUNK if fwd_chunk is None else fwd_chunk
        nz = None if rhs_perm is None else (np.asarray(rhs_perm) != 0.0)
        akey = None if fused_asm is None else tuple(np.asarray(a).tobytes() for a in fused_asm)
        key = (cache_slots, fwd_chunk, None if nz is None else nz.tobytes(), akey)
        cache = self.__dict__.setdefault("_streams", {})
        if key not in cache:
            cache[key] = build_op_streams(self, cache_slots, nz, fwd_chunk, fused_asm)
        return cache[key]

    def level_sets(self):
        """Dependency levels of the row recurrence for the small-batch kernel (one workgroup per sample, lane = row):
        forward -- row i needs every row in its structure; backward (L^T w = y) -- row i needs every row below it in
        column i.  Returns (lev_ptr_f, lev_rows_f, lev_ptr_b, lev_rows_b); rows of a level sorted by decreasing work."""
        if getattr(self, "_levels", None) is None:
            n = self.n
            rp, ec, pp = self.row_ptr.astype(np.int64), self.ent_col.astype(np.int64), self.pair_ptr.astype(np.int64)
            cp, cr = self.col_ptr.astype(np.int64), self.col_row.astype(np.int64)
            lf = np.zeros(n, np.int64)
            for i in range(n):
                cols = ec[rp[i]:rp[i + 1] - 1]
                lf[i] = 0 if len(cols) == 0 else lf[cols].max() + 1
            lb = np.zeros(n, np.int64)
            for i in range(n - 1, -1, -1):
                rows = cr[cp[i]:cp[i + 1]]
                lb[i] = 0 if len(rows) == 0 else lb[rows].max() + 1
            cost_f = (pp[rp[1:]] - pp[rp[:-1]]) + 2 * np.diff(rp)
            cost_b = np.diff(cp) + 1

            def pack(lev, cost):
                order = np.lexsort((-cost, lev))
                ptr = np.zeros(lev.max() + 2, np.int64)
                np.cumsum(np.bincount(lev, minlength=lev.max() + 1), out=ptr[1:])
                return ptr.astype(np.int32), order.astype(np.int32)
            self._levels = pack(lf, cost_f) + pack(lb, cost_b)
        return self._levels

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


# =========================================================================================
# Op streams for the device "schedule interpreter" (csrc/fom_kernels.hip::fom_vm_kernel)
# =========================================================================================
# The factorisation / substitution schedule is flattened into a stream of fixed-size ops that a
# wave executes for its 64 samples.  The wave fetches the ops' global operands one CHUNK (8 ops)
# ahead of executing them, so a value stored while chunk c executes may be fetched no earlier than
# for chunk c+2 (both streams).  This module orders the rows (any topological order of the elimination tree is a
# valid elimination order) and pads with NOPs so that rule always holds; tests/test_host_and_abi.py
# replays the stream with exactly that prefetch semantics.
#
# Per-sample value space G (doubles):  [0, nnzL) entries of L (initially the assembled A values),
# [nnzL, nnzL+n) 1/L_ii,  [nnzL+n, nnzL+2n) y then w.
CHUNK = 8
# forward kinds.  Kind 0 is the only common one and is branch-free on the device:
#   acc -= rc[b] * G[a]        with two constant LDS slots after the row cache: NEG1 (-1.0) and ZERO (0.0),
# so "acc = A_e" is an FMA against NEG1 (every FIN* op leaves acc = 0) and padding is an FMA against ZERO.
OP_FMA, OP_LDX, OP_FMAX, OP_FINOFF, OP_FINDIAG, OP_YSET, OP_FINY = 0, 3, 4, 5, 6, 7, 8
# fused affine assembly (parameter vectors short enough to sit in LDS): the value of A_e is built by the interpreter itself,
#   XFMA acc += imm[d] * x[b]      CADD acc += imm[d]
# instead of being fetched from the pre-pass's output ("acc = A_e" FMA against the NEG1 slot)
OP_XFMA, OP_CADD = 9, 10
# both operands in LDS: acc -= rc[b] * rc[d] (the row cache is a ring: entries of recently finished rows are still in it)
OP_FMALL = 11
# a row entry beyond the LDS row cache is fetched like any other operand: LDX x = G[a]; FMAX acc -= x * G[a]
OPB_NOP, OPB_WFMA, OPB_WSET, OPB_WFIN = 0, 1, 3, 5


class _Emitter:
    def __init__(self, nvals, pad_b=-1, distance=2, chunk=CHUNK):
        self.chunk = chunk
        self.pad_b = pad_b                                   # b field of padding ops (forward: the ZERO slot)
        self.distance = distance                             # chunks between a store and the first fetch of the value
        self.kind, self.a, self.b, self.d = [], [], [], []
        self.store_chunk = np.full(nvals, -10, np.int64)     # chunk in which a value was last stored

    def pos(self):
        return len(self.kind)

    def emit(self, kind, a=-1, b=-1, d=-1, loads=(), stores=()):
        need = 0
        for g in loads:
            need = max(need, (self.store_chunk[g] + self.distance) * self.chunk)
        while self.pos() < need:
            self.kind.append(0); self.a.append(-1); self.b.append(self.pad_b); self.d.append(-1)
        c = self.pos() // self.chunk
        self.kind.append(kind); self.a.append(a); self.b.append(b); self.d.append(d)
        for g in stores:
            self.store_chunk[g] = c

    def arrays(self):
        n = self.pos()
        pad = (-n) % (2 * self.chunk) + 2 * self.chunk          # whole number of chunk pairs + one spare pair
        k = np.asarray(self.kind + [0] * pad, np.int32)
        a = np.asarray(self.a + [-1] * pad, np.int32)
        b = np.asarray(self.b + [self.pad_b] * pad, np.int32)
        d = np.asarray(self.d + [-1] * pad, np.int32)
        return k, a, b, d


def build_op_streams(plan: "CholeskyPlan", cache_slots: int, rhs_nonzero=None, fwd_chunk: int = CHUNK, fused_asm=None):
    """-> dict(fwd=(kind, a, b, d), bwd=(kind, a, b, d), a_list=entries of L that carry an A value, fwd_chunk, imm).
    rhs_nonzero[i] (permuted order): rows whose load F_i may be non-zero (None = all).  fwd_chunk = ops per
    prefetch chunk of the forward stream (8 or 16; the backward stream always uses CHUNK).  fused_asm = (c0, ptr,
    idx, w) from CholeskyPlan.entry_table: the stream then assembles A_e = c0_e + sum_t w_t x[idx_t] itself (XFMA /
    CADD ops with the weights in `imm`), a_list comes back empty and no assembly pre-pass is needed."""
    n, nnzL = plan.n, plan.nnzL
    row_ptr, ent_col = plan.row_ptr.astype(np.int64), plan.ent_col.astype(np.int64)
    pair_ptr = plan.pair_ptr.astype(np.int64)
    IV, YV = nnzL, nnzL + n
    rows_cols = [ent_col[row_ptr[i]:row_ptr[i + 1] - 1] for i in range(n)]
    # children lists by direct dependency: row i needs every row j in its structure
    ndeps = np.array([len(c) for c in rows_cols])
    users = [[] for _ in range(n)]
    for i in range(n):
        for j in rows_cols[i]:
            users[j].append(i)
    # priority = longest remaining chain to the root of the elimination tree
    height = np.zeros(n, np.int64)
    for i in range(n - 1, -1, -1):
        par = plan.parent[i]
        height[i] = 0 if par < 0 else height[par] + (row_ptr[par + 1] - row_ptr[par])
    import heapq
    if os.environ.get("FINROM_FWD_ORDER", "postorder") == "postorder":
        # rows in postorder of the elimination tree (a row right after its subtree: the L_jk it reads were
        # written recently); a row that would need padding yields to the next ready row in postorder
        kids = [[] for _ in range(n)]
        roots = []
        for i in range(n):
            (roots if plan.parent[i] < 0 else kids[plan.parent[i]]).append(i)
        size = np.ones(n, np.int64)
        for i in range(n):
            if plan.parent[i] >= 0:
                size[plan.parent[i]] += size[i]
        po = np.zeros(n, np.int64); cnt = 0
        stack = [(r0, False) for r0 in sorted(roots, key=lambda v: -size[v])][::-1]
        while stack:
            v, seen = stack.pop()
            if seen:
                po[v] = cnt; cnt += 1
                continue
            stack.append((v, True))
            for c in sorted(kids[v], key=lambda u: size[u]):       # pushed small first -> the largest subtree is visited first
                stack.append((c, False))
        height = -po

    # ---------------- forward: factorisation + L y = F -----------------------------------
    NEG1, ZERO = cache_slots, cache_slots + 1
    imm, imm_pos = [], {}

    def imm_index(v):
        # a P1 stiffness on a lattice has a few dozen distinct weights: one small table that stays in the scalar cache
        v = float(v)
        if v not in imm_pos:
            imm_pos[v] = len(imm); imm.append(v)
        return imm_pos[v]
    if fused_asm is not None:
        f_c0, f_ptr, f_idx, f_w = (np.asarray(a) for a in fused_asm)
    fwd_dist = int(os.environ.get("FINROM_FWD_DIST", "2"))
    em = _Emitter(nnzL + 2 * n, pad_b=ZERO, chunk=fwd_chunk, distance=fwd_dist)
    done_chunk = np.zeros(n, np.int64)
    y_nonzero = np.zeros(n, bool)     # rows of L y = F whose solution can be non-zero (F is zero except at the root nodes)
    # The LDS row cache is a RING: a row takes the next len(row) slots, so the entries of the rows finished just before it
    # are still there, and a product L_ik L_jk whose L_jk is among them needs no global operand (OP_FMALL) -- and no
    # producer -> consumer distance either.  In postorder the previous row is usually a child of the current one.
    ring = os.environ.get("FINROM_FWD_RING", "1") != "0"
    slot_of, slot_owner, ring_head = {}, [-1] * cache_slots, 0
    remaining = ndeps.copy()
    eligible = [(-height[i], i) for i in range(n) if remaining[i] == 0]
    heapq.heapify(eligible)
    waiting = []          # (ready_chunk, -height, i): eligible but would need padding right now
    order = []
    while eligible or waiting:
        cur = em.pos() // fwd_chunk
        while waiting and waiting[0][0] <= cur:
            _, nh, i = heapq.heappop(waiting)
            heapq.heappush(eligible, (nh, i))
        if eligible:
            _, i = heapq.heappop(eligible)
            # only rows whose entries have to come from global memory constrain the start (a row still in the ring does not)
            ready = max([done_chunk[j] + fwd_dist for j in rows_cols[i]
                         if not (ring and all(int(e) in slot_of for e in range(row_ptr[j], row_ptr[j + 1] - 1)))], default=0)
            if ready > cur and (eligible or (waiting and waiting[0][0] < ready)):
                heapq.heappush(waiting, (ready, -height[i], i))
                continue
        else:
            _, _, i = heapq.heappop(waiting)
        order.append(i)
        e0, e1 = row_ptr[i], row_ptr[i + 1]
        my_slot = {}                                  # entries of THIS row that sit in the LDS row cache -> slot
        for e in range(e0, e1):
            j = ent_col[e]
            if plan.a_ent[e] >= 0 and fused_asm is not None:
                if f_c0[e] != 0.0:
                    em.emit(OP_CADD, d=imm_index(f_c0[e]))
                for t in range(f_ptr[e], f_ptr[e + 1]):
                    em.emit(OP_XFMA, b=int(f_idx[t]), d=imm_index(f_w[t]))
            elif plan.a_ent[e] >= 0:
                em.emit(OP_FMA, a=e, b=NEG1, loads=(e,))              # acc = 0 - (-1) * A_e
            for q in range(pair_ptr[e], pair_ptr[e + 1]):
                pa, pb = int(plan.pair_a[q]), int(plan.pair_b[q])
                slot = my_slot.get(pa, -1)
                if slot >= 0 and pb in slot_of:
                    em.emit(OP_FMALL, b=slot, d=slot_of[pb])           # L_jk of a recently finished row: still in the ring
                elif slot >= 0:
                    em.emit(OP_FMA, a=pb, b=slot, loads=(pb,))
                else:
                    em.emit(OP_LDX, a=pa, loads=(pa,))
                    em.emit(OP_FMAX, a=pb, loads=(pb,))
            if e == e1 - 1:
                em.emit(OP_FINDIAG, b=i, d=e, stores=(e, IV + i))
            else:
                slot = -1
                if e - e0 < cache_slots:              # (a row longer than the cache keeps its first cache_slots entries)
                    if ring:
                        slot = ring_head; ring_head = (ring_head + 1) % cache_slots
                        if slot_owner[slot] >= 0:
                            slot_of.pop(slot_owner[slot], None)
                        slot_owner[slot] = int(e); slot_of[int(e)] = slot
                    else:
                        slot = int(e - e0)
                    my_slot[int(e)] = slot
                em.emit(OP_FINOFF, a=IV + j, b=slot, d=e, loads=(IV + j,), stores=(e,))
        if rhs_nonzero is None or rhs_nonzero[i]:
            em.emit(OP_YSET, d=i)
            y_nonzero[i] = True
        for e in range(e0, e1 - 1):
            slot = my_slot.get(int(e), -1)
            if not y_nonzero[ent_col[e]]:
                continue          # y_k is structurally zero (no load below row k in the elimination tree): nothing to subtract
            y_nonzero[i] = True
            yk = YV + ent_col[e]
            if slot >= 0:
                em.emit(OP_FMA, a=yk, b=slot, loads=(yk,))
            else:
                em.emit(OP_LDX, a=e, loads=(e,))
                em.emit(OP_FMAX, a=yk, loads=(yk,))
        em.emit(OP_FINY, d=YV + i, stores=(YV + i,))
        done_chunk[i] = (em.pos() - 1) // fwd_chunk
        for u in users[i]:
            remaining[u] -= 1
            if remaining[u] == 0:
                heapq.heappush(eligible, (-height[u], u))
    assert len(order) == n
    fwd = em.arrays()

    # ---------------- backward: L^T w = y (w overwrites y) --------------------------------
    col_ptr = plan.col_ptr.astype(np.int64)
    emb = _Emitter(nnzL + 2 * n, distance=2)     # the substitution kernel fetches one chunk ahead, like the forward interpreter
    ndeps_b = np.diff(col_ptr)
    users_b = [[] for _ in range(n)]
    for i in range(n):
        for c in range(col_ptr[i], col_ptr[i + 1]):
            users_b[plan.col_row[c]].append(i)
    depth = np.zeros(n, np.int64)             # longest chain towards the leaves
    for i in range(n):
        par = plan.parent[i]
        if par >= 0:
            depth[par] = max(depth[par], depth[i] + 1 + (col_ptr[i + 1] - col_ptr[i]))
    if os.environ.get("FINROM_FWD_ORDER", "postorder") == "postorder" and os.environ.get("FINROM_BWD_ORDER", "rpo") == "rpo":
        depth = po.copy()                     # reverse postorder: a row right after its parent's subtree prefix
    done_b = np.zeros(n, np.int64)
    remaining = ndeps_b.copy()
    eligible = [(-depth[i], i) for i in range(n) if remaining[i] == 0]
    heapq.heapify(eligible)
    waiting = []
    cnt = 0
    while eligible or waiting:
        cur = emb.pos() // CHUNK
        while waiting and waiting[0][0] <= cur:
            _, nh, i = heapq.heappop(waiting)
            heapq.heappush(eligible, (nh, i))
        if eligible:
            _, i = heapq.heappop(eligible)
            ready = max([done_b[plan.col_row[c]] + 2 for c in range(col_ptr[i], col_ptr[i + 1])], default=0)
            if ready > cur and (eligible or (waiting and waiting[0][0] < ready)):
                heapq.heappush(waiting, (ready, -depth[i], i))
                continue
        else:
            _, _, i = heapq.heappop(waiting)
        cnt += 1
        emb.emit(OPB_WSET, a=YV + i, loads=(YV + i,))
        for c in range(col_ptr[i], col_ptr[i + 1]):
            le, wr = int(plan.col_ent[c]), YV + int(plan.col_row[c])
            emb.emit(OPB_WFMA, a=le, b=wr, loads=(wr,))
        emb.emit(OPB_WFIN, a=IV + i, d=YV + i, stores=(YV + i,))
        done_b[i] = (emb.pos() - 1) // CHUNK
        for u in users_b[i]:
            remaining[u] -= 1
            if remaining[u] == 0:
                heapq.heappush(eligible, (-depth[u], u))
    assert cnt == n
    bwd = emb.arrays()
    a_list = np.nonzero(plan.a_ent >= 0)[0].astype(np.int32) if fused_asm is None else np.zeros(0, np.int32)
    return {"fwd": fwd, "bwd": bwd, "a_list": a_list, "fwd_chunk": fwd_chunk, "imm": np.asarray(imm, np.float64)}


def build_resolve_stream(plan: "CholeskyPlan", base: int):
    """Op stream (backward-interpreter format: WSET / WFMA / WFIN, both operands from global memory) that
    solves  L L^T v = b  with the STORED factor for a second right-hand side living in the value region
    [base, base + n) (b on entry, v on exit): forward substitution row by row, then backward substitution.
    Used for the adjoint solve of Fin.gradient (fom/forward_solve.py:293-322; A is symmetric, so the adjoint
    operator `_adj_F` is the same factor)."""
    n, nnzL = plan.n, plan.nnzL
    row_ptr, ent_col = plan.row_ptr.astype(np.int64), plan.ent_col.astype(np.int64)
    col_ptr = plan.col_ptr.astype(np.int64)
    IV = nnzL
    em = _Emitter(nnzL + 2 * n + (base - (nnzL + n)) + n, distance=1)
    # forward: rows in elimination order (a valid order; padding covers the chains)
    for i in range(n):
        e0, e1 = row_ptr[i], row_ptr[i + 1]
        em.emit(OPB_WSET, a=base + i, loads=(base + i,))
        for e in range(e0, e1 - 1):
            yk = base + int(ent_col[e])
            em.emit(OPB_WFMA, a=int(e), b=yk, loads=(yk,))
        em.emit(OPB_WFIN, a=IV + i, d=base + i, stores=(base + i,))
    for i in range(n - 1, -1, -1):
        em.emit(OPB_WSET, a=base + i, loads=(base + i,))
        for c in range(col_ptr[i], col_ptr[i + 1]):
            wr = base + int(plan.col_row[c])
            em.emit(OPB_WFMA, a=int(plan.col_ent[c]), b=wr, loads=(wr,))
        em.emit(OPB_WFIN, a=IV + i, d=base + i, stores=(base + i,))
    return em.arrays()


def replay_resolve_stream(plan, stream, Lvals, invd, b_perm, base):
    """NumPy replay of build_resolve_stream (chunk fetched right before it executes)."""
    kind, a, b, d = stream
    n, nnzL = plan.n, plan.nnzL
    G = np.zeros(base + n)
    G[:nnzL] = Lvals; G[nnzL:nnzL + n] = invd; G[base:] = b_perm
    acc = 0.0
    for c in range(len(kind) // CHUNK):
        sl = slice(c * CHUNK, (c + 1) * CHUNK)
        va = G[np.maximum(a[sl], 0)].copy(); vb = G[np.maximum(b[sl], 0)].copy()
        for u in range(CHUNK):
            t = c * CHUNK + u
            if kind[t] == OPB_WSET: acc = va[u]
            elif kind[t] == OPB_WFMA: acc -= va[u] * vb[u]
            elif kind[t] == OPB_WFIN: G[d[t]] = acc * va[u]
    return G[base:].copy()


def replay_op_streams(plan, streams, A_entries, rhs_perm, cache_slots, x=None):
    """NumPy replay of the device interpreter WITH its prefetch semantics (operands of chunk c+1 are
    read before chunk c executes); A_entries[e] = assembled A value of entry e (0 for fill).
    Returns w in the permuted dof order."""
    n, nnzL = plan.n, plan.nnzL
    G = np.zeros(nnzL + 2 * n)
    G[:nnzL] = A_entries
    rc = np.zeros(cache_slots + 2); rc[cache_slots] = -1.0

    def run(stream, backward, CH):
        kind, a, b, d = stream
        nch = len(kind) // CH
        acc = 0.0
        inv = 0.0
        xreg = 0.0

        def fetch(c):
            if c >= nch:
                return None
            sl = slice(c * CH, (c + 1) * CH)
            va = G[np.maximum(a[sl], 0)].copy()
            vb = G[np.maximum(b[sl], 0)].copy() if backward else None
            return va, vb
        ahead = 1                                   # both streams: chunk c+1 is fetched BEFORE chunk c executes
        queue = [fetch(j) for j in range(ahead)]
        for c in range(nch):
            queue.append(fetch(c + ahead))
            cur = queue.pop(0)
            for u in range(CH):
                t = c * CH + u
                k, ld = kind[t], cur[0][u]
                if backward:
                    if k == OPB_WSET: acc = ld
                    elif k == OPB_WFMA: acc -= ld * cur[1][u]
                    elif k == OPB_WFIN: G[d[t]] = acc * ld
                else:
                    if k == OP_FMA: acc -= rc[b[t]] * ld
                    elif k == OP_FMALL: acc -= rc[b[t]] * rc[d[t]]
                    elif k == OP_LDX: xreg = ld
                    elif k == OP_FMAX: acc -= xreg * ld
                    elif k == OP_FINOFF:
                        l = acc * ld; G[d[t]] = l; acc = 0.0
                        if b[t] >= 0: rc[b[t]] = l
                    elif k == OP_FINDIAG:
                        dd = np.sqrt(acc); inv = 1.0 / dd; G[d[t]] = dd; G[nnzL + b[t]] = inv; acc = 0.0
                    elif k == OP_XFMA: acc += streams["imm"][d[t]] * x[b[t]]
                    elif k == OP_CADD: acc += streams["imm"][d[t]]
                    elif k == OP_YSET: acc = rhs_perm[d[t]]
                    elif k == OP_FINY: G[d[t]] = acc * inv; acc = 0.0
    run(streams["fwd"], False, streams.get("fwd_chunk", CHUNK))
    run(streams["bwd"], True, CHUNK)
    return G[nnzL + n:].copy()
