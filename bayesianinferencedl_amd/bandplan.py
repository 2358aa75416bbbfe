"""One-time plan of the FOM's *frontal band sweep* (csrc/fom_band.hip), the throughput path of the batched sparse SPD solve
behind Fin.forward (fom/forward_solve.py:270-291) on the lattice fin of fem.py.

Why a second symbolic phase next to symbolic.py: the schedule interpreter keeps every sample's factor in HBM and fetches one
operand per multiply-add (the interpreter's cost is its instruction chain and ~70 GB of traffic per 100k samples).  The fin
is a long thin domain -- a post m+1 nodes wide and eight fins m/4+1 nodes wide -- so with the natural orderings

    fin f (f = 0..7):  tip -> root, column by column          (half bandwidth  Bf = m/4 + 1)
    post:              row by row from the root y = 0 upwards  (half bandwidth  Bp = m + 1)

the active part of the factorisation (the *front*) is a (B+1) x (B+1) triangle per sample: 105 doubles at m = 12, which
lives in REGISTERS (lane = sample, slots renamed cyclically so that every index is a compile-time constant).  Only the
finished columns of L leave the chip, once, and come back once for the backward substitution.

The one irregularity: eliminating a fin couples its q+1 interface nodes on the post's side wall to each other, and those
sit in consecutive post ROWS, i.e. B, 2B, 3B ... positions apart.  Couplings further apart than B do not fit the band; the
nodes at their far end are carried as *extras*: front members outside the register window (state in LDS), allocated when the
near node enters the window and folded into the window when their own turn comes.  At m = 12 at most 4 extras are alive.

This module computes, for one mesh: the band order, the (up to) three matrix entries every node brings into the window
(diagonal, previous node, node B positions back), where the fin Schur complements go, the extras' life cycles, and checks
by a symbolic elimination that band + extras really cover the factor.  `replay` executes the plan in NumPy with the
device kernel's semantics (tests)."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

_LEFT_YB = (0.75, 1.75, 2.75, 3.75)
_RIGHT_YB = (3.75, 2.75, 1.75, 0.75)


class BandPlanError(ValueError):
    """The mesh / operator does not have the structure the band sweep assumes (callers fall back to the interpreter)."""


class Segment:
    """One band sweep: `npiv` pivots (elimination indices e0 .. e0+npiv-1) over `ntot` segment nodes (the trailing
    ntot - npiv nodes are not eliminated here: the fin's interface nodes), window of NS = B + 1 slots."""

    def __init__(self, g0, e0, npiv, ntot, NS, L0):
        self.g0, self.e0, self.npiv, self.ntot, self.NS, self.L0 = g0, e0, npiv, ntot, NS, L0


class BandPlan:
    def __init__(self, mesh, indptr, indices, nonzero):
        """mesh: fem.FinMesh; (indptr, indices): shared CSR pattern of A; nonzero[e]: entry e of the pattern can be non-zero
        (False = zero in every operator table: the hypotenuse entries of the right-triangle lattice)."""
        m = mesh.m
        self.m, self.n = m, mesh.n
        n = mesh.n
        q, x0, x1, xe, ytop = m // 4, 5 * m // 2, 7 * m // 2, 6 * m, 4 * m
        self.q, self.W = q, m + 1
        self.NSF, self.NSP = q + 2, m + 2                # window slots = half bandwidth + 1
        li, lj = mesh.lattice[:, 0], mesh.lattice[:, 1]
        node_at = {(int(i), int(j)): k for k, (i, j) in enumerate(zip(li, lj))}

        # ---- band order ---------------------------------------------------------------------------------------------
        fins = []                                         # (own nodes in elimination order, interface nodes bottom -> top)
        for side, ybs in ((0, _LEFT_YB), (1, _RIGHT_YB)):
            for yb in ybs:
                j0 = int(round(yb * m))
                cols = range(0, x0) if side == 0 else range(xe, x1, -1)
                own = [node_at[(i, j)] for i in cols for j in range(j0, j0 + q + 1)]
                iface = [node_at[(x0 if side == 0 else x1, j)] for j in range(j0, j0 + q + 1)]
                fins.append((own, iface))
        post = [node_at[(i, j)] for j in range(0, ytop + 1) for i in range(x0, x1 + 1)]
        self.nfins = len(fins)
        self.npf = len(fins[0][0])
        self.npost = len(post)
        order = [v for own, _ in fins for v in own] + post
        if len(order) != n or len(set(order)) != n:
            raise BandPlanError("band order does not cover the mesh")
        self.perm = np.asarray(order, np.int64)           # elimination index -> dof
        self.iperm = np.empty(n, np.int64); self.iperm[self.perm] = np.arange(n)

        # ---- segments and the global numbering of segment nodes (fins carry their interface nodes as trailing nodes) --
        segs, gnode = [], []                              # gnode[g] = dof of segment node g
        g0 = e0 = L0 = 0
        for own, iface in fins:
            segs.append(Segment(g0, e0, len(own), len(own) + len(iface), self.NSF, L0))
            gnode += own + iface
            g0 += len(own) + len(iface); e0 += len(own); L0 += len(own) * self.NSF
        self.post_seg = Segment(g0, e0, len(post), len(post), self.NSP, L0)
        gnode += post
        L0 += len(post) * self.NSP
        self.fin_segs, self.gnode = segs, np.asarray(gnode, np.int64)
        self.G = len(gnode)
        self.nL = L0                                      # doubles of regular L per sample (B off-diagonals + 1/L_jj per pivot)

        # ---- matrix entries each segment node brings: (diag, previous node, node B back) as CSR indices (-1 = none) ----
        A = sp.csr_matrix((np.arange(1, len(indices) + 1), indices, indptr), shape=(n, n))   # value = csr index + 1
        nz = np.asarray(nonzero, bool)

        def csr_of(a, b):
            k = A[a, b] - 1
            return int(k) if k >= 0 and nz[k] else -1
        self.ab_csr = np.full((self.G, 3), -1, np.int64)
        covered = np.zeros(len(indices), bool)
        for seg, (own, iface) in zip(segs, fins):
            nodes = own + iface
            B = seg.NS - 1
            for t, v in enumerate(nodes):
                g = seg.g0 + t
                is_if = t >= len(own)
                for k, back in enumerate((0, 1, B)):
                    if t - back < 0 or (is_if and back != B):
                        continue                          # interface nodes: only their coupling INTO the fin belongs to the fin sweep
                    if is_if and t - back >= len(own):
                        continue
                    c = csr_of(v, nodes[t - back])
                    if c >= 0:
                        self.ab_csr[g, k] = c
                        covered[c] = True
                        covered[A[nodes[t - back], v] - 1] = True
        seg = self.post_seg
        B = seg.NS - 1
        for t, v in enumerate(post):
            for k, back in enumerate((0, 1, B)):
                if t - back < 0:
                    continue
                c = csr_of(v, post[t - back])
                if c >= 0:
                    self.ab_csr[seg.g0 + t, k] = c
                    covered[c] = True
                    covered[A[post[t - back], v] - 1] = True
        if not np.all(covered | ~nz):
            raise BandPlanError("the operator has entries outside the band (not a lattice fin?)")

        # ---- where the fins' Schur complements go ---------------------------------------------------------------------
        # fin f leaves S (q+1 x q+1, lower triangle) on its interface nodes u_0..u_q.  S[t][t] adds to u_t's diagonal entry,
        # S[t][t-1] to its entry B back in the post (u_t and u_{t-1} are one post row apart), S[t][s], t - s >= 2, is a
        # long-range coupling: its own value slot ("special") + an extra.
        ppos = {v: t for t, v in enumerate(post)}
        self.n_special = 0
        self.schur_target = []                            # per fin: [(t, s, AB offset)]
        long_range = []                                   # (near post pos, far post pos, AB offset)
        nAB_reg = 3 * self.G
        for own, iface in fins:
            tg = []
            for t in range(q + 1):
                for s in range(t + 1):
                    pt, ps = ppos[iface[t]], ppos[iface[s]]
                    if t == s:
                        tg.append((t, s, 3 * (self.post_seg.g0 + pt) + 0))
                    elif pt - ps == B:
                        tg.append((t, s, 3 * (self.post_seg.g0 + pt) + 2))
                    elif pt - ps == 1:
                        tg.append((t, s, 3 * (self.post_seg.g0 + pt) + 1))
                    elif pt - ps > B:
                        off = nAB_reg + self.n_special
                        self.n_special += 1
                        tg.append((t, s, off))
                        long_range.append((ps, pt, off))
                    else:
                        raise BandPlanError("interface nodes in an unexpected relative position")
            self.schur_target.append(tg)
        self.nAB = nAB_reg + self.n_special

        # ---- extras: symbolic elimination of the post (band pattern + fin cliques) ---------------------------------------
        npost = self.npost
        adj = [set() for _ in range(npost)]
        for t in range(npost):
            for k, back in enumerate((1, B)):
                if t - back >= 0 and self.ab_csr[seg.g0 + t, k + 1] >= 0:
                    adj[t].add(t - back); adj[t - back].add(t)
        for own, iface in fins:
            ps = [ppos[v] for v in iface]
            for a in ps:
                for b_ in ps:
                    if a != b_:
                        adj[a].add(b_)
        struct = []                                       # struct[p] = sorted later neighbours at elimination time
        adjw = [set(a) for a in adj]
        for p in range(npost):
            later = sorted(v for v in adjw[p] if v > p)
            struct.append(later)
            for a in later:
                adjw[a] |= set(later) - {a}
                adjw[a].discard(p)
        first_ref, last_ref = {}, {}
        for p in range(npost):
            for v in struct[p]:
                if v - p > B:
                    first_ref.setdefault(v, p); last_ref[v] = p
        # an extra x is allocated when its first long-range coupling is SET (entering of the near node = end of step
        # near - NS, or the prologue) and folded into the window at the end of step x - NS
        NS = seg.NS
        alloc = {}
        for near, far, off in long_range:
            alloc[far] = min(alloc.get(far, near), near)
        for x in first_ref:
            if x not in alloc or first_ref[x] < alloc[x]:
                raise BandPlanError("front member outside the band without a long-range coupling that explains it")
            if last_ref[x] != x - NS or any(x not in struct[p] for p in range(first_ref[x], x - NS + 1)):
                raise BandPlanError("extra is not a contiguous front member")
        for x in alloc:
            if x not in first_ref:
                raise BandPlanError("long-range coupling without a front member")
        events = sorted(alloc.items(), key=lambda kv: kv[1])
        slot_free_at, slot_of = [], {}
        for x, near in events:
            start, end = near - NS, x - NS                # allocated over steps (start, end]
            for s_, free_at in enumerate(slot_free_at):
                if free_at <= start:
                    slot_of[x] = s_; slot_free_at[s_] = end
                    break
            else:
                slot_of[x] = len(slot_free_at); slot_free_at.append(end)
        self.NX = len(slot_free_at)
        self.extra_slot = slot_of                         # post position -> extra slot
        # per-pivot tables of the post sweep
        self.act = np.zeros(npost, np.int32)              # bit s: extra slot s is updated at this pivot
        self.lx_ptr = np.zeros(npost + 1, np.int32)       # extras' L values are stored in pivot order, slots ascending
        for p in range(npost):
            for v in struct[p]:
                if v - p > B:
                    self.act[p] |= 1 << slot_of[v]
            self.lx_ptr[p + 1] = self.lx_ptr[p] + bin(int(self.act[p])).count("1")
        self.nLx = int(self.lx_ptr[-1])
        self.ent_extra = np.zeros(npost, np.int32)        # node p was an extra before it entered the window: slot + 1
        for x, s_ in slot_of.items():
            self.ent_extra[x] = s_ + 1
        # couplings set when a node enters: (extra slot, AB offset), CSR by entering post position
        cp = [[] for _ in range(npost)]
        for near, far, off in long_range:
            cp[near].append((slot_of[far], off))
        self.ecp_ptr = np.zeros(npost + 1, np.int32)
        self.ecp_slot, self.ecp_off = [], []
        for t in range(npost):
            for s_, off in sorted(cp[t]):
                self.ecp_slot.append(s_); self.ecp_off.append(off)
            self.ecp_ptr[t + 1] = len(self.ecp_slot)
        self.ecp_slot = np.asarray(self.ecp_slot, np.int32); self.ecp_off = np.asarray(self.ecp_off, np.int32)
        # regular band must hold the rest of the structure
        for p in range(npost):
            if any(v - p <= 0 for v in struct[p]):
                raise BandPlanError("symbolic elimination produced a non-causal structure")
        self.post_struct = struct
        # interface nodes of each fin as elimination indices (backward substitution of the fins reads their w)
        self.iface_elim = np.asarray([[self.iperm[v] for v in iface] for _, iface in fins], np.int64)
        self.fins = fins
        self.post = post

    # ------------------------------------------------------------------------------------------------------------------
    def per_sample_doubles(self):
        """Values per sample in the kernel's global workspace: AB | L | Lx | y (then w)."""
        return self.nAB + self.nL + self.nLx + self.n

    def ab_table(self, c0_csr, W_csr):
        """Sparse-affine map of the pre-pass: AB[e] = c0[e] + sum_t w_t x[idx_t] for every regular value slot (special slots
        start at zero): returns (c0 [nAB], ptr [nAB+1], idx, w)."""
        W_csr = sp.csr_matrix(W_csr)
        flat = self.ab_csr.reshape(-1)
        has = flat >= 0
        c0 = np.zeros(self.nAB)
        c0[:len(flat)][has] = np.asarray(c0_csr)[flat[has]]
        cnt = np.zeros(self.nAB, np.int64)
        cnt[:len(flat)][has] = np.diff(W_csr.indptr)[flat[has]]
        ptr = np.zeros(self.nAB + 1, np.int64); np.cumsum(cnt, out=ptr[1:])
        idx = np.empty(ptr[-1], np.int32); w = np.empty(ptr[-1])
        for e in np.nonzero(has)[0]:
            a = flat[e]
            sl = slice(W_csr.indptr[a], W_csr.indptr[a + 1])
            idx[ptr[e]:ptr[e + 1]] = W_csr.indices[sl]; w[ptr[e]:ptr[e + 1]] = W_csr.data[sl]
        return c0, ptr.astype(np.int32), idx, w

    def compact_slots(self, c0, ptr, idx, w):
        """Physical value slots: logical slots with the SAME affine record (c0, indices, weights -- exact equality) share one
        physical slot, except the slots the fins write to (Schur targets, special slots), which stay private.  With the five /
        nine fin conductivities as parameters 4887 logical slots become ~370 physical ones: the pre-pass writes 3 KB per sample
        instead of 39 KB and the sweep's value-slot reads hit L2 / MALL instead of HBM.  (A nodal field as parameter vector
        has almost no duplicates: nothing is lost there.)
        -> (abmap [nAB] logical -> physical, c0p, ptrp, idxp, wp describing the physical slots)."""
        private = set(off for tg in self.schur_target for _, _, off in tg)
        key_to_phys, abmap = {}, np.empty(self.nAB, np.int64)
        recs = []
        for e in range(self.nAB):
            rec = (float(c0[e]), tuple(int(i) for i in idx[ptr[e]:ptr[e + 1]]), tuple(float(v) for v in w[ptr[e]:ptr[e + 1]]))
            if e in private or e >= 3 * self.G:
                abmap[e] = len(recs); recs.append(rec)
                continue
            if rec not in key_to_phys:
                key_to_phys[rec] = len(recs); recs.append(rec)
            abmap[e] = key_to_phys[rec]
        c0p = np.array([r_[0] for r_ in recs])
        cnt = np.array([len(r_[1]) for r_ in recs], np.int64)
        ptrp = np.zeros(len(recs) + 1, np.int64); np.cumsum(cnt, out=ptrp[1:])
        idxp = np.array([i for r_ in recs for i in r_[1]], np.int32)
        wp = np.array([v for r_ in recs for v in r_[2]], np.float64)
        return abmap.astype(np.int32), c0p, ptrp.astype(np.int32), idxp, wp

    # ------------------------------------------------------------------------------------------------------------------
    def replay(self, AB, F, qoi_only=None):
        """NumPy execution of the plan with the device kernel's data flow (cyclic window slots, extras, fin Schur targets,
        stored L, backward substitution).  AB: assembled value slots [nAB] of one sample; F: load vector over dofs.
        Returns w over dofs.  Test infrastructure for the tables -- the product path is csrc/fom_band.hip.
        qoi_only = (FgQ, row_fin, qptr, qidx, qw) (engine.FomEngine.qoi_only_tables): the kernel's QoI-only form instead -- the
        fins' sweeps carry their observation row's weights as right-hand side, store nothing and have no backward sweep --
        and the observables [n_obs] are returned."""
        AB = np.array(AB, dtype=np.float64, copy=True)
        n = self.n
        L = np.zeros(self.nL); Lx = np.zeros(max(self.nLx, 1)); y = np.zeros(n)
        Fe = np.asarray(F, float)[self.perm]
        NX = max(self.NX, 1)

        def sweep(seg, is_post):
            NS = seg.NS; B = NS - 1
            win = np.zeros((NS, NS)); yw = np.zeros(NS)
            X = np.zeros((NX, NS)); XD = np.zeros(NX); XX = np.zeros((NX, NX)); XY = np.zeros(NX)

            def enter(t):
                u = t % NS
                win[u, :] = 0.0; win[:, u] = 0.0; yw[u] = 0.0
                X[:, u] = 0.0
                if is_post and self.ent_extra[t]:
                    s_ = self.ent_extra[t] - 1
                    for v in range(NS):
                        if v != u:
                            win[u, v] = win[v, u] = X[s_, v]
                    win[u, u] = XD[s_]; yw[u] = XY[s_]
                    for o in range(NX):
                        if o != s_:
                            X[o, u] = XX[max(o, s_), min(o, s_)]
                            XX[max(o, s_), min(o, s_)] = 0.0
                    X[s_, :] = 0.0; XD[s_] = 0.0; XY[s_] = 0.0
                g = seg.g0 + t
                win[u, u] += AB[3 * g]
                if t >= 1:
                    v = (t - 1) % NS
                    win[u, v] += AB[3 * g + 1]; win[v, u] = win[u, v]
                if t >= B:
                    v = (t - B) % NS
                    win[u, v] += AB[3 * g + 2]; win[v, u] = win[u, v]
                if qoi_only is not None:
                    yw[u] += qoi_only[0][g]                # FgQ: the row's weights on a fin's segment nodes, the load on the post's
                elif is_post or t < seg.npiv:
                    yw[u] += Fe[seg.e0 + t] if (is_post or t < seg.npiv) else 0.0
                if is_post:
                    for c in range(self.ecp_ptr[t], self.ecp_ptr[t + 1]):
                        X[self.ecp_slot[c], u] += AB[self.ecp_off[c]]
            for t in range(min(NS, seg.ntot)):
                enter(t)
            for p in range(seg.npiv):
                u = p % NS
                d = win[u, u]
                if not d > 0:
                    raise np.linalg.LinAlgError("not positive definite")
                inv = 1.0 / np.sqrt(d)
                l = np.zeros(NS)
                for s_ in range(1, NS):
                    l[s_] = win[(u + s_) % NS, u] * inv
                base = seg.L0 + p * NS
                yp = yw[u] * inv
                if is_post or qoi_only is None:            # (QoI-only: nothing of a fin's factor or y is kept)
                    L[base:base + B] = l[1:]; L[base + B] = inv
                    y[seg.e0 + p] = yp
                for s_ in range(1, NS):
                    a = (u + s_) % NS
                    yw[a] -= l[s_] * yp
                    for t_ in range(1, s_ + 1):
                        b_ = (u + t_) % NS
                        win[a, b_] -= l[s_] * l[t_]; win[b_, a] = win[a, b_]
                if is_post and self.act[p]:
                    le = np.zeros(NX); k = self.lx_ptr[p]
                    for s_ in range(NX):
                        if self.act[p] >> s_ & 1:
                            le[s_] = X[s_, u] * inv
                            Lx[k] = le[s_]; k += 1
                            XY[s_] -= le[s_] * yp; XD[s_] -= le[s_] ** 2
                            for t_ in range(1, NS):
                                X[s_, (u + t_) % NS] -= le[s_] * l[t_]
                    for a in range(NX):
                        for b_ in range(a):
                            XX[a, b_] -= le[a] * le[b_]
                if p + NS < seg.ntot:
                    enter(p + NS)
            return win, yw

        gfun = []                                         # per fin: the functional's weights on the interface values
        for f, seg in enumerate(self.fin_segs):
            win, yw = sweep(seg, False)
            for t, s_, off in self.schur_target[f]:
                a, b_ = (seg.npiv + t) % seg.NS, (seg.npiv + s_) % seg.NS
                AB[off] += win[a, b_]
            gfun.append([yw[(seg.npiv + t) % seg.NS] for t in range(seg.ntot - seg.npiv)])
        sweep(self.post_seg, True)

        # backward substitution, post first (reverse elimination order), then the fins
        w = y.copy()

        def bsweep(seg, is_post, f=None):
            NS = seg.NS; B = NS - 1
            ww = np.zeros(NS); Wx = np.zeros(NX)
            if not is_post:
                for t in range(seg.ntot - seg.npiv):
                    ww[(seg.npiv + t) % NS] = w[self.iface_elim[f][t]]
            for p in range(seg.npiv - 1, -1, -1):
                u = p % NS
                base = seg.L0 + p * NS
                acc = w[seg.e0 + p]
                for s_ in range(1, NS):
                    if p + s_ < seg.ntot:
                        acc -= L[base + s_ - 1] * ww[(u + s_) % NS]
                if is_post and self.act[p]:
                    k = self.lx_ptr[p]
                    for s_ in range(NX):
                        if self.act[p] >> s_ & 1:
                            acc -= Lx[k] * Wx[s_]; k += 1
                acc *= L[base + B]
                w[seg.e0 + p] = acc; ww[u] = acc
                if is_post and self.ent_extra[p]:
                    Wx[self.ent_extra[p] - 1] = acc
        bsweep(self.post_seg, True)
        if qoi_only is not None:
            _, row_fin, qptr, qidx, qw = qoi_only
            q = np.zeros(len(row_fin))
            for o, f in enumerate(row_fin):
                q[o] = sum(qw[t] * w[qidx[t]] for t in range(qptr[o], qptr[o + 1]))
                if f >= 0:
                    q[o] += sum(gfun[f][t] * w[self.iface_elim[f][t]] for t in range(len(gfun[f])))
            return q
        for f, seg in enumerate(self.fin_segs):
            bsweep(seg, False, f)
        out = np.empty(n); out[self.perm] = w
        return out


def nonzero_entries(c0_csr, W_csr, extra_tables=()):
    """Entries of the CSR pattern that are non-zero in at least one operator table."""
    W = sp.csr_matrix(W_csr)
    nzm = (np.asarray(c0_csr) != 0) | (np.diff(W.indptr) > 0) & (np.asarray(abs(W).sum(axis=1)).ravel() != 0)
    for t in extra_tables:
        nzm |= np.asarray(t) != 0
    return nzm
