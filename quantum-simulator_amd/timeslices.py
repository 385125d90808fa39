"""Timeslicing and the integer tables the kernels consume.

``slice_gate_indices`` is the reference's ASAP layering (BasePlacer.break_into_timeslices,
Placer.py:14-61) on gate indices.  It keeps the reference's data structures where they decide the
result: the successor collections are Python sets (their iteration order fixes the order of the next
ready list, hence the order of commands inside later slices, hence the row order of every cost
matrix), and a successor of two gates of one slice is queued twice and dropped by the busy test.

``CircuitIR`` flattens a circuit into numpy tables:
  edges      counted two-qubit gates (two qubits, not Measure/Reset; HQA_placer.py:66-67), slice order
  lookahead  per edge, the other qubit of the two-qubit commands touching it in the next three slices
             (HQA_placer.py:227-237); depends on the circuit only
  visits     the (slice, gate, qubit) walk of the gate-ordered cost functions (BFS_placer.py:214-237)
             with, per visit, the previous slice in which the qubit was visited
"""
from __future__ import annotations

import weakref
from typing import List, Sequence, Tuple

import numpy as np

from .circuit import OpType


def slice_gate_indices(gate_q: Sequence[Tuple[int, ...]]) -> List[List[int]]:
    n = len(gate_q)
    pred = [set() for _ in range(n)]
    succ = [set() for _ in range(n)]
    latest = {}
    for g, qs in enumerate(gate_q):
        for q in qs:
            p = latest.get(q)
            if p is not None:
                pred[g].add(p)
                succ[p].add(g)
            latest[q] = g
    ready = [g for g in range(n) if not pred[g]]
    placed = set()
    slices = []
    while ready:
        layer, busy = [], set()
        for g in ready:
            qs = gate_q[g]
            if busy.isdisjoint(qs):
                layer.append(g)
                busy.update(qs)
                placed.add(g)
        upcoming = []
        for g in layer:
            for s in succ[g]:
                if s not in placed and pred[s] <= placed:
                    upcoming.append(s)
        slices.append(layer)
        ready = upcoming
    return slices


def native_slices(Q: int, q0: np.ndarray, q1: np.ndarray):
    """Same result as ``slice_gate_indices`` for circuits of one- and two-qubit gates, computed by
    qp_timeslices (host C++).  Returns (order, slice_ptr): gate indices slice after slice + CSR."""
    import ctypes as C
    from . import _lib
    q0 = np.ascontiguousarray(q0, np.int32)
    q1 = np.ascontiguousarray(q1, np.int32)
    n = len(q0)
    order = np.empty(max(n, 1), np.int32)
    ptr = np.empty(n + 2, np.int32)
    T = C.c_int32(0)
    _lib.check(_lib.lib().qp_timeslices(n, _lib.ptr(q0), _lib.ptr(q1), Q, _lib.ptr(order), _lib.ptr(ptr), C.byref(T)))
    return order[:n], ptr[:T.value + 1].copy()


class CircuitIR:
    """Integer view of one circuit.  Build with ``CircuitIR.from_circuit`` or ``from_gate_tables``."""

    def __init__(self, Q: int, gate_q, gate_mr, slices=None, qubits=None, commands=None, tables=None, trusted=False):
        self.Q = int(Q)
        self.gate_q = gate_q if trusted else [tuple(int(x) for x in g) for g in gate_q]
        self.gate_mr = gate_mr if trusted else [bool(x) for x in gate_mr]
        n = len(self.gate_q)
        small = all(len(g) <= 2 for g in self.gate_q)
        if tables is not None:
            q0, q1 = tables
        elif small:
            q0 = np.fromiter((g[0] for g in self.gate_q), np.int32, n)
            q1 = np.fromiter((g[1] if len(g) == 2 else -1 for g in self.gate_q), np.int32, n)
        if slices is None and small:
            order, ptr = native_slices(self.Q, q0, q1)
            slices = [order[ptr[t]:ptr[t + 1]].tolist() for t in range(len(ptr) - 1)]
        self.slices = slices if slices is not None else slice_gate_indices(self.gate_q)
        self.T = len(self.slices)
        self.qubits = qubits          # circuit.qubits order (objects), or None
        self._commands = commands     # command objects in file order, a callable that makes them, or None
        if small:
            self._build_tables_vectorised(q0, q1)
        else:
            self._build_edges()
            self._build_lookahead()
            self._build_visits()

    def _build_tables_vectorised(self, q0, q1):
        """Edges, look-ahead and visits for circuits of one- and two-qubit gates, with numpy only."""
        T, Q = self.T, self.Q
        order = np.fromiter((g for layer in self.slices for g in layer), np.int64, sum(len(x) for x in self.slices))
        sizes = np.fromiter((len(x) for x in self.slices), np.int64, T)
        gptr = np.concatenate([[0], np.cumsum(sizes)])
        gt = np.repeat(np.arange(T, dtype=np.int64), sizes)          # slice of each ordered gate
        gpos = np.arange(len(order), dtype=np.int64) - gptr[gt]       # position inside its slice
        a, b = q0[order].astype(np.int64), q1[order].astype(np.int64)
        mr = np.asarray(self.gate_mr, bool)[order] if len(order) else np.zeros(0, bool)
        two = b >= 0
        counted = two & ~mr
        # edges
        ce = np.flatnonzero(counted)
        self.nE = len(ce)
        self.edge_u = a[ce].astype(np.uint16)
        self.edge_v = b[ce].astype(np.uint16)
        self.edge_t = gt[ce].astype(np.int32)
        self.slice_edge_ptr = np.concatenate([[0], np.cumsum(np.bincount(gt[ce], minlength=T))]).astype(np.int32)
        self.edge_of_gate = dict(zip(order[ce].tolist(), range(self.nE)))
        rec = np.zeros(self.nE, dtype=np.dtype([("u", "<u2"), ("v", "<u2"), ("t", "<i4")]))
        rec["u"], rec["v"], rec["t"] = self.edge_u, self.edge_v, self.edge_t
        self.edge_records = rec
        # look-ahead: per (slice, qubit) the two-qubit command holding it
        la = np.full((self.nE, 6), -1, np.int16)
        if self.nE and T > 1:
            pos = np.full((T, Q), -1, np.int32)
            first = np.zeros((T, Q), np.int32)
            second = np.zeros((T, Q), np.int32)
            tw = np.flatnonzero(two)
            for x in (a[tw], b[tw]):
                pos[gt[tw], x] = gpos[tw]
                first[gt[tw], x] = a[tw]
                second[gt[tw], x] = b[tw]
            u = self.edge_u.astype(np.int64)
            v = self.edge_v.astype(np.int64)
            for dt in (1, 2, 3):
                tt = self.edge_t.astype(np.int64) + dt
                ok = tt < T
                tc = np.where(ok, tt, 0)
                cand = []
                for x in (u, v):
                    p = np.where(ok, pos[tc, x], -1)
                    f1, f2 = first[tc, x], second[tc, x]
                    cand.append((p, np.where((f1 == u) | (f1 == v), f2, f1)))
                (pu, ou), (pv, ov) = cand
                pv = np.where(pv == pu, -1, pv)
                swap = (pu < 0) | ((pv >= 0) & (pv < pu))
                p1, o1 = np.where(swap, pv, pu), np.where(swap, ov, ou)
                p2, o2 = np.where(swap, pu, pv), np.where(swap, ou, ov)
                la[:, 2 * (dt - 1)] = np.where(p1 >= 0, o1, -1)
                la[:, 2 * (dt - 1) + 1] = np.where(p2 >= 0, o2, -1)
        self.lookahead = la
        # visits: (gate, qubit) in slice order; second qubit of a two-qubit gate right after the first
        nv = len(order) + int(two.sum())
        vidx = np.arange(len(order), dtype=np.int64) + np.concatenate([[0], np.cumsum(two)[:-1]]) if len(order) else np.zeros(0, np.int64)
        vq = np.empty(nv, np.int64)
        vt = np.empty(nv, np.int64)
        vp = np.full(nv, 0xFFFF, np.int64)
        vq[vidx], vt[vidx] = a, gt
        vp[vidx] = np.where(counted, b, 0xFFFF)
        v2 = vidx[two] + 1
        vq[v2], vt[v2] = b[two], gt[two]
        # previous slice in which the same qubit was visited: neighbour in a stable sort by qubit
        srt = np.argsort(vq, kind="stable")
        prev = np.full(nv, -1, np.int64)
        same = vq[srt][1:] == vq[srt][:-1]
        prev[srt[1:][same]] = vt[srt[:-1][same]]
        self.nV = nv
        rec = np.zeros(nv, dtype=np.dtype([("q", "<u2"), ("partner", "<u2"), ("t", "<i4"), ("prev_t", "<i4"), ("pad", "<i4")]))
        rec["q"], rec["partner"], rec["t"], rec["prev_t"] = vq, vp, vt, prev
        self.visit_records = rec
        nvis = np.bincount(gt, weights=1 + two.astype(np.int64), minlength=T).astype(np.int64) if len(order) else np.zeros(T, np.int64)
        self.slice_visit_ptr = np.concatenate([[0], np.cumsum(nvis)]).astype(np.int32)

    # -- construction ---------------------------------------------------------------------------
    @classmethod
    def from_circuit(cls, circuit) -> "CircuitIR":
        qubits = list(circuit.qubits)
        tables = getattr(circuit, "gate_tables", None)
        if tables is not None:
            # this package's Circuit: integer tables straight from the reader, command objects only if
            # somebody asks for them (break_into_timeslices)
            tab, mr = tables()
            nq = (tab >= 0).sum(1)
            rows = tab.tolist()
            gate_q = [tuple(r[:k]) for r, k in zip(rows, nq.tolist())]
            small = tab.shape[1] <= 2 or not (nq > 2).any()
            pair = (tab[:, 0].astype(np.int32), (tab[:, 1] if tab.shape[1] > 1 else np.full(len(tab), -1)).astype(np.int32))
            return cls(len(qubits), gate_q, mr.tolist(), qubits=qubits, commands=weakref.WeakMethod(circuit.get_commands),
                       tables=pair if small and len(tab) else None, trusted=True)
        index = {q: i for i, q in enumerate(qubits)}
        commands = list(circuit)
        gate_q = [tuple(index[q] for q in c.qubits) for c in commands]
        gate_mr = [c.op.type in (OpType.Measure, OpType.Reset) or getattr(c.op.type, "name", "") in ("Measure", "Reset")
                   for c in commands]
        return cls(len(qubits), gate_q, gate_mr, qubits=qubits, commands=commands)

    @classmethod
    def from_gate_tables(cls, Q, q0, q1, mr=None, slices=None) -> "CircuitIR":
        q0 = np.ascontiguousarray(q0, np.int32)
        q1 = np.ascontiguousarray(q1, np.int32)
        gate_q = [(a,) if b < 0 else (a, b) for a, b in zip(q0.tolist(), q1.tolist())]
        return cls(Q, gate_q, mr if mr is not None else [False] * len(gate_q), slices=slices, tables=(q0, q1))

    def _build_edges(self):
        eu, ev, et, ptr = [], [], [], [0]
        self.edge_of_gate = {}
        for t, layer in enumerate(self.slices):
            for g in layer:
                qs = self.gate_q[g]
                if len(qs) == 2 and not self.gate_mr[g]:
                    self.edge_of_gate[g] = len(eu)
                    eu.append(qs[0])
                    ev.append(qs[1])
                    et.append(t)
            ptr.append(len(eu))
        self.nE = len(eu)
        self.edge_u = np.asarray(eu, np.uint16)
        self.edge_v = np.asarray(ev, np.uint16)
        self.edge_t = np.asarray(et, np.int32)
        self.slice_edge_ptr = np.asarray(ptr, np.int32)
        rec = np.zeros(self.nE, dtype=np.dtype([("u", "<u2"), ("v", "<u2"), ("t", "<i4")]))
        rec["u"], rec["v"], rec["t"] = self.edge_u, self.edge_v, self.edge_t
        self.edge_records = rec

    def _build_lookahead(self):
        T, Q, nE = self.T, self.Q, self.nE
        la = np.full((nE, 6), -1, np.int16)
        if nE and T > 1:
            # position and operands of the two-qubit command holding each qubit, per slice
            pos = np.full((T, Q), -1, np.int32)
            first = np.zeros((T, Q), np.int32)
            second = np.zeros((T, Q), np.int32)
            for t, layer in enumerate(self.slices):
                for k, g in enumerate(layer):
                    qs = self.gate_q[g]
                    if len(qs) == 2:
                        for q in qs:
                            pos[t, q] = k
                            first[t, q] = qs[0]
                            second[t, q] = qs[1]
            u = self.edge_u.astype(np.int64)
            v = self.edge_v.astype(np.int64)
            for dt in (1, 2, 3):
                tt = self.edge_t.astype(np.int64) + dt
                ok = tt < T
                tc = np.where(ok, tt, 0)
                cand = []
                for x in (u, v):
                    p = np.where(ok, pos[tc, x], -1)
                    f1, f2 = first[tc, x], second[tc, x]
                    other = np.where((f1 == u) | (f1 == v), f2, f1)
                    cand.append((p, other))
                (pu, ou), (pv, ov) = cand
                pv = np.where(pv == pu, -1, pv)  # both qubits in one command: one term
                swap = (pu < 0) | ((pv >= 0) & (pv < pu))
                p1 = np.where(swap, pv, pu)
                o1 = np.where(swap, ov, ou)
                p2 = np.where(swap, pu, pv)
                o2 = np.where(swap, ou, ov)
                la[:, 2 * (dt - 1)] = np.where(p1 >= 0, o1, -1)
                la[:, 2 * (dt - 1) + 1] = np.where(p2 >= 0, o2, -1)
        self.lookahead = la

    def _build_visits(self):
        vq, vp, vt, vprev, ptr = [], [], [], [], [0]
        seen_at = {}
        for t, layer in enumerate(self.slices):
            for g in layer:
                qs = self.gate_q[g]
                counted = len(qs) == 2 and not self.gate_mr[g]
                for k, q in enumerate(qs):
                    vq.append(q)
                    vp.append(qs[1] if (counted and k == 0) else 0xFFFF)
                    vt.append(t)
                    vprev.append(seen_at.get(q, -1))
                for q in qs:
                    seen_at[q] = t
            ptr.append(len(vq))
        self.nV = len(vq)
        rec = np.zeros(self.nV, dtype=np.dtype([("q", "<u2"), ("partner", "<u2"), ("t", "<i4"), ("prev_t", "<i4"),
                                                ("pad", "<i4")]))
        rec["q"], rec["partner"], rec["t"], rec["prev_t"] = vq, vp, vt, vprev
        self.visit_records = rec
        self.slice_visit_ptr = np.asarray(ptr, np.int32)

    def scheduled_edges(self):
        """(sched_ptr[T+1], words): the counted gates as uint32 (u | v << 16) in the padded per-slice
        layout the dense cost kernels read (see qp_schedule_gates).  Inside a slice 32 consecutive
        gates (the lanes that share LDS cycles) get their endpoints in different 4-byte LDS banks of
        the row strip where possible; the cost is a sum over the slice's gates, so their order is free."""
        from . import _lib
        eu = np.ascontiguousarray(self.edge_u, np.uint16)
        ev = np.ascontiguousarray(self.edge_v, np.uint16)
        sep = np.ascontiguousarray(self.slice_edge_ptr, np.int32)
        ch = _lib.QP_SCHED_CHUNK
        n = int(((np.diff(sep.astype(np.int64)) + ch - 1) // ch * ch).sum()) + _lib.QP_SCHED_TAIL
        sched = np.empty(n, np.uint32)
        sptr = np.empty(self.T + 1, np.int32)
        _lib.check(_lib.lib().qp_schedule_gates(self.T, _lib.ptr(sep), _lib.ptr(eu), _lib.ptr(ev), _lib.ptr(sptr),
                                                _lib.ptr(sched), n))
        return sptr, sched

    def visit_mask(self, ldq: int) -> np.ndarray:
        """uint8 [T, ldq]: 0xFF where the qubit is an operand of a command of the slice (what the gate-ordered cost
        loops visit, BFS_placer.py:214-237), 0 elsewhere."""
        vm = np.zeros((self.T, int(ldq)), np.uint8)
        if self.nV:
            vm[self.visit_records["t"], self.visit_records["q"]] = 0xFF
        return vm

    def visit_prev(self, ldq: int) -> np.ndarray:
        """int16 [T, ldq]: the last slice before t that visits the qubit (visit_mask), -1 if none."""
        seen = np.where(self.visit_mask(ldq) != 0, np.arange(self.T, dtype=np.int32)[:, None], -1)
        run = np.maximum.accumulate(seen, axis=0)
        return np.vstack([np.full((1, int(ldq)), -1, np.int32), run[:-1]]).astype(np.int16)

    def grouped_edges(self, ldq: int):
        """(group_ptr, words) for partitions with rows of ``ldq`` <= 512 bytes, or None when the rows are longer:
        the counted gates of every ``64 // (ldq // 16)`` consecutive slices together, endpoints as byte offsets
        into those rows (see qp_schedule_gate_groups)."""
        from . import _lib
        G = int(_lib.lib().qp_group_rows(int(ldq)))
        if G == 0:
            return None
        eu = np.ascontiguousarray(self.edge_u, np.uint16)
        ev = np.ascontiguousarray(self.edge_v, np.uint16)
        sep = np.ascontiguousarray(self.slice_edge_ptr, np.int32)
        ch = _lib.QP_SCHED_CHUNK
        ng = (self.T + G - 1) // G
        per_group = np.diff(sep.astype(np.int64)[np.minimum(np.arange(ng + 1) * G, self.T)])
        n = int(((per_group + ch - 1) // ch * ch).sum()) + _lib.QP_SCHED_TAIL
        sched = np.empty(n, np.uint32)
        gptr = np.empty(ng + 1, np.int32)
        _lib.check(_lib.lib().qp_schedule_gate_groups(self.T, _lib.ptr(sep), _lib.ptr(eu), _lib.ptr(ev), int(ldq),
                                                      _lib.ptr(gptr), _lib.ptr(sched), n))
        return gptr, sched

    # -- helpers ----------------------------------------------------------------------------------
    def slice_ops(self, t: int):
        """(edge index, u, v) of the counted gates of slice t, slice order."""
        a, b = int(self.slice_edge_ptr[t]), int(self.slice_edge_ptr[t + 1])
        return [(e, int(self.edge_u[e]), int(self.edge_v[e])) for e in range(a, b)]

    def active_qubits(self, t: int):
        a, b = int(self.slice_visit_ptr[t]), int(self.slice_visit_ptr[t + 1])
        return self.visit_records["q"][a:b]

    @property
    def commands(self):
        made = self._commands
        if isinstance(made, weakref.WeakMethod):  # held weakly so that the cached IR does not keep its circuit alive
            made = made()
            if made is None:
                raise ValueError("the circuit this IR was built from no longer exists")
        if callable(made):
            self._commands = made()
        return self._commands

    def command_slices(self):
        """Timeslices as lists of command objects (the reference's return type)."""
        if self.commands is None:
            raise ValueError("this IR was not built from a circuit object")
        return [[self.commands[g] for g in layer] for layer in self.slices]


_ir_cache = weakref.WeakKeyDictionary()


def circuit_ir(circuit) -> CircuitIR:
    """IR of a circuit object, cached per object (the reference re-slices the circuit in every
    place and cost call, HQA_placer.py:28,264)."""
    try:
        hit = _ir_cache.get(circuit)
    except TypeError:
        hit = None
    n = sum(1 for _ in circuit)
    if hit is not None and hit[0] == n:
        return hit[1]
    ir = CircuitIR.from_circuit(circuit)
    try:
        _ir_cache[circuit] = (n, ir)
    except TypeError:
        pass
    return ir
