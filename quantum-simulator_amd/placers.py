"""Placer plug-ins with the reference's names and method signatures (main.py:36-48,85-97):

    placer.place_per_timeslice(circuit, multicore_arch) -> list[dict[qubit -> core node]]
    placer.per_timeslice_cost(circuit, partition, multicore_arch) -> float

The sequential control flow of each placement algorithm stays on the host, as in the reference; what
it evaluates — op x core cost matrices, the assignment solve, the communication-cost sums, the
core-distance matrix — comes from libqplace.so (HIP kernels + host C++ behind include/qplace.h).
Nothing here computes a cost on the CPU in Python.
"""
from __future__ import annotations

import random
from bisect import insort
from collections import deque

import numpy as np

from . import engine as E
from ._lib import QP_ABSENT, QP_PLACE_BFS, QP_PLACE_RANDOM, QP_CM_GTRHQA, QP_CM_HQA, QP_CM_TRHQA, QP_V1, QP_V2, QP_V4, check, host_threads, lib, ptr
from .multicore import graph_tables
from .timeslices import CircuitIR, circuit_ir


class _ArchTables:
    """Per-architecture state: node order, all-pairs distances (host), device tables on demand."""

    def __init__(self, network, device=None):
        self.nodes, self.eu, self.ev, self.w = graph_tables(network)
        self.index = {n: i for i, n in enumerate(self.nodes)}
        self.C = len(self.nodes)
        self.device = device
        self.D = E.apsp(self.C, self.eu, self.ev, self.w)
        self._table = None
        self._hops = None

    @property
    def table(self):
        if self._table is None:
            self._table = E.DistanceTable(self.D, self.device)
        return self._table

    @property
    def hop_table(self):
        if self._hops is None:
            self._hops = E.DistanceTable(E.hops(self.C, self.eu, self.ev), self.device)
        return self._hops

    @staticmethod
    def fingerprint(network):
        return (id(network), network.number_of_nodes(),
                tuple(d.get("weight", 1) for _, _, d in network.edges(data=True)))


class _Context:
    """Everything a placer call needs for one (circuit, architecture) pair, cached on the placer."""

    def __init__(self, circuit, arch, device, ir=None):
        self.device = device
        self.ir: CircuitIR = ir if ir is not None else circuit_ir(circuit)
        self.dc = E.DeviceCircuit(self.ir, device)
        self.arch = _ArchTables(arch.network, device)
        self.eng = E.get_engine(device)
        self.qindex = {q: i for i, q in enumerate(self.ir.qubits)}
        self._cm_bufs = None
        self._slab = None
        self._streams = []

    def lane_resources(self, nbytes, lanes):
        """Pinned (device-visible) host memory and HIP streams for the lanes of a placement batch; kept for
        the next batch."""
        import torch
        if self._slab is None or self._slab.numel() < nbytes:
            self._slab = torch.empty(int(nbytes), dtype=torch.uint8, pin_memory=True)
        while len(self._streams) < lanes:
            self._streams.append(torch.cuda.Stream(self.device))
        return self._slab, self._streams[:lanes]

    # -- partitions -------------------------------------------------------------------------------
    def partition_array(self, partition, dense, rows=None):
        """list[dict] -> uint8 [Tp, ldq].  ``dense``: every circuit qubit must be in every used row
        (a missing key is a KeyError in the reference's HQA-family cost)."""
        Q, ldq = self.ir.Q, self.dc.ldq
        n = len(partition) if rows is None else min(rows, len(partition))
        out = np.full((len(partition), ldq), QP_ABSENT if not dense else 0, np.uint8)
        out[:, Q:] = 0
        idx, qidx = self.arch.index, self.qindex
        for t, row in enumerate(partition):
            if dense and t < n and len(row) < Q:
                missing = next(q for q in self.ir.qubits if q not in row)
                raise KeyError(missing)
            for q, node in row.items():
                k = qidx.get(q)
                if k is None:
                    continue
                out[t, k] = QP_ABSENT if node is None else idx[node]
        return out

    def rows_to_partition(self, rows):
        """uint8 rows -> list of dicts keyed by the circuit's qubit objects, valued by node labels,
        in circuit.qubits order (the reference's dicts are built in that order, HQA_placer.py:40-44)."""
        qs, nodes = self.ir.qubits, self.arch.nodes
        return [{qs[i]: nodes[c] for i, c in enumerate(r)} for r in (row[: self.ir.Q].tolist() for row in rows)]

    # -- device calls -----------------------------------------------------------------------------
    def cost_matrix(self, kind, t, a, op_edges, avail):
        """One op x core matrix for the live assignment ``a`` (uint8 [Q])."""
        import torch
        n_ops, n_av, C = len(op_edges), len(avail), self.arch.C
        dev = self.device
        ab = np.zeros((1, self.dc.ldq), np.uint8)
        ab[0, : self.ir.Q] = a
        av = np.zeros((1, C), np.uint8)
        av[0, :n_av] = avail
        out = self.eng.cost_matrix(
            kind, self.dc, self.arch.table,
            torch.tensor([t], dtype=torch.int32, device=dev), torch.from_numpy(ab).to(dev),
            torch.tensor([list(op_edges)], dtype=torch.int32, device=dev),
            torch.tensor([n_ops], dtype=torch.int32, device=dev), torch.from_numpy(av).to(dev),
            torch.tensor([n_av], dtype=torch.int32, device=dev))
        return out[0, :n_ops, :n_av].cpu().numpy()

    def cost(self, arr, variant, table=None):
        import torch
        A = torch.from_numpy(np.ascontiguousarray(arr[None])).to(self.device)
        tab = table if table is not None else self.arch.table
        if variant == "v3":
            out = self.eng.cost_v3(self.dc, tab, A)
        else:
            out = self.eng.cost_v2(self.dc, tab, A, variant)
        return float(out.cpu()[0])


class BasePlacer:
    """Common services; subclasses provide ``place_per_timeslice``."""

    def __init__(self, device=None):
        self.name = "BasePlacer"
        self._device = device
        self._ctx_key = None
        self._ctx = None

    # -- context ----------------------------------------------------------------------------------
    def _context(self, circuit, arch) -> _Context:
        # keyed on the circuit's IR object, which the context keeps alive (an id of a dead object could be reused)
        ir = circuit_ir(circuit)
        key = (id(ir), ir.n_gates if hasattr(ir, "n_gates") else len(ir.gate_q), _ArchTables.fingerprint(arch.network))
        if self._ctx_key != key:
            dev = self._device if self._device is not None else E.default_device()
            self._ctx = _Context(circuit, arch, dev, ir)
            self._ctx_key = key
            self.node_index = self._ctx.arch.index
            self._dist_matrix = self._ctx.arch.D
        return self._ctx

    # -- reference surface ------------------------------------------------------------------------
    def break_into_timeslices(self, circuit):
        """Timeslices as lists of commands, same membership and order as Placer.py:14-61."""
        return circuit_ir(circuit).command_slices()

    def get_weighted_distance(self, multicore_arch, node1, node2):
        """Weighted shortest-path distance node1 -> node2 (Placer.py:63-76), read from the all-pairs
        table qp_apsp_f64 fills (same bits as a fresh search from node1)."""
        fp = _ArchTables.fingerprint(multicore_arch.network)
        if self._ctx is not None and self._ctx_key[2] == fp:
            tabs = self._ctx.arch
        else:
            if getattr(self, "_dist_key", None) != fp:
                self._dist_tabs = _ArchTables(multicore_arch.network)
                self._dist_key = fp
            tabs = self._dist_tabs
        return self._as_distance(tabs.D[tabs.index[node1], tabs.index[node2]])

    def _host_tables(self, multicore_arch):
        """Node order and distance matrix of an architecture; needs no GPU."""
        fp = _ArchTables.fingerprint(multicore_arch.network)
        if self._ctx is not None and self._ctx_key[2] == fp:
            return self._ctx.arch
        if getattr(self, "_dist_key", None) != fp:
            self._dist_tabs = _ArchTables(multicore_arch.network)
            self._dist_key = fp
        return self._dist_tabs

    def _get_average_weighted_distance(self, multicore_arch):
        """HQA_placer.py:242-257 (== TRHQA / GTRHQA): mean finite distance over unordered pairs of cores, summed in
        (i, j > i) order of ``network.nodes``; 1.0 without a finite pair."""
        tabs = self._host_tables(multicore_arch)
        nodes = list(multicore_arch.network.nodes)
        total, pairs = 0, 0
        for i in range(len(nodes)):
            for j in range(i + 1, len(nodes)):
                w = self._as_distance(tabs.D[tabs.index[nodes[i]], tabs.index[nodes[j]]])
                if w != float("inf"):
                    total += w
                    pairs += 1
        return total / pairs if pairs > 0 else 1.0

    @staticmethod
    def _as_distance(x):
        x = float(x)
        return int(x) if x == 0 else x

    def per_timeslice_cost(self, circuit, partition, multicore_arch):
        """BasePlacer.per_timeslice_cost (Placer.py:78-120) on the device."""
        ctx = self._context(circuit, multicore_arch)
        if len(partition) == 0:
            return 0
        arr = ctx.partition_array(partition, dense=False)
        return self._finish(ctx.cost(arr, QP_V1))

    def _finish(self, x):
        if x == float("inf") and np.isfinite(self._ctx.arch.D).all():
            # the reference raises inside its loop here; main.py:99-102 records inf for the repeat
            raise KeyError("qubit missing from the partition where the cost function needs it")
        return x

    def place_per_timeslice(self, circuit, multicore_arch):
        raise NotImplementedError("Subclasses must implement place_per_timeslice method")

    def simulate_broken_links(self, circuit, multicore_arch, num_simulations=10, break_probability=0.1):
        """HQA_placer.py:325-361 == TRHQA_placer.py:628-669 == BFS_placer.py:241-278 == Random_placer.py:255-296:
        ``num_simulations`` times, give every link an infinite weight with
        probability ``break_probability`` (one draw of the global ``random`` stream per link, in
        ``G.edges`` order), place and cost the circuit on the damaged network, and return the list of
        (set of broken links, cost).  The architecture gets its own network back afterwards.  Distances
        through a broken link are infinite, as networkx makes them (the link stays traversable)."""
        import random
        intact = multicore_arch.network
        links = list(intact.edges)
        results = []
        try:
            for _ in range(num_simulations):
                damaged = intact.copy()
                broken = {link for link in links if random.random() < break_probability}
                for a, b in broken:
                    damaged[a][b]["weight"] = float("inf")
                multicore_arch.network = damaged
                partition = self.place_per_timeslice(circuit, multicore_arch)
                results.append((broken, self.per_timeslice_cost(circuit, partition, multicore_arch)))
        finally:
            multicore_arch.network = intact
        return results

    def get_qpu_capacity(self, qpu_node, multicore_arch):
        return multicore_arch.network.nodes[qpu_node]["qpu"].available_qubits


# ------------------------------------------------------------------------------------------------
# HQA / TRHQA / GTRHQA
# ------------------------------------------------------------------------------------------------

def random_initial_assignment(all_qubits, core_nodes, qubits_per_core, seed=None):
    """HQA_placer.py:8-18 (== TRHQA_placer.py, GTRHQA_placer.py): every core ``qubits_per_core`` times, shuffled by
    the global ``random`` stream (seeded first if ``seed`` is given), dealt to the qubits in order."""
    if seed is not None:
        random.seed(seed)
    slots = []
    for c in core_nodes:
        slots.extend([c] * qubits_per_core)
    random.shuffle(slots)
    return dict(zip(all_qubits, slots))


class _HqaFamily(BasePlacer):
    _kind = QP_CM_HQA

    def per_timeslice_cost(self, circuit, partition, multicore_arch):
        """HQA_placer.py:259-287: crossing gates of every slice + moves between rows i-1 and i for
        i >= 2, with row i standing for slice i."""
        ctx = self._context(circuit, multicore_arch)
        T = ctx.ir.T
        if len(partition) < T:
            raise IndexError("list index out of range")
        arr = ctx.partition_array(partition, dense=True, rows=T)
        return ctx.cost(arr, "v3")

    def run_repeats(self, circuit, multicore_arch, n):
        """``n`` repeats of (place, cost) as main.py:88-92 runs them, as one batch: the placements advance
        together (place_batch) and one qp_slice_cost_v3 launch costs them all.  Returns (seconds per
        placement, costs): two lists of length n."""
        import time
        import torch
        ctx = self._context(circuit, multicore_arch)
        t0 = time.perf_counter()
        rows = self.place_batch(circuit, multicore_arch, B=n)
        if not torch.is_tensor(rows):
            pad = np.zeros(rows.shape[:2] + (ctx.dc.ldq,), np.uint8)
            pad[:, :, :ctx.ir.Q] = rows
            rows = torch.from_numpy(pad).to(ctx.device)
        torch.cuda.synchronize(ctx.device)
        dt = (time.perf_counter() - t0) / n
        costs = ctx.eng.cost_v3(ctx.dc, ctx.arch.table, rows).cpu().numpy()
        return [dt] * n, [float(c) for c in costs]

    def per_core_intercore_activity(self, circuit, partition, multicore_arch):
        """HQA_placer.py:288-321 (== TRHQA_placer.py:586-619): {core node: number of inter-core events it
        takes part in}, counted on the device."""
        import torch
        ctx = self._context(circuit, multicore_arch)
        T = ctx.ir.T
        if len(partition) < T:
            raise IndexError("list index out of range")
        arr = ctx.partition_array(partition, dense=True, rows=T)
        A = torch.from_numpy(np.ascontiguousarray(arr[None])).to(ctx.device)
        counts = ctx.eng.core_activity(ctx.dc, A, ctx.arch.C).cpu().numpy()[0]
        return {node: int(counts[i]) for i, node in enumerate(ctx.arch.nodes)}

    @staticmethod
    def _initial(Q, cap):
        # HQA_placer.py:40-44: qubit i (circuit.qubits order) starts on core int(i / cap)
        return np.array([int(i / cap) for i in range(Q)], np.uint8)

    def _check_capacity(self, Q, C, cap):
        if Q > C * cap:
            raise ValueError(f"Not enough space: {Q} qubits need {C * cap} total slots")

    def place_per_timeslice(self, circuit, multicore_arch):
        """HQA_placer.py:20-176 / TRHQA_placer.py:22-228: one placement, through the native loop of the batches
        (one lane): per slice one device cost-matrix launch and one host assignment solve per while-iteration."""
        ctx = self._context(circuit, multicore_arch)
        rows = self.place_batch(circuit, multicore_arch, B=1, threads=1)[0]
        return ctx.rows_to_partition([rows[t] for t in range(rows.shape[0])])

    def place_rows(self, ctx: _Context, cap: int):
        """HQA_placer.py:56-176 / TRHQA_placer.py:62-228 on index arrays, slice by slice in Python around
        ``Engine.cost_matrix`` and ``lsap``: the readable form of the loop, which the tests hold the native one
        against (about ten times slower).  Returns T uint8 rows."""
        ir, C = ctx.ir, ctx.arch.C
        Q, T = ir.Q, ir.T
        self._check_capacity(Q, C, cap)
        D = ctx.arch.D
        rows = [self._initial(Q, cap)]
        for t in range(1, T):
            prev = rows[t - 1]
            a = prev.copy()
            ops = [op for op in ir.slice_ops(t) if prev[op[1]] != prev[op[2]]]
            if not ops:
                rows.append(a)
                continue
            occ = np.bincount(a, minlength=C).astype(np.int64)
            remaining = ops
            budget = len(ops) + 1
            spent = 0
            while remaining and spent < budget:
                spent += 1
                avail = np.flatnonzero(occ <= cap - 2)
                if avail.size == 0:
                    self._make_space(a, occ, cap, C, D)
                    continue
                M = ctx.cost_matrix(self._kind, t, a, [e for e, _, _ in remaining], avail)
                try:
                    rr, cc = E.lsap(M)
                except ValueError:  # the reference swallows the solver's exception and stops the slice
                    break
                taken = set()
                for i, j in zip(rr.tolist(), cc.tolist()):
                    if M[i, j] != np.inf:
                        _, u, v = remaining[i]
                        core = int(avail[j])
                        occ[a[u]] -= 1
                        occ[a[v]] -= 1
                        a[u] = core
                        a[v] = core
                        occ[core] += 2
                        taken.add(i)
                remaining = [op for k, op in enumerate(remaining) if k not in taken]
            rows.append(a)
        return rows

    # -- many instances at once ----------------------------------------------------------------------
    def place_batch(self, circuit, multicore_arch, initial=None, B=None, threads=None, tables=None, tab_index=None):
        """B independent placements of one circuit on one architecture through qp_hqa_place_batch: the whole
        per-slice loop is native; the batch is split over ``threads`` host threads (default: half of this process's
        share of the node's CPUs -- its affinity mask divided by the ranks on the node -- at most 8), each driving its instances on its own stream (cost matrices on the
        device, assignment solves on the host).  ``initial`` is uint8 [B, Q] (row 0 of each instance; default:
        B copies of the block assignment, i.e. B identical repeats as main.py:88-92 runs them).  Returns uint8
        [B, T, Q]; instance b equals what ``place_per_timeslice`` gives when started from ``initial[b]``.
        With ``tables`` (float64 [S, C, C] distance matrices on the host) and ``tab_index`` (int [B]) instance b
        is placed on the network whose distances are ``tables[tab_index[b]]`` (damaged copies of the machine)."""
        import ctypes as C_
        import torch
        ctx = self._context(circuit, multicore_arch)
        cap = multicore_arch.qpu_qubit_num
        ir, C, D = ctx.ir, ctx.arch.C, ctx.arch.D
        Q, T = ir.Q, ir.T
        self._check_capacity(Q, C, cap)
        if initial is None:
            initial = np.tile(self._initial(Q, cap), (B or 1, 1))
        initial = np.ascontiguousarray(initial, np.uint8)
        if initial.ndim != 2 or initial.shape[1] != Q:
            raise ValueError("initial must be [B, Q]")
        B = initial.shape[0]
        out = np.empty((B, self._rows_out(T), Q), np.uint8)
        if threads is None:
            # half of this rank's CPUs, at most 8: every lane polls its stream, and more lanes than
            # hardware queues plus solver threads only contend (64 x qft_n63 on the 8x8 torus: 12.6 ms on one
            # lane, 8.2 ms on 8, 11.6 ms on 16)
            threads = host_threads(8, polling=True)
        lanes = max(1, min(int(threads), B))
        sep = np.ascontiguousarray(ir.slice_edge_ptr, np.int32)
        most = int(np.diff(sep).max()) if T else 0
        need = int(lib().qp_hqa_place_pinned_bytes(ctx.dc.ldq, C, most, B, lanes))
        slab, streams = ctx.lane_resources(need, lanes)
        handles = (C_.c_void_p * lanes)(*[st.cuda_stream for st in streams])
        edges = np.ascontiguousarray(ir.edge_records)
        if tables is None:
            Dh = np.ascontiguousarray(D, np.float64)
            f64 = ctx.arch.table.f64
            torch.cuda.synchronize(ctx.device)  # the circuit and distance tables were uploaded on another stream
            with torch.cuda.device(ctx.device):
                check(lib().qp_hqa_place_batch(self._kind, C_.byref(ctx.dc.c), C_.byref(f64), ptr(sep),
                                               edges.ctypes.data, ptr(Dh), int(cap), ptr(initial), B, ptr(out),
                                               slab.data_ptr(), slab.numel(), lanes, handles))
            return out
        Dh = np.ascontiguousarray(tables, np.float64)
        if Dh.ndim != 3 or Dh.shape[1:] != (C, C):
            raise ValueError("tables must be [S, C, C]")
        tab = np.ascontiguousarray(tab_index, np.int32)
        if tab.shape != (B,):
            raise ValueError("tab_index must have one entry per instance")
        Dz = Dh.copy()
        Dz[:, np.arange(C), np.arange(C)] = 0.0  # the device tables carry a zero diagonal
        d_tab = torch.from_numpy(Dz).to(ctx.device)
        torch.cuda.synchronize(ctx.device)
        with torch.cuda.device(ctx.device):
            check(lib().qp_hqa_place_batch_multi(self._kind, C_.byref(ctx.dc.c), d_tab.data_ptr(), Dh.shape[0], C, ptr(tab),
                                                 ptr(sep), edges.ctypes.data, ptr(Dh), int(cap), ptr(initial), B, ptr(out),
                                                 slab.data_ptr(), slab.numel(), lanes, handles))
        return out

    def simulate_broken_links(self, circuit, multicore_arch, num_simulations=10, break_probability=0.1):
        """HQA_placer.py:325-361 == TRHQA_placer.py:628-669, all simulations at once: the placement draws no
        random numbers, so the links to break are drawn for every simulation first (the same draws in the same
        order as the reference's loop makes them), then all damaged networks get their distance matrices in one
        threaded batch, the circuit is placed on all of them as one placement batch with a table per instance,
        and one launch costs every placement with its own table."""
        import torch
        intact = multicore_arch.network
        links = list(intact.edges)
        ctx = self._context(circuit, multicore_arch)
        S = int(num_simulations)
        if S <= 0:
            return []
        broken = [{link for link in links if random.random() < break_probability} for _ in range(S)]
        arch = ctx.arch
        pos = {}
        for k, (u, v) in enumerate(zip(arch.eu, arch.ev)):
            pos[(arch.nodes[u], arch.nodes[v])] = k
            pos[(arch.nodes[v], arch.nodes[u])] = k
        W = np.tile(np.asarray(arch.w, np.float64), (S, 1))
        for s_, links_ in enumerate(broken):
            for a, b in links_:
                W[s_, pos[(a, b)]] = np.inf
        tables = E.apsp_batch(arch.C, arch.eu, arch.ev, W)
        rows = self.place_batch(circuit, multicore_arch, B=S, tables=tables, tab_index=np.arange(S, dtype=np.int32))
        pad = np.zeros(rows.shape[:2] + (ctx.dc.ldq,), np.uint8)
        pad[:, :, :ctx.ir.Q] = rows
        Dz = tables.copy()
        Dz[:, np.arange(arch.C), np.arange(arch.C)] = 0.0
        costs = ctx.eng.cost_v3_multi(ctx.dc, torch.from_numpy(Dz).to(ctx.device),
                                      torch.arange(S, dtype=torch.int32, device=ctx.device),
                                      torch.from_numpy(pad).to(ctx.device)).cpu().numpy()
        return [(links_, float(c)) for links_, c in zip(broken, costs)]

    @staticmethod
    def _rows_out(T):
        return T

    @staticmethod
    def _make_space(a, occ, cap, C, D):
        """HQA_placer.py:180-204: from every core above cap-2, move up to two of its qubits (lowest
        qubit index first, as they stand when the core's turn comes) to the nearest core with room, ties
        by lower occupancy then node order.  ``a`` and ``occ`` are updated in place; ``D`` may be the
        matrix or its rows as lists.  Plain lists throughout: the per-core numpy calls of a vectorised
        version cost more than the work at these sizes."""
        lim = cap - 2
        occl = occ.tolist()
        if max(occl) <= lim:
            return
        members = [[] for _ in range(C)]
        for q, c in enumerate(a.tolist()):
            members[c].append(q)
        for core in range(C):
            if occl[core] > lim:
                drow = D[core] if isinstance(D, list) else D[core].tolist()
                for q in members[core][:2]:
                    best, bkey = -1, None
                    for tc in range(C):
                        if occl[tc] < cap - 1 and tc != core:
                            key = (drow[tc], occl[tc])
                            if bkey is None or key < bkey:  # strict: the first of equal keys, as a stable sort gives
                                best, bkey = tc, key
                    if best >= 0:
                        a[q] = best
                        members[core].remove(q)
                        insort(members[best], q)
                        occl[core] -= 1
                        occl[best] += 1
                    if occl[core] <= lim:
                        break
        occ[:] = occl


class hqa_placer(_HqaFamily):
    _kind = QP_CM_HQA

    def __init__(self, device=None):
        super().__init__(device)
        self.name = "HQA"


class trhqa_placer(_HqaFamily):
    _kind = QP_CM_TRHQA

    def __init__(self, device=None):
        super().__init__(device)
        self.name = "TRHQA"


class gtrhqa_placer(_HqaFamily):
    _kind = QP_CM_GTRHQA

    def __init__(self, device=None):
        super().__init__(device)
        self.name = "GTRHQA"

    # one damaged network at a time (the device placer takes one distance table per launch)
    simulate_broken_links = BasePlacer.simulate_broken_links

    def place_rows(self, ctx: _Context, cap: int):
        """GTRHQA_placer.py:62-265.  Returns T+1 rows: slice t starts from row t-1 (row 0 for t = 0)
        and its result is appended as row t+1, exactly as the reference's loop does.  Every slice runs
        on the device (qp_gtrhqa_place)."""
        Q = ctx.ir.Q
        out = self._place_on_device(ctx, cap, self._initial(Q, cap)[None])
        return [out[0, t] for t in range(out.shape[1])]

    def _place_on_device(self, ctx: _Context, cap: int, initial: np.ndarray, as_tensor=False):
        import torch
        ir, C = ctx.ir, ctx.arch.C
        Q = ir.Q
        self._check_capacity(Q, C, cap)
        if cap < 2:
            raise ValueError("GTRHQA needs cores that hold at least two qubits")
        initial = np.ascontiguousarray(initial, np.uint8)
        if initial.ndim != 2 or initial.shape[1] != Q:
            raise ValueError("initial must be [B, Q]")
        if initial.size and int(initial.max()) >= C:
            raise ValueError("initial holds a core index outside the architecture")
        ldq = ctx.dc.ldq
        pad = np.zeros((initial.shape[0], ldq), np.uint8)
        pad[:, :Q] = initial
        rows = ctx.eng.gtrhqa_place(ctx.dc, ctx.arch.table, torch.from_numpy(pad).to(ctx.device), cap)
        if as_tensor:
            return rows
        return rows.cpu().numpy()[:, :, :Q]

    def run_repeats(self, circuit, multicore_arch, n):
        """As for the other HQA-family placers, with the rows never leaving the device."""
        import time
        import torch
        ctx = self._context(circuit, multicore_arch)
        t0 = time.perf_counter()
        rows = self.place_batch(circuit, multicore_arch, B=n, as_tensor=True)
        torch.cuda.synchronize(ctx.device)
        dt = (time.perf_counter() - t0) / n
        costs = ctx.eng.cost_v3(ctx.dc, ctx.arch.table, rows).cpu().numpy()
        return [dt] * n, [float(c) for c in costs]

    def place_batch(self, circuit, multicore_arch, initial=None, B=None, threads=8, as_tensor=False):
        """B placements in one launch.  Returns uint8 [B, T+1, Q] (the reference's T+1 rows), or with
        as_tensor the device tensor [B, T+1, ldq] that Engine.cost_v3 takes as it is."""
        ctx = self._context(circuit, multicore_arch)
        cap = multicore_arch.qpu_qubit_num
        if initial is None:
            initial = np.tile(self._initial(ctx.ir.Q, cap), (B or 1, 1))
        return self._place_on_device(ctx, cap, initial, as_tensor=as_tensor)


# ------------------------------------------------------------------------------------------------
# BFS / Random: host placement driven by Python's global ``random`` stream, device cost
# ------------------------------------------------------------------------------------------------

class _SparsePlacer(BasePlacer):
    _which = None  # QP_PLACE_RANDOM / QP_PLACE_BFS of the native form (place_batch)

    def per_timeslice_cost(self, circuit, partition, multicore_arch):
        """BFS_placer.py:202-239 == Random_placer.py:178-213."""
        ctx = self._context(circuit, multicore_arch)
        if len(partition) < ctx.ir.T:
            raise IndexError("list index out of range")
        return self._finish(ctx.cost(ctx.partition_array(partition, dense=False), QP_V2))

    @staticmethod
    def _slice_tables(ctx):
        """What the placement loops read of the circuit, as integer tables (cached on the context): per slice
        the active qubits in the order the reference walks its set of them (``list(set(...))`` of the command
        objects' qubits: fixed by the qubits' hashes and the insertion order), and the commands in slice order."""
        if getattr(ctx, "_sparse_tables", None) is None:
            qidx = ctx.qindex
            mptr, mem, cptr, c0, c1 = [0], [], [0], [], []
            for layer in ctx.ir.command_slices():
                members = set()
                for cmd in layer:
                    members.update(cmd.qubits)
                mem.extend(qidx[q] for q in list(members))
                mptr.append(len(mem))
                for cmd in layer:
                    qs = cmd.qubits
                    if len(qs) == 2:
                        c0.append(qidx[qs[0]])
                        c1.append(qidx[qs[1]])
                    elif len(qs) == 1:
                        c0.append(qidx[qs[0]])
                        c1.append(-1)
                    else:
                        c0.append(-1)
                        c1.append(-1)
                cptr.append(len(c0))
            ctx._sparse_tables = (np.array(mptr, np.int32), np.array(mem, np.uint16), np.array(cptr, np.int32),
                                  np.array(c0, np.int32), np.array(c1, np.int32))
        return ctx._sparse_tables

    @staticmethod
    def _neighbour_tables(ctx, net):
        """Neighbours of every core, nearest first (BFS_placer.py:7-18; Python's stable sort keeps adjacency
        order among equal distances), as positions in ``list(net.nodes())``."""
        if getattr(ctx, "_nb_tables", None) is None:
            nodes = list(net.nodes())
            pos = {n: i for i, n in enumerate(nodes)}
            idx, D = ctx.arch.index, ctx.arch.D
            nptr, nb = [0], []
            for n in nodes:
                order = list(net.neighbors(n))
                order.sort(key=lambda m: D[idx[n], idx[m]])
                nb.extend(pos[m] for m in order)
                nptr.append(len(nb))
            ctx._nb_tables = (np.array(nptr, np.int32), np.array(nb, np.int32))
        return ctx._nb_tables

    def place_batch(self, circuit, multicore_arch, B, states=None):
        """``B`` placements in a row by the native form of ``place_per_timeslice`` (qp_place_sparse), drawing from
        Python's global ``random`` stream exactly as B calls would and leaving it where they would; with
        ``states`` (B results of ``random.getstate()``) repeat r draws from states[r] instead and the global
        stream ends where the last of them ends.  Returns uint8 [B, T, ldq] rows as qp_slice_cost_v2 takes them
        (QP_ABSENT = the qubit is not in the slice's dict)."""
        ctx = self._context(circuit, multicore_arch)
        net = multicore_arch.network
        nodes = list(net.nodes())
        if nodes != list(ctx.arch.nodes):
            raise ValueError("core order of the distance tables differs from network.nodes()")
        cap = int(multicore_arch.qpu_qubit_num)
        mptr, mem, cptr, c0, c1 = self._slice_tables(ctx)
        T, Q, ldq, C = ctx.ir.T, ctx.ir.Q, ctx.dc.ldq, ctx.arch.C
        A = np.empty((B, T, ldq), np.uint8)
        if states is None:
            st = [random.getstate()]
        else:
            if len(states) != B:
                raise ValueError("one generator state per placement")
            st = list(states)
        if any(s[0] != 3 for s in st):
            raise ValueError("unknown random state version")
        mt = np.array([s[1] for s in st], np.uint32)
        if self._which == QP_PLACE_BFS:
            nptr, nb = self._neighbour_tables(ctx, net)
            D = np.ascontiguousarray(ctx.arch.D, np.float64)
            args = (ptr(cptr), ptr(c0), ptr(c1), ptr(nptr), ptr(nb), ptr(D))
        else:
            args = (None,) * 6
        check(lib().qp_place_sparse(self._which, T, Q, ldq, C, cap, ptr(mptr), ptr(mem), *args, ptr(mt), len(st), B, ptr(A)))
        random.setstate((3, tuple(int(x) for x in mt[-1]), st[-1][2]))
        for n in nodes:  # what the last slice of the last placement leaves on the architecture object
            net.nodes[n]["qpu"].available_qubits = cap
        if B and T and self._which == QP_PLACE_RANDOM:  # (the BFS loop only resets the counters at a slice's start)
            used = np.bincount(A[-1, -1, :Q][A[-1, -1, :Q] != QP_ABSENT], minlength=C)
            for i, n in enumerate(nodes):
                net.nodes[n]["qpu"].available_qubits = cap - int(used[i])
        return A

    def run_repeats(self, circuit, multicore_arch, n, states=None):
        """``n`` repeats of (place, cost) as main.py:88-92 runs them: the placements by the native loop, their
        costs by one qp_slice_cost_v2 launch.  Returns (seconds per placement, costs)."""
        import time
        import torch
        ctx = self._context(circuit, multicore_arch)
        t0 = time.perf_counter()
        A = self.place_batch(circuit, multicore_arch, n, states=states)
        dt = (time.perf_counter() - t0) / max(n, 1)
        costs = ctx.eng.cost_v2(ctx.dc, ctx.arch.table, torch.from_numpy(A).to(ctx.device), QP_V2).cpu().numpy()
        return [dt] * n, [float(c) for c in costs]  # inf = the reference's cost loop raises for this repeat


class random_placer(_SparsePlacer):
    def __init__(self, device=None):
        super().__init__(device)
        self.name = "RANDOM"
        self._which = QP_PLACE_RANDOM

    def simulate_broken_links(self, circuit, multicore_arch, num_simulations=10, break_probability=0.1):
        """Random_placer.py:255-296: as the other placers' (one ``random.random()`` per link and simulation), but
        every damaged network is placed with ``place_per_timeslice_weighted_aware``."""
        plain = self.place_per_timeslice
        self.place_per_timeslice = self.place_per_timeslice_weighted_aware
        try:
            return super().simulate_broken_links(circuit, multicore_arch, num_simulations, break_probability)
        finally:
            self.place_per_timeslice = plain

    def place(self, circuit, multicore_arch):
        """Random_placer.py:9-23: one core per qubit, a ``random.sample`` of the cores; None (after the
        reference's message) when there are more qubits than cores."""
        nodes = list(multicore_arch.network.nodes())
        qubits = circuit.qubits
        if len(qubits) > len(nodes):
            print("Error: Not enough unique QPU nodes to place all qubits.")
            return None
        return dict(zip(qubits, random.sample(nodes, len(qubits))))

    def get_placement_statistics(self, circuit, partition, multicore_arch):
        """Random_placer.py:214-253: counts of two-qubit gates and of those split over two cores, with the
        weighted and hop distances of the split ones (slice order) and their means."""
        tabs = self._host_tables(multicore_arch)
        H = E.hops(tabs.C, tabs.eu, tabs.ev)
        ir = circuit_ir(circuit)
        layers = ir.command_slices()
        stats = {"total_timeslices": len(layers), "total_partitions": len(partition), "weighted_distances": [],
                 "unweighted_distances": [], "two_qubit_gates": 0, "cross_qpu_gates": 0}
        for i, layer in enumerate(layers):
            for g, cmd in zip(ir.slices[i], layer):
                if len(cmd.qubits) == 2 and not ir.gate_mr[g]:
                    stats["two_qubit_gates"] += 1
                    a, b = cmd.qubits
                    na, nb = partition[i][a], partition[i][b]
                    if na != nb:
                        stats["cross_qpu_gates"] += 1
                        stats["weighted_distances"].append(self._as_distance(tabs.D[tabs.index[na], tabs.index[nb]]))
                        stats["unweighted_distances"].append(int(H[tabs.index[na], tabs.index[nb]]))
        if stats["weighted_distances"]:
            stats["avg_weighted_distance"] = np.mean(stats["weighted_distances"])
            stats["avg_unweighted_distance"] = np.mean(stats["unweighted_distances"])
        else:
            stats["avg_weighted_distance"] = 0
            stats["avg_unweighted_distance"] = 0
        return stats

    def place_per_timeslice_weighted_aware(self, circuit, multicore_arch, bias_factor=0.3):
        """Random_placer.py:61-142.  Per slice: a two-qubit gate whose qubits are both unplaced goes to a random
        core with two free slots; when there is none, to a pair of cores with a free slot each, drawn (with
        probability ``bias_factor``, one ``random.random()``) by numpy's global ``np.random.choice`` with
        probabilities proportional to 1 / (distance + 0.1) (0.1 for an infinite distance), else by
        ``random.choice`` over the pairs; every qubit still unplaced then draws ``random.choice`` over the cores
        with room, in gate order."""
        net = multicore_arch.network
        cap = multicore_arch.qpu_qubit_num
        nodes = list(net.nodes())
        result = []
        for layer in self.break_into_timeslices(circuit):
            room = {n: cap for n in nodes}
            row = {}
            placed = set()
            for cmd in layer:
                if len(cmd.qubits) != 2:
                    continue
                a, b = cmd.qubits
                if a in placed or b in placed:
                    continue
                homes = [n for n in nodes if room[n] >= 2]
                if homes:
                    home = random.choice(homes)
                    row[a] = row[b] = home
                    room[home] -= 2
                    placed.update((a, b))
                    continue
                open_nodes = [n for n in nodes if room[n] > 0]
                if len(open_nodes) < 2:
                    continue
                pairs, probs = [], []
                for i, n1 in enumerate(open_nodes):
                    for n2 in open_nodes[i + 1:]:
                        w = self.get_weighted_distance(multicore_arch, n1, n2)
                        pairs.append((n1, n2))
                        probs.append(1.0 / (w + 0.1) if w != float("inf") else 0.1)
                if random.random() < bias_factor:
                    total = sum(probs)
                    n1, n2 = pairs[np.random.choice(len(pairs), p=[x / total for x in probs])]
                else:
                    n1, n2 = random.choice(pairs)
                row[a], row[b] = n1, n2
                room[n1] -= 1
                room[n2] -= 1
                placed.update((a, b))
            for cmd in layer:
                for q in cmd.qubits:
                    if q not in placed:
                        open_nodes = [n for n in nodes if room[n] > 0]
                        if open_nodes:
                            n = random.choice(open_nodes)
                            row[q] = n
                            room[n] -= 1
                            placed.add(q)
            for n in nodes:
                net.nodes[n]["qpu"].available_qubits = room[n]
            result.append(row)
        return result

    def place_per_timeslice(self, circuit, multicore_arch):
        """Random_placer.py:25-59: per slice, every active qubit (in ``list(set(...))`` order) draws
        ``random.choice`` over the cores that still have room."""
        net = multicore_arch.network
        cap = multicore_arch.qpu_qubit_num
        nodes = list(net.nodes())
        result = []
        for layer in self.break_into_timeslices(circuit):
            for n in nodes:
                net.nodes[n]["qpu"].available_qubits = cap
            room = {n: cap for n in nodes}
            members = set()
            for cmd in layer:
                members.update(cmd.qubits)
            placed = {}
            open_nodes = [n for n in nodes if room[n] > 0]  # (the reference rebuilds this list per qubit; it only
            for q in list(members):                         # changes when a core fills up)
                if q not in placed and open_nodes:
                    n = random.choice(open_nodes)
                    placed[q] = n
                    room[n] -= 1
                    if room[n] == 0:
                        open_nodes = [m for m in nodes if room[m] > 0]
            for n in nodes:
                net.nodes[n]["qpu"].available_qubits = room[n]
            result.append(placed)
        return result


class bfs_placer(_SparsePlacer):
    def __init__(self, device=None):
        super().__init__(device)
        self.name = "BFS"
        self._which = QP_PLACE_BFS

    def _get_weighted_neighbors_sorted(self, multicore_arch, node):
        """BFS_placer.py:7-18: the neighbours of a core, nearest first (stable among equal distances)."""
        tabs = self._host_tables(multicore_arch)
        nb = list(multicore_arch.network.neighbors(node))
        nb.sort(key=lambda m: self._as_distance(tabs.D[tabs.index[node], tabs.index[m]]))
        return nb

    def place(self, circuit, multicore_arch):
        """BFS_placer.py:20-99: one placement for the whole circuit, on the free slots the architecture's core
        objects hold when it is called (they are not reset).  Two-qubit gates in circuit order: both qubits to the
        first core with two free slots, else (after ``random.shuffle`` of the core list, which stays shuffled) to
        the first core with a free slot and its nearest neighbour with one; the other qubits breadth-first from
        the emptiest cores; what is still left draws ``random.choice``."""
        net = multicore_arch.network
        nodes = list(net.nodes())
        free = {n: net.nodes[n]["qpu"].available_qubits for n in nodes}
        row, placed = {}, set()
        for cmd in circuit:
            if len(cmd.qubits) != 2:
                continue
            a, b = cmd.qubits[0], cmd.qubits[1]
            if a in placed or b in placed:
                continue
            for n in nodes:
                if free[n] >= 2:
                    row[a] = row[b] = n
                    free[n] -= 2
                    placed.update((a, b))
                    break
            if a not in placed:
                random.shuffle(nodes)
                found = False
                for n1 in nodes:
                    if free[n1] >= 1:
                        for n2 in self._get_weighted_neighbors_sorted(multicore_arch, n1):
                            if free[n2] >= 1:
                                row[a], row[b] = n1, n2
                                free[n1] -= 1
                                free[n2] -= 1
                                placed.update((a, b))
                                found = True
                                break
                        if found:
                            break
        waiting = [q for q in list(circuit.qubits) if q not in placed]
        if waiting:
            start = [(n, free[n]) for n in nodes if free[n] > 0]
            start.sort(key=lambda x: x[1], reverse=True)
            queue = deque(n for n, _ in start)
            seen = set(queue)
            while queue and waiting:
                n = queue.popleft()
                row[waiting.pop(0)] = n
                free[n] -= 1
                for m in self._get_weighted_neighbors_sorted(multicore_arch, n):
                    if m not in seen and free[m] > 0:
                        seen.add(m)
                        queue.append(m)
            for q in waiting:
                open_nodes = [n for n in nodes if free[n] > 0]
                if open_nodes:
                    n = random.choice(open_nodes)
                    row[q] = n
                    free[n] -= 1
        for n in free:
            net.nodes[n]["qpu"].available_qubits = free[n]
        return row

    def place_per_timeslice(self, circuit, multicore_arch):
        """BFS_placer.py:101-200.  Two-qubit gates go to the first core with two free slots, else to
        the closest pair of cores with one free slot each; single-qubit gates draw ``random.choice``;
        what is left is spread breadth-first from the emptiest cores."""
        ctx = self._context(circuit, multicore_arch)
        net = multicore_arch.network
        cap = multicore_arch.qpu_qubit_num
        nodes = list(net.nodes())
        idx, D = ctx.arch.index, ctx.arch.D

        def by_distance(n):  # neighbours of n, nearest first (BFS_placer.py:7-18); stable for ties
            nb = list(net.neighbors(n))
            nb.sort(key=lambda m: D[idx[n], idx[m]])
            return nb

        result = []
        for layer in self.break_into_timeslices(circuit):
            for n in nodes:
                net.nodes[n]["qpu"].available_qubits = cap
            room = {n: cap for n in nodes}
            row = {}
            members = set()
            for cmd in layer:
                members.update(cmd.qubits)
            members = list(members)
            done = set()
            for cmd in layer:
                qs = cmd.qubits
                if len(qs) == 2:
                    a, b = qs
                    if a in done or b in done:
                        continue
                    home = next((n for n in nodes if room[n] >= 2), None)
                    if home is not None:
                        row[a] = row[b] = home
                        room[home] -= 2
                        done.update((a, b))
                    else:
                        pairs = [(n1, n2, D[idx[n1], idx[n2]]) for n1 in nodes if room[n1] >= 1
                                 for n2 in by_distance(n1) if room[n2] >= 1]
                        pairs.sort(key=lambda x: x[2])
                        for n1, n2, _ in pairs:
                            if room[n1] >= 1 and room[n2] >= 1:
                                row[a], row[b] = n1, n2
                                room[n1] -= 1
                                room[n2] -= 1
                                done.update((a, b))
                                break
                if len(qs) == 1:
                    q = qs[0]
                    if q not in done:
                        open_nodes = [n for n in nodes if room[n] > 0]
                        if open_nodes:
                            n = random.choice(open_nodes)
                            row[q] = n
                            room[n] -= 1
                            done.add(q)
            waiting = [q for q in members if q not in done]
            start = [(n, room[n]) for n in nodes if room[n] > 0]
            start.sort(key=lambda x: x[1], reverse=True)
            queue = deque(n for n, _ in start)
            seen = set(queue)
            while queue and waiting:
                n = queue.popleft()
                row[waiting.pop(0)] = n
                room[n] -= 1
                for m in by_distance(n):
                    if m not in seen and room[m] > 0:
                        seen.add(m)
                        queue.append(m)
            result.append(row)
        return result


PLACERS = {"HQA": hqa_placer, "TRHQA": trhqa_placer, "GTRHQA": gtrhqa_placer, "RANDOM": random_placer,
           "BFS": bfs_placer}


class qubo_placer(_SparsePlacer):
    """Qubo_placer.py:11-174 with the model built as integer tables, the sampler replaced by
    replica-parallel annealing on the GPU (qp_qubo_anneal) and the lowest-energy replica chosen by
    exact MFMA energies (qp_qubo_energy_i32).  ``num_reads`` and the seed default to the reference's
    (10, 42); reads are rounded up to a multiple of 128 replicas, which costs nothing on the GPU.
    The samples are not neal's (different generator and sweep order); energies are exact."""

    def __init__(self, device=None, num_reads=10, num_sweeps=1000, seed=42):
        super().__init__(device)
        self.name = "QUBO"
        self.num_reads, self.num_sweeps, self.seed = num_reads, num_sweeps, seed
        self.last_energy = None

    def _model(self, ctx, cap):
        from .qubo import DeviceQubo, QuboModel
        key = (id(ctx), cap)
        if getattr(self, "_model_key", None) != key:
            H = E.hops(ctx.arch.C, ctx.arch.eu, ctx.arch.ev)
            self._qm = QuboModel(ctx.ir, ctx.arch.C, cap, H)
            self._dq = DeviceQubo(self._qm, ctx.device)
            self._model_key = key
        return self._qm, self._dq

    def place_per_timeslice(self, circuit, multicore_arch):
        ctx = self._context(circuit, multicore_arch)
        qm, dq = self._model(ctx, multicore_arch.qpu_qubit_num)
        R = (max(1, self.num_reads) + 127) // 128 * 128
        states, _ = dq.anneal(R, self.num_sweeps, seed=self.seed)
        energy = dq.energy(states)
        best = int(energy.argmin().item())
        self.last_energy = int(energy[best].item())
        rows = qm.decode(states[best].cpu().numpy())
        qs, nodes = ctx.ir.qubits, ctx.arch.nodes
        return [{qs[q]: nodes[c] for q, c in enumerate(row) if c is not None} for row in rows]

    def per_timeslice_cost(self, circuit, partition, multicore_arch):
        """Qubo_placer.py:177-207: the gate-ordered cost with unweighted hop counts."""
        ctx = self._context(circuit, multicore_arch)
        if len(partition) < ctx.ir.T:
            raise IndexError("list index out of range")
        arr = ctx.partition_array(partition, dense=False)
        x = ctx.cost(arr, QP_V4, table=ctx.arch.hop_table)
        if x == float("inf"):
            raise KeyError("qubit missing from the partition where the cost function needs it")
        return int(x)

    def intercorecommatrix(self, circuit, partition, multicore_arch, plot_file=None):
        """Qubo_placer.py:208-245: symmetric core x core counts of crossing gates and of qubits that
        changed core since their previous gate, counted on the device.  Returns the numpy matrix (the
        reference only draws it); with ``plot_file`` the same heat-map is saved there."""
        import torch
        ctx = self._context(circuit, multicore_arch)
        if len(partition) < ctx.ir.T:
            raise IndexError("list index out of range")
        arr = ctx.partition_array(partition, dense=False)
        A = torch.from_numpy(np.ascontiguousarray(arr[None])).to(ctx.device)
        M, status = ctx.eng.comm_matrix(ctx.dc, A, ctx.arch.C)
        if int(status.cpu()[0]):
            raise KeyError("qubit missing from the partition where the walk needs it")
        M = M.cpu().numpy()[0].astype(int)
        if plot_file:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            plt.figure(figsize=(8, 6))
            plt.imshow(M, cmap="hot", interpolation="nearest")
            plt.colorbar(label="Intercore Communication Count")
            plt.title("Intercore Communication Matrix")
            plt.xlabel("QPU index")
            plt.ylabel("QPU index")
            plt.savefig(plot_file, bbox_inches="tight")
            plt.close()
        return M


PLACERS["QUBO"] = qubo_placer
