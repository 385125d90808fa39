"""Device-side state and launches: circuit tables and distance tables in HBM, and thin wrappers that
hand torch-owned device buffers to the C ABI (include/qplace.h).  torch is used for device memory,
streams and copies only; every number comes out of libqplace.so.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import (QP_ABSENT, QP_CM_GTRHQA, QP_CM_HQA, QP_CM_TRHQA, QP_DT_F64, QP_DT_U16, QP_V1, QP_V2, QP_V4,
                   QpCircuit, QpDTable, QpGridMetric, check, lib, ptr)
from .timeslices import CircuitIR


def _torch():
    import torch
    return torch


def default_device():
    torch = _torch()
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: the cost kernels run on an MI355X only (there is no CPU path)")
    return torch.device("cuda", torch.cuda.current_device())


def _stream_ptr(device):
    return C.c_void_p(_torch().cuda.current_stream(device).cuda_stream)


def _to_dev(arr: np.ndarray, device):
    torch = _torch()
    flat = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
    if flat.size == 0:
        return torch.zeros(16, dtype=torch.uint8, device=device)
    return torch.from_numpy(flat.copy()).to(device)


# ------------------------------------------------------------------------------------------------
# host-only entry points
# ------------------------------------------------------------------------------------------------

def apsp(C_: int, eu, ev, w) -> np.ndarray:
    """All-pairs weighted core distances (row = source), via qp_apsp_f64."""
    eu = np.ascontiguousarray(eu, np.int32)
    ev = np.ascontiguousarray(ev, np.int32)
    w = np.ascontiguousarray(w, np.float64)
    D = np.empty((C_, C_), np.float64)
    check(lib().qp_apsp_f64(C_, len(eu), ptr(eu), ptr(ev), ptr(w), ptr(D)))
    return D


def apsp_batch(C_: int, eu, ev, W, threads=None) -> np.ndarray:
    """Distance matrices [S, C, C] of S networks that share their links and differ in the weights W [S, E].
    ``threads`` defaults to this process's share of the node's CPUs, at most 16 (``_lib.host_threads``)."""
    if threads is None:
        threads = _lib.host_threads(16)
    eu = np.ascontiguousarray(eu, np.int32)
    ev = np.ascontiguousarray(ev, np.int32)
    W = np.ascontiguousarray(W, np.float64)
    S = W.shape[0]
    D = np.empty((S, C_, C_), np.float64)
    check(lib().qp_apsp_f64_batch(C_, len(eu), ptr(eu), ptr(ev), ptr(W), S, ptr(D), int(threads)))
    return D


def hops(C_: int, eu, ev) -> np.ndarray:
    eu = np.ascontiguousarray(eu, np.int32)
    ev = np.ascontiguousarray(ev, np.int32)
    H = np.empty((C_, C_), np.float64)
    check(lib().qp_hops_f64(C_, len(eu), ptr(eu), ptr(ev), ptr(H)))
    return H


def lsap(cost: np.ndarray):
    """Row/column indices of the optimal assignment, tie-for-tie what scipy returns."""
    cost = np.ascontiguousarray(cost, np.float64)
    nr, nc = cost.shape
    k = min(nr, nc)
    rows = np.empty(k, np.int32)
    cols = np.empty(k, np.int32)
    rc = lib().qp_lsap_f64(nr, nc, ptr(cost), ptr(rows), ptr(cols))
    if rc == _lib.QP_OK:
        return rows, cols
    raise ValueError(lib().qp_last_error().decode())


def lsap_batch(costs: np.ndarray, nr: np.ndarray, nc: np.ndarray, threads=None):
    """costs [n, R, L] (problem k uses the top-left nr[k] x nc[k]); returns rows, cols [n, min(R, L)], status.
    ``threads`` defaults to this process's share of the node's CPUs, at most 8."""
    if threads is None:
        threads = _lib.host_threads(8)
    costs = np.ascontiguousarray(costs, np.float64)
    n, R, L = costs.shape
    nr = np.ascontiguousarray(nr, np.int32)
    nc = np.ascontiguousarray(nc, np.int32)
    k = max(1, min(R, L))
    rows = np.full((n, k), -1, np.int32)
    cols = np.full((n, k), -1, np.int32)
    status = np.zeros(n, np.int32)
    check(lib().qp_lsap_f64_batch(n, ptr(nr), ptr(nc), ptr(costs), R * L, L, ptr(rows), ptr(cols), k, ptr(status),
                                  int(threads)))
    return rows, cols, status


# ------------------------------------------------------------------------------------------------
# device-resident tables
# ------------------------------------------------------------------------------------------------

class DeviceCircuit:
    """A CircuitIR uploaded once; ``.c`` is the QpCircuit the ABI takes."""

    def __init__(self, ir: CircuitIR, device=None, visit_mask=True):
        self.ir = ir
        self.device = device if device is not None else default_device()
        if ir.Q > 32767:
            raise ValueError("at most 32767 qubits")
        self._sep = _to_dev(ir.slice_edge_ptr, self.device)
        self._edges = _to_dev(ir.edge_records, self.device)
        self._la = _to_dev(ir.lookahead, self.device)
        self._svp = _to_dev(ir.slice_visit_ptr, self.device)
        self._visits = _to_dev(ir.visit_records, self.device)
        sptr, sched = ir.scheduled_edges()
        self._sched, self._sched_ptr = _to_dev(sched, self.device), _to_dev(sptr, self.device)
        most = int(np.diff(ir.slice_edge_ptr).max()) if ir.T else 0
        self.ldq = (ir.Q + 15) // 16 * 16
        # rows of at most 512 bytes: the gates of the slices a wave takes per step, laid out together
        grouped = ir.grouped_edges(self.ldq)
        self._gsched = self._gptr = None
        if grouped is not None:
            self._gptr, self._gsched = _to_dev(grouped[0], self.device), _to_dev(grouped[1], self.device)
        # rows of up to 1024 bytes: which qubits every slice visits, for the row walk of the gate-ordered cost
        # (visit_mask=False leaves it out: the cost is then always taken visit by visit; so does a circuit whose
        # mask would take more than 64 MB)
        self._vmask = _to_dev(ir.visit_mask(self.ldq), self.device) if visit_mask and self.ldq <= 1024 and 0 < ir.T * self.ldq <= 1 << 26 else None
        self._vprev = _to_dev(ir.visit_prev(self.ldq), self.device) if self._vmask is not None and ir.T <= 32767 else None
        self.c = QpCircuit(ir.Q, ir.T, ir.nE, ir.nV, most, self.ldq if grouped is not None else 0,
                           self._sep.data_ptr(), self._edges.data_ptr(),
                           self._la.data_ptr(), self._svp.data_ptr(), self._visits.data_ptr(),
                           self._sched.data_ptr(), self._sched_ptr.data_ptr(),
                           self._gsched.data_ptr() if grouped is not None else None,
                           self._gptr.data_ptr() if grouped is not None else None,
                           self._vmask.data_ptr() if self._vmask is not None else None,
                           self.ldq if self._vmask is not None else 0, 0,
                           self._vprev.data_ptr() if self._vprev is not None else None)

    @property
    def csr_bytes(self):
        return 4 * (self.ir.T + 1) + 8 * self.ir.nE


# |D[a][b] - D[b][a]| relative to the entry below which a matrix counts as symmetric: costs are compared
# at 1e-12 relative, a symmetrised entry is off by at most half the asymmetry
SYMMETRY_TOLERANCE = 1e-13
SCALED_GRID_TOLERANCE = 1e-13  # a distance of k equal links against k times the link weight: rounding of k - 1 additions


class DistanceTable:
    """A C x C distance matrix on the host (``D``) and on the device in two forms: fp64 (cost
    matrices) and the packed form qp_dtable_pack chooses (uint16 when every entry is an integer)."""

    def __init__(self, D: np.ndarray, device=None):
        torch = _torch()
        self.device = device if device is not None else default_device()
        self.D = np.ascontiguousarray(D, np.float64)
        self.C = int(self.D.shape[0])
        n = self.C * self.C
        staging = np.zeros((n * 8 + 15) // 16 * 16, np.uint8)
        kind = C.c_int32(0)
        nbytes = C.c_size_t(0)
        check(lib().qp_dtable_pack(ptr(self.D), self.C, ptr(staging), staging.size, C.byref(kind), C.byref(nbytes)))
        self.kind = kind.value
        padded = (nbytes.value + 15) // 16 * 16
        self._packed = torch.from_numpy(staging[:padded].copy()).to(self.device)
        # "table": always looked up; "packed": additionally carries the closed form of a grid metric
        # when the matrix is one, and the dense cost kernel then computes distances from core ids
        # "f64" / "packed" for a real-valued matrix: flagged symmetric when it is one to within rounding
        # (undirected graphs), which lets a table too large for on-chip memory be replaced by pair counts
        self.table_only = QpDTable(self.kind, self.C, self._packed.data_ptr())
        self.grid = QpGridMetric()
        self.grid_scale = 0.0
        if self.kind == QP_DT_U16:
            check(lib().qp_dtable_grid_metric(ptr(self.D), self.C, C.byref(self.grid)))
        else:
            # one real weight on every link (the shipped example's two-core mesh): hop counts by closed form, times it
            scale = C.c_double(0.0)
            check(lib().qp_dtable_scaled_grid_metric(ptr(self.D), self.C, SCALED_GRID_TOLERANCE, C.byref(self.grid), C.byref(scale)))
            self.grid_scale = scale.value
        self.symmetric = lib().qp_dtable_symmetric(ptr(self.D), self.C, SYMMETRY_TOLERANCE)
        check(min(self.symmetric, 0))
        self.packed = QpDTable(self.kind, self.C, self._packed.data_ptr(), self.grid,
                               self.symmetric if self.kind == QP_DT_F64 else 0, 0, self.grid_scale)
        if self.kind == QP_DT_F64:
            self._f64 = self._packed
        else:
            Dz = self.D.copy()
            np.fill_diagonal(Dz, 0.0)
            buf = np.zeros((n * 8 + 15) // 16 * 16, np.uint8)
            buf[:n * 8] = Dz.view(np.uint8).reshape(-1)
            self._f64 = torch.from_numpy(buf).to(self.device)
        self.f64 = QpDTable(QP_DT_F64, self.C, self._f64.data_ptr(), QpGridMetric(), self.symmetric)

    @classmethod
    def from_network(cls, network, device=None, weighted=True):
        from .multicore import graph_tables
        nodes, eu, ev, w = graph_tables(network)
        D = apsp(len(nodes), eu, ev, w) if weighted else hops(len(nodes), eu, ev)
        t = cls(D, device)
        t.nodes = nodes
        return t


# ------------------------------------------------------------------------------------------------
# launches
# ------------------------------------------------------------------------------------------------

class Engine:
    """Launches on one device.  Holds a growable workspace so repeated calls allocate nothing."""

    def __init__(self, device=None):
        self.device = device if device is not None else default_device()
        self._ws = None

    def _on_device(self):
        """The ABI acts on the current HIP device (grid sizing, occupancy queries, launches): make the
        engine's device current for the duration of a call."""
        return _torch().cuda.device(self.device)

    def _workspace(self, nbytes):
        torch = _torch()
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self._ws

    def _check_partitions(self, dc: DeviceCircuit, A):
        torch = _torch()
        if A.dtype != torch.uint8 or A.dim() != 3 or not A.is_contiguous() or A.device != self.device:
            raise ValueError("A must be a contiguous uint8 [B, Tp, ldq] tensor on the engine's device")
        B, Tp, ldq = A.shape
        if ldq % 16 or ldq < dc.ir.Q:
            raise ValueError("row pitch must be a multiple of 16 and >= Q")
        return B, Tp, ldq

    def cost_v3(self, dc: DeviceCircuit, dt: DistanceTable, A, out=None, table="packed"):
        """HQA-family cost of B dense partitions; A is uint8 [B, Tp, ldq] on the device."""
        torch = _torch()
        B, Tp, ldq = self._check_partitions(dc, A)
        if out is None:
            out = torch.empty(B, dtype=torch.float64, device=self.device)
        ws = self._workspace(lib().qp_slice_cost_workspace(dc.ir.T, B))
        tab = {"packed": dt.packed, "table": dt.table_only, "f64": dt.f64}[table]
        with self._on_device():
            check(lib().qp_slice_cost_v3(C.byref(dc.c), C.byref(tab), A.data_ptr(), B, Tp, ldq, out.data_ptr(),
                                         ws.data_ptr(), ws.numel(), _stream_ptr(self.device)))
        return out

    def cost_v3_slices(self, dc: DeviceCircuit, dt: DistanceTable, A, t_begin=0, t_end=None, row0=0, out=None, table="packed"):
        """What each timeslice t_begin <= t < t_end adds to the HQA-family cost: float64 [B, t_end - t_begin].
        Row r of a partition in ``A`` is the assignment of slice ``row0 + r`` (a window of the circuit is enough:
        it must hold row t_begin - 1, from slice 2 on, to row t_end - 1)."""
        torch = _torch()
        B, Tp, ldq = self._check_partitions(dc, A)
        t_end = dc.ir.T if t_end is None else t_end
        if out is None:
            out = torch.empty((B, max(t_end - t_begin, 0)), dtype=torch.float64, device=self.device)
        if t_end <= t_begin:
            return out
        tab = {"packed": dt.packed, "table": dt.table_only, "f64": dt.f64}[table]
        with self._on_device():
            check(lib().qp_slice_cost_v3_slices(C.byref(dc.c), C.byref(tab), A.data_ptr(), B, Tp, ldq, row0, t_begin, t_end,
                                                out.data_ptr(), _stream_ptr(self.device)))
        return out

    def cost_v3_multi(self, dc: DeviceCircuit, tables, tab_index, A, out=None):
        """HQA-family cost of B dense partitions, partition b with the distance table ``tables[tab_index[b]]``:
        ``tables`` float64 [S, C, C] and ``tab_index`` int32 [B] on the device (zero diagonals, inf allowed)."""
        torch = _torch()
        B, Tp, ldq = self._check_partitions(dc, A)
        if tables.dtype != torch.float64 or tables.dim() != 3 or tables.shape[1] != tables.shape[2] or not tables.is_contiguous():
            raise ValueError("tables must be a contiguous float64 [S, C, C] tensor")
        if tab_index.dtype != torch.int32 or tab_index.shape != (B,) or not tab_index.is_contiguous():
            raise ValueError("tab_index must be a contiguous int32 [B] tensor")
        if out is None:
            out = torch.empty(B, dtype=torch.float64, device=self.device)
        with self._on_device():
            check(lib().qp_slice_cost_v3_multi(C.byref(dc.c), tables.data_ptr(), tables.shape[0], tables.shape[1],
                                               tab_index.data_ptr(), A.data_ptr(), B, Tp, ldq, out.data_ptr(),
                                               _stream_ptr(self.device)))
        return out

    def cost_v3_static(self, dc: DeviceCircuit, dt: DistanceTable, A, out, table="packed"):
        """cost_v3 through the entry point that allocates nothing: ``out`` is required and the partial sums
        live in the library's own static device array (raises QplaceError when the batch needs more of it
        than there is)."""
        B, Tp, ldq = self._check_partitions(dc, A)
        tab = {"packed": dt.packed, "table": dt.table_only, "f64": dt.f64}[table]
        with self._on_device():
            check(lib().qp_slice_cost_v3_static(C.byref(dc.c), C.byref(tab), A.data_ptr(), B, Tp, ldq, out.data_ptr(),
                                                _stream_ptr(self.device)))
        return out

    def cost_v2(self, dc: DeviceCircuit, dt: DistanceTable, A, variant=QP_V2, out=None, table="packed"):
        """Gate-ordered cost (V1/V2/V4) of B sparse partitions (0xFF = absent)."""
        torch = _torch()
        B, Tp, ldq = self._check_partitions(dc, A)
        if out is None:
            out = torch.empty(B, dtype=torch.float64, device=self.device)
        ws = self._workspace(lib().qp_slice_cost_workspace(dc.ir.T, B))
        tab = dt.packed if table == "packed" else dt.f64
        with self._on_device():
            check(lib().qp_slice_cost_v2(C.byref(dc.c), C.byref(tab), A.data_ptr(), B, Tp, ldq, variant,
                                         out.data_ptr(), ws.data_ptr(), ws.numel(), _stream_ptr(self.device)))
        return out

    def cost_matrix(self, kind, dc: DeviceCircuit, dt: DistanceTable, t, a, ops, n_ops, avail, n_avail, out=None):
        """Batched op x core matrices; all arguments are device tensors (see qp_cost_matrix)."""
        torch = _torch()
        B, max_ops = ops.shape
        if out is None:
            out = torch.empty((B, max_ops, dt.C), dtype=torch.float64, device=self.device)
        with self._on_device():
            check(lib().qp_cost_matrix(kind, C.byref(dc.c), C.byref(dt.f64), B, t.data_ptr(), a.data_ptr(), a.shape[1],
                                       ops.data_ptr(), n_ops.data_ptr(), max_ops, avail.data_ptr(), n_avail.data_ptr(),
                                       out.data_ptr(), _stream_ptr(self.device)))
        return out


    def core_activity(self, dc: DeviceCircuit, A, n_cores: int):
        """Per-core inter-core event counts of B dense partitions: int64 [B, C]."""
        torch = _torch()
        B, Tp, ldq = self._check_partitions(dc, A)
        out = torch.empty((B, n_cores), dtype=torch.int64, device=self.device)
        with self._on_device():
            check(lib().qp_core_activity(C.byref(dc.c), A.data_ptr(), B, Tp, ldq, int(n_cores), out.data_ptr(),
                                         _stream_ptr(self.device)))
        return out

    def comm_matrix(self, dc: DeviceCircuit, A, n_cores: int):
        """Core x core event counts of B sparse partitions: (int32 [B, C, C], int32 status [B])."""
        torch = _torch()
        B, Tp, ldq = self._check_partitions(dc, A)
        out = torch.empty((B, n_cores, n_cores), dtype=torch.int32, device=self.device)
        status = torch.empty(B, dtype=torch.int32, device=self.device)
        with self._on_device():
            check(lib().qp_comm_matrix(C.byref(dc.c), A.data_ptr(), B, Tp, ldq, int(n_cores), out.data_ptr(),
                                       status.data_ptr(), _stream_ptr(self.device)))
        return out, status

    def gtrhqa_place(self, dc: DeviceCircuit, dt: DistanceTable, initial, cap: int, out=None):
        """B GTRHQA placements on the device.  initial: uint8 [B, ldq] device tensor (row 0 of every
        instance, core indices < C); returns uint8 [B, T+1, ldq]."""
        torch = _torch()
        if initial.dtype != torch.uint8 or initial.dim() != 2 or not initial.is_contiguous() or initial.device != self.device:
            raise ValueError("initial must be a contiguous uint8 [B, ldq] tensor on the engine's device")
        B, ldq = initial.shape
        if ldq % 16 or ldq < dc.ir.Q:
            raise ValueError("row pitch must be a multiple of 16 and >= Q")
        if out is None:
            out = torch.empty((B, dc.ir.T + 1, ldq), dtype=torch.uint8, device=self.device)
        ws = self._workspace(lib().qp_gtrhqa_workspace(dt.C))
        with self._on_device():
            check(lib().qp_gtrhqa_place(C.byref(dc.c), C.byref(dt.f64), initial.data_ptr(), B, ldq, int(cap),
                                        out.data_ptr(), ws.data_ptr(), ws.numel(), _stream_ptr(self.device)))
        return out


def rows_to_array(rows, Q, ldq=None, absent=QP_ABSENT) -> np.ndarray:
    """List of rows (core index or None per qubit) -> uint8 [Tp, ldq]."""
    ldq = ldq if ldq is not None else (Q + 15) // 16 * 16
    out = np.zeros((len(rows), ldq), np.uint8)
    for t, r in enumerate(rows):
        out[t, :Q] = [absent if c is None else c for c in r]
    return out


_engines = {}


def get_engine(device=None) -> Engine:
    device = device if device is not None else default_device()
    key = str(device)
    if key not in _engines:
        _engines[key] = Engine(device)
    return _engines[key]
