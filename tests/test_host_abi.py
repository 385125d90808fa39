"""CPU-side checks: the C-ABI library loads and exports what include/qplace.h declares; its host
entry points (distances, hop counts, assignment solver, table packing) and the host IR agree with the
reference-generated fixtures.  No device call is made here."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ALL_CASES, ROOT, SMALL_CASES, load_case
from oracle import placer_oracle as O
from quantum_simulator_amd import _lib, engine
from quantum_simulator_amd.timeslices import CircuitIR, slice_gate_indices


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "qplace.h")).read()
    declared = set(re.findall(r"\b(qp_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in qplace.h but not exported"
    assert declared == set(_lib.EXPORTS), "ctypes table and header disagree"
    assert _lib.lib().qp_version() == 204


@pytest.mark.parametrize("name", ALL_CASES)
def test_apsp_and_hops_bits(name):
    gc = load_case(name)
    eu = np.array([e[0] for e in gc.edges], np.int32)
    ev = np.array([e[1] for e in gc.edges], np.int32)
    w = np.array([e[2] for e in gc.edges], np.float64)
    D = engine.apsp(gc.C, eu, ev, w)
    assert D.tobytes() == gc.D.tobytes()
    H = engine.hops(gc.C, eu, ev)
    assert (H == gc.hops).all()


def test_apsp_disconnected_and_inf_edges():
    # two components: weighted search fails, hop count fails too -> inf (Placer.py:71-76)
    D = engine.apsp(4, [0, 2], [1, 3], [1.5, 2.5])
    assert D[0, 1] == 1.5 and D[2, 3] == 2.5 and np.isinf(D[0, 2]) and D[1, 1] == 0
    # an inf-weight edge is still a path of length inf for the weighted search
    D = engine.apsp(3, [0, 1], [1, 2], [1.0, float("inf")])
    assert np.isinf(D[0, 2]) and D[0, 1] == 1.0
    assert np.array(O.distance_matrix(3, [(0, 1, 1.0), (1, 2, float("inf"))])).tobytes() == D.tobytes()


@pytest.mark.parametrize("name", ALL_CASES)
def test_lsap_records(name):
    gc = load_case(name)
    for tag, cm, r, c in gc.lsap_samples():
        rr, cc = engine.lsap(cm)
        assert rr.tolist() == r.tolist() and cc.tolist() == c.tolist(), (tag, cm.shape)


def test_lsap_random_vs_scipy_and_batch():
    opt = pytest.importorskip("scipy.optimize")
    rng = np.random.default_rng(11)
    mats = []
    for trial in range(400):
        nr, nc = rng.integers(1, 12, size=2)
        kind = trial % 4
        if kind == 0:
            cm = rng.integers(0, 3, size=(nr, nc)).astype(float)
        elif kind == 1:
            cm = np.maximum(0.1, rng.normal(0.3, 0.6, size=(nr, nc)))
        elif kind == 2:
            cm = rng.random((nr, nc))
        else:
            cm = rng.integers(0, 2, size=(nr, nc)) * 0.45 + 0.1
        r, c = opt.linear_sum_assignment(cm)
        rr, cc = engine.lsap(cm)
        assert rr.tolist() == r.tolist() and cc.tolist() == c.tolist(), cm
        mats.append((cm, r, c))
    R = max(m.shape[0] for m, _, _ in mats)
    L = max(m.shape[1] for m, _, _ in mats)
    costs = np.zeros((len(mats), R, L))
    for k, (m, _, _) in enumerate(mats):
        costs[k, :m.shape[0], :m.shape[1]] = m
    rows, cols, status = engine.lsap_batch(costs, [m.shape[0] for m, _, _ in mats], [m.shape[1] for m, _, _ in mats], 4)
    assert (status == 0).all()
    for k, (m, r, c) in enumerate(mats):
        n = min(m.shape)
        assert rows[k, :n].tolist() == r.tolist() and cols[k, :n].tolist() == c.tolist()


def test_lsap_infeasible_and_invalid():
    with pytest.raises(ValueError):
        engine.lsap(np.array([[np.inf, np.inf], [1.0, 2.0]]))
    with pytest.raises(ValueError):
        engine.lsap(np.array([[np.nan, 1.0], [1.0, 2.0]]))
    r, c = engine.lsap(np.zeros((0, 3)))
    assert len(r) == 0


def test_dtable_pack_kinds():
    import ctypes as C
    D = np.array([[0, 2, 4], [2, 0, 2], [4, 2, 0]], float)
    st = np.zeros(9 * 8, np.uint8)
    kind, nb = C.c_int32(), C.c_size_t()
    _lib.check(_lib.lib().qp_dtable_pack(_lib.ptr(D), 3, _lib.ptr(st), st.size, C.byref(kind), C.byref(nb)))
    assert kind.value == _lib.QP_DT_U16 and nb.value == 18
    assert st[:18].view(np.uint16).tolist() == [0, 2, 4, 2, 0, 2, 4, 2, 0]
    D[0, 1] = 2.5
    D[1, 1] = 7.0  # diagonal is forced to zero
    _lib.check(_lib.lib().qp_dtable_pack(_lib.ptr(D), 3, _lib.ptr(st), st.size, C.byref(kind), C.byref(nb)))
    assert kind.value == _lib.QP_DT_F64 and nb.value == 72
    assert st.view(np.float64)[4] == 0.0 and st.view(np.float64)[1] == 2.5
    D[0, 2] = np.inf
    _lib.check(_lib.lib().qp_dtable_pack(_lib.ptr(D), 3, _lib.ptr(st), st.size, C.byref(kind), C.byref(nb)))
    assert kind.value == _lib.QP_DT_F64


@pytest.mark.parametrize("name", ALL_CASES)
def test_product_timeslices_match_reference(name):
    gc = load_case(name)
    assert slice_gate_indices(gc.gate_q) == gc.slices


@pytest.mark.parametrize("name", SMALL_CASES)
def test_ir_tables(name):
    gc = load_case(name)
    ir = CircuitIR(gc.Q, gc.gate_q, gc.gate_mr)
    assert ir.T == gc.T
    # lookahead slots reproduce the attraction term for arbitrary assignments
    rng = np.random.default_rng(3)
    a = rng.integers(0, gc.C, size=gc.Q).tolist()
    w = [0.5, 0.5, 0.25, 0.25, 0.125, 0.125]
    for e in range(ir.nE):
        u, v, t = int(ir.edge_u[e]), int(ir.edge_v[e]), int(ir.edge_t[e])
        for core in range(min(gc.C, 4)):
            want = O.attraction(u, v, core, a, gc.slices, gc.gate_q, t)
            got = 0.0
            for k in range(6):
                o = int(ir.lookahead[e, k])
                if o >= 0:
                    got += w[k] * 0.9 if a[o] == core else w[k] * 0.3
            assert got == want
    # visits: one per (gate, qubit) in slice order, prev_t is the previous slice touching the qubit
    last = {}
    k = 0
    for t, layer in enumerate(gc.slices):
        for g in layer:
            for q in gc.gate_q[g]:
                rec = ir.visit_records[k]
                assert (rec["q"], rec["t"], rec["prev_t"]) == (q, t, last.get(q, -1))
                k += 1
            for q in gc.gate_q[g]:
                last[q] = t
    assert k == ir.nV
    # the same per (slice, qubit), as the row walk of the gate-ordered cost reads it
    ldq = (gc.Q + 15) // 16 * 16
    vm, vp = ir.visit_mask(ldq), ir.visit_prev(ldq)
    assert vm.shape == vp.shape == (ir.T, ldq) and vm.dtype == np.uint8 and vp.dtype == np.int16
    seen = np.zeros((ir.T, ldq), bool)
    seen[ir.visit_records["t"], ir.visit_records["q"]] = True
    assert ((vm == 0xFF) == seen).all() and ((vm == 0) == ~seen).all()
    assert (vp[ir.visit_records["t"], ir.visit_records["q"]] == ir.visit_records["prev_t"]).all()
    for q in range(ldq):
        last = -1
        for t in range(ir.T):
            assert vp[t, q] == last
            if seen[t, q]:
                last = t


def test_qasm_reader_matches_fixture_tables():
    """Where the shipped circuits are readable, the QASM front end yields the fixtures' gate tables."""
    ref = "/root/reference/circuits"
    if not os.path.isdir(ref):
        pytest.skip("reference circuits not present on this machine")
    from quantum_simulator_amd import circuit_from_qasm
    for name in ("shor5_mesh2x2_u", "bigadder_mesh4x4_u", "qft63_torus8x8_u"):
        gc = load_case(name)
        ir = CircuitIR.from_circuit(circuit_from_qasm(os.path.join(ref, gc.meta["qasm"]), maxwidth=1024))
        assert ir.gate_q == gc.gate_q and ir.gate_mr == gc.gate_mr and ir.slices == gc.slices
        assert [str(q) for q in ir.qubits] == gc.meta["qubits"]


@pytest.mark.parametrize("name", ALL_CASES)
def test_native_timeslices_match_reference(name):
    """qp_timeslices reproduces the reference's slices, order inside each slice included."""
    from quantum_simulator_amd.timeslices import native_slices
    gc = load_case(name)
    q0 = np.array([g[0] for g in gc.gate_q], np.int32)
    q1 = np.array([g[1] if len(g) == 2 else -1 for g in gc.gate_q], np.int32)
    order, ptr = native_slices(gc.Q, q0, q1)
    got = [order[ptr[t]:ptr[t + 1]].tolist() for t in range(len(ptr) - 1)]
    assert got == gc.slices


def test_native_timeslices_random_vs_python_sets():
    """Random circuits, including gate indices that collide in CPython's 8-slot set table."""
    from quantum_simulator_amd.timeslices import native_slices
    rng = np.random.default_rng(2)
    for trial in range(60):
        Q = int(rng.integers(2, 40))
        n = int(rng.integers(1, 600))
        q0 = rng.integers(0, Q, size=n).astype(np.int32)
        q1 = rng.integers(0, Q, size=n).astype(np.int32)
        one = (rng.random(n) < 0.4) | (q1 == q0)
        q1[one] = -1
        gate_q = [(int(a),) if b < 0 else (int(a), int(b)) for a, b in zip(q0, q1)]
        order, ptr = native_slices(Q, q0, q1)
        got = [order[ptr[t]:ptr[t + 1]].tolist() for t in range(len(ptr) - 1)]
        assert got == slice_gate_indices(gate_q), trial
    with pytest.raises(_lib.QplaceError):
        native_slices(4, np.array([0], np.int32), np.array([0], np.int32))  # both operands the same qubit


def test_scheduled_gates_are_a_padded_per_slice_permutation():
    from quantum_simulator_amd import layered_gate_arrays
    q0, q1, _ = layered_gate_arrays(300, 12, 0.5, seed=8)
    keep = np.random.default_rng(1).random(len(q0)) > 0.3
    ir = CircuitIR.from_gate_tables(300, q0[keep], q1[keep])
    plain = ir.edge_u.astype(np.uint32) | (ir.edge_v.astype(np.uint32) << 16)
    sptr, sched = ir.scheduled_edges()
    ch, tail = _lib.QP_SCHED_CHUNK, _lib.QP_SCHED_TAIL
    assert sptr[0] == 0 and len(sched) == sptr[-1] + tail and not sched[sptr[-1]:].any()
    worse = 0
    for t in range(ir.T):
        a, b = ir.slice_edge_ptr[t], ir.slice_edge_ptr[t + 1]
        s0, s1 = sptr[t], sptr[t + 1]
        assert s1 - s0 == (b - a + ch - 1) // ch * ch
        assert sorted(plain[a:b].tolist()) == sorted(sched[s0:s0 + (b - a)].tolist())
        assert not sched[s0 + (b - a):s1].any()  # padding: qubit 0 with itself
        for arr, lo, sign in ((plain, a, 1), (sched, s0, -1)):
            banks = ((arr[lo:lo + 32] & 0xFFFF) >> 2) & 31
            worse += sign * (32 - len(set(banks.tolist())))
    assert worse > 0, "the schedule should spread the first 32 gates' endpoints over more banks"
    with pytest.raises(_lib.QplaceError):  # too little room
        eu, ev = ir.edge_u.astype(np.uint16), ir.edge_v.astype(np.uint16)
        sep = np.ascontiguousarray(ir.slice_edge_ptr, np.int32)
        _lib.check(_lib.lib().qp_schedule_gates(ir.T, _lib.ptr(sep), _lib.ptr(eu), _lib.ptr(ev), _lib.ptr(sptr.copy()),
                                                _lib.ptr(sched), len(sched) - 1))


@pytest.mark.parametrize("Q", [5, 40, 64, 100, 270, 512])
def test_grouped_gates_are_a_padded_per_group_permutation(Q):
    """Short rows: the gates of the G slices a wave takes per step, endpoints as byte offsets into those rows."""
    from quantum_simulator_amd import layered_gate_arrays
    T = 45
    q0, q1, _ = layered_gate_arrays(Q, T, 0.5, seed=Q)
    keep = np.random.default_rng(Q).random(len(q0)) > 0.3
    ir = CircuitIR.from_gate_tables(Q, q0[keep], q1[keep])
    ldq = (Q + 15) // 16 * 16
    G = _lib.lib().qp_group_rows(ldq)
    assert G == 64 // (ldq // 16)
    gptr, sched = ir.grouped_edges(ldq)
    ch, tail = _lib.QP_SCHED_CHUNK, _lib.QP_SCHED_TAIL
    ng = (ir.T + G - 1) // G
    assert len(gptr) == ng + 1 and gptr[0] == 0 and len(sched) == gptr[-1] + tail and not sched[gptr[-1]:].any()
    for j in range(ng):
        want = []
        for t in range(j * G, min(ir.T, (j + 1) * G)):
            off = (t - j * G) * ldq
            for e in range(ir.slice_edge_ptr[t], ir.slice_edge_ptr[t + 1]):
                want.append((off + int(ir.edge_u[e])) | ((off + int(ir.edge_v[e])) << 16))
        s0, s1 = gptr[j], gptr[j + 1]
        assert s1 - s0 == (len(want) + ch - 1) // ch * ch
        assert sorted(want) == sorted(sched[s0:s0 + len(want)].tolist())
        assert not sched[s0 + len(want):s1].any()  # padding: byte 0 with itself
    assert _lib.lib().qp_group_rows(528) == 0 and _lib.lib().qp_group_rows(24) == 0
    assert ir.grouped_edges(1024) is None
    with pytest.raises(_lib.QplaceError):  # too little room
        eu, ev = ir.edge_u.astype(np.uint16), ir.edge_v.astype(np.uint16)
        sep = np.ascontiguousarray(ir.slice_edge_ptr, np.int32)
        _lib.check(_lib.lib().qp_schedule_gate_groups(ir.T, _lib.ptr(sep), _lib.ptr(eu), _lib.ptr(ev), ldq,
                                                      _lib.ptr(gptr.copy()), _lib.ptr(sched), len(sched) - 1))


def _grid_metric(D):
    g = _lib.QpGridMetric()
    D = np.ascontiguousarray(D, np.float64)
    _lib.check(_lib.lib().qp_dtable_grid_metric(_lib.ptr(D), D.shape[0], ctypes.byref(g)))
    return g


def test_grid_metric_detection():
    """A matrix is accepted as a grid metric only if every entry equals the closed form."""
    import random
    from quantum_simulator_amd.engine import apsp
    from quantum_simulator_amd.multicore import multicore, graph_tables

    def table(topo, nx, ny, w):
        random.seed(3)
        arch = multicore(nx, ny, ncp_qubits=2, edge_weight_range=w, network_type=topo)
        nodes, eu, ev, ww = graph_tables(arch.network)
        return apsp(len(nodes), eu, ev, ww)

    g = _grid_metric(table("mesh", 4, 4, (2.0, 2.0)))
    assert (g.valid, g.xbits, g.wx, g.wy, g.wrap_x, g.wrap_y) == (1, 2, 2, 2, 0, 0)
    g = _grid_metric(table("mesh", 16, 16, (2.0, 2.0)))
    assert (g.valid, g.xbits, g.wx, g.wy) == (1, 4, 2, 2)
    g = _grid_metric(table("torus", 8, 8, (3.0, 3.0)))
    assert (g.valid, g.xbits, g.wx, g.wy, g.wrap_x, g.wrap_y) == (1, 3, 3, 3, 8, 8)
    g = _grid_metric(table("line", 9, 1, (1.0, 1.0)))
    assert g.valid and g.wrap_x == 0 and g.wrap_y == 0
    D = table("mesh", 4, 4, (2.0, 2.0))
    want = np.array([[g2 for g2 in row] for row in D])
    for bad in (table("mesh", 4, 4, (1.0, 3.0)), table("mesh", 3, 3, (2.0, 2.0)), table("star", 6, 1, (2.0, 2.0)),
                table("all_to_all", 4, 4, (1.0, 1.0))):
        assert not _grid_metric(bad).valid
    D[3, 9] += 1.0  # one entry off the formula
    assert not _grid_metric(D).valid
    # the closed form reproduces the matrix it was accepted for
    g = _grid_metric(want)
    W = 1 << g.xbits
    for a in range(16):
        for b in range(16):
            assert want[a, b] == g.wx * abs(a % W - b % W) + g.wy * abs(a // W - b // W)


TRICKY_QASM = '''OPENQASM 2.0; // header; with semicolon
include "qelib1.inc";
qreg q[3]; qreg r[2]; creg c[3];
gate foo a,b { cx a,b; h a; }
h q; // broadcast
cx q[0],q[1]; rz(pi/2) q[2]; u3(0.1,(0.2),0.3) r[1];
foo q[0], r[0];
ccx q[0],q[1],q[2];
barrier q;
cx q[1], // split over lines; really
   q[2];
measure q -> c;
if(c==3) x r[0];
reset r[1];
swap q [ 0 ] , r [ 1 ] ;
'''


def _commands(circ):
    return [(x.op.name, x.op.params, [str(q) for q in x.qubits], x.op.type, x.bits) for x in circ]


def test_native_qasm_reader_equals_python_reader():
    """Runs of plain gate statements go through qp_qasm_scan; everything else (declarations, broadcasts,
    user gates, measure, reset, barrier, conditionals, comments in odd places) through the Python reader.
    Both routes must give the same commands in the same order, and the same integer tables."""
    from quantum_simulator_amd import circuit_to_qasm_str, layered_circuit
    from quantum_simulator_amd.circuit import QasmError, circuit_from_qasm_str
    a, b = circuit_from_qasm_str(TRICKY_QASM), circuit_from_qasm_str(TRICKY_QASM, native=False)
    assert _commands(a) == _commands(b) and len(a) == len(b) == 16
    ia, ib = CircuitIR.from_circuit(a), CircuitIR.from_circuit(b)
    assert ia.gate_q == ib.gate_q and ia.gate_mr == ib.gate_mr and ia.slices == ib.slices
    assert [[c.op.name for c in sl] for sl in ia.command_slices()] == [[c.op.name for c in sl] for sl in ib.command_slices()]
    text = circuit_to_qasm_str(layered_circuit(70, 40, 0.5, seed=3))
    a, b = circuit_from_qasm_str(text), circuit_from_qasm_str(text, native=False)
    assert _commands(a) == _commands(b)
    ia, ib = CircuitIR.from_circuit(a), CircuitIR.from_circuit(b)
    assert ia.gate_q == ib.gate_q and ia.slices == ib.slices and ia.edge_records.tobytes() == ib.edge_records.tobytes()
    for bad in ("qreg q[2]; cx q[0],q[5];", "qreg q[2]; cx q[0] q[1];", "qreg q[2]; h p[0];", "qreg q[2]; h q[0]"):
        for native in (True, False):
            with pytest.raises(QasmError):
                circuit_from_qasm_str(bad, native=native)


def test_qasm_scan_stops_where_it_should():
    text = b"h q[0]; cx q[0],q[1];\n// note\nrz(0.5*(pi)) q[1]; measure q[0] -> c[0]; x q[0];"
    n = len(text)
    sizes, base = np.array([2], np.int32), np.array([0], np.int32)
    h = np.empty(8, np.uint64)
    npos, ppos, qidx = np.empty((8, 2), np.int32), np.empty((8, 2), np.int32), np.empty((8, 3), np.int32)
    cnt, stop = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.check(_lib.lib().qp_qasm_scan(text, n, 0, 1, b"q\0", _lib.ptr(sizes), _lib.ptr(base), 8, _lib.ptr(h), _lib.ptr(npos),
                                       _lib.ptr(ppos), _lib.ptr(qidx), ctypes.byref(cnt), ctypes.byref(stop)))
    assert cnt.value == 3 and text[stop.value:].startswith(b"measure")
    assert qidx[:3].tolist() == [[0, -1, -1], [0, 1, -1], [1, -1, -1]]
    assert [text[a:a + l] for a, l in npos[:3].tolist()] == [b"h", b"cx", b"rz"]
    assert ppos[0].tolist() == [-1, -1] and text[ppos[2, 0]:ppos[2, 0] + ppos[2, 1]] == b"0.5*(pi)"
    assert h[0] != h[1]


def test_dtable_symmetric():
    """qp_dtable_symmetric: rounding-level asymmetry passes, anything larger does not; infinities must
    match; NaN never passes; bad arguments are an error, not a verdict."""
    L = _lib.lib()
    rng = np.random.default_rng(5)
    n = 40
    M = rng.uniform(1, 90, size=(n, n))
    S = (M + M.T) / 2
    np.fill_diagonal(S, 0.0)
    assert L.qp_dtable_symmetric(_lib.ptr(S), n, 1e-13) == 1
    R = S * (1.0 + rng.uniform(-2e-16, 2e-16, size=(n, n)))  # last-place noise, as path sums from either end have
    assert L.qp_dtable_symmetric(_lib.ptr(R), n, 1e-13) == 1
    assert L.qp_dtable_symmetric(_lib.ptr(R), n, 0.0) == 0
    X = S.copy()
    X[3, 7] *= 1.0 + 1e-11
    assert L.qp_dtable_symmetric(_lib.ptr(X), n, 1e-13) == 0
    I = S.copy()
    I[2, 5] = I[5, 2] = np.inf
    assert L.qp_dtable_symmetric(_lib.ptr(I), n, 1e-13) == 1
    I[5, 2] = 4.0
    assert L.qp_dtable_symmetric(_lib.ptr(I), n, 1e-13) == 0
    N = S.copy()
    N[1, 2] = N[2, 1] = np.nan
    assert L.qp_dtable_symmetric(_lib.ptr(N), n, 1e-13) == 0
    assert L.qp_dtable_symmetric(None, n, 1e-13) < 0
    assert L.qp_dtable_symmetric(_lib.ptr(S), n, ctypes.c_double(-1.0)) < 0
    # distances over an undirected weighted graph, as the placers make them: accepted
    from quantum_simulator_amd.engine import apsp
    eu = list(range(30)) + [0, 4, 9]
    ev = [(i + 1) % 30 for i in range(30)] + [15, 20, 27]
    D = apsp(30, eu, ev, rng.uniform(1, 3, size=len(eu)))
    assert L.qp_dtable_symmetric(_lib.ptr(D), 30, 1e-13) == 1


def test_make_space_native_equals_python():
    """qp_make_space (the batch loop of HQA / TRHQA placements) against placers._HqaFamily._make_space, which
    the single-instance path uses and the reference fixtures pin: random over-full assignments, ties in
    distance and occupancy (integer distances), rows that need nothing."""
    from quantum_simulator_amd.placers import hqa_placer
    L = _lib.lib()
    rng = np.random.default_rng(12)
    for trial in range(200):
        C = int(rng.integers(2, 40))
        cap = int(rng.integers(2, 7))
        Q = int(rng.integers(1, C * cap + 1))
        lda = Q + int(rng.integers(0, 5))
        n = int(rng.integers(1, 6))
        D = rng.integers(1, 5, size=(C, C)).astype(np.float64) if trial % 2 else rng.uniform(1, 9, size=(C, C))
        np.fill_diagonal(D, 0.0)
        a = np.zeros((n, lda), np.uint8)
        for r in range(n):  # any assignment within capacity; crowded ones are what the step is for
            slots = np.repeat(np.arange(C), cap)
            a[r, :Q] = rng.permutation(slots)[:Q] if r % 2 else np.sort(rng.permutation(slots)[:Q])
        occ = np.stack([np.bincount(a[r, :Q], minlength=C) for r in range(n)]).astype(np.int64)
        rows = np.flatnonzero(rng.random(n) < 0.7).astype(np.int64)
        want_a, want_occ = a.copy(), occ.copy()
        for r in rows:
            hqa_placer._make_space(want_a[r, :Q], want_occ[r], cap, C, D)
        got_a, got_occ = a.copy(), occ.copy()
        assert L.qp_make_space(_lib.ptr(got_a), _lib.ptr(got_occ), _lib.ptr(rows) if rows.size else None, rows.size,
                               Q, lda, C, cap, _lib.ptr(D)) == 0
        assert (got_a == want_a).all() and (got_occ == want_occ).all(), trial
    assert L.qp_make_space(None, None, None, 0, 1, 1, 1, 2, None) < 0


def test_apsp_batch_equals_one_at_a_time():
    """qp_apsp_f64_batch (threaded) == qp_apsp_f64 per network, bit for bit, including links of infinite weight."""
    from quantum_simulator_amd import engine
    rng = np.random.default_rng(5)
    C = 12
    eu = list(range(C)) + [0, 3, 7]
    ev = [(i + 1) % C for i in range(C)] + [6, 9, 11]
    W = rng.uniform(1, 3, size=(9, len(eu)))
    W[1, 2] = W[1, 5] = np.inf
    W[4, :] = np.inf
    W[6] = np.floor(W[6])
    got = engine.apsp_batch(C, eu, ev, W, threads=4)
    for s in range(W.shape[0]):
        assert got[s].tobytes() == engine.apsp(C, eu, ev, W[s]).tobytes()
    assert np.isinf(got[4][0, 1]) and got[4][2, 2] == 0


def test_host_threads_are_divided_by_the_ranks_on_the_node(monkeypatch):
    """Every default thread count comes from qp_host_threads: the CPUs this process may run on divided by
    LOCAL_WORLD_SIZE (the ranks torch.distributed.run started on the node), polling lanes half of that, never
    less than one; qp_set_host_threads overrides, 0 restores."""
    import os
    from quantum_simulator_amd import _lib
    L = _lib.lib()
    cpus = len(os.sched_getaffinity(0))
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    assert L.qp_set_host_threads(0) == 0
    assert _lib.host_threads(1 << 20) == cpus
    assert _lib.host_threads(3) == min(3, cpus)
    assert _lib.host_threads(8, polling=True) == max(1, min(8, cpus // 2))
    for ranks in (2, 8, 64):
        monkeypatch.setenv("LOCAL_WORLD_SIZE", str(ranks))
        share = max(1, cpus // ranks)
        assert _lib.host_threads(1 << 20) == share
        assert _lib.host_threads(16) == min(16, share)
        assert _lib.host_threads(8, polling=True) == max(1, min(8, share // 2))
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "not a number")
    assert _lib.host_threads(1 << 20) == cpus
    try:
        assert L.qp_set_host_threads(5) == 0
        assert _lib.host_threads(16) == 5 and _lib.host_threads(2) == 2 and _lib.host_threads(8, polling=True) == 2
        assert L.qp_set_host_threads(-1) < 0
    finally:
        L.qp_set_host_threads(0)


def test_bench_starts_its_own_ranks(monkeypatch):
    """``python bench.py --gpus N`` with no launcher around it: the parent builds a torch.distributed.run command
    for N ranks on 127.0.0.1 with the same arguments, runs it as a child and returns its exit code."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert bench.launch_ranks(4) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 0 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    k = cmd.index(os.path.join(root, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # and main() takes that road only when no launcher has set WORLD_SIZE
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 7


def test_scaled_grid_metric_detection():
    """qp_dtable_scaled_grid_metric: one real weight on every link of a mesh / line / torus whose row length is a
    power of two gives scale = that weight and the hop metric; integer matrices (qp_dtable_grid_metric's own
    case), random weights, other topologies and matrices with an infinite entry are refused."""
    import random
    from quantum_simulator_amd.multicore import graph_tables, multicore
    L = _lib.lib()

    def probe(D):
        g, s = _lib.QpGridMetric(), ctypes.c_double(-1.0)
        D = np.ascontiguousarray(D, np.float64)
        assert L.qp_dtable_scaled_grid_metric(_lib.ptr(D), D.shape[0], 1e-13, ctypes.byref(g), ctypes.byref(s)) == 0
        return g, s.value

    def table(topo, nx, ny, wr):
        random.seed(5)
        arch = multicore(nx, ny, ncp_qubits=4, edge_weight_range=wr, network_type=topo)
        nodes, eu, ev, w = graph_tables(arch.network)
        return engine.apsp(len(nodes), eu, ev, w)

    for topo, nx, ny, w, xbits, wrap in (("mesh", 2, 1, 1.8444218515250481, 0, (0, 0)), ("mesh", 4, 4, 0.1, 2, (0, 0)),
                                         ("mesh", 16, 16, 1.0 / 3.0, 4, (0, 0)), ("torus", 8, 8, 1.37, 3, (8, 8)),
                                         ("line", 64, 1, 2.5, None, (0, 0))):
        g, s = probe(table(topo, nx, ny, (w, w)))
        assert g.valid == 1 and s == w and (g.wrap_x, g.wrap_y) == wrap, (topo, g.valid, s)
        if nx > 2 and ny > 2:
            assert g.xbits == xbits and g.wx == 1 and g.wy == 1
    for D in (table("mesh", 4, 4, (2.0, 2.0)), table("mesh", 4, 4, (1.0, 3.0)), table("mesh", 3, 3, (1.5, 1.5)),
              table("star", 8, 1, (1.5, 1.5)), table("all_to_all", 2, 2, (1.5, 1.5))):
        g, s = probe(D)
        assert g.valid == 0 and s == 0.0, (g.valid, s)
    Dinf = table("mesh", 4, 4, (1.5, 1.5))
    Dinf[0, 5] = np.inf
    g, s = probe(Dinf)
    assert g.valid == 0 and s == 0.0
    assert L.qp_dtable_scaled_grid_metric(None, 4, 1e-13, None, None) < 0
