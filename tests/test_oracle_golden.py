"""The CPU restatement (oracle/placer_oracle.py) against fixtures produced by the reference itself."""
import random

import numpy as np
import pytest

import os

from conftest import ALL_CASES, ROOT, SMALL_CASES, load_case
from oracle import placer_oracle as O

BIG = [c for c in ALL_CASES if c not in SMALL_CASES]


def _rows_eq(rows, arr):
    got = np.array([[0xFF if c is None else c for c in r] for r in rows], np.uint8)
    return got.shape == arr.shape and bool((got == arr).all())


@pytest.mark.parametrize("name", ALL_CASES)
def test_timeslices_match_reference(name):
    gc = load_case(name)
    assert O.timeslices(gc.gate_q) == gc.slices


@pytest.mark.parametrize("name", ALL_CASES)
def test_distance_matrix_bits(name):
    gc = load_case(name)
    D = np.array(O.distance_matrix(gc.C, gc.edges))
    assert D.tobytes() == gc.D.tobytes()
    H = np.array(O.hop_matrix(gc.C, gc.edges))
    assert (H == gc.hops).all()


@pytest.mark.parametrize("name", ALL_CASES)
def test_cost_v3_bits(name):
    gc = load_case(name)
    D = gc.D.tolist()
    for p in ("HQA", "TRHQA", "GTRHQA"):
        if gc.has(p):
            got = O.cost_v3(gc.slices, gc.gate_q, gc.gate_mr, gc.part(p).tolist(), D)
            assert float(got).hex() == gc.meta["placers"][p]["cost_hex"], p


@pytest.mark.parametrize("name", ALL_CASES)
def test_cost_v2_and_v1_bits(name):
    gc = load_case(name)
    D = gc.D.tolist()
    for p in ("BFS", "RANDOM"):
        pm = gc.meta["placers"][p]
        rows = gc.part_rows(p)
        try:
            got = O.cost_v2(gc.slices, gc.gate_q, gc.gate_mr, rows, D)
            assert pm["cost_error"] is None
            assert float(got).hex() == pm["cost_hex"], p
        except O.CostError as e:
            assert pm["cost_error"] == str(e)
        try:
            got = O.cost_v1(gc.slices, gc.gate_q, gc.gate_mr, rows, D)
            assert pm["v1_error"] is None
            assert float(got) == gc.cost_v1(p), p
        except O.CostError as e:
            assert pm["v1_error"] == str(e)
    for p in ("HQA", "GTRHQA"):
        if gc.has(p):
            got = O.cost_v1(gc.slices, gc.gate_q, gc.gate_mr, gc.part_rows(p), D)
            assert float(got) == gc.cost_v1(p), p


@pytest.mark.parametrize("name", ALL_CASES)
def test_lsap_matches_scipy_records(name):
    gc = load_case(name)
    n = 0
    for tag, cm, r, c in gc.lsap_samples():
        rr, cc = O.lsap(cm.tolist())
        assert rr == r.tolist() and cc == c.tolist(), (tag, cm.shape)
        n += 1
    if gc.has("HQA") and gc.meta["placers"]["HQA"]["n_lsap"]:
        assert n > 0


def test_lsap_random_against_scipy():
    scipy_opt = pytest.importorskip("scipy.optimize")
    rng = np.random.default_rng(5)
    for trial in range(300):
        nr, nc = rng.integers(1, 9, size=2)
        if trial % 3 == 0:
            cm = rng.integers(0, 3, size=(nr, nc)).astype(float)  # heavy ties
        elif trial % 3 == 1:
            cm = np.maximum(0.1, rng.normal(0.3, 0.5, size=(nr, nc)))  # clamp ties like the placers
        else:
            cm = rng.random((nr, nc))
        r, c = scipy_opt.linear_sum_assignment(cm)
        rr, cc = O.lsap(cm.tolist())
        assert rr == r.tolist() and cc == c.tolist(), cm


@pytest.mark.parametrize("name", SMALL_CASES)
@pytest.mark.parametrize("placer", ["HQA", "TRHQA", "GTRHQA"])
def test_place_hqa_family_small(name, placer):
    _check_place(name, placer)


@pytest.mark.slow
@pytest.mark.parametrize("name", [c for c in BIG if "gen256" not in c])
@pytest.mark.parametrize("placer", ["HQA", "TRHQA", "GTRHQA"])
def test_place_hqa_family_big(name, placer):
    _check_place(name, placer)


def _check_place(name, placer):
    gc = load_case(name)
    if not gc.has(placer):
        pytest.skip("capacity too small for this placer in the fixture")
    D = gc.D.tolist()
    if placer == "GTRHQA":
        rows = O.place_gtrhqa(gc.slices, gc.gate_q, gc.gate_mr, gc.Q, gc.C, gc.cap, D)
    else:
        rec = []
        rows = O.place_hqa_family(gc.slices, gc.gate_q, gc.gate_mr, gc.Q, gc.C, gc.cap, D, placer.lower(), rec)
        # every recorded matrix the reference handed to scipy must occur, entry for entry
        mine = {(len(m), np.array(m).tobytes()) for _, _, _, m, _ in rec}
        for tag, cm, r, c in gc.lsap_samples():
            if tag == placer:
                assert (cm.shape[0], cm.tobytes()) in mine, (name, placer)
    assert _rows_eq(rows, gc.part(placer))


@pytest.mark.parametrize("name", SMALL_CASES + ["test500_mesh8x8_u"])
def test_place_random_and_bfs(name):
    gc = load_case(name)
    D = gc.D.tolist()
    keys = gc.qubit_objects()
    rng = random.Random()
    rng.seed(gc.meta["placers"]["BFS"]["place_seed"])
    rows = O.place_bfs(gc.slices, gc.gate_q, gc.Q, gc.C, gc.cap, gc.edges, D, qubit_key=keys, rng=rng)
    assert _rows_eq(rows, gc.part("BFS"))
    rng.seed(gc.meta["placers"]["RANDOM"]["place_seed"])
    rows = O.place_random(gc.slices, gc.gate_q, gc.Q, gc.C, gc.cap, qubit_key=keys, rng=rng)
    assert _rows_eq(rows, gc.part("RANDOM"))


def _sparse_tables(gc):
    """Integer tables qp_place_sparse reads: per slice the active qubits in the order ``list(set(...))`` of the
    qubit objects yields them, the commands in slice order, every core's neighbours nearest first."""
    keys = gc.qubit_objects()
    back = {id(k): i for i, k in enumerate(keys)}
    mptr, mem, cptr, c0, c1 = [0], [], [0], [], []
    for sl in gc.slices:
        members = set()
        for g in sl:
            members.update(keys[q] for q in gc.gate_q[g])
        mem.extend(back[id(k)] for k in list(members))
        mptr.append(len(mem))
        for g in sl:
            qs = gc.gate_q[g]
            c0.append(qs[0] if len(qs) in (1, 2) else -1)
            c1.append(qs[1] if len(qs) == 2 else -1)
        cptr.append(len(c0))
    adj = O.adjacency(gc.C, gc.edges)
    nptr, nb = [0], []
    for c in range(gc.C):
        ns = [(v, gc.D[c][v]) for v, _ in adj[c]]
        ns.sort(key=lambda x: x[1])
        nb.extend(v for v, _ in ns)
        nptr.append(len(nb))
    i32 = lambda x: np.array(x, np.int32)  # noqa: E731
    return i32(mptr), np.array(mem, np.uint16), i32(cptr), i32(c0), i32(c1), i32(nptr), i32(nb)


@pytest.mark.parametrize("name", ALL_CASES)
def test_native_random_and_bfs_placement(name):
    """qp_place_sparse (host C++): three placements in a row from the fixture's seed.  The first equals the
    reference's own placement (the fixture), all three equal three calls of the restatement on one stream, and
    the generator is left in the state those calls leave it in; per-repeat states give each repeat its own."""
    from quantum_simulator_amd._lib import QP_PLACE_BFS, QP_PLACE_RANDOM, check, lib, ptr
    gc = load_case(name)
    mptr, mem, cptr, c0, c1, nptr, nb = _sparse_tables(gc)
    D = np.ascontiguousarray(gc.D, np.float64)
    Dl = gc.D.tolist()
    keys = gc.qubit_objects()
    T, Q, C, cap = len(gc.slices), gc.Q, gc.C, gc.cap
    ldq = (Q + 15) // 16 * 16
    n = 3
    for placer, which in (("RANDOM", QP_PLACE_RANDOM), ("BFS", QP_PLACE_BFS)):
        rng = random.Random()
        rng.seed(gc.meta["placers"][placer]["place_seed"])
        st0 = rng.getstate()
        want, states = [], []
        for _ in range(n):
            states.append(rng.getstate())
            if placer == "BFS":
                want.append(O.place_bfs(gc.slices, gc.gate_q, Q, C, cap, gc.edges, Dl, qubit_key=keys, rng=rng))
            else:
                want.append(O.place_random(gc.slices, gc.gate_q, Q, C, cap, qubit_key=keys, rng=rng))
        for per_repeat in (False, True):
            src = states if per_repeat else [st0]
            mt = np.array([s[1] for s in src], np.uint32)
            A = np.full((n, T, ldq), 7, np.uint8)
            bfs = (ptr(cptr), ptr(c0), ptr(c1), ptr(nptr), ptr(nb), ptr(D)) if placer == "BFS" else (None,) * 6
            check(lib().qp_place_sparse(which, T, Q, ldq, C, cap, ptr(mptr), ptr(mem), *bfs, ptr(mt), len(src), n, ptr(A)))
            assert (A[0, :, :Q] == gc.part(placer)).all(), placer
            for r in range(n):
                assert _rows_eq(want[r], A[r, :, :Q]), (placer, r)
            assert (A[:, :, Q:] == 0).all()
            assert tuple(int(x) for x in mt[-1]) == rng.getstate()[1]


def test_known_answer_386():
    """results.csv:2 of the reference: HQA == TRHQA == 386 x (the single edge weight) on the shipped
    example config; with unit weight the shipped code gives exactly 386."""
    gc = load_case("gen256_mesh2x1_i")
    assert gc.cost("HQA") == 386.0 and gc.cost("TRHQA") == 386.0
    D = gc.D.tolist()
    assert O.cost_v3(gc.slices, gc.gate_q, gc.gate_mr, gc.part("HQA").tolist(), D) == 386.0
    published = 769.4877543309925
    assert abs(published / 386 - 1.99349) < 1e-5  # an edge weight inside the config's [1, 2]


def test_qubo_energy_tiny_bruteforce():
    """2 qubits joined by one gate, 2 cores, capacity 2, one slice: the minimum-energy states put both
    qubits on one core with two slack bits set there."""
    slices = [[0]]
    gate_q = [(0, 1)]
    act, inter = O.qubo_slices(slices, gate_q, ["q[0]", "q[1]"])
    C, cap = 2, 2
    H = [[0, 1], [1, 0]]
    best, arg = None, []
    for bits in range(1 << 8):
        x = [[[(bits >> (j * 2 + i)) & 1 for i in range(2)] for j in range(C)]]
        y = [[[(bits >> (4 + j * 2 + p)) & 1 for p in range(cap)] for j in range(C)]]
        e = O.qubo_energy(act, inter, C, cap, H, x, y)
        if best is None or e < best:
            best, arg = e, [(x, y)]
        elif e == best:
            arg.append((x, y))
    assert best == 0
    for x, y in arg:
        cores = [j for j in range(C) if x[0][j][0]]
        assert len(cores) == 1 and x[0][cores[0]][1] == 1
        assert sum(y[0][cores[0]]) == 2
        assert O.qubo_decode(act, C, cap, x, 2) == [[cores[0], cores[0]]]


@pytest.mark.parametrize("name", ALL_CASES)
def test_c_oracle_costs_bits(name):
    """oracle/cost_oracle.c forms the same left-to-right sums: same bits as the reference's floats."""
    from oracle import c_oracle
    gc = load_case(name)
    gt = c_oracle.GateTables(gc.slices, gc.gate_q, gc.gate_mr)
    for p in ("HQA", "TRHQA", "GTRHQA"):
        if gc.has(p):
            assert float(c_oracle.cost_v3(gt, gc.Q, gc.part(p), gc.D)).hex() == gc.meta["placers"][p]["cost_hex"], p
    for p in ("BFS", "RANDOM"):
        pm = gc.meta["placers"][p]
        want = float("inf") if pm["cost_error"] else gc.cost(p)
        assert c_oracle.cost_v2(gt, gc.Q, gc.part(p), gc.D, 0) == want, p
        want = float("inf") if pm["v1_error"] else gc.cost_v1(p)
        assert c_oracle.cost_v2(gt, gc.Q, gc.part(p), gc.D, 1) == want, (p, "v1")


def test_core_activity_restatement_matches_reference_fixture():
    """oracle.core_activity against what the unmodified reference returned (activity.json)."""
    import json
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "activity.json")))
    for name in ALL_CASES:
        gc = load_case(name)
        for pname, want in gold[name].items():
            got = O.core_activity(gc.slices, gc.gate_q, gc.gate_mr, gc.part(pname).tolist(), gc.C)
            assert got == want, (name, pname)


@pytest.mark.parametrize("name", ["qft18_mesh4x4_u", "shor5_star4_u", "bigadder_torus3x3_u"])
def test_cpu_baseline_variants_match_reference_costs(name):
    """The two extra CPU forms bench.py times (a fresh shortest-path search per term, as the reference does;
    NumPy gathers on a precomputed matrix) against the reference's own costs in the fixtures."""
    from conftest import case_arch, load_case
    from oracle import cpu_variants
    gc = load_case(name)
    arch = case_arch(gc)
    nodes = list(arch.network.nodes)
    for p in ("HQA", "TRHQA"):
        if not gc.has(p):
            continue
        rows = gc.part(p).tolist()
        got, searches = cpu_variants.cost_v3_reference_structure(arch.network, nodes, gc.slices, gc.gate_q, gc.gate_mr,
                                                                 rows, gc.T)
        assert got == gc.cost(p) and searches > 0
        vec = cpu_variants.NumpyCost(gc.slices, gc.gate_q, gc.gate_mr, gc.D).cost(gc.part(p))
        assert abs(vec - gc.cost(p)) <= 1e-12 * abs(gc.cost(p))


def _qubo_cost_fixture():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "qubo_cost.json")))


def _fixture_partition(gc, pname, entry):
    """The partition a qubo_cost.json entry was made from: a stored one, or the BFS one with the recorded
    (slice, qubit) removed."""
    arr = gc.part(pname.split("_")[0]).copy()
    if entry["damage"]:
        t, q = entry["damage"]
        arr[t, q] = 0xFF
    return arr


def _comm_from_upper(upper, C):
    M = np.zeros((C, C), np.int64)
    for a, b, v in upper:
        M[a, b] = v
        M[b, a] = v
    return M


@pytest.mark.parametrize("name", ALL_CASES)
def test_qubo_cost_and_comm_matrix_restatements_match_reference_fixture(name):
    """oracle.cost_v4 / oracle.comm_matrix against what the unmodified reference's qubo_placer returned for the
    same partitions (qubo_cost.json: per_timeslice_cost, Qubo_placer.py:177-207; intercorecommatrix, :208-245),
    including the partitions on which the reference raises."""
    gold = _qubo_cost_fixture()[name]
    gc = load_case(name)
    H = gc.hops.tolist()
    assert set(gold) >= {"BFS", "RANDOM", "HQA"}
    for pname, entry in gold.items():
        arr = _fixture_partition(gc, pname, entry)
        rows = [[None if c == 0xFF else int(c) for c in row] for row in arr]
        if entry["cost"] is None:
            with pytest.raises(O.CostError):
                O.cost_v4(gc.slices, gc.gate_q, gc.gate_mr, rows, H)
        else:
            got = O.cost_v4(gc.slices, gc.gate_q, gc.gate_mr, rows, H)
            assert got == entry["cost"] and float(got).is_integer(), (pname, got, entry["cost"])
        if "comm_upper" not in entry:
            continue
        if entry["comm_upper"] is None:
            with pytest.raises(O.CostError):
                O.comm_matrix(gc.slices, gc.gate_q, gc.gate_mr, rows, gc.C)
        else:
            got = np.array(O.comm_matrix(gc.slices, gc.gate_q, gc.gate_mr, rows, gc.C))
            assert (got == _comm_from_upper(entry["comm_upper"], gc.C)).all(), pname
            assert int(got.sum()) == entry["comm_total"]


def test_sparse_qubo_model_and_cpu_annealer():
    """oracle.qubo_cpu (the CPU baseline of the QUBO path): its sparse form of the objective equals
    placer_oracle.qubo_energy (Qubo_placer.py:84-133 term by term) on random states, and the C annealer's summed
    energy changes equal the energy difference between its final and initial state."""
    from oracle import qubo_cpu
    from quantum_simulator_amd import layered_gate_arrays
    from quantum_simulator_amd.timeslices import CircuitIR
    rng = np.random.default_rng(5)
    for Q, T, C, cap in ((6, 3, 4, 3), (9, 4, 3, 4), (12, 2, 5, 2)):
        q0, q1, _ = layered_gate_arrays(Q, T, 0.6, seed=Q + T)
        keep = rng.random(len(q0)) > 0.25
        ir = CircuitIR.from_gate_tables(Q, q0[keep], q1[keep])
        names = [f"q[{i}]" for i in range(Q)]
        act, inter = O.qubo_slices(ir.slices, ir.gate_q, names)
        H = rng.integers(1, 5, size=(C, C))
        H = np.triu(H, 1)
        H = H + H.T
        sm = qubo_cpu.SparseModel(act, inter, C, cap, H)
        assert sm.N == sum(C * (len(a) + cap) for a in act)
        for _ in range(6):
            s = rng.integers(0, 2, size=sm.N).astype(np.int8)
            x = [[[int(s[sm.var_x(t, j, i)]) for i in range(len(act[t]))] for j in range(C)] for t in range(len(act))]
            y = [[[int(s[sm.var_y(t, j, p)]) for p in range(cap)] for j in range(C)] for t in range(len(act))]
            assert sm.energy(s) == O.qubo_energy(act, inter, C, cap, H.tolist(), x, y)
        s = rng.integers(0, 2, size=sm.N).astype(np.int8)
        e0 = sm.energy(s)
        dE, accepted = sm.anneal(s, 50, 0.01, 1.2, seed=7)
        assert accepted > 0 and sm.energy(s) == e0 + dE
        assert sm.energy(s) <= e0
