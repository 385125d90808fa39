"""The Placer plug-ins end to end on the GPU box: same partitions and costs as the reference's own
placers produced for the fixtures (partitions identical; costs exact for integer weights, 1e-12 /
summation-bound otherwise)."""
import random

import numpy as np
import pytest

from conftest import ALL_CASES, SMALL_CASES, case_arch, case_circuit, load_case

pytestmark = pytest.mark.gpu

MEDIUM = SMALL_CASES + ["qft63_mesh4x4_cap4_u", "test500_mesh8x8_u"]


def _cost_ok(got, want, exact, nterms):
    if exact:
        return got == want
    return abs(got - want) <= max(1e-12, nterms * 2.0 ** -53) * max(1.0, abs(want))


@pytest.mark.parametrize("name", MEDIUM)
@pytest.mark.parametrize("pname", ["HQA", "TRHQA", "GTRHQA"])
def test_hqa_family_place_and_cost(name, pname):
    from quantum_simulator_amd.placers import PLACERS
    gc = load_case(name)
    if not gc.has(pname):
        pytest.skip("capacity too small in this fixture")
    circ, arch = case_circuit(gc), case_arch(gc)
    placer = PLACERS[pname]()
    part = placer.place_per_timeslice(circ, arch)
    assert len(part) == gc.meta["placers"][pname]["Tp"]
    ctx = placer._context(circ, arch)
    got = ctx.partition_array(part, dense=True)[:, :gc.Q]
    assert (got == gc.part(pname)).all(), "partition differs from the reference's"
    # dict surface: keyed by the circuit's qubits in circuit.qubits order, valued by node labels
    assert list(part[0].keys()) == circ.qubits
    cost = placer.per_timeslice_cost(circ, part, arch)
    assert _cost_ok(cost, gc.cost(pname), name.endswith("_i"), ctx.ir.nE + ctx.ir.T * gc.Q)
    assert placer._dist_matrix.tobytes() == gc.D.tobytes()
    if pname != "GTRHQA" and gc.T <= 200:
        # the slice-by-slice Python form of the same loop (one cost-matrix call and one solve per iteration)
        rows = np.array(placer.place_rows(ctx, arch.qpu_qubit_num))
        assert (rows == gc.part(pname)).all(), "place_rows differs from the reference's partition"


@pytest.mark.parametrize("name", MEDIUM)
def test_bfs_and_random_place_and_cost(name):
    from quantum_simulator_amd.placers import PLACERS, BasePlacer
    gc = load_case(name)
    circ, arch = case_circuit(gc), case_arch(gc)
    exact = name.endswith("_i")
    for pname in ("BFS", "RANDOM"):
        placer = PLACERS[pname]()
        pm = gc.meta["placers"][pname]
        random.seed(pm["place_seed"])
        part = placer.place_per_timeslice(circ, arch)
        ctx = placer._context(circ, arch)
        got = ctx.partition_array(part, dense=False)[:, :gc.Q]
        assert (got == gc.part(pname)).all(), pname
        if pm["cost_error"] is None:
            cost = placer.per_timeslice_cost(circ, part, arch)
            assert _cost_ok(cost, gc.cost(pname), exact, 2 * ctx.ir.nV), pname
        else:
            with pytest.raises(Exception):
                placer.per_timeslice_cost(circ, part, arch)
        base = BasePlacer()
        if pm["v1_error"] is None:
            assert _cost_ok(base.per_timeslice_cost(circ, part, arch), gc.cost_v1(pname), exact, 2 * ctx.ir.nV)


def test_distance_and_timeslice_surface():
    from quantum_simulator_amd.placers import BasePlacer
    gc = load_case("bigadder_mesh4x4_u")
    circ, arch = case_circuit(gc), case_arch(gc)
    b = BasePlacer()
    ts = b.break_into_timeslices(circ)
    cmds = list(circ)
    assert [[cmds.index(c) for c in sl] for sl in ts[:5]] == gc.slices[:5]
    assert len(ts) == gc.T
    assert b.get_weighted_distance(arch, 0, 5) == gc.D[0, 5]
    assert b.get_weighted_distance(arch, 3, 3) == 0


def test_not_enough_space_raises():
    from quantum_simulator_amd.placers import hqa_placer
    from quantum_simulator_amd.multicore import multicore
    gc = load_case("qft18_mesh4x4_u")
    circ = case_circuit(gc)
    arch = multicore(2, 2, ncp_qubits=2, edge_weight_range=(1, 1), network_type="mesh")
    with pytest.raises(ValueError):
        hqa_placer().place_per_timeslice(circ, arch)


def test_example_config_end_to_end(tmp_path):
    """The shipped example config's flow (mesh 2x1, 130 qubits per core, weights in [1, 2], five
    placers, 3 repeats, one RNG stream) against the costs the reference code produced for the same
    seed; the circuit is re-serialised from the fixture's gate tables."""
    from conftest import GOLDEN_INDEX
    from quantum_simulator_amd import circuit_to_qasm_str, driver
    gold = GOLDEN_INDEX["example_config"]
    gc = load_case("gen256_mesh2x1_u")
    folder = tmp_path / "gen"
    folder.mkdir()
    (folder / "256.qasm").write_text(circuit_to_qasm_str(case_circuit(gc)))
    cfg_path = tmp_path / "cfg.txt"
    cfg_path.write_text(
        f"mode = simulate\ntopology = {gold['topology']}\nnum_cores_x = {gold['nx']}\nnum_cores_y = {gold['ny']}\n"
        f"ncm_qubits = 4\nncp_qubits = {gold['cap']}\nedge_weight_range = {gold['wr'][0]},{gold['wr'][1]}\n"
        f"circuit_folders = {folder}\nplacers = {', '.join(gold['placers'])}\nrepeats = {gold['repeats']}\n"
        f"out_csv = {tmp_path / 'out.csv'}\nplot_file = \nseed = {gold['seed']}\n")
    cfg = driver.parse_config_file(str(cfg_path))
    res = driver.run_simulation(cfg, verbose=False)[0]
    for p in gold["placers"]:
        want = float.fromhex(gold["results"][p]["mean_hex"])
        got = res[f"{p}_cost_mean"]
        assert abs(got - want) <= 3e-11 * abs(want), (p, got, want)  # up to ~2.6e5 terms summed left to right
    w = float.fromhex(gold["edge_weight_hex"])
    assert round(res["HQA_cost_mean"] / w) == 386 and round(res["TRHQA_cost_mean"] / w) == 386
    header = (tmp_path / "out.csv").read_text().splitlines()[0]
    assert header == "file," + ",".join(f"{p}_time_mean,{p}_cost_mean" for p in gold["placers"])
    # the repeats of every placer ran as one batch (RANDOM / BFS: native placement loop on the one random
    # stream, one cost launch); one repeat at a time (as main.py does) gives the same costs
    cfg2 = dict(cfg, batch_repeats=0)
    res2 = driver.run_simulation(cfg2, verbose=False)[0]
    for p in gold["placers"]:
        assert res2[f"{p}_cost_mean"] == res[f"{p}_cost_mean"], p


@pytest.mark.parametrize("name", ["qft18_mesh4x4_u", "bigadder_torus3x3_u", "shor5_star4_u", "qft63_torus8x8_i", "test500_mesh8x8_u"])
@pytest.mark.parametrize("pname", ["RANDOM", "BFS"])
def test_sparse_place_batch_is_the_sequence_of_calls(name, pname):
    """RANDOM / BFS: place_batch (native loop) gives the rows of as many place_per_timeslice calls on the same
    random stream and leaves the stream where they leave it; run_repeats gives their costs; with per-repeat
    generator states every repeat is the call made from that state."""
    import random
    from quantum_simulator_amd.placers import PLACERS
    gc = load_case(name)
    circ, arch = case_circuit(gc), case_arch(gc)
    pl = PLACERS[pname]()
    ctx = pl._context(circ, arch)
    n = 3
    random.seed(gc.meta["placers"][pname]["place_seed"])
    st0 = random.getstate()
    parts, states = [], []
    for _ in range(n):
        states.append(random.getstate())
        parts.append(pl.place_per_timeslice(circ, arch))
    end = random.getstate()
    want = np.stack([ctx.partition_array(p, dense=False) for p in parts])
    assert (want[0, :, :gc.Q] == gc.part(pname)).all()  # the reference's own placement from this seed
    costs = [pl.per_timeslice_cost(circ, p, arch) for p in parts]
    random.setstate(st0)
    got = pl.place_batch(circ, arch, n)
    assert got.tobytes() == want.tobytes()
    assert random.getstate() == end
    random.setstate(st0)
    _, c2 = pl.run_repeats(circ, arch, n)
    assert c2 == costs and random.getstate() == end
    random.seed(12345)
    got = pl.place_batch(circ, arch, n, states=states)
    assert got.tobytes() == want.tobytes() and random.getstate() == end


@pytest.mark.parametrize("name", ["qft18_mesh4x4_u", "bigadder_mesh4x4_i", "qft18_mesh4x4_cap4_u", "shor5_star4_u"])
@pytest.mark.parametrize("pname", ["HQA", "TRHQA", "GTRHQA"])
def test_place_batch_lockstep(name, pname):
    """Lock-step batches: identical repeats reproduce the reference partition; instances started from
    different (shuffled) assignments equal the sequential restatement started from the same rows."""
    from oracle import placer_oracle as O
    from quantum_simulator_amd.placers import PLACERS
    gc = load_case(name)
    if not gc.has(pname):
        pytest.skip("capacity too small in this fixture")
    circ, arch = case_circuit(gc), case_arch(gc)
    placer = PLACERS[pname]()
    same = placer.place_batch(circ, arch, B=3)
    for b in range(3):
        assert (same[b] == gc.part(pname)).all()
    rng = np.random.default_rng(4)
    slots = np.repeat(np.arange(gc.C), gc.cap)
    inits = np.stack([rng.permutation(slots)[:gc.Q] for _ in range(5)]).astype(np.uint8)
    got = placer.place_batch(circ, arch, initial=inits)
    D = gc.D.tolist()
    for b in range(5):
        if pname == "GTRHQA":
            want = O.place_gtrhqa(gc.slices, gc.gate_q, gc.gate_mr, gc.Q, gc.C, gc.cap, D, init=inits[b].tolist())
        else:
            want = O.place_hqa_family(gc.slices, gc.gate_q, gc.gate_mr, gc.Q, gc.C, gc.cap, D, pname.lower(),
                                      init=inits[b].tolist())
        assert (got[b] == np.array(want, np.uint8)).all(), (pname, b)


@pytest.mark.parametrize("name", ["qft63_torus8x8_u", "qft63_torus8x8_i"])
@pytest.mark.parametrize("pname", ["HQA", "TRHQA", "GTRHQA"])
def test_place_batch_large_torus_case(name, pname):
    """The large shipped configuration (qft_n63 on the 8x8 torus, capacity 2): a batch of 8 repeats gives
    the reference's own partition in every instance, and its cost."""
    from quantum_simulator_amd.placers import PLACERS
    gc = load_case(name)
    circ, arch = case_circuit(gc), case_arch(gc)
    placer = PLACERS[pname]()
    rows = placer.place_batch(circ, arch, B=8)
    want = gc.part(pname)
    for b in range(8):
        assert (rows[b][:, :gc.Q] == want).all(), (pname, b)
    _, costs = placer.run_repeats(circ, arch, 8)
    ctx = placer._context(circ, arch)
    for c in costs:
        assert _cost_ok(c, gc.cost(pname), name.endswith("_i"), ctx.ir.nE + ctx.ir.T * gc.Q)


GTR_SHAPES = [  # Q, T, C, cap, instances, disconnected graph, integer edge weights
    (40, 30, 8, 6, 4, False, False), (200, 20, 32, 8, 5, False, False), (33, 12, 5, 8, 3, False, False),
    (64, 15, 16, 4, 4, False, False), (48, 10, 8, 6, 3, True, False), (300, 8, 64, 6, 3, True, False),
    (1024, 5, 256, 8, 3, False, False), (16, 9, 8, 2, 2, False, False),
    (40, 30, 8, 6, 4, False, True), (200, 20, 32, 8, 5, False, True), (1024, 5, 256, 8, 3, False, True),
    (300, 8, 64, 6, 3, True, True), (64, 15, 16, 4, 4, False, True),
]


@pytest.mark.parametrize("Q,T,C,cap,B,disconnected,integer", GTR_SHAPES)
def test_gtrhqa_device_placement_matches_oracle(Q, T, C, cap, B, disconnected, integer):
    """qp_gtrhqa_place (every slice on the device) against the sequential restatement, row for row,
    from shuffled starting assignments; includes graphs with unreachable cores (infinite distances),
    capacities that leave few or no cores with room, and integer edge weights (many equal bids; the
    kernel then forms its bids from a 16-bit copy of the table)."""
    import torch
    from oracle import placer_oracle as O
    from quantum_simulator_amd import engine, layered_gate_arrays
    from quantum_simulator_amd.timeslices import CircuitIR
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(Q * 31 + T)
    q0, q1, _ = layered_gate_arrays(Q, T, 0.6, seed=Q + T)
    keep = rng.random(len(q0)) > 0.25
    ir = CircuitIR.from_gate_tables(Q, q0[keep], q1[keep])
    if disconnected:  # two rings that do not touch
        h = C // 2
        eu = list(range(C))
        ev = [(i + 1) % h for i in range(h)] + [h + (i + 1) % (C - h) for i in range(C - h)]
    else:  # ring + chords
        eu = list(range(C)) + [int(x) for x in rng.integers(0, C, size=C // 2)]
        ev = [(i + 1) % C for i in range(C)] + [int(x) for x in rng.integers(0, C, size=C // 2)]
    ok = [k for k in range(len(eu)) if eu[k] != ev[k]]
    eu, ev = [eu[k] for k in ok], [ev[k] for k in ok]
    w = rng.integers(1, 4, size=len(eu)).astype(float) if integer else rng.uniform(1, 3, size=len(eu))
    D = engine.apsp(C, eu, ev, w)
    assert np.isinf(D).any() == disconnected
    dc, dt, eng = engine.DeviceCircuit(ir, dev), engine.DistanceTable(D, dev), engine.Engine(dev)
    slots = np.repeat(np.arange(C), cap)
    inits = np.stack([rng.permutation(slots)[:Q] for _ in range(B)]).astype(np.uint8)
    inits[0] = np.arange(Q) // cap  # the reference's own starting assignment
    pad = np.zeros((B, dc.ldq), np.uint8)
    pad[:, :Q] = inits
    got = eng.gtrhqa_place(dc, dt, torch.from_numpy(pad).to(dev), cap).cpu().numpy()
    assert got.shape == (B, ir.T + 1, dc.ldq)
    Dl = D.tolist()
    for b in range(B):
        want = np.array(O.place_gtrhqa(ir.slices, ir.gate_q, ir.gate_mr, Q, C, cap, Dl, init=inits[b].tolist()), np.uint8)
        assert (got[b, :, :Q] == want).all(), (b, np.argwhere(got[b, :, :Q] != want)[:5])


def test_core_activity_matches_reference_fixture():
    """per_core_intercore_activity on the stored HQA / TRHQA / GTRHQA partitions against what the
    unmodified reference returned for them (tests/golden/activity.json) and against the restatement."""
    import json
    import os
    from conftest import ROOT
    from oracle import placer_oracle as O
    from quantum_simulator_amd.placers import PLACERS
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "activity.json")))
    seen = 0
    for name in ALL_CASES:
        gc = load_case(name)
        circ, arch = case_circuit(gc), case_arch(gc)
        nodes = list(arch.network.nodes)
        placer = PLACERS["HQA"]()
        ctx = placer._context(circ, arch)
        for pname, want in gold[name].items():
            rows = gc.part(pname)
            part = ctx.rows_to_partition(rows)
            got = placer.per_core_intercore_activity(circ, part, arch)
            assert [got[n] for n in nodes] == want, (name, pname)
            assert O.core_activity(gc.slices, gc.gate_q, gc.gate_mr, rows.tolist(), gc.C) == want
            seen += 1
    assert seen >= 40


@pytest.mark.parametrize("Q,T,C,B", [(40, 30, 8, 4), (200, 70, 64, 3), (1024, 130, 256, 2), (33, 1, 5, 2), (700, 300, 200, 5),
                                     (1000, 2, 17, 3), (513, 1, 100, 2), (200, 400, 64, 3), (63, 600, 16, 2), (17, 2100, 5, 3)])
def test_core_activity_and_comm_matrix_random(Q, T, C, B):
    import torch
    from oracle import placer_oracle as O
    from quantum_simulator_amd import engine, layered_gate_arrays
    from quantum_simulator_amd.timeslices import CircuitIR
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(Q + T)
    q0, q1, _ = layered_gate_arrays(Q, T, 0.5, seed=Q * 3 + T)
    keep = rng.random(len(q0)) > 0.3
    ir = CircuitIR.from_gate_tables(Q, q0[keep], q1[keep])
    dc, eng = engine.DeviceCircuit(ir, dev), engine.Engine(dev)
    A = rng.integers(0, C, size=(B, ir.T + 1, dc.ldq), dtype=np.uint8)
    A[0, 1:] = A[0, :1]
    got = eng.core_activity(dc, torch.from_numpy(A).to(dev), C).cpu().numpy()
    for b in range(B):
        assert got[b].tolist() == O.core_activity(ir.slices, ir.gate_q, ir.gate_mr, A[b, :, :Q].tolist(), C)
    if C > 255:
        return
    # sparse partitions for the gate-ordered walk; instance 1 loses one active qubit (the reference raises)
    S = np.full((B, ir.T, dc.ldq), 0xFF, np.uint8)
    for b in range(B):
        for t in range(ir.T):
            act = ir.active_qubits(t)
            S[b, t, act] = rng.integers(0, C, size=len(act))
    if B > 1 and ir.nV:
        v = ir.visit_records[int(rng.integers(0, ir.nV))]
        S[1, v["t"], v["q"]] = 0xFF
    M, status = eng.comm_matrix(dc, torch.from_numpy(S).to(dev), C)
    M, status = M.cpu().numpy(), status.cpu().numpy()
    for b in range(B):
        rows = [[None if c == 0xFF else int(c) for c in r[:Q]] for r in S[b]]
        try:
            want = O.comm_matrix(ir.slices, ir.gate_q, ir.gate_mr, rows, C)
        except O.CostError:
            assert status[b] == 1
            continue
        assert status[b] == 0 and M[b].tolist() == want
        assert (M[b] == M[b].T).all()


@pytest.mark.parametrize("Q,T,C,B", [(1024, 40, 255, 420), (600, 130, 30, 130), (777, 33, 100, 500), (513, 70, 2, 240)])
def test_comm_matrix_row_walk(Q, T, C, B):
    """Batches on a circuit that visits most of every row take the communication counts from the pair counters of the
    row walk: equal to the visit-by-visit kernel on every partition, to the oracle on a sample, error cases included."""
    import torch
    from oracle import placer_oracle as O
    from quantum_simulator_amd import engine, layered_gate_arrays
    from quantum_simulator_amd.timeslices import CircuitIR
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(Q + T)
    q0, q1, _ = layered_gate_arrays(Q, T, 0.7, seed=Q + T)
    keep = rng.random(len(q0)) >= 0.1
    ir = CircuitIR.from_gate_tables(Q, q0[keep], q1[keep])
    assert ir.nV >= 128 * ir.T and B * ir.T >= 16384
    walk, byvisit, eng = engine.DeviceCircuit(ir, dev), engine.DeviceCircuit(ir, dev, visit_mask=False), engine.Engine(dev)
    A = rng.integers(0, C, size=(B, ir.T, walk.ldq), dtype=np.uint8)
    A[:, :, Q:] = rng.integers(0, 256, size=(B, ir.T, walk.ldq - Q), dtype=np.uint8)
    stay = rng.random((B, ir.T, Q)) < 0.6
    for t in range(1, ir.T):
        A[:, t, :Q] = np.where(stay[:, t], A[:, t - 1, :Q], A[:, t, :Q])
    vm = ir.visit_mask(walk.ldq)[:, :Q] != 0
    for b in range(20):  # only the visited qubits are placed
        A[b, :, :Q][~vm] = 0xFF
    for b in range(20, 40):  # a visited qubit is absent somewhere, from the first slice on in half of them
        v = ir.visit_records[int(rng.integers(0, ir.nV))]
        A[b, v["t"], v["q"]] = 0xFF
        if b >= 30:
            A[b, :v["t"], v["q"]] = 0xFF
    if C < 255:
        for b in range(40, 50):  # a core that does not exist, under a visit or not
            t, q = int(rng.integers(0, ir.T)), int(rng.integers(0, Q))
            A[b, t, q] = int(rng.integers(C, 255))
    Ad = torch.from_numpy(A).to(dev)
    M, status = eng.comm_matrix(walk, Ad, C)
    Mr, sr = eng.comm_matrix(byvisit, Ad, C)
    M, status, Mr, sr = M.cpu().numpy(), status.cpu().numpy(), Mr.cpu().numpy(), sr.cpu().numpy()
    assert (status == sr).all(), np.flatnonzero(status != sr)[:8]
    assert status[20:40].any() and not status[:20].any()
    ok = status == 0
    assert (M[ok] == Mr[ok]).all()
    for b in (0, 19, 20, 29, 30, 39, 41, 49, B - 1):
        rows = [[None if c == 0xFF else int(c) for c in r[:Q]] for r in A[b]]
        if 40 <= b < 50 and status[b] == 1:
            continue  # a visited core that does not exist is reported even where the walk would never index with it
        try:
            want = O.comm_matrix(ir.slices, ir.gate_q, ir.gate_mr, rows, C)
        except O.CostError:
            assert status[b] == 1, b
            continue
        assert status[b] == 0 and M[b].tolist() == want, b


def test_gtrhqa_device_placement_large_batch():
    """Large batches run on small workgroups, small batches on one 16-wave workgroup per instance: both
    must give the oracle's rows."""
    import torch
    from oracle import placer_oracle as O
    from quantum_simulator_amd import engine, layered_gate_arrays
    from quantum_simulator_amd.timeslices import CircuitIR
    dev = torch.device("cuda", 0)
    Q, T, C, cap, B = 96, 25, 16, 8, 700
    rng = np.random.default_rng(12)
    q0, q1, _ = layered_gate_arrays(Q, T, 0.6, seed=5)
    ir = CircuitIR.from_gate_tables(Q, q0, q1)
    eu = list(range(C)) + [0, 3, 5]
    ev = [(i + 1) % C for i in range(C)] + [8, 11, 13]
    D = engine.apsp(C, eu, ev, rng.uniform(1, 3, size=len(eu)))
    dc, dt, eng = engine.DeviceCircuit(ir, dev), engine.DistanceTable(D, dev), engine.Engine(dev)
    slots = np.repeat(np.arange(C), cap)
    pad = np.zeros((B, dc.ldq), np.uint8)
    for b in range(B):
        pad[b, :Q] = rng.permutation(slots)[:Q]
    big = eng.gtrhqa_place(dc, dt, torch.from_numpy(pad).to(dev), cap).cpu().numpy()
    small = np.concatenate([eng.gtrhqa_place(dc, dt, torch.from_numpy(pad[s:s + 100].copy()).to(dev), cap).cpu().numpy()
                            for s in range(0, B, 100)])
    assert big.tobytes() == small.tobytes()
    Dl = D.tolist()
    for b in (0, 1, 350, 699):
        want = np.array(O.place_gtrhqa(ir.slices, ir.gate_q, ir.gate_mr, Q, C, cap, Dl, init=pad[b, :Q].tolist()), np.uint8)
        assert (big[b, :, :Q] == want).all(), b


def test_gtrhqa_full_size_invariants():
    """The benchmark's circuit (1024 qubits x 1000 slices, 16x16 mesh, capacity 8), 24 instances: properties
    that hold for any correct GTRHQA run whatever the size — no core ever holds more than its capacity,
    row t+1 differs from row t-1 only in qubits of slice t's gates, both qubits of a moved gate land on
    one core that had room, and instance 0 (the reference's own start) equals the sequential restatement
    on the first slices."""
    import random
    import torch
    from oracle import placer_oracle as O
    from quantum_simulator_amd import engine, layered_gate_arrays
    from quantum_simulator_amd.multicore import graph_tables, multicore
    from quantum_simulator_amd.timeslices import CircuitIR
    dev = torch.device("cuda", 0)
    Q, T, C, cap, B = 1024, 1000, 256, 8, 24
    q0, q1, _ = layered_gate_arrays(Q, T, 0.5, seed=2025)
    ir = CircuitIR.from_gate_tables(Q, q0, q1)
    random.seed(1234)
    arch = multicore(16, 16, ncm_qubits=4, ncp_qubits=cap, edge_weight_range=(1.0, 3.0), network_type="mesh")
    nodes, eu, ev, w = graph_tables(arch.network)
    D = engine.apsp(C, eu, ev, w)
    dc, dt, eng = engine.DeviceCircuit(ir, dev), engine.DistanceTable(D, dev), engine.Engine(dev)
    rng = np.random.default_rng(8)
    slots = np.repeat(np.arange(C), cap)
    init = np.stack([rng.permutation(slots)[:Q] for _ in range(B)]).astype(np.uint8)
    init[0] = np.arange(Q) // cap
    rows = eng.gtrhqa_place(dc, dt, torch.from_numpy(init).to(dev), cap)
    occ = torch.zeros((B, T + 1, C), dtype=torch.int32, device=dev)
    occ.scatter_add_(2, rows[:, :, :Q].long(), torch.ones((B, T + 1, Q), dtype=torch.int32, device=dev))
    assert int(occ.max().item()) <= cap
    r = rows[:, :, :Q].cpu().numpy()
    sep = ir.slice_edge_ptr
    for t in (0, 1, 2, 17, 500, T - 1):
        prev = r[:, t - 1] if t > 0 else r[:, 0]
        new = r[:, t + 1]
        u, v = ir.edge_u[sep[t]:sep[t + 1]].astype(int), ir.edge_v[sep[t]:sep[t + 1]].astype(int)
        touched = np.zeros(Q, bool)
        touched[u] = touched[v] = True
        assert (new[:, ~touched] == prev[:, ~touched]).all()
        moved = (new[:, u] != prev[:, u]) | (new[:, v] != prev[:, v])
        assert (new[:, u][moved] == new[:, v][moved]).all()          # a placed gate sits on one core
        assert (prev[:, u] != prev[:, v])[moved].all()               # only gates that were split get placed
        room = np.stack([np.bincount(prev[b], minlength=C) <= cap - 2 for b in range(B)])
        bb = np.nonzero(moved)[0]
        assert room[bb, new[:, u][moved]].all()                      # and only onto cores that had room
    want = O.place_gtrhqa(ir.slices[:6], ir.gate_q, ir.gate_mr, Q, C, cap, D.tolist())
    assert (r[0, :7] == np.array(want, np.uint8)).all()


@pytest.mark.parametrize("key", ["qft18_mesh4x4_u/HQA", "shor5_mesh2x2_i/HQA", "qft29_torus8x8_u/HQA",
                                 "qft18_mesh4x4_u/TRHQA", "bigadder_mesh4x4_u/BFS", "qft63_mesh4x4_cap4_u/RANDOM",
                                 "qft18_mesh4x4_u/RANDOM", "shor5_mesh2x2_i/RANDOM", "layered12x8_mesh2x2_cap3/RANDOM",
                                 "layered20x6_torus3x3_cap3/RANDOM"])
def test_simulate_broken_links_matches_reference_fixture(key, tmp_path):
    """simulate_broken_links under the same seed as the unmodified reference
    (tests/golden/broken_links.json): same links broken in every simulation, same cost (infinite where
    the damaged network forces a move across a broken link)."""
    import json
    import os
    import random
    from conftest import ROOT
    from quantum_simulator_amd.placers import PLACERS
    name, pname = key.split("/")
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "broken_links.json")))[key]
    if "qasm" in gold:  # a generated circuit whose text travels in the fixture, with its architecture's parameters
        from quantum_simulator_amd.circuit import circuit_from_qasm
        from quantum_simulator_amd.multicore import multicore
        (tmp_path / "c.qasm").write_text(gold["qasm"])
        circ = circuit_from_qasm(str(tmp_path / "c.qasm"), maxwidth=1024)
        m = gold["arch"]
        random.seed(m["arch_seed"])
        arch = multicore(m["nx"], m["ny"], ncm_qubits=4, ncp_qubits=m["cap"], edge_weight_range=tuple(m["weight_range"]),
                         network_type=m["topology"])
    else:
        gc = load_case(name)
        circ, arch = case_circuit(gc), case_arch(gc)
    before = arch.network
    eidx = {e: i for i, e in enumerate(arch.network.edges)}
    random.seed(gold["seed"])
    np.random.seed(gold["seed"])  # (random_placer's weight-biased choice of a core pair draws from numpy's generator)
    drawn = [0]
    plain = np.random.choice

    def counted(*a, **k):
        drawn[0] += 1
        return plain(*a, **k)
    np.random.choice = counted
    try:
        got = PLACERS[pname]().simulate_broken_links(circ, arch, num_simulations=gold["num_simulations"],
                                                     break_probability=gold["break_probability"])
    finally:
        np.random.choice = plain
    assert arch.network is before
    assert drawn[0] == gold.get("numpy_choices", 0)
    assert len(got) == len(gold["sims"])
    for (broken, cost), want in zip(got, gold["sims"]):
        assert sorted(eidx[e] for e in broken) == want["broken"]
        if want["cost"] == "inf":
            assert cost == float("inf")
        else:
            w = float.fromhex(want["cost"])
            assert abs(cost - w) <= 1e-12 * abs(w), (cost, w)


_TWO_RANK_WORKER = """
import json, os, sys
sys.path.insert(0, {root!r})
import torch
import torch.distributed as dist
from quantum_simulator_amd import driver
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=2)
cfg = driver.parse_config_file({cfg!r})
cfg["out_csv"] = {out!r} + os.environ["RANK"]
out = []
for batched in (1, 0):
    res = driver.run_simulation(dict(cfg, batch_repeats=batched), verbose=False)
    out.append([[r[k] for k in sorted(r) if k.endswith("cost_mean") or k.endswith("cost_min")] for r in res])
print("RESULT " + json.dumps(out))
dist.destroy_process_group()
"""


def test_driver_two_ranks_share_the_gpu(tmp_path):
    """The driver on two gloo ranks that share cuda:0, real placers: the repeats of RANDOM / BFS (seeded per
    unit when there is more than one rank) and of HQA give the same table whether a rank runs its repeats as
    one batch or one at a time, and both ranks end with the same table."""
    import json
    import os
    import subprocess
    import sys
    from quantum_simulator_amd import circuit_to_qasm_str, layered_circuit
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    folder = tmp_path / "circ"
    folder.mkdir()
    for k, (q, t) in enumerate([(12, 9), (20, 6)]):
        (folder / f"c{k}.qasm").write_text(circuit_to_qasm_str(layered_circuit(q, t, 0.5, seed=k)))
    cfg = tmp_path / "cfg.txt"
    cfg.write_text(f"mode = simulate\ntopology = mesh\nnum_cores_x = 3\nnum_cores_y = 2\nncm_qubits = 4\nncp_qubits = 4\n"
                   f"edge_weight_range = 1.0,3.0\ncircuit_folders = {folder}\nplacers = RANDOM, BFS, HQA\nrepeats = 5\n"
                   f"out_csv = {tmp_path / 'o.csv'}\nplot_file = \nseed = 11\n")
    script = tmp_path / "w.py"
    script.write_text(_TWO_RANK_WORKER.format(root=root, cfg=str(cfg), out=str(tmp_path / "o.csv")))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out
        outs.append(json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][0][7:]))
    assert outs[0] == outs[1]
    batched, single = outs[0]
    assert batched == single and len(batched) == 2
    assert all(np.isfinite(x) for row in batched for x in row)


_SHARDED_COST_WORKER = """
import json, os, sys
sys.path.insert(0, {root!r})
import numpy as np
import torch
import torch.distributed as dist
from quantum_simulator_amd import engine, layered_gate_arrays
from quantum_simulator_amd.driver import sharded_cost_v3, slice_shard
from quantum_simulator_amd.timeslices import CircuitIR
dist.init_process_group("gloo")
dev = torch.device("cuda", 0)
Q, T, C, B = 200, 61, 64, 4
q0, q1, _ = layered_gate_arrays(Q, T, 0.5, seed=5)
ir = CircuitIR.from_gate_tables(Q, q0, q1)
rng = np.random.default_rng(8)
eu = list(range(C)); ev = [(i + 1) % C for i in range(C)]
D = engine.apsp(C, eu, ev, rng.uniform(1, 3, size=C))
dc, dt, eng = engine.DeviceCircuit(ir, dev), engine.DistanceTable(D, dev), engine.Engine(dev)
A = rng.integers(0, C, size=(B, ir.T, dc.ldq), dtype=np.uint8)
t0, t1, row0 = slice_shard(ir.T, dist.get_rank(), dist.get_world_size())
mine = torch.from_numpy(np.ascontiguousarray(A[:, row0:t1])).to(dev)   # this rank holds only its window of rows
got = sharded_cost_v3(eng, dc, dt, mine).cpu().numpy()
whole = eng.cost_v3(dc, dt, torch.from_numpy(A).to(dev)).cpu().numpy()
print("RESULT " + json.dumps([got.tolist(), whole.tolist()]))
dist.destroy_process_group()
"""


def test_one_cost_evaluation_split_by_timeslices_over_two_ranks(tmp_path):
    """A cost evaluation split by timeslices: two gloo ranks that share cuda:0 each hold the rows of their half of
    the circuit plus the row before it, cost their slices, and one all-reduce adds the halves: the value of the
    whole evaluation to 1e-12, the same on both ranks."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text(_SHARDED_COST_WORKER.format(root=root))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29579")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out
        outs.append(json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][0][7:]))
    assert outs[0][0] == outs[1][0]
    for got, whole in zip(*outs[0]):
        assert abs(got - whole) <= 1e-12 * abs(whole), (got, whole)


@pytest.mark.parametrize("Q,T,C,S", [(63, 70, 64, 9), (40, 30, 16, 5), (300, 12, 200, 4), (1024, 9, 256, 3)])
def test_cost_v3_with_a_table_per_partition(Q, T, C, S):
    """qp_slice_cost_v3_multi: every partition with its own distance table (damaged copies of one network, some
    links infinite) against the single-table entry point and the oracle."""
    import torch
    from oracle import placer_oracle as O
    from quantum_simulator_amd import engine, layered_gate_arrays
    from quantum_simulator_amd.timeslices import CircuitIR
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(Q + S)
    q0, q1, _ = layered_gate_arrays(Q, T, 0.5, seed=Q)
    keep = rng.random(len(q0)) > 0.2
    ir = CircuitIR.from_gate_tables(Q, q0[keep], q1[keep])
    eu = list(range(C)) + [int(x) for x in rng.integers(0, C, size=C // 2)]
    ev = [(i + 1) % C for i in range(C)] + [int(x) for x in rng.integers(0, C, size=C // 2)]
    ok = [k for k in range(len(eu)) if eu[k] != ev[k]]
    eu, ev = [eu[k] for k in ok], [ev[k] for k in ok]
    W = np.tile(rng.uniform(1, 3, size=len(eu)), (S, 1))
    for s in range(1, S):
        W[s, rng.random(len(eu)) < 0.15] = np.inf
    tables = engine.apsp_batch(C, eu, ev, W)
    dc, eng = engine.DeviceCircuit(ir, dev), engine.Engine(dev)
    B = 2 * S + 1
    tab = (np.arange(B) % S).astype(np.int32)
    A = rng.integers(0, C, size=(B, ir.T, dc.ldq), dtype=np.uint8)
    A[0, 1:] = A[0, :1]
    Ad = torch.from_numpy(A).to(dev)
    got = eng.cost_v3_multi(dc, torch.from_numpy(tables).to(dev), torch.from_numpy(tab).to(dev), Ad).cpu().numpy()
    assert got.tobytes() == eng.cost_v3_multi(dc, torch.from_numpy(tables).to(dev), torch.from_numpy(tab).to(dev), Ad).cpu().numpy().tobytes()
    for b in range(B):
        D = tables[tab[b]]
        want = O.cost_v3(ir.slices, ir.gate_q, ir.gate_mr, A[b, :, :Q].tolist(), D.tolist())
        if np.isinf(want):
            assert np.isinf(got[b])
        else:
            assert abs(got[b] - want) <= 1e-12 * max(1.0, abs(want)), (b, got[b], want)
        one = float(eng.cost_v3(dc, engine.DistanceTable(D, dev), Ad[b:b + 1], table="f64").cpu()[0])
        assert (np.isinf(one) and np.isinf(got[b])) or abs(one - got[b]) <= 1e-12 * max(1.0, abs(one))


def test_cost_v3_with_a_table_per_partition_more_partitions_than_workgroups():
    """Tables of more than 128 cores are read from device memory by workgroups that each walk several
    partitions (the grid is capped at eight workgroups per compute unit): 2600 partitions on 130 cores, every
    result against a NumPy evaluation of the same sums, and two launches bit for bit the same."""
    import torch
    from quantum_simulator_amd import engine, layered_gate_arrays
    from quantum_simulator_amd.timeslices import CircuitIR
    dev = torch.device("cuda", 0)
    Q, T, C, S, B = 48, 6, 130, 3, 2600
    rng = np.random.default_rng(77)
    q0, q1, _ = layered_gate_arrays(Q, T, 0.5, seed=3)
    ir = CircuitIR.from_gate_tables(Q, q0, q1)
    eu = list(range(C - 1))
    ev = [i + 1 for i in range(C - 1)]
    W = rng.uniform(1, 3, size=(S, len(eu)))
    tables = engine.apsp_batch(C, eu, ev, W)
    dc, eng = engine.DeviceCircuit(ir, dev), engine.Engine(dev)
    tab = rng.integers(0, S, size=B).astype(np.int32)
    A = rng.integers(0, C, size=(B, ir.T, dc.ldq), dtype=np.uint8)
    Ad, d_tab, d_tables = torch.from_numpy(A).to(dev), torch.from_numpy(tab).to(dev), torch.from_numpy(tables).to(dev)
    got = eng.cost_v3_multi(dc, d_tables, d_tab, Ad).cpu().numpy()
    assert got.tobytes() == eng.cost_v3_multi(dc, d_tables, d_tab, Ad).cpu().numpy().tobytes()
    # the same sums with NumPy gathers
    want = np.zeros(B)
    Dz = tables.copy()
    Dz[:, np.arange(C), np.arange(C)] = 0.0
    Al = A[:, :, :Q].astype(np.int64)
    for t in range(ir.T):
        ops = ir.slice_ops(t)
        if ops:
            u = np.array([o[1] for o in ops]); v = np.array([o[2] for o in ops])
            want += Dz[tab[:, None], Al[:, t][:, u], Al[:, t][:, v]].sum(axis=1)
        if t >= 2:
            want += Dz[tab[:, None], Al[:, t - 1], Al[:, t]].sum(axis=1)
    assert np.allclose(got, want, rtol=1e-12, atol=0)


@pytest.mark.parametrize("pname", ["HQA", "TRHQA"])
def test_place_batch_with_a_table_per_instance(pname):
    """A placement batch in which every instance has its own damaged network equals the placements made one
    damaged network at a time."""
    import random
    from quantum_simulator_amd import engine
    from quantum_simulator_amd.placers import PLACERS
    gc = load_case("qft18_mesh4x4_u")
    circ, arch = case_circuit(gc), case_arch(gc)
    placer = PLACERS[pname]()
    ctx = placer._context(circ, arch)
    rng = np.random.default_rng(3)
    S = 6
    W = np.tile(np.asarray(ctx.arch.w, np.float64), (S, 1))
    for s in range(1, S):
        W[s, rng.random(W.shape[1]) < 0.12] = np.inf
    tables = engine.apsp_batch(ctx.arch.C, ctx.arch.eu, ctx.arch.ev, W)
    got = placer.place_batch(circ, arch, B=S, tables=tables, tab_index=np.arange(S, dtype=np.int32))
    assert (got[0] == gc.part(pname)).all()  # table 0 is the intact network
    intact = arch.network
    links = list(intact.edges)
    try:
        for s in range(1, S):
            damaged = intact.copy()
            for k, (a, b) in enumerate(links):
                damaged[a][b]["weight"] = float(W[s, k])
            arch.network = damaged
            one = PLACERS[pname]()
            rows = one.place_rows(one._context(circ, arch), arch.qpu_qubit_num)
            assert (got[s] == np.array(rows)).all(), (pname, s)
    finally:
        arch.network = intact


_RCCL_WORKER = """
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
import torch
import torch.distributed as dist
from quantum_simulator_amd import driver
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
sums = np.array([[[30.0, 2, 1.0, 2]], [[0.0, 0, 0.25, 1]]])
mins = np.array([[10.0], [np.inf]])
# world size 1 short-circuits in reduce_table: run the two collectives it would run, on device tensors
s = torch.from_numpy(sums).to("cuda")
m = torch.from_numpy(mins).to("cuda")
dist.all_reduce(s, op=dist.ReduceOp.SUM)
dist.all_reduce(m, op=dist.ReduceOp.MIN)
torch.cuda.synchronize()
assert s.cpu().numpy().tobytes() == sums.tobytes() and m.cpu().numpy().tobytes() == mins.tobytes()
s2, m2 = driver.reduce_table(sums, mins, torch.device("cuda", 0))
assert s2.tobytes() == sums.tobytes() and m2.tobytes() == mins.tobytes()
dist.barrier()
dist.destroy_process_group()
print("RCCL ok")
"""


def test_rccl_all_reduce_of_the_result_table(tmp_path):
    """The job's only collectives (one SUM and one MIN all-reduce of a few doubles) through the RCCL backend on
    device tensors.  One GPU is all a test box has, so the group has one rank: what runs is RCCL's
    initialisation and the two all-reduce calls the driver makes; the two-rank logic is covered on gloo."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text(_RCCL_WORKER.format(root=root))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29581", RANK="0", WORLD_SIZE="1")
    out = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         timeout=280)
    assert out.returncode == 0 and "RCCL ok" in out.stdout, out.stdout


@pytest.mark.gpu
def test_bench_two_ranks_without_a_launcher():
    """``python bench.py --gpus 2`` started plainly (no torch.distributed.run around it, as a driver would): the
    parent starts the two ranks itself and rank 0's JSON line comes through with n_gpus = 2.  The box has one
    GPU, so the ranks share it over gloo (RCCL refuses two ranks on one device); the path through the launcher,
    the barriers and the reductions is the one an 8-GPU run takes."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    env["QP_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--partitions", "64", "--slices", "200", "--no-cpu", "--no-qubo", "--single"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-3000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["scaling"] == "weak"
    d = res["distributed"]
    assert d["backend"] == "gloo" and d["world_size"] == 2 and [r["rank"] for r in d["ranks"]] == [0, 1]
    assert d["host_threads_per_rank"]["local_world_size"] == 2
    assert len(res["ms_per_step_per_rank"]) == 2 and res["value"] > 0
