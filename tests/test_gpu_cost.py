"""GPU parity tests: the HIP kernels, called through the C ABI, against the CPU oracle and the
reference-generated fixtures.  Integer distance tables must agree exactly; fp64 tables within 1e-12
relative (the sums are formed in a different order from the reference's sequential loop)."""
import math

import numpy as np
import pytest

from conftest import ALL_CASES, SMALL_CASES, load_case
from oracle import placer_oracle as O

pytestmark = pytest.mark.gpu

RTOL = 1e-12


@pytest.fixture(scope="module")
def gpu():
    import torch
    from quantum_simulator_amd import engine
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X; none is visible")
    dev = torch.device("cuda", 0)
    return torch, engine, engine.Engine(dev), dev


def _close(got, want, exact, nterms=0):
    """Exact for integer tables.  Otherwise 1e-12 relative, widened to the reference's own
    left-to-right summation error bound (nterms * 2**-53) when it adds more than ~9000 terms."""
    if np.isinf(want):
        return np.isinf(got)
    if exact:
        return got == want
    return abs(got - want) <= max(RTOL, nterms * 2.0 ** -53) * max(1.0, abs(want))


def _pad(arr, ldq):
    out = np.zeros(arr.shape[:-1] + (ldq,), np.uint8)
    out[..., :arr.shape[-1]] = arr
    return out


@pytest.mark.parametrize("name", ALL_CASES)
def test_cost_v3_golden(gpu, name):
    torch, engine, eng, dev = gpu
    gc = load_case(name)
    from quantum_simulator_amd.timeslices import CircuitIR
    ir = CircuitIR(gc.Q, gc.gate_q, gc.gate_mr, slices=gc.slices)
    dc = engine.DeviceCircuit(ir, dev)
    dt = engine.DistanceTable(gc.D, dev)
    exact = name.endswith("_i")
    assert (dt.kind == engine.QP_DT_U16) == exact
    for p in ("HQA", "TRHQA", "GTRHQA"):
        if not gc.has(p):
            continue
        A = torch.from_numpy(_pad(gc.part(p), dc.ldq)[None].copy()).to(dev)
        terms = O.cost_v3_terms(gc.slices, gc.gate_q, gc.gate_mr, gc.part(p).tolist(), gc.D.tolist())
        for table in ("packed", "table", "f64"):
            got = float(eng.cost_v3(dc, dt, A, table=table).cpu()[0])
            assert _close(got, gc.cost(p), exact, len(terms)), (p, table, got, gc.cost(p))
            # against the correctly rounded sum the kernel is inside 1e-12 however many terms there are
            assert _close(got, math.fsum(terms), exact), (p, table, got, math.fsum(terms))
    # the three partitions of equal length as one batch, plus a repeat for run-to-run determinism
    parts = [gc.part(p) for p in ("HQA", "TRHQA") if gc.has(p)]
    if parts:
        A = torch.from_numpy(np.stack([_pad(x, dc.ldq) for x in parts] * 3)).to(dev)
        r1 = eng.cost_v3(dc, dt, A).cpu().numpy()
        r2 = eng.cost_v3(dc, dt, A).cpu().numpy()
        assert r1.tobytes() == r2.tobytes()
        assert r1[0] == r1[len(parts)] == r1[2 * len(parts)]


@pytest.mark.parametrize("name", ALL_CASES)
def test_cost_v3_golden_in_large_batches(gpu, name):
    """The reference's own partitions and costs through the kernels that only batches of 64 partitions or more reach
    (four partitions per wave step): every fixture partition 67 times over, interleaved with the other placers'
    partitions of the same length and with shuffled copies in between, so that the four partitions of a wave step
    differ; every copy must cost what the reference's run of it cost."""
    torch, engine, eng, dev = gpu
    gc = load_case(name)
    from quantum_simulator_amd.timeslices import CircuitIR
    ir = CircuitIR(gc.Q, gc.gate_q, gc.gate_mr, slices=gc.slices)
    dc = engine.DeviceCircuit(ir, dev)
    dt = engine.DistanceTable(gc.D, dev)
    exact = name.endswith("_i")
    rng = np.random.default_rng(len(name))
    by_len = {}
    for p in ("HQA", "TRHQA", "GTRHQA"):
        if gc.has(p):
            by_len.setdefault(gc.part(p).shape[0], []).append(p)
    for Tp, placers in by_len.items():
        rows, want = [], []
        for k in range(67):
            for p in placers:
                rows.append(_pad(gc.part(p), dc.ldq))
                want.append(gc.cost(p))
            noise = _pad(gc.part(placers[0]), dc.ldq).copy()  # a partition of the same shape with its rows shuffled
            rng.shuffle(noise, axis=0)
            rows.append(noise)
            want.append(None)
        A = torch.from_numpy(np.stack(rows)).to(dev)
        nterms = ir.nE + ir.T * gc.Q
        for table in ("packed", "table"):
            got = eng.cost_v3(dc, dt, A, table=table).cpu().numpy()
            for b, w in enumerate(want):
                if w is not None:
                    assert _close(got[b], w, exact, nterms), (table, b, got[b], w)


@pytest.mark.parametrize("name", ALL_CASES)
def test_cost_v2_v1_v4_golden(gpu, name):
    torch, engine, eng, dev = gpu
    gc = load_case(name)
    from quantum_simulator_amd.timeslices import CircuitIR
    ir = CircuitIR(gc.Q, gc.gate_q, gc.gate_mr, slices=gc.slices)
    dc = engine.DeviceCircuit(ir, dev)
    dt = engine.DistanceTable(gc.D, dev)
    ht = engine.DistanceTable(gc.hops.astype(np.float64), dev)
    exact = name.endswith("_i")
    for p in ("BFS", "RANDOM"):
        A = torch.from_numpy(_pad(gc.part(p), dc.ldq)[None].copy()).to(dev)
        got = float(eng.cost_v2(dc, dt, A, engine.QP_V2).cpu()[0])
        assert _close(got, gc.cost(p), exact, 2 * ir.nV), (p, got, gc.cost(p))
        got = float(eng.cost_v2(dc, dt, A, engine.QP_V1).cpu()[0])
        assert _close(got, gc.cost_v1(p), exact, 2 * ir.nV), (p, "v1", got, gc.cost_v1(p))
        try:
            want = O.cost_v4(gc.slices, gc.gate_q, gc.gate_mr, gc.part_rows(p), gc.hops.tolist())
        except O.CostError:
            want = float("inf")
        got = float(eng.cost_v2(dc, ht, A, engine.QP_V4).cpu()[0])
        assert got == want, (p, "v4", got, want)
    for p in ("HQA", "GTRHQA"):
        if gc.has(p):
            A = torch.from_numpy(_pad(gc.part(p), dc.ldq)[None].copy()).to(dev)
            got = float(eng.cost_v2(dc, dt, A, engine.QP_V1).cpu()[0])
            assert _close(got, gc.cost_v1(p), exact, 2 * ir.nV), (p, "v1 dense", got)


@pytest.mark.parametrize("name", ALL_CASES)
def test_cost_v4_and_comm_matrix_match_the_reference_runs(gpu, name):
    """qp_slice_cost_v2(QP_V4) and qp_comm_matrix against what the unmodified reference's qubo_placer returned
    for the same partitions (tests/golden/qubo_cost.json, written by oracle/gen_golden_qubo_cost.py:
    per_timeslice_cost, Qubo_placer.py:177-207, and the matrix intercorecommatrix draws, :208-245): BFS's and
    RANDOM's sparse partitions, HQA's dense one, and a BFS partition with one placed qubit removed, on which the
    reference raises (inf / status 1 here).  All partitions of a case go through one call."""
    import json
    import os
    torch, engine, eng, dev = gpu
    from conftest import GOLDEN
    from quantum_simulator_amd.timeslices import CircuitIR
    gold = json.load(open(os.path.join(GOLDEN, "qubo_cost.json")))[name]
    gc = load_case(name)
    ir = CircuitIR(gc.Q, gc.gate_q, gc.gate_mr, slices=gc.slices)
    dc = engine.DeviceCircuit(ir, dev)
    ht = engine.DistanceTable(gc.hops.astype(np.float64), dev)
    names, parts = [], []
    for pname, entry in gold.items():
        arr = gc.part(pname.split("_")[0]).copy()[:ir.T]
        if entry["damage"]:
            t, q = entry["damage"]
            arr[t, q] = 0xFF
        names.append(pname)
        parts.append(_pad(arr, dc.ldq))
        parts[-1][:, gc.Q:] = 0
    A = torch.from_numpy(np.stack(parts)).to(dev)
    cost = eng.cost_v2(dc, ht, A, engine.QP_V4).cpu().numpy()
    M, status = eng.comm_matrix(dc, A, gc.C)
    M, status = M.cpu().numpy(), status.cpu().numpy()
    for b, pname in enumerate(names):
        entry = gold[pname]
        if entry["cost"] is None:
            assert np.isinf(cost[b]), (pname, cost[b], entry["cost_raises"])
        else:
            assert cost[b] == entry["cost"], (pname, cost[b], entry["cost"])
        if "comm_upper" not in entry:
            continue
        if entry["comm_upper"] is None:
            assert status[b] == 1, (pname, entry["comm_raises"])
            continue
        want = np.zeros((gc.C, gc.C), np.int64)
        for a, c, v in entry["comm_upper"]:
            want[a, c] = want[c, a] = v
        assert status[b] == 0 and (M[b] == want).all(), pname
        assert int(M[b].sum()) == entry["comm_total"]


SHAPES = [  # Q, T, C, B, integer weights
    (5, 9, 4, 3, False), (17, 40, 16, 5, True), (63, 70, 64, 4, False), (64, 33, 64, 7, True),
    (100, 21, 100, 3, False), (130, 50, 128, 2, False), (500, 12, 200, 2, True), (1024, 37, 256, 3, True),
    (1024, 6, 256, 2, False), (1000, 5, 16, 2, False), (1500, 7, 64, 2, False), (33, 1, 8, 2, False),
    (33, 2, 8, 2, True), (48, 3, 8, 2, False),
]


def _random_setup(Q, T, C, integer, seed):
    from quantum_simulator_amd import layered_gate_arrays
    from quantum_simulator_amd.timeslices import CircuitIR
    rng = np.random.default_rng(seed)
    q0, q1, _ = layered_gate_arrays(Q, T, 0.5, seed=seed)
    # drop a random third of the gates so slices are ragged (some qubits idle, some slices short)
    keep = rng.random(len(q0)) > 0.33
    ir = CircuitIR.from_gate_tables(Q, q0[keep], q1[keep])
    # ring + chords, random weights
    eu = list(range(C)) + [int(x) for x in rng.integers(0, C, size=C // 2)]
    ev = [(i + 1) % C for i in range(C)] + [int(x) for x in rng.integers(0, C, size=C // 2)]
    ok = [k for k in range(len(eu)) if eu[k] != ev[k]]
    eu, ev = [eu[k] for k in ok], [ev[k] for k in ok]
    w = rng.integers(1, 6, size=len(eu)).astype(float) if integer else rng.uniform(1, 3, size=len(eu))
    return ir, eu, ev, w, rng


@pytest.mark.parametrize("Q,T,C,B,integer", SHAPES)
def test_cost_v3_random(gpu, Q, T, C, B, integer):
    torch, engine, eng, dev = gpu
    ir, eu, ev, w, rng = _random_setup(Q, T, C, integer, seed=Q * 7 + T)
    D = engine.apsp(C, eu, ev, w)
    dc, dt = engine.DeviceCircuit(ir, dev), engine.DistanceTable(D, dev)
    assert (dt.kind == engine.QP_DT_U16) == integer
    Tp = ir.T + (1 if Q % 2 else 0)
    A = rng.integers(0, C, size=(B, Tp, dc.ldq), dtype=np.uint8)
    A[:, :, Q:] = rng.integers(0, 256, size=(B, Tp, dc.ldq - Q), dtype=np.uint8)  # pad bytes are garbage
    A[0, 1:] = A[0, :1]  # one partition that never moves
    got = eng.cost_v3(dc, dt, torch.from_numpy(A).to(dev)).cpu().numpy()
    Dl = D.tolist()
    for b in range(B):
        # 1e-12 against the correctly rounded sum; against the reference's left-to-right sum its own error bound
        # counts too once there are tens of thousands of terms (two cores: every term is the same number)
        terms = O.cost_v3_terms(ir.slices, ir.gate_q, ir.gate_mr, A[b, :, :Q].tolist(), Dl)
        assert _close(got[b], math.fsum(terms), integer), (b, got[b], math.fsum(terms))
        want = O.cost_v3(ir.slices, ir.gate_q, ir.gate_mr, A[b, :, :Q].tolist(), Dl)
        assert _close(got[b], want, integer, len(terms)), (b, got[b], want)


SHORT_ROWS = [  # Q, T, C, B, integer weights, pmove: rows of 16 to 512 bytes (1 to 32 pieces, 64 to 2 rows per wave step)
    (9, 700, 8, 5, True, 0.5), (30, 333, 16, 6, False, 0.3), (40, 6100, 20, 3, True, 0.2), (70, 150, 64, 9, False, 0.9),
    (128, 77, 100, 4, False, 0.5), (130, 200, 256, 3, True, 0.5), (260, 1300, 200, 2, False, 0.1), (500, 64, 130, 5, False, 1.0),
    (512, 3, 256, 4, True, 1.0), (18, 138, 16, 700, False, 0.3),
    # fp64 tables too large for LDS on partitions of 24 KB or more: core pairs are counted
    (64, 500, 256, 4, False, 0.9), (512, 700, 144, 3, False, 0.4), (100, 5000, 180, 2, False, 0.05),
    # ... and on shorter partitions: gathers from device memory
    (60, 90, 150, 3, False, 0.5),
]


@pytest.mark.parametrize("Q,T,C,B,integer,pmove", SHORT_ROWS)
def test_cost_v3_short_rows_in_groups(gpu, Q, T, C, B, integer, pmove):
    """Rows of at most 512 bytes: a wave takes several slices per step and the gates come from the grouped list
    (items of up to 256 steps, so the 6100-slice case has more than one item per partition); every table form
    (16-bit and fp64 in LDS at either pitch, fp64 in device memory or replaced by pair counts) against the vectorised
    CPU sum, and the same partitions again for a caller that passes no grouped list."""
    from oracle.cpu_variants import NumpyCost
    torch, engine, eng, dev = gpu
    ir, eu, ev, w, rng = _random_setup(Q, T, C, integer, seed=Q * 13 + T)
    D = engine.apsp(C, eu, ev, w)
    dc, dt = engine.DeviceCircuit(ir, dev), engine.DistanceTable(D, dev)
    assert dc.ldq <= 512 and dc.c.group_ldq == dc.ldq and dc.c.group_sched
    Tp = ir.T + (1 if Q % 2 else 0)
    A = np.empty((B, Tp, dc.ldq), np.uint8)
    A[:, 0] = rng.integers(0, C, size=(B, dc.ldq), dtype=np.uint8)
    for t in range(1, Tp):
        move = rng.random((B, dc.ldq)) < pmove
        A[:, t] = np.where(move, rng.integers(0, C, size=(B, dc.ldq), dtype=np.uint8), A[:, t - 1])
    A[:, :, Q:] = rng.integers(0, 256, size=(B, Tp, dc.ldq - Q), dtype=np.uint8)  # pad bytes are garbage
    Ad = torch.from_numpy(A).to(dev)
    got = eng.cost_v3(dc, dt, Ad).cpu().numpy()
    assert got.tobytes() == eng.cost_v3(dc, dt, Ad).cpu().numpy().tobytes()
    nc = NumpyCost(ir.slices, ir.gate_q, ir.gate_mr, D)
    nterms = ir.nE + ir.T * Q
    for b in range(min(B, 12)):
        want = nc.cost(A[b, :ir.T, :Q])
        assert _close(got[b], want, integer, nterms), (b, got[b], want)
    saved = dc.c.group_sched, dc.c.group_ptr, dc.c.group_ldq
    dc.c.group_sched, dc.c.group_ptr, dc.c.group_ldq = None, None, 0
    try:
        plain = eng.cost_v3(dc, dt, Ad).cpu().numpy()
    finally:
        dc.c.group_sched, dc.c.group_ptr, dc.c.group_ldq = saved
    assert all(_close(plain[b], got[b], integer, nterms) for b in range(B)), (plain, got)
    # a row pitch other than the one the list was laid out for: the list is not used
    if dc.ldq + 16 <= 512:
        wide = np.zeros((B, Tp, dc.ldq + 16), np.uint8)
        wide[:, :, :dc.ldq] = A
        again = eng.cost_v3(dc, dt, torch.from_numpy(wide).to(dev)).cpu().numpy()
        assert all(_close(again[b], got[b], integer, nterms) for b in range(B)), (again, got)


SHORT_ROWS_QUADS = [  # Q, T, C, B, integer weights, pmove: short rows, an LDS table indexed by packed core pairs, 64 partitions or more
    (9, 700, 8, 64, True, 0.5), (30, 333, 16, 67, False, 0.3), (40, 6100, 20, 65, True, 0.2), (70, 150, 64, 130, False, 0.9),
    (128, 77, 100, 64, True, 0.5), (260, 1300, 60, 66, False, 0.1), (500, 64, 130, 70, True, 1.0), (512, 3, 16, 64, False, 1.0),
    (63, 498, 64, 301, False, 1.0), (18, 138, 16, 9000, False, 0.05), (33, 1, 8, 64, True, 1.0), (33, 2, 8, 66, False, 1.0),
    # fewer steps in the whole batch than the GPU holds waves (a run per step, none without one)
    (195, 73, 33, 386, False, 1.0), (97, 1080, 2, 76, False, 1.0), (8, 6202, 48, 83, True, 0.02),
    # fp64 tables of 65 to 128 cores: at a pitch of 128 entries
    (63, 300, 100, 70, False, 1.0), (128, 77, 128, 64, False, 0.5), (40, 700, 65, 66, False, 0.3),
]


@pytest.mark.parametrize("Q,T,C,B,integer,pmove", SHORT_ROWS_QUADS)
def test_cost_v3_short_rows_four_partitions(gpu, Q, T, C, B, integer, pmove):
    """Rows of at most 512 bytes, from 64 partitions on, a table that sits in LDS indexed by packed core pairs (up to
    16 cores, up to 64 cores, 16-bit tables of more): a wave step takes the rows of four consecutive partitions
    (interleaved strip words, one look-up per byte lane); B is not always a multiple of four, rows are ragged, items
    are parts of a partition (6100 slices) or several quads (9000 partitions of 138 slices).  Every partition against
    the vectorised CPU sum, and against the same partitions in batches too small for this form."""
    from oracle.cpu_variants import NumpyCost
    torch, engine, eng, dev = gpu
    ir, eu, ev, w, rng = _random_setup(Q, T, C, integer, seed=Q * 19 + T)
    D = engine.apsp(C, eu, ev, w)
    dc, dt = engine.DeviceCircuit(ir, dev), engine.DistanceTable(D, dev)
    assert (dt.kind == engine.QP_DT_U16) == integer
    assert dc.ldq <= 512 and dc.c.group_ldq == dc.ldq and dc.c.group_sched
    Tp = ir.T + (1 if Q % 2 else 0)
    A = np.empty((B, Tp, dc.ldq), np.uint8)
    A[:, 0] = rng.integers(0, C, size=(B, dc.ldq), dtype=np.uint8)
    for t in range(1, Tp):
        move = rng.random((B, dc.ldq)) < pmove
        A[:, t] = np.where(move, rng.integers(0, C, size=(B, dc.ldq), dtype=np.uint8), A[:, t - 1])
    A[:, :, Q:] = rng.integers(0, 256, size=(B, Tp, dc.ldq - Q), dtype=np.uint8)  # pad bytes are garbage
    A[1, 1:] = A[1, :1]  # one partition that never moves
    Ad = torch.from_numpy(A).to(dev)
    got = eng.cost_v3(dc, dt, Ad, table="table").cpu().numpy()
    assert got.tobytes() == eng.cost_v3(dc, dt, Ad, table="table").cpu().numpy().tobytes()
    nc = NumpyCost(ir.slices, ir.gate_q, ir.gate_mr, D)
    nterms = ir.nE + ir.T * Q
    for b in list(range(min(B, 140))) + list(range(max(0, B - 9), B)):
        want = nc.cost(A[b, :ir.T, :Q])
        assert _close(got[b], want, integer, nterms), (b, got[b], want)
    small = torch.cat([eng.cost_v3(dc, dt, Ad[i:i + 60].contiguous(), table="table") for i in range(0, min(B, 600), 60)]).cpu().numpy()
    assert all(_close(small[b], got[b], integer, nterms) for b in range(len(small))), (small, got)


@pytest.mark.parametrize("Q,T,C,B,integer", [(18, 40, 16, 50001, False), (100, 70, 64, 26001, True), (40, 300, 9, 25000, True)])
def test_cost_v3_many_short_partitions(gpu, Q, T, C, B, integer):
    """Tens of thousands of partitions of a circuit of few steps: a wave then walks several consecutive partitions
    as one stream.  Every partition must cost, bit for bit, what it costs in a batch small enough for one
    partition per item, and a sample of them what the vectorised CPU sum says."""
    from oracle.cpu_variants import NumpyCost
    torch, engine, eng, dev = gpu
    ir, eu, ev, w, rng = _random_setup(Q, T, C, integer, seed=Q * 17 + T)
    D = engine.apsp(C, eu, ev, w)
    dc, dt = engine.DeviceCircuit(ir, dev), engine.DistanceTable(D, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(Q * T)
    A = torch.randint(0, C, (B, ir.T, dc.ldq), device=dev, generator=g, dtype=torch.int16).to(torch.uint8)
    keep = torch.rand((B, ir.T, dc.ldq), device=dev, generator=g) < 0.6
    for t in range(1, ir.T):
        A[:, t] = torch.where(keep[:, t], A[:, t - 1], A[:, t])
    got = eng.cost_v3(dc, dt, A).cpu().numpy()
    # (the last batch takes the odd partition along: fewer than 64 partitions take another kernel, whose fp64 sums
    # round in another order)
    starts = list(range(0, B - 999, 1000))
    small = torch.cat([eng.cost_v3(dc, dt, A[i:(i + 1000 if i != starts[-1] else B)].contiguous()) for i in starts]).cpu().numpy()
    assert got.tobytes() == small.tobytes()
    nc = NumpyCost(ir.slices, ir.gate_q, ir.gate_mr, D)
    for b in list(range(5)) + [int(x) for x in rng.integers(0, B, size=40)] + [B - 1]:
        want = nc.cost(A[b, :, :Q].cpu().numpy())
        assert _close(got[b], want, integer, ir.nE + ir.T * Q), (b, got[b], want)


GROUPED = [  # Q, T, C, B, integer weights: rows over 512 bytes, table in LDS at pitch 256, 64 partitions or more
    (1000, 70, 256, 67, True), (1024, 9, 64, 64, False), (600, 41, 40, 130, False), (513, 3, 200, 65, True),
    (777, 2, 64, 66, False), (1024, 1, 256, 64, True), (900, 30, 81, 70, True), (1024, 12, 96, 64, True), (640, 25, 97, 65, True),
    (1024, 9, 100, 64, False), (600, 41, 128, 70, False), (900, 30, 65, 66, False),
]


@pytest.mark.parametrize("Q,T,C,B,integer", GROUPED)
def test_cost_v3_four_partitions_per_wave(gpu, Q, T, C, B, integer):
    """From 64 partitions on, long rows with an LDS-resident table are costed four partitions per wave step
    (one interleaved strip of core ids); B is not always a multiple of four.  Every partition against the
    vectorised CPU sum (exact for integer tables, 1e-12 otherwise), the first and last few against the
    sequential oracle."""
    from oracle.cpu_variants import NumpyCost
    torch, engine, eng, dev = gpu
    ir, eu, ev, w, rng = _random_setup(Q, T, C, integer, seed=Q * 11 + T)
    D = engine.apsp(C, eu, ev, w)
    dc, dt = engine.DeviceCircuit(ir, dev), engine.DistanceTable(D, dev)
    assert (dt.kind == engine.QP_DT_U16) == integer
    A = np.empty((B, ir.T, dc.ldq), np.uint8)
    A[:, 0] = rng.integers(0, C, size=(B, dc.ldq), dtype=np.uint8)
    for t in range(1, ir.T):
        move = rng.random((B, dc.ldq)) < 0.4
        A[:, t] = np.where(move, rng.integers(0, C, size=(B, dc.ldq), dtype=np.uint8), A[:, t - 1])
    A[:, :, Q:] = rng.integers(0, 256, size=(B, ir.T, dc.ldq - Q), dtype=np.uint8)  # pad bytes are garbage
    got = eng.cost_v3(dc, dt, torch.from_numpy(A).to(dev), table="table").cpu().numpy()
    nc = NumpyCost(ir.slices, ir.gate_q, ir.gate_mr, D)
    for b in range(B):
        want = nc.cost(A[b, :, :Q])
        assert _close(got[b], want, integer), (b, got[b], want)
    Dl = D.tolist()
    for b in list(range(3)) + list(range(B - 5, B)):
        want = O.cost_v3(ir.slices, ir.gate_q, ir.gate_mr, A[b, :, :Q].tolist(), Dl)
        assert _close(got[b], want, integer), (b, got[b], want)


@pytest.mark.parametrize("Q,T,C,B,integer", [(100, 21, 100, 3, False), (600, 40, 64, 3, True), (1500, 7, 256, 2, False),
                                             (17, 2, 8, 2, True), (5, 1, 4, 2, False)])
def test_cost_v3_slice_by_slice_and_in_windows(gpu, Q, T, C, B, integer):
    """Per-slice terms against the oracle's sum over one slice at a time; their total against the batch kernel;
    and the circuit costed in three contiguous pieces, each from a window of rows that starts one row before the
    piece (what a rank of a multi-GPU job holds): the pieces reproduce the per-slice values bit for bit."""
    from quantum_simulator_amd.driver import slice_shard
    torch, engine, eng, dev = gpu
    ir, eu, ev, w, rng = _random_setup(Q, T, C, integer, seed=Q * 3 + T)
    D = engine.apsp(C, eu, ev, w)
    dc, dt = engine.DeviceCircuit(ir, dev), engine.DistanceTable(D, dev)
    T = ir.T
    A = np.empty((B, T, dc.ldq), np.uint8)
    A[:, 0] = rng.integers(0, C, size=(B, dc.ldq), dtype=np.uint8)
    for t in range(1, T):
        move = rng.random((B, dc.ldq)) < 0.3
        A[:, t] = np.where(move, rng.integers(0, C, size=(B, dc.ldq), dtype=np.uint8), A[:, t - 1])
    Ad = torch.from_numpy(A).to(dev)
    got = eng.cost_v3_slices(dc, dt, Ad).cpu().numpy()
    assert got.shape == (B, T)
    Dl = D.tolist()
    for b in range(B):
        rows = A[b, :, :Q].tolist()
        for t in range(T):
            want = 0
            for g in ir.slices[t]:
                qs = ir.gate_q[g]
                if len(qs) == 2 and not ir.gate_mr[g] and rows[t][qs[0]] != rows[t][qs[1]]:
                    want += Dl[rows[t][qs[0]]][rows[t][qs[1]]]
            if t >= 2:
                want += sum(Dl[rows[t - 1][q]][rows[t][q]] for q in range(Q) if rows[t][q] != rows[t - 1][q])
            assert _close(got[b, t], want, integer), (b, t, got[b, t], want)
    total = eng.cost_v3(dc, dt, Ad).cpu().numpy()
    for b in range(B):
        assert _close(got[b].sum(), total[b], integer), (b, got[b].sum(), total[b])
    pieces = []
    for rank in range(3):
        t0, t1, row0 = slice_shard(T, rank, 3)
        window = torch.from_numpy(np.ascontiguousarray(A[:, row0:max(t1, row0 + 1)])).to(dev)
        pieces.append(eng.cost_v3_slices(dc, dt, window, t0, t1, row0).cpu().numpy())
    assert np.concatenate(pieces, axis=1).tobytes() == got.tobytes()
    from quantum_simulator_amd._lib import QplaceError
    with pytest.raises(QplaceError):
        eng.cost_v3_slices(dc, dt, Ad[:, :T - 1].contiguous() if T > 1 else Ad, 0, T + (T == 1), 0)  # a window that is one row short


PAIR_SHAPES = [  # Q, T, C, B, chord edges dropped so the graph falls apart, moves per row
    (1024, 40, 256, 3, False, 0.3), (600, 70, 200, 2, False, 0.05), (1021, 5, 140, 2, False, 1.0),
    (700, 33, 256, 2, True, 0.3), (1024, 130, 255, 5, False, 0.02),
]


@pytest.mark.parametrize("Q,T,C,B,split,pmove", PAIR_SHAPES)
def test_cost_v3_pair_counts(gpu, Q, T, C, B, split, pmove):
    """Real-valued table of more than 134 cores, rows longer than 512 bytes: the default path counts
    unordered core pairs on chip and multiplies by the symmetrised table once per partition; forcing
    look-ups ("table") takes the one-pass-per-band kernel.  Both against the oracle at 1e-12 relative
    (the tolerance the symmetrisation is sized for), including infinite distances."""
    torch, engine, eng, dev = gpu
    ir, eu, ev, w, rng = _random_setup(Q, T, C, False, seed=Q * 13 + T)
    if split:  # two rings that do not touch: distances between them are infinite
        h = C // 2
        eu = list(range(C))
        ev = [(i + 1) % h if i < h else h + (i + 1 - h) % (C - h) for i in range(C)]
        w = rng.uniform(1, 3, size=C)
    D = engine.apsp(C, eu, ev, w)
    dc, dt = engine.DeviceCircuit(ir, dev), engine.DistanceTable(D, dev)
    assert dt.kind == engine.QP_DT_F64 and dt.symmetric == 1 and dt.packed.symmetric == 1 and dt.table_only.symmetric == 0
    assert np.isinf(D).any() == split
    A = np.empty((B, ir.T, dc.ldq), np.uint8)
    A[:, 0] = rng.integers(0, C, size=(B, dc.ldq), dtype=np.uint8)
    for t in range(1, ir.T):
        move = rng.random((B, dc.ldq)) < pmove
        A[:, t] = np.where(move, rng.integers(0, C, size=(B, dc.ldq), dtype=np.uint8), A[:, t - 1])
    if split:  # partition 0 stays inside the first ring: finite cost
        A[0] %= C // 2
    Ad = torch.from_numpy(A).to(dev)
    got = eng.cost_v3(dc, dt, Ad).cpu().numpy()
    band = eng.cost_v3(dc, dt, Ad, table="table").cpu().numpy()
    Dl = D.tolist()
    for b in range(B):
        want = O.cost_v3(ir.slices, ir.gate_q, ir.gate_mr, A[b, :, :Q].tolist(), Dl)
        if np.isinf(want):
            assert np.isinf(got[b]) and np.isinf(band[b]), (b, got[b], band[b])
        else:
            assert abs(got[b] - want) <= 1e-12 * abs(want), (b, got[b], want)
            assert abs(band[b] - want) <= 1e-12 * abs(want), (b, band[b], want)
    if split:
        assert np.isfinite(got[0]) and np.isinf(got[1:]).all()
    # run to run the sums are formed in the same order
    again = eng.cost_v3(dc, dt, Ad).cpu().numpy()
    assert again.tobytes() == got.tobytes()


def test_cost_v3_asymmetric_table_is_looked_up(gpu):
    """A table that is not symmetric (nothing the placers produce: their graphs are undirected) is never
    symmetrised: the flag stays clear and directed look-ups give the directed sums."""
    torch, engine, eng, dev = gpu
    Q, T, C, B = 1024, 12, 160, 2
    ir, eu, ev, w, rng = _random_setup(Q, T, C, False, seed=99)
    D = engine.apsp(C, eu, ev, w)
    D = D * (1.0 + 0.5 * np.triu(np.ones((C, C)), 1))  # entries above the diagonal 50% larger
    dc, dt = engine.DeviceCircuit(ir, dev), engine.DistanceTable(D, dev)
    assert dt.kind == engine.QP_DT_F64 and dt.symmetric == 0 and dt.packed.symmetric == 0
    A = rng.integers(0, C, size=(B, ir.T, dc.ldq), dtype=np.uint8)
    got = eng.cost_v3(dc, dt, torch.from_numpy(A).to(dev)).cpu().numpy()
    Dl = D.tolist()
    for b in range(B):
        want = O.cost_v3(ir.slices, ir.gate_q, ir.gate_mr, A[b, :, :Q].tolist(), Dl)
        assert abs(got[b] - want) <= 1e-12 * abs(want), (b, got[b], want)


GRIDS = [  # topology, nx, ny, integer edge weight, closed form expected
    ("mesh", 4, 4, 2.0, True), ("mesh", 2, 8, 3.0, True), ("mesh", 16, 16, 2.0, True), ("mesh", 3, 4, 2.0, True),
    ("mesh", 4, 3, 2.0, False), ("mesh", 8, 2, 5.0, True), ("mesh", 2, 1, 1.0, True), ("line", 7, 1, 1.0, True),
    ("line", 200, 1, 3.0, True), ("torus", 8, 8, 2.0, True), ("torus", 4, 16, 1.0, True), ("torus", 16, 16, 1.0, True),
    ("torus", 3, 8, 1.0, False), ("star", 9, 1, 2.0, False), ("all_to_all", 4, 4, 1.0, False),
]


@pytest.mark.parametrize("topo,nx,ny,wt,closed", GRIDS)
@pytest.mark.parametrize("Q,T", [(40, 23), (600, 9), (1024, 40), (1021, 5)])
def test_cost_v3_grid_metric(gpu, topo, nx, ny, wt, closed, Q, T):
    """Integer-weight meshes, lines and tori: the kernel that computes distances from core ids, the
    kernel that looks them up in the uint16 table, and the oracle agree exactly."""
    torch, engine, eng, dev = gpu
    from quantum_simulator_amd import layered_gate_arrays
    from quantum_simulator_amd.multicore import multicore
    from quantum_simulator_amd.timeslices import CircuitIR
    arch = multicore(nx, ny, ncp_qubits=4, edge_weight_range=(wt, wt), network_type=topo)
    dt = engine.DistanceTable.from_network(arch.network, dev)
    assert dt.kind == engine.QP_DT_U16
    assert bool(dt.grid.valid) == closed, (dt.grid.valid, dt.grid.xbits, dt.grid.wrap_x, dt.grid.wrap_y)
    C = dt.C
    rng = np.random.default_rng(Q + T + C)
    q0, q1, _ = layered_gate_arrays(Q, T, 0.7, seed=Q + C)
    keep = rng.random(len(q0)) > 0.2
    ir = CircuitIR.from_gate_tables(Q, q0[keep], q1[keep])
    dc = engine.DeviceCircuit(ir, dev)
    B = 3
    A = rng.integers(0, C, size=(B, ir.T, dc.ldq), dtype=np.uint8)
    A[:, :, Q:] = rng.integers(0, 256, size=(B, ir.T, dc.ldq - Q), dtype=np.uint8)
    Ad = torch.from_numpy(A).to(dev)
    got = eng.cost_v3(dc, dt, Ad).cpu().numpy()
    tab = eng.cost_v3(dc, dt, Ad, table="table").cpu().numpy()
    assert got.tobytes() == tab.tobytes()
    Dl = dt.D.tolist()
    want = O.cost_v3(ir.slices, ir.gate_q, ir.gate_mr, A[0, :, :Q].tolist(), Dl)
    assert got[0] == want


@pytest.mark.parametrize("topo,nx,ny,wt,closed,Q,T", [
    ("mesh", 2, 1, 1.8444218515250481, True, 256, 60),   # the shipped example's machine: one link (example_config.txt:1-12)
    ("mesh", 2, 1, 1.5, True, 40, 7),
    ("mesh", 4, 4, 0.1, True, 100, 30),                  # 0.1 + 0.1 + 0.1 != 0.3: entries differ from hops * w by rounding
    ("mesh", 16, 16, 2.75, True, 1024, 9),
    ("mesh", 16, 16, 1.0 / 3.0, True, 600, 12),
    ("torus", 8, 8, 1.37, True, 63, 40),
    ("line", 64, 1, 2.5, True, 200, 11),
    ("line", 256, 1, 0.7, True, 1500, 4),
    ("mesh", 3, 3, 1.5, False, 30, 10),                  # no power-of-two row length: look-ups
    ("star", 8, 1, 1.5, False, 30, 10),
])
def test_cost_v3_scaled_grid_metric(gpu, topo, nx, ny, wt, closed, Q, T):
    """One real weight on every link (what multicore.py:159-161 draws for a one-link machine, or any
    edge_weight_range (w, w)): the table is w times an integer grid metric up to the rounding of its
    entries, and the dense cost then counts hops by closed form and multiplies once.  Against the table
    look-ups of the same launch and against the oracle's sum of the table's own entries: 1e-12."""
    torch, engine, eng, dev = gpu
    from quantum_simulator_amd import layered_gate_arrays
    from quantum_simulator_amd.multicore import multicore
    from quantum_simulator_amd.timeslices import CircuitIR
    arch = multicore(nx, ny, ncp_qubits=4, edge_weight_range=(wt, wt), network_type=topo)
    dt = engine.DistanceTable.from_network(arch.network, dev)
    assert dt.kind == engine.QP_DT_F64
    assert bool(dt.grid.valid) == closed and (dt.grid_scale == wt) == closed, (dt.grid.valid, dt.grid_scale)
    C = dt.C
    rng = np.random.default_rng(Q + T + C)
    q0, q1, _ = layered_gate_arrays(Q, T, 0.7, seed=Q + C)
    keep = rng.random(len(q0)) > 0.2
    ir = CircuitIR.from_gate_tables(Q, q0[keep], q1[keep])
    dc = engine.DeviceCircuit(ir, dev)
    B = 3
    A = rng.integers(0, C, size=(B, ir.T, dc.ldq), dtype=np.uint8)
    A[:, :, Q:] = rng.integers(0, 256, size=(B, ir.T, dc.ldq - Q), dtype=np.uint8)
    Ad = torch.from_numpy(A).to(dev)
    got = eng.cost_v3(dc, dt, Ad).cpu().numpy()
    tab = eng.cost_v3(dc, dt, Ad, table="table").cpu().numpy()
    Dl = dt.D.tolist()
    for b in range(B):
        assert _close(got[b], tab[b], False), (b, got[b], tab[b])
        terms = O.cost_v3_terms(ir.slices, ir.gate_q, ir.gate_mr, A[b, :, :Q].tolist(), Dl)
        assert _close(got[b], math.fsum(terms), False), (b, got[b], math.fsum(terms))
    # the entry point without a workspace takes the same road
    out = torch.empty(B, dtype=torch.float64, device=dev)
    eng.cost_v3_static(dc, dt, Ad, out)
    assert out.cpu().numpy().tobytes() == got.tobytes()


@pytest.mark.parametrize("Q,T,C,B,integer", [s for s in SHAPES if s[2] <= 255][:10])
def test_cost_v2_random(gpu, Q, T, C, B, integer):
    torch, engine, eng, dev = gpu
    ir, eu, ev, w, rng = _random_setup(Q, T, C, integer, seed=Q * 11 + T)
    D = engine.apsp(C, eu, ev, w)
    H = engine.hops(C, eu, ev)
    dc, dt, ht = engine.DeviceCircuit(ir, dev), engine.DistanceTable(D, dev), engine.DistanceTable(H, dev)
    A = np.full((B, ir.T, dc.ldq), 0xFF, np.uint8)
    for b in range(B):
        for t in range(ir.T):
            act = ir.active_qubits(t)
            A[b, t, act] = rng.integers(0, C, size=len(act))
    # instance 1: drop one active qubit somewhere (an error in V2, maybe not in V1)
    if B > 1 and ir.nV:
        v = ir.visit_records[int(rng.integers(0, ir.nV))]
        A[1, v["t"], v["q"]] = 0xFF
    Ad = torch.from_numpy(A).to(dev)
    Dl, Hl = D.tolist(), H.tolist()
    for variant, fn, tab, tabl in ((engine.QP_V2, O.cost_v2, dt, Dl), (engine.QP_V1, O.cost_v1, dt, Dl),
                                   (engine.QP_V4, O.cost_v4, ht, Hl)):
        got = eng.cost_v2(dc, tab, Ad, variant).cpu().numpy()
        for b in range(B):
            rows = [[None if c == 0xFF else int(c) for c in r[:Q]] for r in A[b]]
            try:
                want = float(fn(ir.slices, ir.gate_q, ir.gate_mr, rows, tabl))
            except O.CostError:
                want = float("inf")
            assert _close(got[b], want, integer or variant == engine.QP_V4), (variant, b, got[b], want)
    # V1 with a short partition stops early; V2 reports inf
    if ir.T > 2:
        short = torch.from_numpy(A[:, :ir.T - 1].copy()).to(dev)
        got = eng.cost_v2(dc, dt, short, engine.QP_V2).cpu().numpy()
        assert np.isinf(got).all()
        got = eng.cost_v2(dc, dt, short, engine.QP_V1).cpu().numpy()
        rows = [[None if c == 0xFF else int(c) for c in r[:Q]] for r in A[0, :ir.T - 1]]
        want = float(O.cost_v1(ir.slices, ir.gate_q, ir.gate_mr, rows, Dl))
        assert _close(got[0], want, integer)


@pytest.mark.parametrize("Q,T,C,B,drop,integer", [
    (1024, 37, 255, 1024, 0.05, True), (1000, 21, 16, 1100, 0.3, True), (600, 50, 100, 1030, 0.0, True),
    (513, 9, 2, 1024, 0.2, True), (777, 130, 64, 1024, 0.1, True), (1024, 260, 200, 260, 0.1, True),
    (1024, 37, 255, 2000, 0.05, False), (1000, 300, 16, 230, 0.3, False), (600, 129, 100, 520, 0.0, False),
    (513, 70, 2, 1024, 0.2, False),
    # rows of at most 512 bytes: several partitions per wave step (16-bit tables; fp64 up to 127 cores)
    (63, 260, 64, 4100, 0.1, True), (63, 260, 64, 4100, 0.1, False), (200, 220, 255, 1500, 0.2, True),
    (512, 120, 30, 1100, 0.0, False), (17, 240, 9, 9001, 0.0, True), (100, 250, 16, 2500, 0.3, False),
    (300, 30, 100, 7400, 0.1, True), (200, 220, 127, 1500, 0.2, False), (256, 100, 65, 2700, 0.0, False),
    # the same kernel on longer rows and up to 16 cores, one partition per step
    (1024, 70, 16, 1000, 0.1, True), (530, 140, 9, 500, 0.2, False), (1000, 66, 3, 1030, 0.0, True)])
def test_cost_v2_row_walk(gpu, Q, T, C, B, drop, integer):
    """Batches on a circuit that visits most qubits in most slices are walked row by row (16-bit tables looked up,
    symmetric fp64 tables through core-pair counts; few partitions in chunks of slices): equal to the visit-by-visit
    kernel on every partition and to the oracle on a sample, absent qubits and the errors they cause included."""
    from quantum_simulator_amd import layered_gate_arrays
    from quantum_simulator_amd.timeslices import CircuitIR
    torch, engine, eng, dev = gpu
    rng = np.random.default_rng(Q + T)
    q0, q1, _ = layered_gate_arrays(Q, T, 0.7, seed=Q + T)
    keep = rng.random(len(q0)) >= drop
    ir = CircuitIR.from_gate_tables(Q, q0[keep], q1[keep])
    assert ir.nV * 8 >= ir.T * ((Q + 15) // 16 * 16)
    eu = list(range(C - 1)) + [int(x) for x in rng.integers(0, C, size=C)]
    ev = list(range(1, C)) + [int(x) for x in rng.integers(0, C, size=C)]
    ok = [k for k in range(len(eu)) if eu[k] != ev[k]]
    eu, ev = [eu[k] for k in ok], [ev[k] for k in ok]
    D = engine.apsp(C, eu, ev, rng.integers(1, 9, size=len(eu)).astype(float) if integer else rng.uniform(1, 3, size=len(eu)))
    H = engine.hops(C, eu, ev)
    walk, byvisit = engine.DeviceCircuit(ir, dev), engine.DeviceCircuit(ir, dev, visit_mask=False)
    assert walk.c.visit_mask and walk.c.visit_prev and not byvisit.c.visit_mask
    dt, ht = engine.DistanceTable(D, dev), engine.DistanceTable(H, dev)
    assert (dt.kind == engine.QP_DT_U16) == integer and ht.kind == engine.QP_DT_U16
    assert integer or dt.packed.symmetric
    Tp = ir.T + (Q % 2)
    A = rng.integers(0, C, size=(B, Tp, walk.ldq), dtype=np.uint8)
    A[:, :, Q:] = rng.integers(0, 256, size=(B, Tp, walk.ldq - Q), dtype=np.uint8)
    # qubits that stay where they are most of the time, as a placer leaves them
    stay = rng.random((B, Tp, Q)) < 0.6
    for t in range(1, Tp):
        A[:, t, :Q] = np.where(stay[:, t], A[:, t - 1, :Q], A[:, t, :Q])
    # partitions 0..39: only the visited qubits are placed; 40..59: some unvisited ones are absent;
    # 60..79: a visited qubit is absent somewhere (an error unless it had no core before and is alone in its gate)
    vm = ir.visit_mask(walk.ldq)[:, :Q] != 0
    for b in range(40):
        A[b, :ir.T, :Q][~vm] = 0xFF
    for b in range(40, 60):
        hole = (rng.random((ir.T, Q)) < 0.1) & ~vm
        A[b, :ir.T, :Q][hole] = 0xFF
    for b in range(60, 80):
        v = ir.visit_records[int(rng.integers(0, ir.nV))]
        A[b, v["t"], v["q"]] = 0xFF
        if b >= 70:  # from the first slice on: no previous core either
            A[b, :v["t"], v["q"]] = 0xFF
    Ad = torch.from_numpy(A).to(dev)
    Dl, Hl = D.tolist(), H.tolist()
    for variant, fn, tab, tabl in ((engine.QP_V2, O.cost_v2, dt, Dl), (engine.QP_V4, O.cost_v4, ht, Hl), (engine.QP_V1, O.cost_v1, dt, Dl)):
        got = eng.cost_v2(walk, tab, Ad, variant).cpu().numpy()
        ref = eng.cost_v2(byvisit, tab, Ad, variant).cpu().numpy()
        exact = integer or variant == engine.QP_V4
        if exact:
            assert got.tobytes() == ref.tobytes(), (variant, np.flatnonzero(got != ref)[:8])
        else:  # pair counts times the symmetrised table against a sum in visit order
            assert (np.isinf(got) == np.isinf(ref)).all()
            fin = np.isfinite(ref)
            assert np.allclose(got[fin], ref[fin], rtol=1e-12, atol=0)
        assert np.isfinite(got[:60]).all()
        assert variant == engine.QP_V1 or np.isinf(got[60:80]).any()  # (V1 fails only on a qubit that loses its core)
        for b in (0, 41, 60, 65, 70, 75, 79, B - 1):
            rows = [[None if c == 0xFF else int(c) for c in r[:Q]] for r in A[b, :ir.T]]
            try:
                want = float(fn(ir.slices, ir.gate_q, ir.gate_mr, rows, tabl))
            except O.CostError:
                want = float("inf")
            assert _close(got[b], want, exact, nterms=ir.nV + ir.nE), (variant, b, got[b], want)
    # V1 on partitions that end early stops there
    if ir.T > 2:
        short = torch.from_numpy(A[:, :ir.T - 1].copy()).to(dev)
        got = eng.cost_v2(walk, dt, short, engine.QP_V1).cpu().numpy()
        ref = eng.cost_v2(byvisit, dt, short, engine.QP_V1).cpu().numpy()
        assert (np.isinf(got) == np.isinf(ref)).all()
        fin = np.isfinite(ref)
        assert np.allclose(got[fin], ref[fin], rtol=0 if integer else 1e-12, atol=0)


def test_cost_v3_rejects_bad_arguments(gpu):
    torch, engine, eng, dev = gpu
    ir, eu, ev, w, rng = _random_setup(20, 5, 4, True, seed=1)
    dc, dt = engine.DeviceCircuit(ir, dev), engine.DistanceTable(engine.apsp(4, eu, ev, w), dev)
    A = torch.zeros((1, ir.T - 1, dc.ldq), dtype=torch.uint8, device=dev)
    from quantum_simulator_amd._lib import QplaceError
    with pytest.raises(QplaceError):
        eng.cost_v3(dc, dt, A)  # fewer rows than slices: IndexError in the reference
    with pytest.raises(ValueError):
        eng.cost_v3(dc, dt, torch.zeros((1, ir.T, 24), dtype=torch.uint8, device=dev))


@pytest.mark.parametrize("name", SMALL_CASES)
@pytest.mark.parametrize("kind", ["hqa", "trhqa", "gtrhqa"])
def test_cost_matrix_bits(gpu, name, kind):
    """Every matrix the placement loop builds on this case, bit for bit (the oracle's matrices are
    themselves pinned to the ones the reference handed to scipy)."""
    torch, engine, eng, dev = gpu
    gc = load_case(name)
    if not gc.has(kind.upper()):
        pytest.skip("capacity too small")
    from quantum_simulator_amd.timeslices import CircuitIR
    ir = CircuitIR(gc.Q, gc.gate_q, gc.gate_mr, slices=gc.slices)
    dc, dt = engine.DeviceCircuit(ir, dev), engine.DistanceTable(gc.D, dev)
    rec = []
    Dl = gc.D.tolist()
    if kind == "gtrhqa":
        O.place_gtrhqa(gc.slices, gc.gate_q, gc.gate_mr, gc.Q, gc.C, gc.cap, Dl, rec)
    else:
        O.place_hqa_family(gc.slices, gc.gate_q, gc.gate_mr, gc.Q, gc.C, gc.cap, Dl, kind, rec)
    rec = rec[:400]
    if not rec:
        pytest.skip("no matrices on this case")
    B = len(rec)
    max_ops = max(len(r[1]) for r in rec)
    t = np.array([r[0] for r in rec], np.int32)
    a = np.zeros((B, dc.ldq), np.uint8)
    ops = np.zeros((B, max_ops), np.int32)
    n_ops = np.zeros(B, np.int32)
    avail = np.zeros((B, gc.C), np.uint8)
    n_avail = np.zeros(B, np.int32)
    for b, (tt, rops, rav, M, live) in enumerate(rec):
        a[b, :gc.Q] = live
        ops[b, :len(rops)] = [ir.edge_of_gate[g] for g, _, _ in rops]
        n_ops[b] = len(rops)
        avail[b, :len(rav)] = rav
        n_avail[b] = len(rav)
    code = {"hqa": engine.QP_CM_HQA, "trhqa": engine.QP_CM_TRHQA, "gtrhqa": engine.QP_CM_GTRHQA}[kind]
    out = eng.cost_matrix(code, dc, dt, *[torch.from_numpy(x).to(dev) for x in (t, a, ops, n_ops, avail, n_avail)])
    out = out.cpu().numpy()
    for b, (tt, rops, rav, M, live) in enumerate(rec):
        want = np.array(M)
        got = out[b, :len(rops), :len(rav)]
        assert got.tobytes() == want.tobytes(), (name, kind, b)


@pytest.mark.parametrize("integer", [True, False])
def test_cost_v3_full_size_against_c_oracle(gpu, integer):
    """The benchmark's shape (1024 qubits x 1000 slices, 256 cores) on 48 partitions, dense and sparse
    movement, against the C restatement (sequential sums): exact for the uint16 table."""
    torch, engine, eng, dev = gpu
    import random
    from oracle import c_oracle
    from quantum_simulator_amd import layered_gate_arrays
    from quantum_simulator_amd.multicore import multicore, graph_tables
    from quantum_simulator_amd.timeslices import CircuitIR
    Q, T, B = 1024, 1000, 48
    q0, q1, _ = layered_gate_arrays(Q, T, 0.5, seed=99)
    ir = CircuitIR.from_gate_tables(Q, q0, q1)
    random.seed(5)
    arch = multicore(16, 16, ncp_qubits=8, edge_weight_range=(2.0, 2.0) if integer else (1.0, 3.0), network_type="mesh")
    nodes, eu, ev, w = graph_tables(arch.network)
    D = engine.apsp(256, eu, ev, w)
    dc, dt = engine.DeviceCircuit(ir, dev), engine.DistanceTable(D, dev)
    rng = np.random.default_rng(1)
    A = np.empty((B, T, Q), np.uint8)
    A[:, 0] = np.arange(Q) // 8
    for t in range(1, T):
        pm = np.where(np.arange(B) % 3 == 0, 0.02, 0.95)[:, None]
        move = rng.random((B, Q)) < pm
        A[:, t] = np.where(move, rng.integers(0, 256, size=(B, Q)), A[:, t - 1])
    got = eng.cost_v3(dc, dt, torch.from_numpy(A).to(dev)).cpu().numpy()
    if integer:  # closed-form distances (the default for this table) and table look-ups
        assert dt.grid.valid
        assert got.tobytes() == eng.cost_v3(dc, dt, torch.from_numpy(A).to(dev), table="table").cpu().numpy().tobytes()
    gt = c_oracle.GateTables(ir.slices, ir.gate_q, ir.gate_mr)
    nterms = ir.nE + T * Q
    for b in range(B):
        want = c_oracle.cost_v3(gt, Q, A[b], D)
        assert _close(got[b], want, integer, nterms), (b, got[b], want)
    if not integer:
        # against the correctly rounded sum of the same terms the kernel stays inside 1e-12 however many
        # terms there are (the wider bound above only covers the reference's own left-to-right rounding)
        Dl = D.tolist()
        worst_fsum = worst_seq = 0.0
        for b in (0, 1, 2, 47):  # two rarely-moving and two mostly-moving partitions
            terms = O.cost_v3_terms(ir.slices, ir.gate_q, ir.gate_mr, A[b].tolist(), Dl)
            exact = math.fsum(terms)
            assert abs(got[b] - exact) <= RTOL * exact, (b, got[b], exact)
            worst_fsum = max(worst_fsum, abs(got[b] - exact) / exact)
            worst_seq = max(worst_seq, abs(got[b] - c_oracle.cost_v3(gt, Q, A[b], D)) / exact)
        print(f"\nfull-size fp64 cost: max relative deviation from fsum {worst_fsum:.3e}, "
              f"from the sequential sum {worst_seq:.3e} ({len(terms)} terms)")
    # size-independent property: the cost is additive over a split of the timeline when the second
    # part starts with a repeated row (no movement is charged for rows 0 and 1 of either part)
    assert got.tobytes() == eng.cost_v3(dc, dt, torch.from_numpy(A).to(dev)).cpu().numpy().tobytes()


WIDE = [  # Q, T, topology, nx, ny, integer weight (0 = uniform(1,3)), B, share of qubits that move per row
    (2048, 70, "mesh", 4, 4, 2.0, 3, 0.1),       # closed form and uint16 table in LDS
    (1100, 300, "mesh", 16, 16, 0, 2, 0.05),     # fp64 table of 256 cores: core pairs are counted in LDS
    (2048, 260, "mesh", 16, 16, 0, 5, 1.0),      # the same with every qubit moving, eight rows per workgroup step
    (5000, 40, "mesh", 14, 14, 0, 3, 0.5),       # 196 cores, 313 pieces per row
    (16384, 6, "torus", 16, 16, 0, 2, 0.3),      # the longest rows that are counted: one row per workgroup step
    (3000, 20, "torus", 8, 8, 1.0, 2, 1.0),      # rings; every qubit moves
    (1500, 33, "star", 9, 1, 0, 2, 0.3),         # fp64 table in LDS, pieces not a power of two
    (4099, 9, "mesh", 4, 3, 3.0, 2, 0.5),        # integer table without a closed form
    (20000, 7, "mesh", 4, 4, 2.0, 2, 0.2),       # four pieces per thread
    (30001, 4, "line", 7, 1, 0, 2, 0.2),
    (32767, 3, "mesh", 16, 16, 0, 1, 0.02),      # the longest row the circuit tables allow
    # closed forms on rows of 1040 to 2048 bytes: a wave per row, two 1 KB pieces per lane; the last piece ragged, missing
    # for most lanes, or full; rings
    (1100, 50, "torus", 8, 8, 1.0, 3, 1.0), (1025, 9, "line", 16, 1, 3.0, 2, 0.5), (1999, 33, "mesh", 8, 4, 1.0, 70, 0.3),
    (2040, 300, "mesh", 16, 16, 2.0, 4, 0.1),
]


@pytest.mark.parametrize("Q,T,topo,nx,ny,wt,B,pmove", WIDE)
def test_cost_v3_wide_rows(gpu, Q, T, topo, nx, ny, wt, B, pmove):
    """Rows longer than 1024 bytes (a workgroup per item, groups of threads per run of rows) against the C
    restatement: exact for integer tables, closed form == table look-ups bit for bit."""
    torch, engine, eng, dev = gpu
    import random
    from oracle import c_oracle
    from quantum_simulator_amd import layered_gate_arrays
    from quantum_simulator_amd.multicore import multicore
    from quantum_simulator_amd.timeslices import CircuitIR
    random.seed(Q)
    arch = multicore(nx, ny, ncp_qubits=4, edge_weight_range=(wt, wt) if wt else (1.0, 3.0), network_type=topo)
    dt = engine.DistanceTable.from_network(arch.network, dev)
    assert (dt.kind == engine.QP_DT_U16) == bool(wt)
    C = dt.C
    rng = np.random.default_rng(Q + T)
    q0, q1, _ = layered_gate_arrays(Q, T, 0.6, seed=Q + C)
    keep = rng.random(len(q0)) > 0.25
    ir = CircuitIR.from_gate_tables(Q, q0[keep], q1[keep])
    dc = engine.DeviceCircuit(ir, dev)
    assert dc.ldq > 1024
    A = np.empty((B, ir.T, dc.ldq), np.uint8)
    A[:, 0] = rng.integers(0, C, size=(B, dc.ldq), dtype=np.uint8)
    for t in range(1, ir.T):
        move = rng.random((B, dc.ldq)) < pmove
        A[:, t] = np.where(move, rng.integers(0, C, size=(B, dc.ldq), dtype=np.uint8), A[:, t - 1])
    A[:, :, Q:] = rng.integers(0, 256, size=(B, ir.T, dc.ldq - Q), dtype=np.uint8)  # pad bytes are garbage
    Ad = torch.from_numpy(A).to(dev)
    got = eng.cost_v3(dc, dt, Ad).cpu().numpy()
    if wt:
        assert got.tobytes() == eng.cost_v3(dc, dt, Ad, table="table").cpu().numpy().tobytes()
    elif C * C * 8 > 150 * 1024 and dc.ldq <= 16384:
        # counted pairs times the symmetrised table against look-ups of the directed entries
        looked_up = eng.cost_v3(dc, dt, Ad, table="table").cpu().numpy()
        assert all(_close(looked_up[b], got[b], False) for b in range(B)), (looked_up, got)
    assert got.tobytes() == eng.cost_v3(dc, dt, Ad).cpu().numpy().tobytes()
    # a caller that passes no padded gate list: the gates come from the edge records (another order of terms)
    saved = dc.c.edges_sched, dc.c.sched_ptr
    dc.c.edges_sched, dc.c.sched_ptr = None, None
    try:
        plain = eng.cost_v3(dc, dt, Ad).cpu().numpy()
    finally:
        dc.c.edges_sched, dc.c.sched_ptr = saved
    assert all(_close(plain[b], got[b], bool(wt)) for b in range(B)), (plain, got)
    gt = c_oracle.GateTables(ir.slices, ir.gate_q, ir.gate_mr)
    nterms = ir.nE + ir.T * Q
    for b in range(B):
        want = c_oracle.cost_v3(gt, Q, A[b, :, :Q], dt.D)
        assert _close(got[b], want, bool(wt), nterms), (b, got[b], want)


def test_degenerate_circuits(gpu):
    """No gates at all (zero slices) and one-qubit gates only (slices without counted gates): what the
    reference's loops do with them — nothing, or movement only."""
    torch, engine, eng, dev = gpu
    from quantum_simulator_amd.timeslices import CircuitIR
    rng = np.random.default_rng(3)
    C, cap = 4, 3
    D = engine.apsp(C, [0, 1, 2], [1, 2, 3], [1.5, 2.0, 2.5])
    dt = engine.DistanceTable(D, dev)
    # zero slices
    ir = CircuitIR.from_gate_tables(5, np.array([], np.int32), np.array([], np.int32))
    dc = engine.DeviceCircuit(ir, dev)
    A = torch.from_numpy(rng.integers(0, C, size=(2, 3, dc.ldq), dtype=np.uint8)).to(dev)
    assert eng.cost_v3(dc, dt, A).cpu().tolist() == [0.0, 0.0]
    assert eng.cost_v2(dc, dt, A, engine.QP_V2).cpu().tolist() == [0.0, 0.0]
    # partitions of a circuit without slices have no rows at all: the counts are zero, not an error
    none = torch.empty((2, 0, dc.ldq), dtype=torch.uint8, device=dev)
    assert int(eng.core_activity(dc, none, C).sum().item()) == 0
    M, status = eng.comm_matrix(dc, none, C)
    assert int(M.sum().item()) == 0 and int(status.sum().item()) == 0
    assert eng.core_activity(dc, A, C).cpu().numpy().sum() == 0
    init = torch.from_numpy(rng.integers(0, C, size=(2, dc.ldq), dtype=np.uint8)).to(dev)
    rows = eng.gtrhqa_place(dc, dt, init, cap)
    assert rows.shape == (2, 1, dc.ldq) and (rows[:, 0] == init).all()
    # one-qubit gates only: Q = 1100 takes the long-row kernels, Q = 20 the short-row ones
    for Q in (20, 1100):
        q0 = rng.integers(0, Q, size=6 * Q).astype(np.int32)
        ir = CircuitIR.from_gate_tables(Q, q0, np.full(len(q0), -1, np.int32))
        assert ir.nE == 0 and ir.T > 2
        dc = engine.DeviceCircuit(ir, dev)
        An = rng.integers(0, C, size=(2, ir.T, dc.ldq), dtype=np.uint8)
        got = eng.cost_v3(dc, dt, torch.from_numpy(An).to(dev)).cpu().numpy()
        for b in range(2):
            want = O.cost_v3(ir.slices, ir.gate_q, ir.gate_mr, An[b, :, :Q].tolist(), D.tolist())
            assert _close(got[b], want, False)
        init = np.zeros((2, dc.ldq), np.uint8)
        init[:, :Q] = rng.integers(0, C, size=(2, Q))
        rows = eng.gtrhqa_place(dc, dt, torch.from_numpy(init).to(dev), 600).cpu().numpy()
        assert (rows == init[:, None, :]).all()  # nothing to place: every row repeats the start
