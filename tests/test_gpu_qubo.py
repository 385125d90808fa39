"""QUBO path on the GPU.  The reference's solver stack (pyqubo, neal) is not available anywhere this
code runs and the reference ships no fixture for it, so sample-for-sample parity is unpinned; what is
checked: the MFMA energies and fields equal the integer CPU restatement of the objective
(Qubo_placer.py:84-133) exactly, the annealer's bookkeeping equals those energies, brute-force optima
on tiny models, decode/repair against the oracle, and solution quality."""
import itertools

import numpy as np
import pytest

from oracle import placer_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X; none is visible")
    return torch, torch.device("cuda", 0)


def _model(Q, T, C, cap, seed, topo="mesh"):
    from quantum_simulator_amd import layered_gate_arrays
    from quantum_simulator_amd.engine import hops
    from quantum_simulator_amd.multicore import build_topology, graph_tables
    from quantum_simulator_amd.qubo import QuboModel
    from quantum_simulator_amd.timeslices import CircuitIR
    rng = np.random.default_rng(seed)
    q0, q1, _ = layered_gate_arrays(Q, T, 0.5, seed=seed)
    keep = rng.random(len(q0)) > 0.3
    ir = CircuitIR.from_gate_tables(Q, q0[keep], q1[keep])
    side = int(round(C ** 0.5))
    G = build_topology(topo, side, side) if topo == "mesh" else build_topology(topo, C)
    nodes, eu, ev, _ = graph_tables(G)
    H = hops(len(nodes), eu, ev)
    names = [f"q[{i}]" for i in range(Q)]
    return QuboModel(ir, len(nodes), cap, H, names), names


def _oracle_energy(qm, names, state):
    act, inter = O.qubo_slices(qm.ir.slices, qm.ir.gate_q, names)
    assert act == qm.act and inter == qm.inter
    x = [[[int(state[qm.var_x(t, j, i)]) for i in range(qm.n[t])] for j in range(qm.C)] for t in range(qm.T)]
    y = [[[int(state[qm.var_y(t, j, p)]) for p in range(qm.cap)] for j in range(qm.C)] for t in range(qm.T)]
    return O.qubo_energy(act, inter, qm.C, qm.cap, qm.H.tolist(), x, y), x


@pytest.mark.parametrize("Q,T,C,cap,seed", [(6, 3, 4, 3, 1), (20, 5, 4, 6, 2), (40, 4, 9, 5, 3), (33, 6, 16, 3, 4),
                                            (400, 2, 16, 32, 9), (500, 2, 16, 32, 11)])
def test_energy_and_fields_exact(gpu, Q, T, C, cap, seed):
    """Random states against the integer restatement.  The two large cases have slices of 384 and 288, and of more
    than 384 state bytes per core: up to 384 the contraction keeps the coupling matrix in registers (12 K steps
    with extra half tiles on every wave, 9 with two), above that it stages both operands through LDS."""
    torch, dev = gpu
    from quantum_simulator_amd.qubo import DeviceQubo
    qm, names = _model(Q, T, C, cap, seed)
    dq = DeviceQubo(qm, dev)
    rng = np.random.default_rng(seed)
    R = 128
    X = np.zeros((R, qm.Npad), np.int8)
    real = qm.real_mask()
    X[:, real] = rng.integers(0, 2, size=(R, int(real.sum())))
    X[0] = 0
    X[1, real] = 1
    Xd = torch.from_numpy(X).to(dev)
    e = dq.energy(Xd).cpu().numpy()
    for r in list(range(6)) + [R - 1]:
        want, _ = _oracle_energy(qm, names, X[r])
        assert e[r] == want, (r, e[r], want)
    # fields: flipping v changes the energy by (1 - 2 s_v) (a_v + F_v)
    F = dq.fields(Xd).cpu().numpy()
    a = qm.linear()
    for r in (2, 5):
        base, _ = _oracle_energy(qm, names, X[r])
        for v in rng.choice(np.flatnonzero(real), size=12, replace=False):
            s2 = X[r].copy()
            s2[v] ^= 1
            want, _ = _oracle_energy(qm, names, s2)
            assert want - base == (1 - 2 * int(X[r, v])) * (a[v] + F[v, r]), (r, v)


def test_energy_bruteforce_optimum_tiny(gpu):
    """2 qubits, one gate, 2 cores of capacity 2, one slice: all 2^8 states through the MFMA path; the
    minimum is 0 and is reached exactly by 'both qubits on one core, that core's two slack bits set'."""
    torch, dev = gpu
    from quantum_simulator_amd.qubo import DeviceQubo, QuboModel
    from quantum_simulator_amd.timeslices import CircuitIR
    ir = CircuitIR(2, [(0, 1)], [False])
    qm = QuboModel(ir, 2, 2, np.array([[0, 1], [1, 0]]), ["q[0]", "q[1]"])
    dq = DeviceQubo(qm, dev)
    X = np.zeros((256, qm.Npad), np.int8)
    order = [qm.var_x(0, j, i) for j in range(2) for i in range(2)] + [qm.var_y(0, j, p) for j in range(2) for p in range(2)]
    for s in range(256):
        for b, v in enumerate(order):
            X[s, v] = (s >> b) & 1
    e = dq.energy(torch.from_numpy(X).to(dev)).cpu().numpy()
    assert e.min() == 0
    best = np.flatnonzero(e == 0)
    assert len(best) == 2
    for s in best:
        rows = qm.decode(X[s])
        assert rows[0][0] == rows[0][1]
    for s in (0, 37, 255):
        x = [[[int(X[s, qm.var_x(0, j, i)]) for i in range(2)] for j in range(2)]]
        y = [[[int(X[s, qm.var_y(0, j, p)]) for p in range(2)] for j in range(2)]]
        assert e[s] == O.qubo_energy([[0, 1]], [[(0, 1)]], 2, 2, [[0, 1], [1, 0]], x, y)


@pytest.mark.parametrize("Q,T,C,cap,seed", [(20, 5, 4, 6, 2), (40, 6, 16, 4, 7), (24, 7, 9, 4, 5)])
def test_anneal_bookkeeping_and_quality(gpu, Q, T, C, cap, seed):
    torch, dev = gpu
    from quantum_simulator_amd.qubo import DeviceQubo
    qm, names = _model(Q, T, C, cap, seed)
    dq = DeviceQubo(qm, dev)
    R = 128
    X0, d0 = dq.anneal(R, 0, seed=9)                 # zero sweeps: the random initial states
    assert int(d0.abs().max()) == 0
    e0 = dq.energy(X0)
    X1, d1 = dq.anneal(R, 300, seed=9)               # same seed: same start
    e1 = dq.energy(X1)
    assert torch.equal(e1 - e0, d1), "summed accepted dE must equal the exact energy difference"
    X2, d2 = dq.anneal(R, 300, seed=9)
    assert torch.equal(X1, X2) and torch.equal(d1, d2), "same seed, same result"
    # continuing from given states works and padding stays zero
    X3, d3 = dq.anneal(R, 50, seed=11, init=X1)
    assert torch.equal(dq.energy(X3) - e1, d3)
    assert int(X3.cpu().numpy()[:, ~qm.real_mask()].sum()) == 0
    # quality: annealing beats both its random start and the block placement (active qubit i of each
    # slice on core i // cap, slack bits filled), which satisfies every constraint
    rows0 = []
    for t in range(qm.T):
        row = [None] * Q
        for i, q in enumerate(qm.act[t]):
            row[q] = i // cap
        rows0.append(row)
    block = torch.from_numpy(np.tile(qm.state_from_rows(rows0), (R, 1))).to(dev)
    e_block = int(dq.energy(block)[0])
    want, _ = _oracle_energy(qm, names, qm.state_from_rows(rows0))
    assert e_block == want
    assert int(e1.min()) < int(e0.min()) and int(e1.min()) <= e_block
    # decode/repair of the best replica: every active qubit placed, no core over capacity
    best = int(e1.argmin())
    state = X1[best].cpu().numpy()
    _, x = _oracle_energy(qm, names, state)
    rows = qm.decode(state)
    act, _ = O.qubo_slices(qm.ir.slices, qm.ir.gate_q, names)
    assert rows == O.qubo_decode(act, qm.C, qm.cap, x, Q)
    for t, row in enumerate(rows):
        placed = [c for c in row if c is not None]
        assert len(placed) == qm.n[t]
        if qm.n[t] <= qm.C * qm.cap:
            assert max(np.bincount(placed, minlength=qm.C)) <= max(qm.cap, -(-qm.n[t] // qm.C))


def test_decode_repair_matches_oracle(gpu):
    torch, dev = gpu
    qm, names = _model(30, 4, 4, 5, 8)
    rng = np.random.default_rng(0)
    act, _ = O.qubo_slices(qm.ir.slices, qm.ir.gate_q, names)
    for trial in range(20):
        state = np.zeros(qm.Npad, np.int8)
        for t in range(qm.T):
            for i in range(qm.n[t]):
                for j in range(qm.C):
                    state[qm.var_x(t, j, i)] = rng.random() < 0.3   # zero, one or several cores per qubit
        x = [[[int(state[qm.var_x(t, j, i)]) for i in range(qm.n[t])] for j in range(qm.C)] for t in range(qm.T)]
        assert qm.decode(state) == O.qubo_decode(act, qm.C, qm.cap, x, 30)


def test_qubo_placer_end_to_end(gpu):
    from conftest import case_arch, case_circuit, load_case
    from quantum_simulator_amd.placers import qubo_placer
    gc = load_case("shor5_mesh2x2_u")
    circ, arch = case_circuit(gc), case_arch(gc)
    placer = qubo_placer(num_sweeps=400)
    part = placer.place_per_timeslice(circ, arch)
    assert len(part) == gc.T
    names = gc.meta["qubits"]
    qobj = circ.qubits
    for t, row in enumerate(part):
        active = {q for g in gc.slices[t] for q in gc.gate_q[g]}
        assert {qobj.index(q) for q in row} == active
        load = {}
        for c in row.values():
            load[c] = load.get(c, 0) + 1
        assert max(load.values()) <= gc.cap
    cost = placer.per_timeslice_cost(circ, part, arch)
    rows = [[None] * gc.Q for _ in part]
    for t, row in enumerate(part):
        for q, c in row.items():
            rows[t][qobj.index(q)] = c
    assert cost == O.cost_v4(gc.slices, gc.gate_q, gc.gate_mr, rows, gc.hops.tolist())
    again = qubo_placer(num_sweeps=400).place_per_timeslice(circ, arch)
    assert again == part


def test_anneal_not_worse_than_plain_cpu_annealing(gpu):
    """Same model, same number of sweeps, same temperature range: the best energy over the GPU's replicas
    is at least as low as the best of a plain sequential CPU annealer (oracle.qubo_cpu_anneal, which reads
    the coefficients off the restated objective, not off anything the product builds)."""
    torch, dev = gpu
    from quantum_simulator_amd.qubo import DeviceQubo
    qm, names = _model(6, 3, 4, 3, 1)
    act, inter = O.qubo_slices(qm.ir.slices, qm.ir.gate_q, names)
    H = qm.H.tolist()
    # position of every variable in the compact (unpadded) order x[t][j][i], then y[t][j][p]
    index = []
    for t in range(qm.T):
        index += [qm.var_x(t, j, i) for j in range(qm.C) for i in range(qm.n[t])]
        index += [qm.var_y(t, j, p) for j in range(qm.C) for p in range(qm.cap)]
    N = len(index)

    def energy_of(s):
        st = np.zeros(qm.Npad, np.int8)
        st[index] = s
        x = [[[int(st[qm.var_x(t, j, i)]) for i in range(qm.n[t])] for j in range(qm.C)] for t in range(qm.T)]
        y = [[[int(st[qm.var_y(t, j, p)]) for p in range(qm.cap)] for j in range(qm.C)] for t in range(qm.T)]
        return O.qubo_energy(act, inter, qm.C, qm.cap, H, x, y)

    sweeps = 200
    hot, cold = qm.beta_range()
    cpu_best = O.qubo_cpu_anneal(energy_of, N, sweeps, hot, cold, replicas=8, seed=3)
    dq = DeviceQubo(qm, dev)
    X, _ = dq.anneal(128, sweeps, seed=3)
    gpu_best = int(dq.energy(X).min())
    assert gpu_best <= cpu_best, (gpu_best, cpu_best)


def _energy_of(dq, qm, torch, dev, states):
    """Energies of a few states through the MFMA path (batch padded to 128 replicas)."""
    X = np.zeros((128, qm.Npad), np.int8)
    X[:len(states)] = states
    return dq.energy(torch.from_numpy(X).to(dev)).cpu().numpy()[:len(states)]


def test_three_qubit_gate_couples_all_pairs(gpu):
    """A gate on three qubits makes all three pairs interact (Qubo_placer.py:58-66): every qubit has degree 2.
    Hand-derived energies: all three on one core with three slack bits set: 0.  Two on core 0 (two slack
    bits), one on core 1 (one slack bit): core 0 contributes 10 * (2 + 2 - 2 * 1), core 1 10 * 2: 40.
    One qubit per core on three cores: 10 * 6 = 60."""
    torch, dev = gpu
    from quantum_simulator_amd.qubo import DeviceQubo, QuboModel
    from quantum_simulator_amd.timeslices import CircuitIR
    ir = CircuitIR(3, [(0, 1, 2)], [False])
    names = ["q[0]", "q[1]", "q[2]"]
    H = np.array([[0, 1, 2], [1, 0, 1], [2, 1, 0]])
    qm = QuboModel(ir, 3, 3, H, names)
    assert qm.inter[0] == [(0, 1), (0, 2), (1, 2)] and qm.deg[0].tolist() == [2, 2, 2]
    dq = DeviceQubo(qm, dev)
    rows = [[[0, 0, 0]], [[0, 0, 1]], [[0, 1, 2]]]
    states = [qm.state_from_rows(r) for r in rows]
    e = _energy_of(dq, qm, torch, dev, states)
    assert e.tolist() == [0, 40, 60]
    for s, want in zip(states, (0, 40, 60)):
        assert _oracle_energy(qm, names, s)[0] == want


def test_long_moves_beyond_int8(gpu):
    """A coupling of 10 * hops above 127 (more than an int8 matrix entry holds): on a line of 15 cores a qubit
    that is on core 0 in slice 0 and on core 14 in slice 1 costs 10 * 14 = 140 and nothing else (one-hot, slack
    bit set); moving it to core 13 costs 130, to core 12 120."""
    torch, dev = gpu
    from quantum_simulator_amd.engine import hops
    from quantum_simulator_amd.qubo import DeviceQubo, QuboModel
    from quantum_simulator_amd.timeslices import CircuitIR
    ir = CircuitIR(1, [(0,), (0,)], [False, False])
    assert ir.T == 2
    C = 15
    H = hops(C, list(range(C - 1)), list(range(1, C)))
    assert H[0, 14] == 14
    qm = QuboModel(ir, C, 1, H, ["q[0]"])
    assert 10 * int(qm.H.max()) > 127  # more than an int8 coefficient holds
    dq = DeviceQubo(qm, dev)
    states = [qm.state_from_rows([[0], [c]]) for c in (14, 13, 12, 0)]
    e = _energy_of(dq, qm, torch, dev, states)
    assert e.tolist() == [140, 130, 120, 0]
    for s, want in zip(states, (140, 130, 120, 0)):
        assert _oracle_energy(qm, ["q[0]"], s)[0] == want
    # the local field of the slice-1 bit on core 14 sees the slice-0 bit
    X = np.zeros((128, qm.Npad), np.int8)
    X[0] = states[0]
    F = dq.fields(torch.from_numpy(X).to(dev)).cpu().numpy()
    assert F[qm.var_x(1, 14, 0), 0] == 140 - 8  # 10 * 14 from the move, -8 from the core's slack bit


def test_active_qubits_are_ordered_by_name(gpu):
    """Active qubits are sorted by str(qubit) (Qubo_placer.py:48): 'q[10]' comes before 'q[2]', so in a slice
    with a gate on q[2], q[10] variable 0 is q[10]; the coupling to the next slice follows the qubit, not the
    position.  Slice 0: gate (q[2], q[10]), q[10] on core 0 and q[2] on core 1 (one slack bit each): the
    Laplacian term is 10 * (1 + 1) = 20.  Slice 1: gate (q[2], q[3]), both on core 1: 0.  q[2] stays on core
    1: no move.  Total 20 (coupling position 0 of slice 0 with position 0 of slice 1 would give 30)."""
    torch, dev = gpu
    from quantum_simulator_amd.qubo import DeviceQubo, QuboModel
    from quantum_simulator_amd.timeslices import CircuitIR
    names = [f"q[{i}]" for i in range(11)]
    ir = CircuitIR(11, [(2, 10), (2, 3)], [False, False])
    assert ir.T == 2
    qm = QuboModel(ir, 2, 2, np.array([[0, 1], [1, 0]]), names)
    assert qm.act == [[10, 2], [2, 3]]
    assert qm.nxt[0].tolist() == [-1, 0]
    dq = DeviceQubo(qm, dev)
    row0 = [None] * 11
    row0[10], row0[2] = 0, 1
    row1 = [None] * 11
    row1[2], row1[3] = 1, 1
    s = qm.state_from_rows([row0, row1])
    assert s[qm.var_x(0, 0, 0)] == 1 and s[qm.var_x(0, 1, 1)] == 1  # q[10] is variable 0 of slice 0
    e = _energy_of(dq, qm, torch, dev, [s])
    assert e.tolist() == [20]
    assert _oracle_energy(qm, names, s)[0] == 20
    rows = qm.decode(s)
    assert rows[0][10] == 0 and rows[0][2] == 1 and rows[1][2] == 1 and rows[1][3] == 1
    # q[2] moved to core 0 in slice 1 (q[3] with it): one hop, +10
    row1b = [None] * 11
    row1b[2], row1b[3] = 0, 0
    assert _energy_of(dq, qm, torch, dev, [qm.state_from_rows([row0, row1b])]).tolist() == [30]


def test_benchmark_size_model(gpu):
    """The QUBO benchmark's own model: 256-qubit layered circuit x 8 slices on a 4x4 mesh, capacity 20
    (35 328 variables, 35 row tiles per slice), 4096 replicas, so that the 128 x 256 tile and the launch
    order of the replica tiles are the ones checked.  Energies of the all-zero state, the block placement,
    random states and the best annealed state equal the restated objective; fields equal energy differences
    of single flips; the annealer's summed dE equals the energy difference over 50 sweeps."""
    torch, dev = gpu
    from quantum_simulator_amd import layered_gate_arrays
    from quantum_simulator_amd.engine import hops
    from quantum_simulator_amd.multicore import build_topology, graph_tables
    from quantum_simulator_amd.qubo import DeviceQubo, QuboModel
    from quantum_simulator_amd.timeslices import CircuitIR
    Q, T, cap, R = 256, 8, 20, 4096
    q0, q1, _ = layered_gate_arrays(Q, T, 0.5, seed=4242)
    ir = CircuitIR.from_gate_tables(Q, q0, q1)
    nodes, eu, ev, _ = graph_tables(build_topology("mesh", 4, 4))
    names = [f"q[{i}]" for i in range(Q)]
    qm = QuboModel(ir, len(nodes), cap, hops(len(nodes), eu, ev), names)
    assert qm.N == 35328 and qm.Npad == 8 * 4608 and int(qm.P[0]) == 288
    dq = DeviceQubo(qm, dev)
    rng = np.random.default_rng(7)
    real = qm.real_mask()
    X0, d0 = dq.anneal(R, 0, seed=5)          # the annealer's random start
    e0 = dq.energy(X0)
    X1, d1 = dq.anneal(R, 50, seed=5)
    e1 = dq.energy(X1)
    assert torch.equal(e1 - e0, d1), "summed accepted dE must equal the exact energy difference"
    X = X1.clone()
    rows0 = [[(qm.pos[t][q] // cap if q in qm.pos[t] else None) for q in range(Q)] for t in range(qm.T)]
    block = qm.state_from_rows(rows0)
    X[0] = 0
    X[1] = torch.from_numpy(block).to(dev)
    dense = np.zeros(qm.Npad, np.int8)
    dense[real] = rng.integers(0, 2, size=int(real.sum()))
    X[2] = torch.from_numpy(dense).to(dev)
    best = int(e1[3:].argmin()) + 3
    check = [0, 1, 2, best, 127, 128, 255, 256, 2049, R - 1]  # both halves of a 256-replica tile, several tiles
    e = dq.energy(X).cpu().numpy()
    Xh = X.cpu().numpy()
    for r in check:
        want, _ = _oracle_energy(qm, names, Xh[r])
        assert e[r] == want, (r, e[r], want)
    assert e[0] == qm.const == 4 * int(qm.n.sum())
    F = dq.fields(X).cpu().numpy()
    a = qm.linear()
    for r in (2, best, R - 1):
        base, _ = _oracle_energy(qm, names, Xh[r])
        for v in rng.choice(np.flatnonzero(real), size=6, replace=False):
            s2 = Xh[r].copy()
            s2[v] ^= 1
            want, _ = _oracle_energy(qm, names, s2)
            assert want - base == (1 - 2 * int(Xh[r, v])) * (a[v] + F[v, r]), (r, v)
