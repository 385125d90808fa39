"""qp_slice_cost_v3_static (the entry point that allocates nothing) against qp_slice_cost_v3, through the
C ABI, bit for bit: two calls in a row on the same GPU, the second with a smaller input, so that what the
first call left in the static scratch array cannot leak into the second; then the size limit."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(torch, engine, dev, Q, T, ncores, B, integer, seed):
    from quantum_simulator_amd import layered_gate_arrays
    from quantum_simulator_amd.timeslices import CircuitIR
    rng = np.random.default_rng(seed)
    q0, q1, _ = layered_gate_arrays(Q, T, 0.5, seed=seed)
    keep = rng.random(len(q0)) > 0.2
    ir = CircuitIR.from_gate_tables(Q, q0[keep], q1[keep])
    eu = list(range(ncores))
    ev = [(i + 1) % ncores for i in range(ncores)]
    w = rng.integers(1, 5, size=ncores).astype(float) if integer else rng.uniform(1, 3, size=ncores)
    D = engine.apsp(ncores, eu, ev, w)
    dc, dt = engine.DeviceCircuit(ir, dev), engine.DistanceTable(D, dev)
    A = torch.from_numpy(rng.integers(0, ncores, size=(B, ir.T, dc.ldq), dtype=np.uint8)).to(dev)
    return dc, dt, A


# (Q, T, cores, B, integer weights): the fp64 pair-count path (which also keeps its symmetrised table in
# the scratch), the LDS-table paths for short and 1 KB rows, rows longer than 1 KB; each followed by a
# smaller problem of another kind
CASES = [
    [(1024, 60, 256, 9, False), (1024, 7, 200, 2, False)],
    [(1024, 40, 64, 6, True), (40, 9, 8, 3, True)],
    [(2048, 12, 100, 4, False), (300, 5, 16, 1, False)],
    [(500, 33, 256, 5, False), (1024, 3, 256, 1, True)],
    # short rows, 64 partitions or more: four partitions per wave step, a quad summed from the parts several waves
    # left in the scratch (long partitions: a run per wave; few steps in all: a run per step)
    [(63, 498, 64, 130, False), (40, 700, 20, 65, True)],
    [(195, 73, 33, 386, False), (18, 138, 16, 700, False)],
]


@pytest.mark.parametrize("pair", CASES)
def test_static_entry_point_matches_workspace_entry_point(pair):
    import torch
    from quantum_simulator_amd import engine
    from quantum_simulator_amd._lib import lib
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X; none is visible")
    dev = torch.device("cuda", 0)
    eng = engine.Engine(dev)
    assert lib().qp_static_scratch_bytes() == 4 << 20
    for k, (Q, T, ncores, B, integer) in enumerate(pair):
        dc, dt, A = _setup(torch, engine, dev, Q, T, ncores, B, integer, seed=31 * Q + T + k)
        for table in ("packed", "table"):
            want = eng.cost_v3(dc, dt, A, table=table).cpu().numpy()
            out = torch.full((B,), -1.0, dtype=torch.float64, device=dev)
            got = eng.cost_v3_static(dc, dt, A, out, table=table).cpu().numpy()
            assert got.tobytes() == want.tobytes(), (k, table, got, want)
            assert np.isfinite(got).all() and (got >= 0).all()


def test_static_entry_point_refuses_what_does_not_fit():
    import torch
    from quantum_simulator_amd import engine
    from quantum_simulator_amd._lib import QplaceError, lib
    dev = torch.device("cuda", 0)
    eng = engine.Engine(dev)
    # the kernel for this shape (rows of 48 bytes, a whole partition per item) keeps one partial sum per
    # partition: more partitions than the 4 MiB the library holds has doubles
    dc, dt, _ = _setup(torch, engine, dev, 40, 5, 8, 1, True, seed=5)
    B = (4 << 20) // 8 + 1000
    A = torch.zeros((B, dc.ir.T, dc.ldq), dtype=torch.uint8, device=dev)
    out = torch.empty(B, dtype=torch.float64, device=dev)
    with pytest.raises(QplaceError) as err:
        eng.cost_v3_static(dc, dt, A, out)
    assert err.value.code == -3 and "static" in str(err.value)
    # the same batch in two halves fits
    half = eng.cost_v3_static(dc, dt, A[:B // 2], out[:B // 2]).cpu().numpy()
    assert (half == 0.0).all()
    # and what the workspace entry point computes for a shape does not depend on how much workspace it is given:
    # with too little it refuses, it never takes another kernel (whose rounding could differ)
    dc2, dt2, A2 = _setup(torch, engine, dev, 1024, 20, 256, 4, False, seed=9)
    want = eng.cost_v3(dc2, dt2, A2).cpu().numpy()
    small = torch.empty(1024, dtype=torch.uint8, device=dev)
    rc = lib().qp_slice_cost_v3(C.byref(dc2.c), C.byref(dt2.packed), A2.data_ptr(), 4, A2.shape[1], A2.shape[2],
                                out.data_ptr(), small.data_ptr(), small.numel(), None)
    torch.cuda.synchronize(dev)
    assert rc == -3
    again = torch.empty(4, dtype=torch.float64, device=dev)
    assert eng.cost_v3_static(dc2, dt2, A2, again).cpu().numpy().tobytes() == want.tobytes()


def test_static_entry_point_inside_a_captured_graph():
    """After one ordinary call (which resolves the scratch address and the kernels' attributes), the entry point is
    recorded into a HIP graph and replayed on new inputs: nothing in it allocates or synchronises."""
    import torch
    from quantum_simulator_amd import engine
    dev = torch.device("cuda", 0)
    eng = engine.Engine(dev)
    for Q, T, ncores, B, integer in ((1024, 30, 256, 6, False), (63, 50, 64, 16, False), (1024, 12, 256, 5, True)):
        dc, dt, A = _setup(torch, engine, dev, Q, T, ncores, B, integer, seed=Q + T)
        out = torch.full((B,), -1.0, dtype=torch.float64, device=dev)
        eng.cost_v3_static(dc, dt, A, out)  # warm-up outside the capture
        want0 = eng.cost_v3(dc, dt, A).cpu().numpy()
        assert out.cpu().numpy().tobytes() == want0.tobytes()
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            eng.cost_v3_static(dc, dt, A, out)
        for rep in range(2):
            A.copy_(torch.randint(0, ncores, A.shape, device=dev, dtype=torch.int16).to(torch.uint8))
            out.fill_(-1.0)
            g.replay()
            torch.cuda.synchronize(dev)
            want = eng.cost_v3(dc, dt, A).cpu().numpy()
            assert out.cpu().numpy().tobytes() == want.tobytes(), (Q, rep)
