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
