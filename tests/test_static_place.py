"""The reference's small placer helpers (whole-circuit ``place`` of the RANDOM and BFS placers, the shuffled
starting assignment, placement statistics, mean core distance) against what the unmodified reference returned
under the same seeds (tests/golden/static_place.json, written by oracle/gen_golden_static.py).  Host code only:
no GPU is needed."""
import json
import os
import random

import pytest

from conftest import ROOT, case_circuit, load_case

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "static_place.json")))
CIRCUIT_OF = {"shor5_mesh4x4": "shor5_mesh2x2_u", "qft18_mesh4x4": "qft18_mesh4x4_u",
              "qft63_mesh4x4_cap3": "qft63_torus8x8_u", "qft29_torus3x3_cap3": "qft29_torus8x8_u"}


def _arch(g):
    from quantum_simulator_amd.multicore import multicore
    random.seed(g["arch_seed"])
    return multicore(g["nx"], g["ny"], ncm_qubits=4, ncp_qubits=g["cap"], edge_weight_range=tuple(g["weight_range"]),
                     network_type=g["topology"])


@pytest.mark.parametrize("key", sorted(GOLD))
def test_static_helpers_match_reference(key, capsys):
    from quantum_simulator_amd import placers as P
    g = GOLD[key]
    circ = case_circuit(load_case(CIRCUIT_OF[key]))
    qs = circ.qubits
    arch = _arch(g)
    nodes = list(arch.network.nodes())

    init = P.random_initial_assignment(qs, nodes, g["cap"], seed=g["seed"])
    assert [init.get(q) for q in qs] == g["random_initial_assignment"]
    assert float(P.hqa_placer()._get_average_weighted_distance(arch)).hex() == g["average_weighted_distance"]

    random.seed(g["seed"])
    m = P.random_placer().place(circ, arch)
    assert (None if m is None else [m[q] for q in qs]) == g["random_place"]
    if m is None:
        assert "Not enough unique QPU nodes" in capsys.readouterr().out

    arch = _arch(g)
    random.seed(g["seed"])
    m = P.bfs_placer().place(circ, arch)
    assert [m.get(q) for q in qs] == g["bfs_place"]
    assert [arch.network.nodes[n]["qpu"].available_qubits for n in nodes] == g["bfs_place_free_after"]


@pytest.mark.parametrize("key", sorted(k for k in GOLD if GOLD[k]["statistics"] is not None))
def test_placement_statistics_match_reference(key):
    """get_placement_statistics of a RANDOM per-slice placement made under the fixture's seed."""
    from quantum_simulator_amd import placers as P
    g = GOLD[key]
    circ = case_circuit(load_case(CIRCUIT_OF[key]))
    arch = _arch(g)
    rp = P.random_placer()
    random.seed(g["seed"])
    part = rp.place_per_timeslice(circ, arch)
    st = rp.get_placement_statistics(circ, part, arch)
    want = g["statistics"]
    assert sorted(st) == sorted(want)
    for k, w in want.items():
        if isinstance(w, int):
            assert st[k] == w, k
        elif isinstance(w, list):
            assert [float(x).hex() for x in st[k]] == w, k
        else:
            assert float(st[k]).hex() == w, k
