"""Loader of the golden fixtures (tests/golden/*.npz + index.json): circuits as integer gate tables, the
core graph with its edge weights, and what the unmodified reference placers returned for them.  Data
only; used by the tests and by the shipped-circuit leg of bench.py."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

with open(os.path.join(GOLDEN, "index.json")) as _f:
    GOLDEN_INDEX = json.load(_f)
ALL_CASES = sorted(GOLDEN_INDEX["cases"])
SMALL_CASES = [c for c in ALL_CASES if GOLDEN_INDEX["cases"][c]["G"] < 3000]


class GoldenCase:
    """One fixture file: integer circuit tables, core graph, reference outputs."""

    def __init__(self, name):
        self.name = name
        self.meta = GOLDEN_INDEX["cases"][name]
        z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.z = z
        gq = z["gate_q"]
        self.gate_q = [tuple(int(x) for x in row if x >= 0) for row in gq]
        self.gate_mr = [bool(x) for x in z["gate_mr"]]
        sp, sg = z["slice_ptr"], z["slice_gate"]
        self.slices = [[int(g) for g in sg[sp[t]:sp[t + 1]]] for t in range(len(sp) - 1)]
        self.Q, self.C, self.T, self.cap = (self.meta[k] for k in ("Q", "C", "T", "cap"))
        self.edges = [(int(u), int(v), float(w)) for (u, v), w in zip(z["edge_uv"], z["edge_w"])]
        self.D = z["D"]
        self.hops = z["hops"]

    def part(self, placer):
        return self.z["part_" + placer]

    def part_rows(self, placer):
        return [[None if c == 0xFF else int(c) for c in row] for row in self.part(placer)]

    def cost(self, placer):
        return float(self.z["cost_" + placer][0])

    def cost_v1(self, placer):
        return float(self.z["costv1_" + placer][0])

    def has(self, placer):
        return placer in self.meta["placers"]

    def lsap_samples(self):
        for k, tag in enumerate(self.meta["lsap_tags"]):
            yield tag, self.z[f"lsap{k}_cost"], self.z[f"lsap{k}_row"], self.z[f"lsap{k}_col"]

    def qubit_objects(self):
        from quantum_simulator_amd.circuit import Qubit
        out = []
        for s in self.meta["qubits"]:
            reg, idx = s[:-1].split("[")
            out.append(Qubit(reg, int(idx)))
        return out


_cache = {}


def load_case(name):
    if name not in _cache:
        _cache[name] = GoldenCase(name)
    return _cache[name]


def case_circuit(gc):
    """A Circuit object equivalent to the fixture's gate tables (names are irrelevant to placers)."""
    from quantum_simulator_amd.circuit import Circuit, OpType
    circ = Circuit()
    regs = {}
    for s in gc.meta["qubits"]:
        reg, idx = s[:-1].split("[")
        regs[reg] = max(regs.get(reg, 0), int(idx) + 1)
    for reg, n in regs.items():
        circ.add_q_register(reg, n)
    qs = circ.qubits
    assert [str(q) for q in qs] == gc.meta["qubits"]
    for g, mr in zip(gc.gate_q, gc.gate_mr):
        if mr:
            circ.add_command("measure", [qs[i] for i in g], optype=OpType.Measure)
        else:
            circ.add_command("cx" if len(g) == 2 else "x", [qs[i] for i in g])
    return circ


def case_arch(gc):
    """The fixture's architecture rebuilt the way the reference's driver builds it: seed the global
    RNG, construct; the edge weights must come out as recorded."""
    import random
    from quantum_simulator_amd.multicore import multicore, graph_tables
    m = gc.meta
    random.seed(m["arch_seed"])
    arch = multicore(m["nx"], m["ny"], ncm_qubits=4, ncp_qubits=m["cap"],
                     edge_weight_range=tuple(m["weight_range"]), network_type=m["topology"])
    nodes, eu, ev, w = graph_tables(arch.network)
    assert [float(x).hex() for x in w] == m["edge_w_hex"]
    assert [(int(a), int(b)) for a, b in zip(eu, ev)] == [(e[0], e[1]) for e in gc.edges]
    return arch
