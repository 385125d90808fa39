import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test that takes more than a few seconds")


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


@pytest.fixture(params=ALL_CASES)
def golden_case(request):
    return load_case(request.param)


@pytest.fixture(params=SMALL_CASES)
def small_case(request):
    return load_case(request.param)
