#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — writes tests/golden/static_place.json.

Runs the *unmodified* reference's small placer helpers under pinned seeds and records what they return:
``random_initial_assignment`` (HQA_placer.py:8-18), ``random_placer.place`` (Random_placer.py:9-23),
``bfs_placer.place`` (BFS_placer.py:20-99), ``random_placer.get_placement_statistics`` (Random_placer.py:214-253)
and ``_get_average_weighted_distance`` (HQA_placer.py:242-257).  Qubits are recorded as positions in
``circuit.qubits``, cores as node labels.  Only runs where /root/reference exists; nothing of it is copied."""
import contextlib
import io
import json
import os
import random
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402

# circuit (relative to REF/circuits), topology, nx, ny, cap, weight range, architecture seed, call seed
RUNS = {
    "shor5_mesh4x4": ("small/shor_n5_transpiled.qasm", "mesh", 4, 4, 2, (1.0, 3.0), 51, 7),
    "qft18_mesh4x4": ("medium/qft_n18_transpiled.qasm", "mesh", 4, 4, 2, (1.0, 3.0), 21, 8),
    "qft63_mesh4x4_cap3": ("large/qft_n63_transpiled.qasm", "mesh", 4, 4, 3, (1.0, 3.0), 52, 9),
    "qft29_torus3x3_cap3": ("large/qft_n29_transpiled.qasm", "torus", 3, 3, 3, (1.0, 3.0), 53, 10),
}


def main():
    R = G._load_reference()
    out = {}
    for key, (qf, topo, nxx, nyy, cap, wr, aseed, seed) in RUNS.items():
        path = os.path.join(G.REF, "circuits", qf)
        if not os.path.exists(path):
            raise SystemExit(f"missing {path}")
        circ = R["qasm"].circuit_from_qasm(path, maxwidth=1024)
        qidx = {q: i for i, q in enumerate(circ.qubits)}
        rec = dict(qasm=qf, topology=topo, nx=nxx, ny=nyy, cap=cap, weight_range=list(wr), arch_seed=aseed, seed=seed)
        with contextlib.redirect_stdout(io.StringIO()):
            arch = G._build_arch(R, topo, nxx, nyy, cap, wr, aseed)
            nodes = list(arch.network.nodes())
            init = R["HQA"].random_initial_assignment(circ.qubits, nodes, cap, seed=seed)
            rec["random_initial_assignment"] = [init.get(q) for q in circ.qubits]
            rec["average_weighted_distance"] = float(R["HQA"].hqa_placer()._get_average_weighted_distance(arch)).hex()
            random.seed(seed)
            m = R["RANDOM"].random_placer().place(circ, arch)
            rec["random_place"] = None if m is None else [m[q] for q in circ.qubits]
            arch = G._build_arch(R, topo, nxx, nyy, cap, wr, aseed)  # fresh free-slot counters
            random.seed(seed)
            m = R["BFS"].bfs_placer().place(circ, arch)
            rec["bfs_place"] = [m.get(q) for q in circ.qubits]
            rec["bfs_place_free_after"] = [arch.network.nodes[n]["qpu"].available_qubits for n in nodes]
            arch = G._build_arch(R, topo, nxx, nyy, cap, wr, aseed)
            rp = R["RANDOM"].random_placer()
            random.seed(seed)
            part = rp.place_per_timeslice(circ, arch)
            try:
                st = rp.get_placement_statistics(circ, part, arch)
            except KeyError:  # a slice with more active qubits than the architecture has slots
                st = None
        rec["statistics"] = None if st is None else {
            k: (v if isinstance(v, int) else [float(x).hex() for x in v] if isinstance(v, list) else float(v).hex())
            for k, v in st.items()}
        out[key] = rec
        print(key, "bfs unplaced:", sum(x is None for x in rec["bfs_place"]), "cross gates:",
              None if st is None else st["cross_qpu_gates"], flush=True)
    with open(os.path.join(G.OUT, "static_place.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))


if __name__ == "__main__":
    main()
