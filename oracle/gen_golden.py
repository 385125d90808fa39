#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — writes the golden fixtures under tests/golden/.

Runs the *unmodified* reference placers (imported from /root/reference through the ``pytket`` name
adapter in oracle/refshim) on pinned inputs and records inputs and outputs as data:

* the circuit as integer gate tables (qubit indices into ``circuit.qubits`` order, file order);
* the reference's own timeslices (``BasePlacer.break_into_timeslices``, Placer.py:14-61) as gate
  index lists;
* core graph edges + weights in ``G.edges()`` order, the distance matrix the placer cached
  (``_dist_matrix``, HQA_placer.py:47-54) and networkx hop counts;
* per placer: the partition as a uint8 ``[Tp, Q]`` array of core indices (0xFF = qubit not in the
  slice's dict) and the float cost of its ``per_timeslice_cost``; for BFS/RANDOM partitions also
  the ``BasePlacer.per_timeslice_cost`` value (Placer.py:78-120);
* a sample of the cost matrices handed to ``scipy.optimize.linear_sum_assignment`` together with
  the row/column indices it returned (HQA_placer.py:126, TRHQA_placer.py:171).

Only runs where /root/reference exists; nothing from the reference is copied.  Usage:
    python oracle/gen_golden.py [case ...]
"""
import contextlib
import hashlib
import io
import json
import os
import random
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

import numpy as np  # noqa: E402
import networkx as nx  # noqa: E402


def _load_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present; goldens can only be regenerated next to it")
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(HERE, "refshim"))
    import multicore as M
    import Placer
    import HQA_placer
    import TRHQA_placer
    import GTRHQA_placer
    import BFS_placer
    import Random_placer
    from pytket import qasm, OpType
    return dict(M=M, Placer=Placer, HQA=HQA_placer, TRHQA=TRHQA_placer, GTRHQA=GTRHQA_placer,
                BFS=BFS_placer, RANDOM=Random_placer, qasm=qasm, OpType=OpType)


# name, qasm (relative to REF/circuits), topology, nx, ny, cap, weight_range, seed
CASES = [
    ("shor5_mesh2x2_u", "small/shor_n5_transpiled.qasm", "mesh", 2, 2, 2, (1.0, 3.0), 11),
    ("shor5_mesh2x2_i", "small/shor_n5_transpiled.qasm", "mesh", 2, 2, 2, (2.0, 2.0), 11),
    ("shor5_line3_u", "small/shor_n5_transpiled.qasm", "line", 3, 1, 2, (1.0, 3.0), 12),
    ("shor5_star4_u", "small/shor_n5_transpiled.qasm", "star", 4, 1, 3, (1.0, 3.0), 13),
    ("shor5_tree2_i", "small/shor_n5_transpiled.qasm", "tree", 2, 1, 2, (3.0, 3.0), 14),
    ("shor5_all4_u", "small/shor_n5_transpiled.qasm", "all_to_all", 2, 2, 2, (1.0, 10.0), 15),
    ("qft18_mesh4x4_u", "medium/qft_n18_transpiled.qasm", "mesh", 4, 4, 2, (1.0, 3.0), 21),
    ("qft18_mesh4x4_i", "medium/qft_n18_transpiled.qasm", "mesh", 4, 4, 2, (2.0, 2.0), 21),
    ("qft18_mesh4x4_cap4_u", "medium/qft_n18_transpiled.qasm", "mesh", 4, 4, 4, (1.0, 3.0), 23),
    ("bigadder_mesh4x4_u", "medium/bigadder_n18_transpiled.qasm", "mesh", 4, 4, 2, (1.0, 3.0), 22),
    ("bigadder_mesh4x4_i", "medium/bigadder_n18_transpiled.qasm", "mesh", 4, 4, 2, (1.0, 1.0), 22),
    ("bigadder_torus3x3_u", "medium/bigadder_n18_transpiled.qasm", "torus", 3, 3, 2, (1.0, 3.0), 24),
    ("qft29_torus8x8_u", "large/qft_n29_transpiled.qasm", "torus", 8, 8, 2, (1.0, 3.0), 31),
    ("qft29_torus8x8_i", "large/qft_n29_transpiled.qasm", "torus", 8, 8, 2, (2.0, 2.0), 31),
    ("qft63_torus8x8_u", "large/qft_n63_transpiled.qasm", "torus", 8, 8, 2, (1.0, 3.0), 32),
    ("qft63_torus8x8_i", "large/qft_n63_transpiled.qasm", "torus", 8, 8, 2, (2.0, 2.0), 32),
    ("qft63_mesh4x4_cap4_u", "large/qft_n63_transpiled.qasm", "mesh", 4, 4, 4, (1.0, 3.0), 33),
    ("test500_mesh8x8_u", "test_circuit.qasm", "mesh", 8, 8, 8, (1.0, 3.0), 41),
    ("gen256_mesh2x1_u", "gen/256.qasm", "mesh", 2, 1, 130, (1.0, 2.0), 51),
    ("gen256_mesh2x1_i", "gen/256.qasm", "mesh", 2, 1, 130, (1.0, 1.0), 51),
]
PLACERS = ["HQA", "TRHQA", "GTRHQA", "BFS", "RANDOM"]
MAX_LSAP_SAMPLES = 48


def _build_arch(R, topo, nxx, nyy, cap, wr, seed):
    random.seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        arch = R["M"].multicore(nxx, nyy, ncm_qubits=4, ncp_qubits=cap, edge_weight_range=wr,
                                network_type=topo)
    return arch


def _partition_array(partition, qidx, node_index):
    arr = np.full((len(partition), len(qidx)), 0xFF, np.uint8)
    for t, d in enumerate(partition):
        for q, c in d.items():
            arr[t, qidx[q]] = node_index[c]
    return arr


def run_case(R, case):
    name, qf, topo, nxx, nyy, cap, wr, seed = case
    t_start = time.time()
    circ = R["qasm"].circuit_from_qasm(os.path.join(REF, "circuits", qf), maxwidth=1024)
    OpType = R["OpType"]
    allq = list(circ.qubits)
    qidx = {q: i for i, q in enumerate(allq)}
    gates = list(circ)
    gid = {id(g): i for i, g in enumerate(gates)}
    G = len(gates)
    maxar = max(len(g.qubits) for g in gates)
    gate_q = np.full((G, maxar), -1, np.int32)
    for i, g in enumerate(gates):
        for k, q in enumerate(g.qubits):
            gate_q[i, k] = qidx[q]
    gate_mr = np.array([g.op.type in (OpType.Measure, OpType.Reset) for g in gates], np.uint8)

    base = R["Placer"].BasePlacer()
    ts = base.break_into_timeslices(circ)
    slice_ptr = np.zeros(len(ts) + 1, np.int32)
    slice_gate = []
    for t, sl in enumerate(ts):
        slice_gate.extend(gid[id(g)] for g in sl)
        slice_ptr[t + 1] = len(slice_gate)
    slice_gate = np.array(slice_gate, np.int32)

    arch = _build_arch(R, topo, nxx, nyy, cap, wr, seed)
    nodes = list(arch.network.nodes)
    node_index = {n: i for i, n in enumerate(nodes)}
    edges = list(arch.network.edges(data="weight"))
    edge_uv = np.array([[node_index[u], node_index[v]] for u, v, _ in edges], np.int32).reshape(-1, 2)
    edge_w = np.array([w for _, _, w in edges], np.float64)
    C = len(nodes)
    hops = np.zeros((C, C), np.int32)
    for i, a in enumerate(nodes):
        for j, b in enumerate(nodes):
            hops[i, j] = nx.shortest_path_length(arch.network, a, b)

    data = dict(gate_q=gate_q, gate_mr=gate_mr, slice_ptr=slice_ptr, slice_gate=slice_gate,
                edge_uv=edge_uv, edge_w=edge_w, hops=hops)
    meta = dict(name=name, qasm=qf, topology=topo, nx=nxx, ny=nyy, cap=cap, weight_range=list(wr),
                arch_seed=seed, Q=len(allq), C=C, T=len(ts), G=G, placers={},
                nodes_are_range=bool(nodes == list(range(C))), qubits=[str(q) for q in allq],
                edge_w_hex=[float(w).hex() for w in edge_w])

    lsap_rec = []

    def make_hook(tag, real):
        def hook(cm):
            r, c = real(cm)
            if len(lsap_rec) < MAX_LSAP_SAMPLES or (cm.shape[0] > 1 and len(lsap_rec) < 2 * MAX_LSAP_SAMPLES
                                                      and len(np.unique(cm)) < cm.size // 2):
                lsap_rec.append((tag, np.array(cm, np.float64), np.array(r, np.int32), np.array(c, np.int32)))
            return r, c
        return hook

    for pname in PLACERS:
        if sum(1 for _ in allq) > C * cap and pname in ("HQA", "TRHQA", "GTRHQA"):
            continue
        mod = R[pname]
        placer = getattr(mod, {"HQA": "hqa_placer", "TRHQA": "trhqa_placer", "GTRHQA": "gtrhqa_placer",
                               "BFS": "bfs_placer", "RANDOM": "random_placer"}[pname])()
        pseed = seed * 1000 + PLACERS.index(pname)
        random.seed(pseed)
        real = getattr(mod, "_hungarian", None)
        n_before = len(lsap_rec)
        if real is not None:
            setattr(mod, "_hungarian", make_hook(pname, real))
        try:
            t0 = time.time()
            part = placer.place_per_timeslice(circ, arch)
            t1 = time.time()
            err = None
            try:
                cost = placer.per_timeslice_cost(circ, part, arch)
            except Exception as ex:  # main.py:99-102 turns this into inf for the repeat
                cost, err = float("inf"), type(ex).__name__
            t2 = time.time()
        finally:
            if real is not None:
                setattr(mod, "_hungarian", real)
        parr = _partition_array(part, qidx, node_index)
        data[f"part_{pname}"] = parr
        data[f"cost_{pname}"] = np.array([cost], np.float64)
        pm = dict(cost_repr=repr(cost), cost_hex=float(cost).hex(), Tp=len(part), place_seed=pseed,
                  place_s=round(t1 - t0, 4), cost_s=round(t2 - t1, 4), cost_error=err,
                  part_sha1=hashlib.sha1(parr.tobytes()).hexdigest(), n_lsap=len(lsap_rec) - n_before)
        if pname in ("HQA", "TRHQA", "GTRHQA"):
            data["D"] = np.array(placer._dist_matrix, np.float64)
        # BasePlacer cost (V1) on the same partition
        try:
            v1 = base.per_timeslice_cost(circ, part, arch)
            pm["v1_error"] = None
        except Exception as ex:
            v1, pm["v1_error"] = float("inf"), type(ex).__name__
        data[f"costv1_{pname}"] = np.array([v1], np.float64)
        pm["costv1_repr"] = repr(v1)
        meta["placers"][pname] = pm

    if "D" not in data:  # capacity too small for the HQA family: fill D the way they would
        D = np.zeros((C, C))
        for i, a in enumerate(nodes):
            for j, b in enumerate(nodes):
                if i != j:
                    D[i, j] = base.get_weighted_distance(arch, a, b)
        data["D"] = D

    for k, (tag, cm, r, c) in enumerate(lsap_rec):
        data[f"lsap{k}_cost"] = cm
        data[f"lsap{k}_row"] = r
        data[f"lsap{k}_col"] = c
    meta["lsap_tags"] = [t for t, _, _, _ in lsap_rec]
    meta["gen_s"] = round(time.time() - t_start, 2)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)
    return meta


def run_example_config(R):
    """The shipped example_config.txt flow (main.py:16-130) with one seed for the whole run:
    arch built once, placers in config order, 3 repeats each, mean over finite costs."""
    cfg = dict(topology="mesh", nx=2, ny=1, cap=130, wr=(1.0, 2.0), placers=["RANDOM", "BFS", "TRHQA", "HQA", "GTRHQA"],
               repeats=3, seed=2024, qasm="gen/256.qasm")
    random.seed(cfg["seed"])
    with contextlib.redirect_stdout(io.StringIO()):
        arch = R["M"].multicore(cfg["nx"], cfg["ny"], ncm_qubits=4, ncp_qubits=cfg["cap"],
                                edge_weight_range=cfg["wr"], network_type=cfg["topology"])
    circ = R["qasm"].circuit_from_qasm(os.path.join(REF, "circuits", cfg["qasm"]), maxwidth=1024)
    cls = {"HQA": "hqa_placer", "TRHQA": "trhqa_placer", "GTRHQA": "gtrhqa_placer", "BFS": "bfs_placer",
           "RANDOM": "random_placer"}
    out = {}
    for p in cfg["placers"]:
        placer = getattr(R[p], cls[p])()
        costs = []
        for _ in range(cfg["repeats"]):
            try:
                part = placer.place_per_timeslice(circ, arch)
                costs.append(placer.per_timeslice_cost(circ, part, arch))
            except Exception:
                costs.append(float("inf"))
        fin = [c for c in costs if np.isfinite(c)]
        out[p] = dict(costs_hex=[float(c).hex() for c in costs],
                      mean_hex=float(np.mean(fin)).hex() if fin else float("inf").hex(),
                      mean_repr=repr(float(np.mean(fin))) if fin else "inf")
    w = list(arch.network.edges(data="weight"))[0][2]
    cfg["edge_weight_hex"] = float(w).hex()
    cfg["results"] = out
    cfg["wr"] = list(cfg["wr"])
    return cfg


def main():
    R = _load_reference()
    os.makedirs(OUT, exist_ok=True)
    want = set(sys.argv[1:])
    idx_path = os.path.join(OUT, "index.json")
    index = json.load(open(idx_path)) if os.path.exists(idx_path) else {"cases": {}, "versions": {}}
    import scipy
    index["versions"] = dict(python=sys.version.split()[0], numpy=np.__version__, scipy=scipy.__version__,
                             networkx=nx.__version__)
    for case in CASES:
        if want and case[0] not in want:
            continue
        meta = run_case(R, case)
        index["cases"][case[0]] = meta
        print(case[0], {p: m["cost_repr"] for p, m in meta["placers"].items()}, f'{meta["gen_s"]}s', flush=True)
    if not want or "example_config" in want:
        index["example_config"] = run_example_config(R)
        print("example_config", {p: r["mean_repr"] for p, r in index["example_config"]["results"].items()})
    with open(idx_path, "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
