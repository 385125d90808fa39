"""Times qp_gtrhqa_place on the benchmark circuit: time_gtrhqa.py [uniform|integer] [B ...]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from quantum_simulator_amd import layered_gate_arrays  # noqa: E402
from quantum_simulator_amd.engine import DeviceCircuit, DistanceTable, Engine, apsp  # noqa: E402
from quantum_simulator_amd.multicore import graph_tables, multicore  # noqa: E402
from quantum_simulator_amd.timeslices import CircuitIR  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    Q, T, C, cap = 1024, 1000, 256, 8
    q0, q1, _ = layered_gate_arrays(Q, T, 0.5, seed=2025)
    ir = CircuitIR.from_gate_tables(Q, q0, q1)
    dc, eng = DeviceCircuit(ir, dev), Engine(dev)
    random.seed(1234)
    args = sys.argv[1:]
    wr = (2.0, 2.0)
    if args and args[0] in ("uniform", "integer"):
        wr = (1.0, 3.0) if args[0] == "uniform" else (2.0, 2.0)
        args = args[1:]
    print("edge weights", wr, flush=True)
    arch = multicore(16, 16, ncm_qubits=4, ncp_qubits=cap, edge_weight_range=wr, network_type="mesh")
    nodes, eu, ev, w = graph_tables(arch.network)
    dt = DistanceTable(apsp(len(nodes), eu, ev, w), dev)
    rng = np.random.default_rng(0)
    for B in [int(x) for x in args] or [64, 256]:
        slots = np.repeat(np.arange(C), cap)
        init = np.stack([rng.permutation(slots)[:Q] for _ in range(B)]).astype(np.uint8)
        init[0] = np.arange(Q) // cap
        d_init = torch.from_numpy(init).to(dev)
        out = torch.empty((B, T + 1, dc.ldq), dtype=torch.uint8, device=dev)
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.gtrhqa_place(dc, dt, d_init, cap, out=out)
            torch.cuda.synchronize()
            dtm = time.perf_counter() - t0
            print(f"B={B} rep={rep}: {dtm * 1e3:.1f} ms, {B * T / dtm:.3e} slice-steps/s", flush=True)
        A = out
        moved = (A[:, 2:, :Q] != A[:, 1:-1, :Q]).float().mean().item()
        cost = eng.cost_v3(dc, dt, A)
        print(f"  moved fraction between consecutive rows {moved:.3f}; mean cost {cost.mean().item():.1f}", flush=True)


if __name__ == "__main__":
    main()
