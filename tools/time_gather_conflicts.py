"""What bank conflicts of the gate-endpoint gathers cost the 1 KB-row cost kernels: the same launch (1024 qubits x T slices,
256 gates per slice, random-walk partitions) on the benchmark's layered circuit and on a circuit whose gates pair qubit
4 k with qubit 4 k + 2 (every run of 32 gates then has its first endpoints in 32 different LDS banks and its second
endpoints in the same dwords: no gather ever waits for another).  time_gather_conflicts.py [B=2048] [T=500]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.argv_saved = sys.argv
import bench  # noqa: E402
from quantum_simulator_amd import layered_gate_arrays  # noqa: E402
from quantum_simulator_amd.engine import DeviceCircuit, DistanceTable, Engine, apsp  # noqa: E402
from quantum_simulator_amd.multicore import graph_tables, multicore  # noqa: E402
from quantum_simulator_amd.timeslices import CircuitIR  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    dev = torch.device("cuda", 0)
    Q, C, cap = 1024, 256, 8
    eng = Engine(dev)
    circuits = {}
    q0, q1, _ = layered_gate_arrays(Q, T, 0.5, seed=2025)
    circuits["layered (the benchmark's)"] = CircuitIR.from_gate_tables(Q, q0, q1)
    # per slice: 256 two-qubit gates (4k, 4k+2), then one-qubit gates on the other 512 qubits, so that every slice is one layer
    a0, a1 = [], []
    for _ in range(T):
        for k in range(256):
            a0.append(4 * k)
            a1.append(4 * k + 2)
        for k in range(256):
            a0.extend([4 * k + 1, 4 * k + 3])
            a1.extend([-1, -1])
    circuits["gates (4k, 4k+2): conflict-free gathers"] = CircuitIR.from_gate_tables(Q, np.array(a0, np.int32), np.array(a1, np.int32))
    tables = {}
    for name, wr in (("uniform: fp64 pair counts", (1.0, 3.0)), ("integer: closed form", (2.0, 2.0))):
        random.seed(1234)
        arch = multicore(16, 16, ncm_qubits=4, ncp_qubits=cap, edge_weight_range=wr, network_type="mesh")
        nodes, eu, ev, w = graph_tables(arch.network)
        tables[name] = DistanceTable(apsp(len(nodes), eu, ev, w), dev)
    for cname, ir in circuits.items():
        assert ir.T == T, (cname, ir.T)
        dc = DeviceCircuit(ir, dev)
        A = bench.make_partitions(torch, dev, B, ir.T, Q, dc.ldq, C, cap, 0.95, seed=7)
        out = torch.empty(B, dtype=torch.float64, device=dev)
        for tname, dt in tables.items():
            for table in ("packed", "table"):
                if table == "table" and "uniform" in tname:
                    continue
                for _ in range(30):
                    eng.cost_v3(dc, dt, A, out=out, table=table)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    eng.cost_v3(dc, dt, A, out=out, table=table)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 30
                print(f"{cname:45s} {tname:28s} {table:7s} {ms:7.3f} ms  {B * T * 1024 / ms / 1e9:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
