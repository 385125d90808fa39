"""Times the gate-ordered (sparse-partition) cost on the benchmark circuit, visit by visit and (16-bit table, 1024
partitions or more) as a row walk: time_v2.py [B [Q T mesh_x mesh_y]]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import make_partitions  # noqa: E402
from quantum_simulator_amd import layered_gate_arrays  # noqa: E402
from quantum_simulator_amd.engine import QP_V2, DeviceCircuit, DistanceTable, Engine, apsp  # noqa: E402
from quantum_simulator_amd.multicore import graph_tables, multicore  # noqa: E402
from quantum_simulator_amd.timeslices import CircuitIR  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    Q, T, nx, ny = (int(x) for x in sys.argv[2:6]) if len(sys.argv) > 5 else (1024, 1000, 15, 17)
    q0, q1, _ = layered_gate_arrays(Q, T, 0.5, seed=2025)
    ir = CircuitIR.from_gate_tables(Q, q0, q1)
    eng = Engine(dev)
    forms = (("row walk where it applies", DeviceCircuit(ir, dev)), ("visit by visit", DeviceCircuit(ir, dev, visit_mask=False)))
    A = make_partitions(torch, dev, B, ir.T, Q, forms[0][1].ldq, nx * ny, 8, 0.95, seed=777)
    random.seed(1234)
    for wr in ((2.0, 2.0), (1.0, 3.0)):
        arch = multicore(nx, ny, ncm_qubits=4, ncp_qubits=8, edge_weight_range=wr, network_type="mesh")
        nodes, eu, ev, w = graph_tables(arch.network)
        dt = DistanceTable(apsp(len(nodes), eu, ev, w), dev)
        outs = []
        for name, dc in forms:
            out = torch.empty(B, dtype=torch.float64, device=dev)
            for _ in range(3):
                eng.cost_v2(dc, dt, A, QP_V2, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                eng.cost_v2(dc, dt, A, QP_V2, out=out)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            outs.append(out.cpu())
            print(f"weights {wr}, {name}: B={B} {ms:.3f} ms  {B * T / ms / 1e6:.3f} G evals/s  "
                  f"{ir.nV * B / ms / 1e6:.2f} G visits/s  {A.numel() / ms / 1e9:.3f} TB/s of rows", flush=True)
        assert torch.equal(outs[0], outs[1]) or wr != (2.0, 2.0), "forms disagree"

if __name__ == "__main__":
    main()
