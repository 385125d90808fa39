"""Times the QUBO energy evaluation (and optionally the fields) on the benchmark model:
time_qubo.py [reps] [fields]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from quantum_simulator_amd import layered_gate_arrays  # noqa: E402
from quantum_simulator_amd.engine import hops  # noqa: E402
from quantum_simulator_amd.multicore import build_topology, graph_tables  # noqa: E402
from quantum_simulator_amd.qubo import DeviceQubo, QuboModel  # noqa: E402
from quantum_simulator_amd.timeslices import CircuitIR  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda", 0)
    Q, T, cap, R = 256, 8, 20, 4096
    q0, q1, _ = layered_gate_arrays(Q, T, 0.5, seed=4242)
    ir = CircuitIR.from_gate_tables(Q, q0, q1)
    nodes, eu, ev, _ = graph_tables(build_topology("mesh", 4, 4))
    qm = QuboModel(ir, len(nodes), cap, hops(len(nodes), eu, ev))
    dq = DeviceQubo(qm, dev)
    X, _ = dq.anneal(R, 20, seed=1)
    dq.energy(X)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        e = dq.energy(X)
    e1.record()
    torch.cuda.synchronize()
    print(f"energy: {e0.elapsed_time(e1) / reps:.4f} ms  best {int(e.min())}")
    if len(sys.argv) > 2:
        dq.fields(X)
        torch.cuda.synchronize()
        e0.record()
        dq.fields(X)
        e1.record()
        torch.cuda.synchronize()
        print(f"fields: {e0.elapsed_time(e1):.3f} ms")


if __name__ == "__main__":
    main()
