"""Writes two synthetic layered circuits as QASM and a config next to them, then runs the config-file driver
on it with every placer (the flow of `python main.py --config ...` in the reference): demo_driver.py [dir]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quantum_simulator_amd import circuit_to_qasm_str, layered_circuit  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "demo")
    folder = os.path.join(out, "circuits")
    os.makedirs(folder, exist_ok=True)
    for name, (q, t) in {"layered_18x30": (18, 30), "layered_40x20": (40, 20)}.items():
        with open(os.path.join(folder, name + ".qasm"), "w") as f:
            f.write(circuit_to_qasm_str(layered_circuit(q, t, 0.5, seed=q)))
    cfg = os.path.join(out, "config.txt")
    with open(cfg, "w") as f:
        f.write(f"mode = simulate\ntopology = mesh\nnum_cores_x = 4\nnum_cores_y = 4\nncm_qubits = 4\nncp_qubits = 4\n"
                f"edge_weight_range = 1,3\ncircuit_folders = {folder}\n"
                f"placers = RANDOM, BFS, TRHQA, HQA, GTRHQA, QUBO\nrepeats = 3\n"
                f"out_csv = {os.path.join(out, 'results.csv')}\nplot_file = {os.path.join(out, 'results.png')}\nseed = 7\n")
    rc = subprocess.call([sys.executable, "-m", "quantum_simulator_amd.driver", "--config", cfg], cwd=ROOT)
    print(open(os.path.join(out, "results.csv")).read())
    sys.exit(rc)


if __name__ == "__main__":
    main()
