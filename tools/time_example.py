"""Runs the shipped example configuration's flow (256-qubit circuit from the fixture, mesh 2x1, five placers,
3 repeats) through the config-file driver and prints the per-placer mean times and costs beside the
reference's published row (results.csv:2, unknown hardware): time_example.py"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quantum_simulator_amd.golden import GOLDEN_INDEX, case_circuit, load_case  # noqa: E402
from quantum_simulator_amd import circuit_to_qasm_str, driver  # noqa: E402

PUBLISHED_S = {"RANDOM": 5.0229, "BFS": 5.5221, "TRHQA": 4.8258, "HQA": 5.3829, "GTRHQA": 4.7455}


def main():
    gold = GOLDEN_INDEX["example_config"]
    gc = load_case("gen256_mesh2x1_u")
    with tempfile.TemporaryDirectory() as tmp:
        folder = os.path.join(tmp, "gen")
        os.makedirs(folder)
        with open(os.path.join(folder, "256.qasm"), "w") as f:
            f.write(circuit_to_qasm_str(case_circuit(gc)))
        cfg_path = os.path.join(tmp, "cfg.txt")
        with open(cfg_path, "w") as f:
            f.write(f"mode = simulate\ntopology = {gold['topology']}\nnum_cores_x = {gold['nx']}\nnum_cores_y = {gold['ny']}\n"
                    f"ncm_qubits = 4\nncp_qubits = {gold['cap']}\nedge_weight_range = {gold['wr'][0]},{gold['wr'][1]}\n"
                    f"circuit_folders = {folder}\nplacers = {', '.join(gold['placers'])}\nrepeats = {gold['repeats']}\n"
                    f"out_csv = {os.path.join(tmp, 'out.csv')}\nplot_file = \nseed = {gold['seed']}\n")
        cfg = driver.parse_config_file(cfg_path)
        for run in range(2):  # the second run has the library loaded and the GPU warm
            t0 = time.perf_counter()
            res = driver.run_simulation(cfg, verbose=False)[0]
            wall = time.perf_counter() - t0
        print(f"whole run (parse, 5 placers x {gold['repeats']} repeats, costs, CSV): {wall:.2f} s")
        for p in gold["placers"]:
            print(f"{p:7s} time_mean {res[p + '_time_mean']:.4f} s (reference's published row: {PUBLISHED_S[p]:.2f} s)  "
                  f"cost_mean {res[p + '_cost_mean']!r}")


if __name__ == "__main__":
    main()
