"""Where the host time of a driver run goes: the cfg3 sweep of tools/time_cfg3.py for one placer under cProfile
(after an untimed first run that loads the library and starts HIP): prof_driver.py [placer=GTRHQA] [repeats=64]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from quantum_simulator_amd import driver  # noqa: E402
from quantum_simulator_amd.golden import load_case  # noqa: E402
from time_cfg3 import qasm_of  # noqa: E402


def main():
    placer = sys.argv[1] if len(sys.argv) > 1 else "GTRHQA"
    repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    out = os.path.join(ROOT, "gpurun_out", "cfg3")
    folder = os.path.join(out, "circuits")
    os.makedirs(folder, exist_ok=True)
    for case in ("qft29_torus8x8_u", "qft63_torus8x8_u"):
        with open(os.path.join(folder, case.split("_")[0] + ".qasm"), "w") as f:
            f.write(qasm_of(load_case(case)))
    cfg_path = os.path.join(out, f"config_{placer}.txt")
    with open(cfg_path, "w") as f:
        f.write(f"mode = simulate\ntopology = torus\nnum_cores_x = 8\nnum_cores_y = 8\nncm_qubits = 4\nncp_qubits = 2\n"
                f"edge_weight_range = 1.0,3.0\ncircuit_folders = {folder}\nplacers = {placer}\nrepeats = {repeats}\n"
                f"out_csv = {os.path.join(out, placer + '.csv')}\nplot_file = {os.path.join(out, placer + '.png')}\nseed = 7\n")
    cfg = driver.parse_config_file(cfg_path)
    driver.run_simulation(cfg, verbose=False)
    t0 = time.perf_counter()
    pr = cProfile.Profile()
    pr.enable()
    driver.run_simulation(cfg, verbose=False)
    pr.disable()
    print(f"{placer}: {time.perf_counter() - t0:.3f} s", flush=True)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)


if __name__ == "__main__":
    main()
