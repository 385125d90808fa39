"""The large-circuit sweep (qft_n29 and qft_n63 gate tables from the fixtures, 8x8 torus, every non-QUBO placer,
64 repeats) through the config-file driver, one placer at a time, wall time each: time_cfg3.py [repeats=64] [dir]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quantum_simulator_amd.golden import load_case  # noqa: E402
from quantum_simulator_amd import driver  # noqa: E402


def qasm_of(gc):
    """The fixture's gate tables as OpenQASM 2.0 (two-qubit gates as cx, one-qubit gates as x, measures kept)."""
    names = gc.meta["qubits"]
    regs = {}
    for s in names:
        reg, idx = s[:-1].split("[")
        regs[reg] = max(regs.get(reg, 0), int(idx) + 1)
    lines = ["OPENQASM 2.0;", 'include "qelib1.inc";'] + [f"qreg {r}[{n}];" for r, n in regs.items()] + [f"creg c[{len(names)}];"]
    for g, mr in zip(gc.gate_q, gc.gate_mr):
        if mr:
            lines.append(f"measure {names[g[0]]} -> c[{g[0]}];")
        else:
            lines.append(("cx " if len(g) == 2 else "x ") + ",".join(names[i] for i in g) + ";")
    return "\n".join(lines) + "\n"


def main():
    repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "cfg3")
    folder = os.path.join(out, "circuits")
    os.makedirs(folder, exist_ok=True)
    for case in ("qft29_torus8x8_u", "qft63_torus8x8_u"):
        with open(os.path.join(folder, case.split("_")[0] + ".qasm"), "w") as f:
            f.write(qasm_of(load_case(case)))
    for placer in ("GTRHQA", "GTRHQA", "HQA", "TRHQA", "BFS", "RANDOM"):  # the first pass also loads the library and starts HIP
        cfg_path = os.path.join(out, f"config_{placer}.txt")
        with open(cfg_path, "w") as f:
            f.write(f"mode = simulate\ntopology = torus\nnum_cores_x = 8\nnum_cores_y = 8\nncm_qubits = 4\nncp_qubits = 2\n"
                    f"edge_weight_range = 1.0,3.0\ncircuit_folders = {folder}\nplacers = {placer}\nrepeats = {repeats}\n"
                    f"out_csv = {os.path.join(out, placer + '.csv')}\nplot_file = {os.path.join(out, placer + '.png')}\nseed = 7\n")
        cfg = driver.parse_config_file(cfg_path)
        t0 = time.perf_counter()
        res = driver.run_simulation(cfg, verbose=False)
        dt = time.perf_counter() - t0
        print(f"{placer:7s} {repeats} repeats x 2 circuits: {dt:7.3f} s   " +
              "  ".join(f"{os.path.basename(r['file'])}: mean cost {r[placer + '_cost_mean']:.1f}" for r in res), flush=True)


if __name__ == "__main__":
    main()
