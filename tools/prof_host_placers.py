"""cProfile of the host-side placers (BFS, RANDOM) and of their cost call on a fixture circuit:
prof_host_placers.py [case]"""
import cProfile
import os
import pstats
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quantum_simulator_amd.golden import case_arch, case_circuit, load_case  # noqa: E402
from quantum_simulator_amd.placers import PLACERS  # noqa: E402


def main():
    gc = load_case(sys.argv[1] if len(sys.argv) > 1 else "qft63_torus8x8_u")
    circ, arch = case_circuit(gc), case_arch(gc)
    for name in ("BFS", "RANDOM"):
        pl = PLACERS[name]()
        random.seed(1)
        part = pl.place_per_timeslice(circ, arch)
        pl.per_timeslice_cost(circ, part, arch)
        t0 = time.perf_counter()
        for _ in range(10):
            part = pl.place_per_timeslice(circ, arch)
        t1 = time.perf_counter()
        for _ in range(10):
            pl.per_timeslice_cost(circ, part, arch)
        t2 = time.perf_counter()
        print(f"{name}: place {(t1 - t0) * 100:.2f} ms, cost {(t2 - t1) * 100:.2f} ms per call", flush=True)
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(5):
            part = pl.place_per_timeslice(circ, arch)
            pl.per_timeslice_cost(circ, part, arch)
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(14)


if __name__ == "__main__":
    main()
