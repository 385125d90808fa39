"""Times the placement batches (HQA / TRHQA: native lanes, device cost matrices + host assignment solves;
GTRHQA: all on the device) on golden-fixture circuits: time_place_batch.py [case] [B] [threads,threads,...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from quantum_simulator_amd.golden import case_arch, case_circuit, load_case  # noqa: E402
from quantum_simulator_amd.placers import PLACERS  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "qft18_mesh4x4_u"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    sweep = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [None]
    gc = load_case(name)
    circ, arch = case_circuit(gc), case_arch(gc)
    rng = np.random.default_rng(0)
    slots = np.repeat(np.arange(gc.C), gc.cap)
    inits = np.stack([rng.permutation(slots)[:gc.Q] for _ in range(B)]).astype(np.uint8)
    for pname in ("HQA", "TRHQA", "GTRHQA"):
        if not gc.has(pname):
            continue
        placer = PLACERS[pname]()
        placer.place_batch(circ, arch, initial=inits[:2])
        for th in (sweep if pname != "GTRHQA" else [None]):
            kw = {} if th is None else {"threads": th}
            placer.place_batch(circ, arch, initial=inits, **kw)
            t0 = time.perf_counter()
            placer.place_batch(circ, arch, initial=inits, **kw)
            dt = time.perf_counter() - t0
            print(f"{name} {pname}: B={B} T={gc.T} threads={th} {dt:.4f} s  {B * gc.T / dt:.3e} slice-steps/s", flush=True)
        if pname != "GTRHQA":  # identical repeats, as the driver runs them
            t0 = time.perf_counter()
            placer.place_batch(circ, arch, B=B)
            dt = time.perf_counter() - t0
            print(f"{name} {pname}: B={B} identical repeats {dt:.4f} s  {B * gc.T / dt:.3e} slice-steps/s", flush=True)


if __name__ == "__main__":
    main()
