import cProfile, pstats, sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from conftest import case_arch, case_circuit, load_case
from quantum_simulator_amd.placers import PLACERS
name, B = "qft18_mesh4x4_u", 256
gc = load_case(name); circ, arch = case_circuit(gc), case_arch(gc)
rng = np.random.default_rng(0)
slots = np.repeat(np.arange(gc.C), gc.cap)
inits = np.stack([rng.permutation(slots)[:gc.Q] for _ in range(B)]).astype(np.uint8)
pl = PLACERS["HQA"]()
pl.place_batch(circ, arch, initial=inits[:2])
pr = cProfile.Profile(); pr.enable()
pl.place_batch(circ, arch, initial=inits)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
