"""cProfile of one HQA-family lock-step batch: prof_place_batch.py [case] [B] [placer]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from quantum_simulator_amd.golden import case_arch, case_circuit, load_case
from quantum_simulator_amd.placers import PLACERS
name = sys.argv[1] if len(sys.argv) > 1 else "qft18_mesh4x4_u"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
gc = load_case(name); circ, arch = case_circuit(gc), case_arch(gc)
rng = np.random.default_rng(0)
slots = np.repeat(np.arange(gc.C), gc.cap)
inits = np.stack([rng.permutation(slots)[:gc.Q] for _ in range(B)]).astype(np.uint8)
pl = PLACERS[sys.argv[3] if len(sys.argv) > 3 else "HQA"]()
pl.place_batch(circ, arch, initial=inits[:2])
pr = cProfile.Profile(); pr.enable()
pl.place_batch(circ, arch, initial=inits)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
