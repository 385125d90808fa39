#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — times the CPU restatement (oracle/placer_oracle.py, precomputed distance
matrix) on golden cases, so that CPU numbers measured on a GPU box can be related to the reference's own
Dijkstra-per-look-up timings, which were taken once in the container that holds the reference.  Prints one line
per (case, placer)."""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_case  # noqa: E402
from oracle import placer_oracle as O  # noqa: E402

CASES = [("gen256_mesh2x1_u", "HQA"), ("gen256_mesh2x1_u", "TRHQA"), ("qft18_mesh4x4_u", "HQA"),
         ("qft18_mesh4x4_u", "TRHQA"), ("qft63_torus8x8_u", "HQA"), ("qft63_torus8x8_u", "GTRHQA")]

for name, pname in CASES:
    gc = load_case(name)
    D = gc.D.tolist()
    t0 = time.perf_counter()
    if pname == "GTRHQA":
        part = O.place_gtrhqa(gc.slices, gc.gate_q, gc.gate_mr, gc.Q, gc.C, gc.cap, D)
    else:
        part = O.place_hqa_family(gc.slices, gc.gate_q, gc.gate_mr, gc.Q, gc.C, gc.cap, D, pname.lower())
    t1 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        cost = O.cost_v3(gc.slices, gc.gate_q, gc.gate_mr, part, D)
    t2 = time.perf_counter()
    T = len(gc.slices)
    print(f"{name:22s} {pname:7s} place {t1 - t0:7.3f} s   cost {(t2 - t1) / reps:8.4f} s   T={T}   "
          f"{T / ((t2 - t1) / reps):9.0f} slice-cost evals/s   cost={cost!r}")
