#!/usr/bin/env python3
"""Headline benchmark: timeslice cost evaluations per second on MI355X.

Workload (``config.workload``): the synthetic layered circuit with 1024 qubits x 1000 timeslices
(256 two-qubit gates per slice) on a 16x16 mesh of cores, 4096 partitions per GPU, costed with the
HQA-family definition (the cost GTRHQA reports).  By default the partitions are GTRHQA's own placements of
the circuit, computed on the device from seeded shuffles of the block assignment (``--source walk`` uses a
seeded random walk instead).  One step = one pass of the cost path over all partitions of the rank:
qp_slice_cost_v3 through the C ABI, inputs resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--weights uniform|integer] [--partitions B]
                    [--source gtrhqa|walk] [--single] [--table packed|table] [--no-cpu] [--no-qubo]

For N > 1 either launch it with torch.distributed.run (one rank per GPU) or run it plainly: without WORLD_SIZE in
the environment it starts its N ranks itself (as children, before anything touches the GPU) and relays rank 0's
line.  Ranks work on disjoint partitions with
no collective in the timed region; the final mean/min reduction over ranks (one RCCL all-reduce of a
few doubles) is done once after it and reported as ``reduce_ms``.

Rank 0 prints one JSON line.  ``roofline`` prices the dominant kernel of the headline variant against
HBM: algorithmic bytes per launch / mean launch time from HIP events on the launch stream.
The headline variant is the reference's default arithmetic (uniform random edge weights, fp64 distance
table); ``other_variants`` repeats the measurement with integer weights (closed-form grid metric) and with
the same integer table looked up in LDS, ``roofline_by_weights`` lists the three side by side; ``gtrhqa_place``, ``cost_matrix`` and ``qubo`` time the other device paths, ``shipped_circuits``
whole placement + cost batches of every placer on two of the reference's own circuits, ``shipped_circuits_cost_at_scale``
the cost launch alone on half a GB of partitions of three of them (rows of 32 to 256 bytes).
``cpu_baseline`` times the CPU oracle (a C port of the reference's per_timeslice_cost loop) on a bounded
sample of the same partitions on every host core, each result checked against the GPU's.
"""
import argparse
import json
import math
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def make_partitions(torch, dev, B, T, Q, ldq, C, cap, pmove, seed):
    """Seeded random-walk partitions: row 0 is the placers' block assignment (qubit i on core
    i // cap); in every later row each qubit moves to a uniformly random core with probability
    ``pmove`` (default: the move rate GTRHQA itself produces on this circuit)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    A = torch.zeros((B, T, ldq), dtype=torch.uint8, device=dev)
    row = (torch.arange(Q, device=dev) // cap).clamp(max=C - 1).to(torch.uint8).expand(B, Q).contiguous()
    A[:, 0, :Q] = row
    for t in range(1, T):
        move = torch.rand((B, Q), device=dev, generator=g) < pmove
        dest = torch.randint(0, C, (B, Q), device=dev, generator=g, dtype=torch.int16).to(torch.uint8)
        row = torch.where(move, dest, row)
        A[:, t, :Q] = row
    return A


def qubo_cpu_baseline(torch, dev, qm, dq, names, seconds):
    """The CPU figure beside spin-flips/s: oracle/qubo_anneal.c, a plain sequential single-flip Metropolis annealer
    (what neal's sampler does, Qubo_placer.py:139-140) on the same model written out as a sparse matrix by the oracle
    (oracle/qubo_cpu.py), same temperature range, one replica per thread.  The energy of every final state is
    checked against the GPU's exact-energy kernel."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import placer_oracle as O, qubo_cpu
    act, inter = O.qubo_slices(qm.ir.slices, qm.ir.gate_q, names)
    sm = qubo_cpu.SparseModel(act, inter, qm.C, qm.cap, qm.H)
    assert sm.N == qm.N
    # compact (oracle) order -> position in the product's padded state vector
    index = np.empty(sm.N, np.int64)
    for t in range(qm.T):
        for j in range(qm.C):
            for i in range(int(qm.n[t])):
                index[sm.var_x(t, j, i)] = qm.var_x(t, j, i)
            for p in range(qm.cap):
                index[sm.var_y(t, j, p)] = qm.var_y(t, j, p)
    hot, cold = qm.beta_range()
    rng = np.random.default_rng(11)

    def one(seed, sweeps):
        st = np.zeros(sm.N, np.int8)
        e0 = sm.energy(st)
        d, acc = sm.anneal(st, sweeps, hot, cold, seed)
        return st, e0 + d

    t0 = time.perf_counter()
    one(1, 2)
    per_sweep = (time.perf_counter() - t0) / 2
    sweeps = int(max(4, min(1000, seconds / 2 / max(per_sweep, 1e-6))))
    t0 = time.perf_counter()
    st1, e1 = one(2, sweeps)
    t_one = time.perf_counter() - t0
    cores = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "1024"))))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as pool:
        finals = list(pool.map(lambda k: one(100 + k, sweeps), range(cores)))
    t_all = time.perf_counter() - t0
    # every final state through the GPU's exact energy (128 states per call)
    states = [st1] + [f[0] for f in finals]
    want = [e1] + [f[1] for f in finals]
    X = np.zeros((128 * ((len(states) + 127) // 128), qm.Npad), np.int8)
    for k, st in enumerate(states):
        X[k, index] = st
    got = dq.energy(torch.from_numpy(X).to(dev)).cpu().numpy()[:len(states)]
    if [int(g) for g in got] != [int(w) for w in want]:
        raise SystemExit(f"parity failure: CPU annealer energies {want[:4]} vs GPU energy kernel {got[:4].tolist()}")
    return {"value": sm.N * sweeps * cores / t_all, "unit": "spin-flip attempts/s", "cores": cores, "kind": "port",
            "one_core_attempts_per_s": sm.N * sweeps / t_one, "sweeps": sweeps,
            "best_energy": int(min(want)),
            "sample": f"{cores} replicas x {sweeps} sweeps over the same {sm.N} variables through oracle/qubo_anneal.c "
                      f"(sequential single-flip Metropolis on the objective as a sparse matrix of {len(sm.col)} couplings, "
                      f"gcc -O2), one replica per thread, {t_all:.1f} s; the energy of every final state equals the GPU "
                      f"energy kernel's"}


def bench_qubo(torch, dev, rank, sweeps, replicas, cpu_seconds=0.0):
    """cfg4: 256-qubit layered circuit x 8 slices on a 4x4 mesh (16 cores, capacity 20): N = 35 328
    binary variables, ``replicas`` annealing replicas per GPU.  Returns spin-flip attempts/s of the
    annealer and the int8 MFMA rate of the exact-energy contraction."""
    import time as _t
    from quantum_simulator_amd import layered_gate_arrays
    from quantum_simulator_amd.engine import hops
    from quantum_simulator_amd.multicore import build_topology, graph_tables
    from quantum_simulator_amd.qubo import DeviceQubo, QuboModel
    from quantum_simulator_amd.timeslices import CircuitIR
    Q, T, cap = 256, 8, 20
    q0, q1, _ = layered_gate_arrays(Q, T, 0.5, seed=4242)
    ir = CircuitIR.from_gate_tables(Q, q0, q1)
    nodes, eu, ev, _ = graph_tables(build_topology("mesh", 4, 4))
    qm = QuboModel(ir, len(nodes), cap, hops(len(nodes), eu, ev))
    dq = DeviceQubo(qm, dev)
    R = replicas
    dq.anneal(R, 1, seed=1 + rank)
    torch.cuda.synchronize(dev)
    t0 = _t.perf_counter()
    X, dE = dq.anneal(R, sweeps, seed=1 + rank)
    torch.cuda.synchronize(dev)
    t_sa = _t.perf_counter() - t0
    e = dq.energy(X)
    torch.cuda.synchronize(dev)
    reps = 5
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(torch.cuda.current_stream(dev))
    for _ in range(reps):
        e = dq.energy(X)
    ev1.record(torch.cuda.current_stream(dev))
    torch.cuda.synchronize(dev)
    ms = ev0.elapsed_time(ev1) / reps
    nnz, mfma_macs, dense_macs = qm.coupling_counts()
    sec = ms * 1e-3
    cpu = None
    if cpu_seconds > 0:
        try:
            cpu = qubo_cpu_baseline(torch, dev, qm, dq, [f"q[{i}]" for i in range(Q)], cpu_seconds)
        except MemoryError as exc:
            cpu = {"error": f"{type(exc).__name__}: {exc}"}
    return {"cpu_baseline": cpu, "workload": f"cfg4: layered {Q}q x {T} slices, mesh 4x4, cap {cap}: N={qm.N} variables, {R} replicas/GPU",
            "sweeps": sweeps, "spin_flip_attempts_per_s": qm.N * R * sweeps / t_sa, "anneal_s": t_sa,
            "best_energy": int(e.min().item()), "energy_ms": ms,
            "useful_macs": nnz * R, "mfma_macs": mfma_macs * R, "dense_macs": dense_macs * R,
            "energy_int8_tops": 2.0 * mfma_macs * R / sec / 1e12, "useful_tops": 2.0 * nnz * R / sec / 1e12,
            "mfma_peak_int8_tops": 5000.0, "mfma_frac": 2.0 * mfma_macs * R / sec / 1e12 / 5000.0,
            "mfma_frac_useful": 2.0 * nnz * R / sec / 1e12 / 5000.0,
            "note": "energy: the W x W same-core couplings of every core by int8 MFMA (mfma_macs; useful_macs = the "
                    "non-zero couplings of the whole model), the same-qubit and slice-to-slice couplings from core bit "
                    "masks; dense_macs = what contracting the whole block-tridiagonal matrix would run.  Annealing "
                    "sweeps are LDS-latency/ALU bound (no HBM or MFMA roof applies)"}


def bench_cost_matrix(torch, dev, dc, ir, D, B, steps):
    """Secondary metric: op x core cost-matrix entries per second (TRHQA entries: two distance
    look-ups, six look-ahead slots, the reference's clamp arithmetic).  B instances, each at a different
    timeslice of the cfg5 circuit with all of its gates as ops and every core available; inputs and
    the fp64 output are resident in HBM.  The kernel is bound by its 8-byte-per-entry output write."""
    from quantum_simulator_amd.engine import DistanceTable, Engine, QP_CM_TRHQA
    import numpy as np
    C = D.shape[0]
    dt = DistanceTable(D, dev)
    eng = Engine(dev)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    tsel = (np.arange(B) * 7) % (ir.T - 4) + 1
    n_ops = (ir.slice_edge_ptr[tsel + 1] - ir.slice_edge_ptr[tsel]).astype(np.int32)
    max_ops = int(n_ops.max())
    ops = np.zeros((B, max_ops), np.int32)
    for b in range(B):
        ops[b, :n_ops[b]] = np.arange(ir.slice_edge_ptr[tsel[b]], ir.slice_edge_ptr[tsel[b] + 1])
    a = torch.randint(0, C, (B, dc.ldq), device=dev, generator=g, dtype=torch.int16).to(torch.uint8)
    avail = torch.arange(C, device=dev, dtype=torch.int16).to(torch.uint8).expand(B, C).contiguous()
    args = [torch.from_numpy(tsel.astype(np.int32)).to(dev), a, torch.from_numpy(ops).to(dev),
            torch.from_numpy(n_ops).to(dev), avail, torch.full((B,), C, dtype=torch.int32, device=dev)]
    out = torch.empty((B, max_ops, C), dtype=torch.float64, device=dev)
    eng.cost_matrix(QP_CM_TRHQA, dc, dt, *args, out=out)
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(torch.cuda.current_stream(dev))
    for _ in range(steps):
        eng.cost_matrix(QP_CM_TRHQA, dc, dt, *args, out=out)
    ev1.record(torch.cuda.current_stream(dev))
    torch.cuda.synchronize(dev)
    ms = ev0.elapsed_time(ev1) / steps
    entries = int(n_ops.sum()) * C
    gbs = entries * 8 / (ms * 1e-3) / 1e9
    return {"kind": "TRHQA", "instances": B, "entries_per_launch": entries, "kernel_ms": ms,
            "entries_per_s": entries / (ms * 1e-3),
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "note": "8 bytes written per entry; inputs are re-read from cache"}}


def bench_shipped(dev, rank=0, world=1):
    """The reference's shipped circuits (gate tables from tests/golden; the QASM files themselves do not travel):
    batches of whole placements + their costs, as the driver runs a config's repeats.  cfg1 = the 256-qubit QFT of
    example_config.txt on its 2x1 mesh (3 repeats, as shipped), cfg2 = qft_n18 on the 4x4 mesh (256 repeats), cfg3 =
    qft_n63 on the 8x8 torus (64 repeats).  The repeats of HQA / TRHQA / GTRHQA are identical computations (the
    reference's own loop runs the same deterministic placement `repeats` times, main.py:88-92); every one of them is
    computed here, none is reused, and the rows say so (identical_repeats).  With more than one rank the repeats are
    dealt round-robin to the ranks, as the driver deals them, and the per-placer sums and minima go through the
    driver's final reduction (one SUM and one MIN all-reduce of a few doubles): every rank ends with the job's
    mean and min cost."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from quantum_simulator_amd import driver
    from quantum_simulator_amd.golden import case_arch, case_circuit, load_case
    from quantum_simulator_amd.placers import PLACERS
    rows = []
    names = ("HQA", "TRHQA", "GTRHQA", "BFS", "RANDOM")
    for cfg, name, B in (("cfg1", "gen256_mesh2x1_u", 3), ("cfg2", "qft18_mesh4x4_u", 256), ("cfg3", "qft63_torus8x8_u", 64)):
        gc = load_case(name)
        circ, arch = case_circuit(gc), case_arch(gc)
        mine = len([r for r in range(B) if r % world == rank])
        sums = np.zeros((1, len(names), 4))
        mins = np.full((1, len(names)), np.inf)
        walls = []
        for pi, pname in enumerate(names):
            placer = PLACERS[pname](dev)
            random.seed(1000 + rank)  # RANDOM / BFS draw from the global stream: every rank its own
            placer.run_repeats(circ, arch, 2)
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            costs = placer.run_repeats(circ, arch, mine)[1] if mine else []
            torch.cuda.synchronize(dev)
            walls.append(time.perf_counter() - t0)
            fin = [c for c in costs if math.isfinite(c)]
            sums[0, pi] = [sum(fin), len(fin), walls[-1], 1]
            if fin:
                mins[0, pi] = min(fin)
        t0 = time.perf_counter()
        sums, mins = driver.reduce_table(sums, mins, dev if world > 1 and dist.get_backend() == "nccl" else None)
        reduce_ms = (time.perf_counter() - t0) * 1e3
        wall = torch.tensor(walls, dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(wall, op=dist.ReduceOp.MAX)
        for pi, pname in enumerate(names):
            w = float(wall[pi].item())
            rows.append({"config": cfg, "case": name, "placer": pname, "repeats": B, "ranks": world, "slices": gc.T,
                         "seconds": w, "slice_steps_per_s": B * gc.T / w,
                         "identical_repeats": pname in ("HQA", "TRHQA", "GTRHQA"),
                         "mean_cost": float(sums[0, pi, 0] / sums[0, pi, 1]) if sums[0, pi, 1] else math.inf,
                         "min_cost": float(mins[0, pi]), "finite_repeats": int(sums[0, pi, 1]),
                         "reduce_ms": reduce_ms if world > 1 else None})
    return rows


def bench_shipped_cost(torch, dev, steps=50):
    """Cost evaluation at scale on the reference's shipped circuits (gate tables and networks from tests/golden):
    for the circuit and network of cfg1 (256-qubit QFT, mesh 2x1, capacity 130), cfg2 (qft_n18, mesh 4x4) and cfg3
    (qft_n63, torus 8x8), about half a GB of partitions -- GTRHQA's own placements from seeded shuffles of the block
    assignment, computed on the device -- costed in one launch with the network's fp64 distance table.  Rows are
    16 to 256 bytes here: a wave takes 4 to 32 slices per step (k_cost_v3_groups)."""
    import numpy as np
    from quantum_simulator_amd.golden import case_arch, case_circuit, load_case
    from quantum_simulator_amd.placers import PLACERS
    rows = []
    for cfg, name in (("cfg1", "gen256_mesh2x1_u"), ("cfg2", "qft18_mesh4x4_u"), ("cfg3", "qft63_torus8x8_u")):
        gc = load_case(name)
        circ, arch = case_circuit(gc), case_arch(gc)
        ctx = PLACERS["GTRHQA"](dev)._context(circ, arch)
        ir, dc, dt, eng = ctx.ir, ctx.dc, ctx.arch.table, ctx.eng
        C, cap, Q = ctx.arch.C, int(gc.meta["cap"]), ir.Q
        B = max(64, (1 << 29) // ((ir.T + 1) * dc.ldq))
        rng = np.random.default_rng(4242)
        slots = np.repeat(np.arange(C), cap)
        init = np.zeros((B, dc.ldq), np.uint8)
        for b in range(B):
            init[b, :Q] = rng.permutation(slots)[:Q] if b else np.arange(Q) // cap
        A = eng.gtrhqa_place(dc, dt, torch.from_numpy(init).to(dev), cap)
        out = torch.empty(B, dtype=torch.float64, device=dev)
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < 0.2:  # untimed: the launches are short, the clocks must be up first
            for _ in range(20):
                eng.cost_v3(dc, dt, A, out=out)
            torch.cuda.synchronize(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(dev))
        for _ in range(steps):
            eng.cost_v3(dc, dt, A, out=out)
        ev1.record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
        ms = ev0.elapsed_time(ev1) / steps
        alg = B * ir.T * dc.ldq + 4 * ir.nE + 4 * (ir.T + 1) + 8 * B + 8 * C * C
        # the first partition is the block assignment: its cost is what the reference's GTRHQA run reports
        rows.append({"config": cfg, "case": name, "qubits": Q, "slices": ir.T, "cores": C, "row_bytes": dc.ldq,
                     "partitions": B, "kernel_ms": ms, "evals_per_s": B * ir.T / (ms * 1e-3),
                     "roofline": {"bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "alg_bytes_per_launch": alg},
                     "cost_of_the_block_assignment_run": float(out[0].item()),
                     "reference_gtrhqa_cost": gc.cost("GTRHQA") if gc.has("GTRHQA") else None})
        del A
        torch.cuda.empty_cache()
    return rows


def bench_broken_links(torch, dev, S=1024):
    """simulate_broken_links as one batch (HQA_placer.py:325-361): S damaged copies of the 8x8 torus (every link
    broken with probability 0.1), qft_n63 placed on each by HQA, every placement costed with its own table.
    Reports the three phases: distance matrices (host threads), placements (native lanes + device cost
    matrices), and the cost launch with one table per partition."""
    import numpy as np
    from quantum_simulator_amd import engine
    from quantum_simulator_amd.golden import case_arch, case_circuit, load_case
    from quantum_simulator_amd.placers import PLACERS
    gc = load_case("qft63_torus8x8_u")
    circ, arch = case_circuit(gc), case_arch(gc)
    placer = PLACERS["HQA"](dev)
    ctx = placer._context(circ, arch)
    rng = np.random.default_rng(99)
    W = np.tile(np.asarray(ctx.arch.w, np.float64), (S, 1))
    W[rng.random(W.shape) < 0.1] = np.inf
    from quantum_simulator_amd._lib import host_threads
    threads = host_threads(16)
    t0 = time.perf_counter()
    tables = engine.apsp_batch(ctx.arch.C, ctx.arch.eu, ctx.arch.ev, W, threads=threads)
    t_apsp = time.perf_counter() - t0
    tab = np.arange(S, dtype=np.int32)
    placer.place_batch(circ, arch, B=8, tables=tables[:8], tab_index=tab[:8])
    t0 = time.perf_counter()
    rows = placer.place_batch(circ, arch, B=S, tables=tables, tab_index=tab)
    t_place = time.perf_counter() - t0
    pad = np.zeros(rows.shape[:2] + (ctx.dc.ldq,), np.uint8)
    pad[:, :, :ctx.ir.Q] = rows
    Dz = tables.copy()
    Dz[:, np.arange(ctx.arch.C), np.arange(ctx.arch.C)] = 0.0
    d_tab, d_idx, d_A = torch.from_numpy(Dz).to(dev), torch.from_numpy(tab).to(dev), torch.from_numpy(pad).to(dev)
    out = torch.empty(S, dtype=torch.float64, device=dev)
    ctx.eng.cost_v3_multi(ctx.dc, d_tab, d_idx, d_A, out=out)
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    ev0.record(torch.cuda.current_stream(dev))
    for _ in range(reps):
        ctx.eng.cost_v3_multi(ctx.dc, d_tab, d_idx, d_A, out=out)
    ev1.record(torch.cuda.current_stream(dev))
    torch.cuda.synchronize(dev)
    ms = ev0.elapsed_time(ev1) / reps
    costs = out.cpu().numpy()
    bytes_ = S * (gc.T * ctx.dc.ldq + 8 * ctx.arch.C ** 2) + 8 * ctx.ir.nE
    return {"workload": f"{S} damaged copies of the 8x8 torus (links broken with p = 0.1), qft_n63 ({gc.T} slices) placed by "
                        f"HQA on each, one distance table per partition",
            "apsp_s": t_apsp, "apsp_threads": threads, "networks_per_s": S / t_apsp,
            "place_s": t_place, "place_slice_steps_per_s": S * gc.T / t_place,
            "cost_ms": ms, "cost_evals_per_s": S * gc.T / (ms * 1e-3),
            "cost_gbs": bytes_ / (ms * 1e-3) / 1e9, "finite_costs": int(np.isfinite(costs).sum()),
            "note": "cost launch: one workgroup per partition stages that partition's 32 KB table in LDS; bytes = rows + tables"}


def launch_ranks(n):
    """``python bench.py --gpus N`` without a launcher around it: this process starts N ranks of the same command
    line under torch.distributed.run (one per GPU, rendezvous on 127.0.0.1 at a free port), hands their output
    through and exits with their code.  Nothing here touches the GPU (no torch import): the ranks are children,
    not replacements of this process."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--weights", choices=["uniform", "integer"], default="uniform",
                    help="edge weights of the headline number: uniform(1,3) -> fp64 distance table, what the reference's "
                         "own configs draw (multicore.py:159-161); integer (2,2) -> exact uint16 table, on this mesh the "
                         "closed-form grid metric (a special case).  The other one is run too: other_variants")
    ap.add_argument("--single", action="store_true", help="run only the --weights variant")
    ap.add_argument("--table", choices=["packed", "table"], default="packed",
                    help="form of the distance table for the headline variant: what the library chooses, or look-ups "
                         "even where a closed form exists (for profiling the look-up kernel on its own)")
    ap.add_argument("--partitions", type=int, default=4096, help="partitions per GPU")
    ap.add_argument("--qubits", type=int, default=1024)
    ap.add_argument("--slices", type=int, default=1000)
    ap.add_argument("--source", choices=["gtrhqa", "walk"], default="gtrhqa",
                    help="where the partitions come from: the GTRHQA placer itself (on the device, from shuffled "
                         "starting assignments) or a seeded random walk with move probability --pmove")
    ap.add_argument("--pmove", type=float, default=0.95)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-qubo", action="store_true")
    ap.add_argument("--qubo-sweeps", type=int, default=1000)
    ap.add_argument("--qubo-replicas", type=int, default=4096)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus differ")
    # one rank per GPU.  For a rehearsal of the multi-rank control flow on a box with fewer GPUs than ranks,
    # QP_BENCH_BACKEND=gloo lets ranks share a device (RCCL refuses two ranks on one GPU; gloo does not care)
    backend = os.environ.get("QP_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and backend == "nccl":
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({ndev} visible)")
    dev = torch.device("cuda", local_rank % max(ndev, 1))
    torch.cuda.set_device(dev)
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # what ran where, for whoever reads the line: the collective backend, its world size and every rank's device
    props = torch.cuda.get_device_properties(dev)
    mine = {"rank": rank, "device": f"cuda:{dev.index}", "name": props.name,
            "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", "")) or None}
    if dist.is_initialized():
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        distributed = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": gathered}
    else:
        distributed = {"backend": None, "world_size": 1, "ranks": [mine]}

    from quantum_simulator_amd import layered_gate_arrays
    from quantum_simulator_amd._lib import host_threads
    from quantum_simulator_amd.engine import DeviceCircuit, DistanceTable, Engine, apsp, QP_DT_U16
    distributed["host_threads_per_rank"] = {"share": host_threads(1 << 20), "polling_lanes": host_threads(8, polling=True),
                                            "local_world_size": int(os.environ.get("LOCAL_WORLD_SIZE", "1"))}
    from quantum_simulator_amd.multicore import graph_tables, multicore
    from quantum_simulator_amd.timeslices import CircuitIR

    Q, T, B = args.qubits, args.slices, args.partitions
    CX = CY = 16
    cap = 8
    circ_seed, weight_seed = 2025, 1234
    q0, q1, _ = layered_gate_arrays(Q, T, 0.5, seed=circ_seed)
    ir = CircuitIR.from_gate_tables(Q, q0, q1)
    dc, eng = DeviceCircuit(ir, dev), Engine(dev)
    C = CX * CY
    out = torch.empty(B, dtype=torch.float64, device=dev)
    alg_bytes = B * ir.T * Q + 4 * ir.nE + 4 * (ir.T + 1) + 8 * B + 8 * C * C

    archs = {}

    def distances(weights):
        random.seed(weight_seed)
        wr = (1.0, 3.0) if weights == "uniform" else (2.0, 2.0)
        arch = multicore(CX, CY, ncm_qubits=4, ncp_qubits=cap, edge_weight_range=wr, network_type="mesh")
        nodes, eu, ev, w = graph_tables(arch.network)
        archs[weights] = arch
        return wr, apsp(len(nodes), eu, ev, w)

    place = None
    if args.source == "gtrhqa":
        # the placer's own output: B GTRHQA placements of the circuit (T+1 rows each) from seeded shuffles
        # of the block assignment, instance 0 from the block assignment itself; all slices on the device
        rng = np.random.default_rng(777 + rank)
        slots = np.repeat(np.arange(C), cap)
        init = np.zeros((B, dc.ldq), np.uint8)
        for b in range(B):
            init[b, :Q] = rng.permutation(slots)[:Q] if b else np.arange(Q) // cap
        d_init = torch.from_numpy(init).to(dev)
        dt0 = DistanceTable(distances(args.weights)[1], dev)
        A = torch.empty((B, ir.T + 1, dc.ldq), dtype=torch.uint8, device=dev)
        eng.gtrhqa_place(dc, dt0, d_init, cap, out=A)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        eng.gtrhqa_place(dc, dt0, d_init, cap, out=A)
        torch.cuda.synchronize(dev)
        t_place = time.perf_counter() - t0
        moved = float((A[:64, 2:ir.T, :Q] != A[:64, 1:ir.T - 1, :Q]).float().mean().item())
        place = {"kernel": "k_gtrhqa_place", "instances": B, "seconds": t_place,
                 "slice_steps_per_s": B * ir.T / t_place, "moved_fraction_between_rows": moved,
                 "note": "whole placements on the device (bidding rounds included); these rows are the cost workload"}
        source = (f"GTRHQA placements of this circuit computed on the device from seeded shuffles of the block "
                  f"assignment ({ir.T + 1} rows each; {moved:.1%} of the qubits change core between consecutive rows)")
    else:
        A = make_partitions(torch, dev, B, ir.T, Q, dc.ldq, C, cap, args.pmove, seed=777 + rank)
        source = (f"random walk from the block assignment, pmove={args.pmove} "
                  f"(GTRHQA's own output on this circuit changes >95% of the qubits between rows)")
    cpu_sample = A[:64, :, :Q].cpu().numpy() if (world == 1 and not args.no_cpu) else None

    def run_variant(weights, table="packed"):
        """Times args.steps passes of the cost path with one weight setting and one form of the distance
        table ("packed": what the library chooses; "table": look-ups even when a closed form exists)."""
        wr, D = distances(weights)
        dt = DistanceTable(D, dev)
        closed = table == "packed" and dt.kind == QP_DT_U16 and bool(dt.grid.valid)
        if dt.kind == QP_DT_U16:
            how = ("none: closed-form grid metric (v_sad_u8 on packed core ids), verified against the matrix"
                   if closed else "uint16, LDS-resident")
            # names as rocprofv3 prints them; from 64 partitions on, table look-ups take four partitions per wave step
            kname = "k_cost_v3_grid_row<false, 4>" if closed else ("k_cost_v3_row4<1, 4>" if B >= 64 else "k_cost_v3_row<1, 0, 4>")
        elif C * C * 8 <= 128 * 1024:
            how, kname = "f64, LDS-resident", "k_cost_v3_row<0, 0, 4>"
        elif table == "packed" and dt.symmetric:
            how = ("f64, 512 KB (too large for LDS), symmetric to rounding: core pairs are counted in LDS "
                   "(uint32 per unordered pair), one product with the table per partition")
            kname = "k_cost_v3_pairs<4, false>"
        else:
            how, kname = "f64, 512 KB: LDS-resident in 4 bands of source cores, one pass per band", "k_cost_v3_row<0, 3, 4> x4"
        # untimed: about 0.3 s of the same launches first, so that a box that has just been idle is at its
        # running clocks before the W warm-up steps (one reading on a fresh box came out 15 % slow without it)
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < 0.3:
            for _ in range(20):
                eng.cost_v3(dc, dt, A, out=out, table=table)
            torch.cuda.synchronize(dev)
        for _ in range(args.warmup):
            eng.cost_v3(dc, dt, A, out=out, table=table)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        stream = torch.cuda.current_stream(dev)
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(args.steps):
            eng.cost_v3(dc, dt, A, out=out, table=table)
        ev1.record(stream)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
        dev_ms = ev0.elapsed_time(ev1) / args.steps  # mean device time of one launch (+ the chunk-sum kernel)
        tmax = torch.tensor([wall], dtype=torch.float64, device=dev)
        stats = torch.stack([out.sum(), torch.tensor(float(B), dtype=torch.float64, device=dev)])
        mn = out.min().reshape(1)
        reduce_ms = None
        per_rank = [wall / args.steps * 1e3]
        if world > 1:
            walls = [torch.zeros_like(tmax) for _ in range(world)]
            dist.all_gather(walls, tmax)
            per_rank = [float(w.item()) / args.steps * 1e3 for w in walls]
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            torch.cuda.synchronize(dev)
            r0 = time.perf_counter()
            dist.all_reduce(stats, op=dist.ReduceOp.SUM)   # final mean cost = sum / count (main.py:103-104)
            dist.all_reduce(mn, op=dist.ReduceOp.MIN)      # final min cost
            torch.cuda.synchronize(dev)
            reduce_ms = (time.perf_counter() - r0) * 1e3
        wall = float(tmax.item())
        achieved = alg_bytes / (dev_ms * 1e-3) / 1e9
        return {"weights": f"{weights} {wr}", "value": world * B * ir.T * args.steps / wall, "unit": "evals/s",
                "ms_per_step": wall / args.steps * 1e3, "ms_per_step_per_rank": per_rank,
                "dtype": "u8 core ids / u32 sums" if dt.kind == QP_DT_U16 else "f64",
                "table": how, "_closed": closed,
                "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": kname,
                             "alg_bytes_per_launch": alg_bytes, "kernel_ms": dev_ms},
                "mean_cost": float(stats[0].item() / stats[1].item()), "min_cost": float(mn.item()),
                "reduce_ms": reduce_ms, "_D": D, "_kind": dt.kind, "_costs": out.cpu().numpy().copy()}

    primary = run_variant(args.weights, table=args.table)
    others = []
    if not args.single:
        for weights in (("integer",) if args.weights == "uniform" else ("uniform",)):
            v = run_variant(weights)
            others.append(v)
        for v in [primary] + list(others):
            if v["_closed"]:  # the same integer table with the closed form switched off
                others.append(run_variant("integer", table="table"))
                break

    cm = None
    if not args.no_qubo:
        cm = bench_cost_matrix(torch, dev, dc, ir, primary["_D"], min(B, 2048), 5)
        if world > 1:
            fl = torch.tensor([cm["entries_per_s"]], dtype=torch.float64, device=dev)
            dist.all_reduce(fl, op=dist.ReduceOp.SUM)
            cm["entries_per_s"] = float(fl.item())
    activity = None
    if not args.no_qubo:
        # per-core inter-core activity counts (HQA_placer.py:288-321) of the same partitions, a slice of the batch
        nb = min(B, 1024)
        Asub = A[:nb]
        eng.core_activity(dc, Asub, C)
        torch.cuda.synchronize(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(dev))
        for _ in range(5):
            counts = eng.core_activity(dc, Asub, C)
        ev1.record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
        ms = ev0.elapsed_time(ev1) / 5
        activity = {"kernel": "k_cost_v3_pairs<4, true>" if 512 < dc.ldq <= 1024 else "k_core_activity", "partitions": nb, "kernel_ms": ms, "evals_per_s": nb * ir.T / (ms * 1e-3),
                    "gbs": nb * ir.T * dc.ldq / (ms * 1e-3) / 1e9, "events_in_partition_0": int(counts[0].sum().item())}
        del Asub, counts
    gate_ordered = None
    if not args.no_qubo and world == 1:
        # the gate-ordered cost (BFS_placer.py:202-239) of a slice of the batch on a 15 x 17 mesh (that cost keeps core
        # id 255 for "qubit absent"): read row by row against visit by visit
        try:
            from quantum_simulator_amd.engine import QP_V2
            random.seed(weight_seed)
            arch255 = multicore(15, 17, ncm_qubits=4, ncp_qubits=cap, edge_weight_range=(1.0, 3.0), network_type="mesh")
            nodes, eu, ev, w = graph_tables(arch255.network)
            dt255 = DistanceTable(apsp(len(nodes), eu, ev, w), dev)
            nb = min(B, 1024)
            Asub = A[:nb, :ir.T].clamp(max=254).contiguous()
            gate_ordered = {"partitions": nb, "cores": 255, "table": "f64, symmetric to rounding"}
            for key, circuit in (("row_walk", dc), ("visit_by_visit", DeviceCircuit(ir, dev, visit_mask=False))):
                o2 = torch.empty(nb, dtype=torch.float64, device=dev)
                for _ in range(2):
                    eng.cost_v2(circuit, dt255, Asub, QP_V2, out=o2)
                torch.cuda.synchronize(dev)
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record(torch.cuda.current_stream(dev))
                for _ in range(5):
                    eng.cost_v2(circuit, dt255, Asub, QP_V2, out=o2)
                ev1.record(torch.cuda.current_stream(dev))
                torch.cuda.synchronize(dev)
                ms = ev0.elapsed_time(ev1) / 5
                gate_ordered[key] = {"kernel": "k_cost_v2_pairs<4>" if key == "row_walk" else "k_cost_v2<0, false, 256>",
                                     "ms": ms, "evals_per_s": nb * ir.T / (ms * 1e-3), "gbs": nb * ir.T * dc.ldq / (ms * 1e-3) / 1e9,
                                     "mean_cost": float(o2.mean().item())}
            del Asub
        except Exception as exc:  # noqa: BLE001 - a side table must not cost the headline line
            gate_ordered = {"error": f"{type(exc).__name__}: {exc}"}
    qubo = None
    if not args.no_qubo:
        del A
        torch.cuda.empty_cache()
        qubo = bench_qubo(torch, dev, rank, args.qubo_sweeps, args.qubo_replicas,
                          cpu_seconds=args.cpu_seconds / 2 if (world == 1 and not args.no_cpu) else 0.0)
        if world > 1:
            fl = torch.tensor([qubo["spin_flip_attempts_per_s"]], dtype=torch.float64, device=dev)
            dist.all_reduce(fl, op=dist.ReduceOp.SUM)
            qubo["spin_flip_attempts_per_s"] = float(fl.item())
    shipped = None
    if not args.no_qubo:
        try:
            shipped = bench_shipped(dev, rank, world)
        except Exception as exc:  # noqa: BLE001 - a side table must not cost the headline line
            if world > 1:
                raise  # the other ranks are waiting in a collective
            shipped = {"error": f"{type(exc).__name__}: {exc}"}
    broken = None
    shipped_cost = None
    if world == 1 and not args.no_qubo:
        try:
            broken = bench_broken_links(torch, dev)
        except Exception as exc:  # noqa: BLE001 - a side table must not cost the headline line
            broken = {"error": f"{type(exc).__name__}: {exc}"}
        try:
            shipped_cost = bench_shipped_cost(torch, dev)
        except Exception as exc:  # noqa: BLE001
            shipped_cost = {"error": f"{type(exc).__name__}: {exc}"}
    if rank == 0:
        D, kind, gpu_costs = primary.pop("_D"), primary.pop("_kind"), primary.pop("_costs")
        res = {
            "metric": "timeslice cost evals/sec", "value": primary["value"], "unit": "evals/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": primary["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": primary["dtype"],
            "data": "synthetic",
            "config": {"workload": f"cfg5: layered {Q}q x {ir.T} slices ({ir.nE // ir.T} cx/slice), mesh {CX}x{CY}, "
                                   f"cap {cap}, {primary['weights']} edge weights, {B} partitions/GPU, "
                                   f"cost V3 (HQA/TRHQA/GTRHQA definition)",
                       "partitions": source,
                       "seeds": {"circuit": circ_seed, "weights": weight_seed, "partitions": "777+rank"},
                       "table": primary["table"]},
            "roofline": primary["roofline"], "mean_cost": primary["mean_cost"], "min_cost": primary["min_cost"],
            "distributed": distributed,
            "reduce_ms": primary["reduce_ms"], "gtrhqa_place": place, "cost_matrix": cm, "core_activity": activity, "gate_ordered_cost": gate_ordered, "qubo": qubo,
            "shipped_circuits": shipped, "shipped_circuits_cost_at_scale": shipped_cost, "broken_links": broken,
        }
        # HBM bytes per launch as measured with rocprofv3 PMC counters in a separate run of this workload
        # (profiles/r03_traffic.json, one entry per kernel name; FETCH_SIZE doubled as gfx950 requires for
        # 16-byte-per-lane streams).  An entry is used only for the kernel and the workload it was measured on.
        try:
            with open(os.path.join(ROOT, "profiles", "r03_traffic.json")) as f:
                measured = json.load(f)
        except (OSError, ValueError):
            measured = {}
        for v in [primary] + others:
            tr = measured.get(v["roofline"]["kernel"])
            if tr and (tr.get("Q"), tr.get("T"), tr.get("B"), tr.get("source")) == (Q, ir.T, B, args.source):
                v["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
                v["roofline"]["traffic_source"] = tr.get("profile")
        for row in shipped_cost or []:  # the same for the shipped circuits' launches (keyed by config, case and batch)
            tr = measured.get("shipped:" + row["config"])
            if tr and (tr.get("case"), tr.get("partitions")) == (row["case"], row["partitions"]):
                row["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
                row["roofline"]["traffic_source"] = tr.get("profile")
                row["roofline"]["kernel"] = tr.get("kernel")
        by_w = {}
        for v in [primary] + others:
            key = ("uniform_fp64_table" if v["dtype"] == "f64" else
                   "integer_closed_form" if v["_closed"] else "integer_uint16_table")
            by_w[key] = {"frac": v["roofline"]["frac"], "kernel": v["roofline"]["kernel"], "kernel_ms": v["roofline"]["kernel_ms"],
                         "evals_per_s": v["value"], "traffic": v["roofline"]["traffic"]}
        res["roofline_by_weights"] = by_w
        res["ms_per_step_per_rank"] = primary["ms_per_step_per_rank"]
        ref1 = os.environ.get("QP_BENCH_REF_1GPU")
        if ref1:  # evals/s of a one-GPU run of the same command, given by whoever runs the sweep
            res["scaling_efficiency"] = primary["value"] / (world * float(ref1))
        primary.pop("_closed")
        for o in others:
            for k in ("_D", "_kind", "_costs", "_closed"):
                o.pop(k)
        if others:
            res["other_variants"] = others
        if world == 1 and not args.no_cpu:
            # CPU baseline on this box's host cores: the C restatement of the reference's cost loop
            # (oracle/cost_oracle.c, same left-to-right sums, one core), on a bounded sample of this
            # run's partitions, each also checked against the GPU result; plus the pure-Python port.
            from oracle import c_oracle, placer_oracle as O
            gt = c_oracle.GateTables(ir.slices, ir.gate_q, ir.gate_mr)
            nterms = ir.nE + (ir.T - 2) * Q
            tol = 0.0 if kind == QP_DT_U16 else max(1e-12, nterms * 2.0 ** -53)
            sample = cpu_sample
            n_c, t_c = 0, 0.0
            while t_c < args.cpu_seconds / 2:  # one core
                k = n_c % len(sample)
                c0 = time.perf_counter()
                want = c_oracle.cost_v3(gt, Q, sample[k], D)
                t_c += time.perf_counter() - c0
                if abs(float(gpu_costs[k]) - want) > tol * abs(want):
                    raise SystemExit(f"parity failure on sample partition {k}: gpu {gpu_costs[k]!r} oracle {want!r}")
                n_c += 1
            # the same on every core this process may use (the C call releases the GIL): one thread per core,
            # each walking the sample for a fixed number of evaluations sized from the one-core rate
            from concurrent.futures import ThreadPoolExecutor
            cores = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "1024"))))
            per_thread = max(2, int(n_c / t_c * args.cpu_seconds / 2))

            def worker(w):
                for i in range(per_thread):
                    k = (w * 7 + i) % len(sample)
                    want = c_oracle.cost_v3(gt, Q, sample[k], D)
                    if abs(float(gpu_costs[k]) - want) > tol * abs(want):
                        raise SystemExit(f"parity failure on sample partition {k}")
                return per_thread

            c0 = time.perf_counter()
            with ThreadPoolExecutor(cores) as pool:
                n_all = sum(pool.map(worker, range(cores)))
            t_all = time.perf_counter() - c0
            Dl = D.tolist()
            n_py, t_py = 0, 0.0
            while t_py < 3.0 and n_py < len(sample):
                rows = sample[n_py].tolist()
                c0 = time.perf_counter()
                want = O.cost_v3(ir.slices, ir.gate_q, ir.gate_mr, rows, Dl)
                t_py += time.perf_counter() - c0
                if abs(float(gpu_costs[n_py]) - want) > tol * abs(want):
                    raise SystemExit(f"parity failure (python port) on sample partition {n_py}")
                n_py += 1
            # the reference's own structure: no distance matrix, a fresh weighted shortest-path search per term
            # (Placer.py:63-76), on the first slices of one partition; and NumPy gathers on a precomputed matrix
            from oracle import cpu_variants
            ref_rate = ref_note = None
            try:
                arch0 = archs[args.weights]
                node_list = list(arch0.network.nodes)
                rows0 = sample[0].tolist()
                n_sl, t_ref = 3, 0.0
                while True:
                    c0 = time.perf_counter()
                    got_ref, searches = cpu_variants.cost_v3_reference_structure(arch0.network, node_list, ir.slices,
                                                                                 ir.gate_q, ir.gate_mr, rows0, n_sl)
                    t_ref = time.perf_counter() - c0
                    if t_ref > 1.5 or n_sl >= ir.T:
                        break
                    n_sl = min(ir.T, n_sl * 2)
                want_ref = O.cost_v3(ir.slices[:n_sl], ir.gate_q, ir.gate_mr, rows0[:n_sl], Dl)
                if got_ref != want_ref:
                    raise SystemExit(f"parity failure (search per term): {got_ref!r} vs {want_ref!r}")
                ref_rate = n_sl / t_ref
                ref_note = f"{searches} shortest-path searches for the first {n_sl} slices of one partition, {t_ref:.1f} s, one core"
            except ImportError:
                pass
            npc = cpu_variants.NumpyCost(ir.slices, ir.gate_q, ir.gate_mr, D)
            n_np, t_np = 0, 0.0
            while t_np < 2.0:
                k = n_np % len(sample)
                c0 = time.perf_counter()
                want = npc.cost(sample[k])
                t_np += time.perf_counter() - c0
                if abs(float(gpu_costs[k]) - want) > max(tol, 1e-12) * abs(want):
                    raise SystemExit(f"parity failure (numpy) on sample partition {k}")
                n_np += 1
            res["cpu_baseline"] = {"reference_structure_evals_per_s": ref_rate, "reference_structure_sample": ref_note,
                                   "numpy_evals_per_s": n_np * ir.T / t_np,
                                   "value": n_all * ir.T / t_all, "unit": "evals/s", "cores": cores, "kind": "port",
                                   "sample": f"{n_all} evaluations over {len(sample)} of this run's partitions x {ir.T} "
                                             f"slices through oracle/cost_oracle.c (gcc -O2, precomputed distance "
                                             f"matrix, sequential sums) on {cores} threads, {t_all:.1f} s; every result "
                                             f"checked against the GPU's",
                                   "one_core_evals_per_s": n_c * ir.T / t_c,
                                   "python_port_evals_per_s": n_py * ir.T / t_py if n_py else None,
                                   "host_cpus": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
        print(json.dumps(res), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
