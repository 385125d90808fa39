"""Config-file driver: the reference's ``python main.py --config file`` flow (main.py:16-165) on this
package's placers, with the work optionally sharded over the ranks of a torch.distributed job.

Config format and required keys are the reference's (config_parser.py:20-84): ``key = value`` lines,
``#`` comments; keys mode, topology, num_cores_x, num_cores_y, ncm_qubits, ncp_qubits,
edge_weight_range, circuit_folders, placers, repeats, out_csv, plot_file.  Optional additions
(``batch_repeats = 0`` runs every repeat on its own as main.py does; by default the repeats of a placer run
as one batch: HQA, TRHQA and GTRHQA, which draw no random numbers, on the device; RANDOM and BFS in a native
loop that draws from ``random`` as the calls would, with one cost launch for all of them):
``seed`` (seeds Python's global ``random`` once before the architecture is built; the reference never
seeds, so its runs are not reproducible) and ``device``.

Output: the same CSV (``file, <P>_time_mean, <P>_cost_mean`` per placer, means over finite repeats,
main.py:103-104,117-130).  A repeat that raises records inf, as in main.py:99-102.

Sharding: the units (file, placer, repeat) are dealt round-robin to ranks; each rank keeps per
(file, placer) the sums [finite cost sum, finite count, time sum, time count] and the minimum cost,
and one all-reduce (SUM) plus one (MIN) at the end gives every rank the full table.  With more than
one rank the RNG-driven placers (RANDOM, BFS) seed ``random`` per unit from (seed, file, placer,
repeat), because one shared stream cannot be split; deterministic placers are unaffected.
"""
from __future__ import annotations

import csv
import math
import os
import random
import time
import zlib

import numpy as np

TOPOLOGY_NAMES = ["mesh", "line", "star", "tree", "torus", "complete"]
REQUIRED = ["mode", "topology", "num_cores_x", "num_cores_y", "ncm_qubits", "ncp_qubits", "edge_weight_range",
            "circuit_folders", "placers", "repeats", "out_csv", "plot_file"]




def _scalar_or_list(text: str):
    """A config value as config_parser.py:4-17 reads it: whatever ``int()`` accepts is an int; otherwise a value
    with a comma is the list of its non-empty items, as floats when ``float()`` accepts every one of them (a
    single item then stands alone); anything else stays a string."""
    text = text.strip()
    try:
        return int(text)
    except ValueError:
        pass
    if "," not in text:
        return text
    items = [piece for piece in (part.strip() for part in text.split(",")) if piece]
    try:
        numbers = [float(piece) for piece in items]
        return numbers if len(numbers) > 1 else numbers[0]
    except (ValueError, IndexError):
        return items


def _as_name_list(value):
    pieces = value if isinstance(value, list) else str(value).split(",")
    return [str(piece).strip() for piece in pieces if str(piece).strip()]


def parse_config_file(path: str) -> dict:
    """config_parser.py:20-84: same keys, same checks in the same order (mode: present, then its value; topology:
    present, then its value; then the rest), same normalisation of the weight range, the folder list (with the
    warning for a folder that does not exist) and the placer list."""
    if not os.path.exists(path):
        raise AssertionError(f"Config file not found: {path}")
    cfg = {}
    with open(path) as handle:
        for raw in handle:
            entry = raw.split("#", 1)[0] if raw.lstrip().startswith("#") else raw
            key, sep, value = entry.partition("=")
            if sep and key.strip():
                cfg[key.strip()] = _scalar_or_list(value)
    checks = {"mode": (["simulate"], "mode must be 'simulate' or ..."),
              "topology": (TOPOLOGY_NAMES, "Invalid topology, must be one of " + ", ".join(TOPOLOGY_NAMES))}
    for key in REQUIRED:  # presence, and for mode and topology the value, key by key (config_parser.py:35-52)
        if key not in cfg:
            raise AssertionError(f"Missing required parameter: {key}")
        if key in checks and cfg[key] not in checks[key][0]:
            raise AssertionError(checks[key][1])
    span = cfg["edge_weight_range"]
    if not (isinstance(span, list) and len(span) >= 2):
        raise AssertionError("edge_weight_range must be two numbers separated by a comma, e.g. 1.0,10.0")
    cfg["edge_weight_range"] = (float(span[0]), float(span[1]))
    cfg["circuit_folders"] = [os.path.expanduser(name) for name in _as_name_list(cfg["circuit_folders"])]
    for folder in cfg["circuit_folders"]:
        if not os.path.isdir(folder):
            print(f"Warning: circuit folder does not exist: {folder}")
    cfg["placers"] = _as_name_list(cfg["placers"])
    return cfg


def build_arch(cfg):
    """main.py:50-61, including its fall-back: any constructor error (e.g. topology ``complete``, which
    the parser accepts and the topology builder does not know) yields a 4x4 all-to-all machine with 16
    qubits per core and weights in (1, 10)."""
    from .multicore import multicore
    try:
        return multicore(int(cfg.get("num_cores_x", 8)), int(cfg.get("num_cores_y", 8)),
                         ncm_qubits=int(cfg.get("ncm_qubits", 4)), ncp_qubits=int(cfg.get("ncp_qubits", 2)),
                         edge_weight_range=tuple(cfg.get("edge_weight_range", (1.0, 3.0))),
                         network_type=cfg.get("topology", "all_to_all"))
    except Exception as exc:  # noqa: BLE001 - mirrors the reference's blanket fall-back
        print("Warning: multicore init failed with args, falling back to default ctor:", exc)
        return multicore(4, 4)


def list_qasm(folders):
    out = []
    for folder in folders:
        if not os.path.exists(folder):
            print(f"Warning: circuit folder not found: {folder}")
            continue
        for fn in sorted(os.listdir(folder)):
            if fn.lower().endswith(".qasm"):
                out.append(os.path.join(folder, fn))
    if not out:
        raise RuntimeError("No .qasm files found in circuit_folders.")
    return out


def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except ImportError:
        pass
    return None


def reduce_table(local: np.ndarray, local_min: np.ndarray, device=None):
    """All-reduce the per-(file, placer) sums (SUM) and minima (MIN) over the job's ranks; identity
    when not distributed.  Arrays are float64; the payload is a few hundred bytes."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return local, local_min
    import torch
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    s = torch.from_numpy(np.ascontiguousarray(local)).to(dev)
    m = torch.from_numpy(np.ascontiguousarray(local_min)).to(dev)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(m, op=dist.ReduceOp.MIN)
    return s.cpu().numpy(), m.cpu().numpy()


def slice_shard(T: int, rank: int, world: int):
    """Contiguous piece of the timeslices 0..T-1 for ``rank`` of ``world``: (t_begin, t_end, row0), where
    row0 is the first partition row the piece reads (the row before the piece from slice 2 on: the movement
    term of HQA_placer.py:277-285 compares a slice with the one before it)."""
    t_begin, t_end = T * rank // world, T * (rank + 1) // world
    return t_begin, t_end, (t_begin - 1 if t_begin >= 2 else t_begin)


def sharded_cost_v3(engine, dc, dt, rows, table="packed"):
    """HQA-family cost of partitions too long for one GPU's turn: every rank holds ``rows``, the uint8
    [B, t_end - row0, ldq] window of its slice_shard of the circuit, costs its own slices and one SUM
    all-reduce of B doubles adds the pieces (identity when not distributed).  Returns float64 [B]."""
    dist = _dist()
    rank = dist.get_rank() if dist else 0
    world = dist.get_world_size() if dist else 1
    t_begin, t_end, row0 = slice_shard(dc.ir.T, rank, world)
    part = engine.cost_v3_slices(dc, dt, rows, t_begin, t_end, row0, table=table).sum(dim=1)
    if dist is not None and world > 1:
        if dist.get_backend() != "nccl":
            part = part.cpu()
        dist.all_reduce(part, op=dist.ReduceOp.SUM)
    return part


def run_simulation(cfg, device=None, verbose=True):
    from ._lib import QplaceError
    from .circuit import circuit_from_qasm
    from .placers import PLACERS

    dist = _dist()
    rank = dist.get_rank() if dist else 0
    world = dist.get_world_size() if dist else 1
    repeats = int(cfg["repeats"])
    names = [p for p in cfg["placers"]]
    seed = cfg.get("seed")
    if seed is None and world > 1:
        # every rank must build the same machine: the edge weights come off the global random stream, so
        # without a configured seed rank 0 draws one and all ranks use it
        box = [random.randrange(1 << 62) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        seed = box[0]
    if seed is not None:
        random.seed(int(seed))
    if device is None and cfg.get("device") not in (None, ""):
        import torch
        device = torch.device(str(cfg["device"]))
    arch = build_arch(cfg)
    files = list_qasm(cfg["circuit_folders"])
    placers = {n: PLACERS[n](device) for n in names if n in PLACERS}
    for n in names:
        if n not in placers and verbose and rank == 0:
            print(f"Placer '{n}' not recognized, skipping.")

    # [cost sum, cost count, time sum, time count] per (file, placer)
    sums = np.zeros((len(files), len(names), 4))
    mins = np.full((len(files), len(names)), np.inf)
    def record(fi, pi, dt, cost):
        if math.isfinite(cost):
            sums[fi, pi, 0] += cost
            sums[fi, pi, 1] += 1
            mins[fi, pi] = min(mins[fi, pi], cost)
        if math.isfinite(dt):
            sums[fi, pi, 2] += dt
            sums[fi, pi, 3] += 1

    batched = str(cfg.get("batch_repeats", "1")).strip().lower() not in ("0", "false", "no")
    unit = 0
    for fi, path in enumerate(files):
        circuit = None
        for pi, name in enumerate(names):
            if name not in placers:
                continue
            mine = [rep for rep in range(repeats) if (unit + rep) % world == rank]
            unit += repeats
            if not mine:
                continue
            if circuit is None:
                circuit = circuit_from_qasm(path, maxwidth=1024)
            # the repeats of a placer run as one batch: HQA / TRHQA / GTRHQA placements advance together on the
            # device; RANDOM / BFS placements run one after the other in the native loop (same draws from the
            # random stream as one call per repeat) and are costed by one launch
            run_batch = getattr(placers[name], "run_repeats", None) if batched else None
            if run_batch is not None:
                try:
                    kw = {}
                    if world > 1 and name in ("RANDOM", "BFS"):  # every repeat draws from its own seeded stream
                        kw["states"] = []
                        for rep in mine:
                            random.seed(zlib.crc32(f"{seed}|{path}|{name}|{rep}".encode()))
                            kw["states"].append(random.getstate())
                    times, costs = run_batch(circuit, arch, len(mine), **kw)
                    for dt, cost in zip(times, costs):
                        record(fi, pi, dt, cost)
                    continue
                except QplaceError:
                    raise  # the library or the device failed: never paper over that with a slower path
                except Exception as exc:  # noqa: BLE001 - what a repeat of the reference would raise (main.py:99-102):
                    if verbose:           # run them one at a time so that each records its own inf
                        print(f"[{name}] batched run raised ({type(exc).__name__}: {exc}); repeating one at a time")
            for rep in mine:
                if world > 1 and name in ("RANDOM", "BFS"):
                    random.seed(zlib.crc32(f"{seed}|{path}|{name}|{rep}".encode()))
                try:
                    t0 = time.perf_counter()
                    part = placers[name].place_per_timeslice(circuit, arch)
                    dt = time.perf_counter() - t0
                    cost = float(placers[name].per_timeslice_cost(circuit, part, arch))
                except Exception as exc:  # noqa: BLE001 - main.py:99-102
                    if verbose:
                        print(f"[{name}] run failed: {exc}")
                    dt, cost = math.inf, math.inf
                record(fi, pi, dt, cost)
    sums, mins = reduce_table(sums, mins, device)

    results = []
    for fi, path in enumerate(files):
        entry = {"file": path}
        for pi, name in enumerate(names):
            if name not in placers:
                continue
            cs, cn, ts, tn = sums[fi, pi]
            entry[f"{name}_time_mean"] = float(ts / tn) if tn else math.inf
            entry[f"{name}_cost_mean"] = float(cs / cn) if cn else math.inf
            entry[f"{name}_cost_min"] = float(mins[fi, pi])
        results.append(entry)
    if rank == 0:
        write_csv(cfg["out_csv"], names, results)
        if verbose:
            print(f"\nWrote all results to {cfg['out_csv']}")
        plot_results(cfg.get("plot_file"), names, results)
    return results


def write_csv(path, names, results):
    cols = ["file"]
    for n in names:
        cols += [f"{n}_time_mean", f"{n}_cost_mean"]
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=cols, extrasaction="ignore")
        w.writeheader()
        for r in results:
            w.writerow({c: r.get(c, math.inf) for c in cols})


def plot_results(path, names, results):
    """Grouped bars of mean runtime and mean cost per file (main.py:132-162); skipped without matplotlib."""
    if not path:
        return
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except ImportError:
        return
    files = [r["file"] for r in results]
    order = np.argsort(files)
    x = np.arange(len(files))
    width = 0.8 / max(1, len(names))
    fig, axes = plt.subplots(1, 2, figsize=(14, 6))
    for ax, key, title, ylabel in ((axes[0], "time_mean", "Average Runtime in s", "#time"),
                                   (axes[1], "cost_mean", "Average Cost", "#expected intercore communication")):
        for k, n in enumerate(names):
            vals = [results[i].get(f"{n}_{key}", math.inf) for i in order]
            ax.bar(x + (k - (len(names) - 1) / 2) * width, vals, width=width, label=n)
        ax.set_title(title)
        ax.set_xticks(x)
        ax.set_xticklabels([files[i] for i in order], rotation=45, ha="right")
        ax.set_ylabel(ylabel)
        ax.legend()
    fig.tight_layout()
    fig.savefig(path, dpi=150, bbox_inches="tight")
    plt.close(fig)


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="Placer comparison driver; run with --config <file>.")
    ap.add_argument("--config", help="Path to text config file")
    args = ap.parse_args(argv)
    if not args.config:
        print("No config file provided. Please run with --config <path_to_config_file>")
        return
    cfg = parse_config_file(args.config)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    if cfg.get("mode") == "simulate":
        run_simulation(cfg)
    else:
        raise RuntimeError(f"Unknown mode in config: {cfg.get('mode')}")


if __name__ == "__main__":
    main()
