"""TEST INFRASTRUCTURE ONLY — two more CPU forms of the HQA-family cost (HQA_placer.py:259-287), timed by
the ``cpu_baseline`` leg of bench.py beside the C restatement:

* ``cost_v3_reference_structure``: the reference's own structure — every term is a fresh
  ``nx.shortest_path_length(G, a, b, weight='weight')`` (BasePlacer.get_weighted_distance, Placer.py:63-76),
  no distance matrix anywhere.  One Dijkstra per term makes a whole partition of the benchmark take minutes,
  so it evaluates the first ``max_slices`` slices only and returns the partial sum with the terms it added.
* ``cost_v3_numpy``: what a NumPy user would write on a precomputed distance matrix: two fancy-indexed
  gathers and a sum per partition.

Neither is part of the product."""
import numpy as np


def cost_v3_reference_structure(network, nodes, slices, gate_q, gate_mr, part, max_slices):
    """Partial HQA-family cost over slices 0..max_slices-1 (crossing gates of those slices, movement into
    slices 2..max_slices-1), each distance from a fresh weighted shortest-path search, summed left to right
    as HQA_placer.py:259-287 sums them.  ``part[i][q]`` is a core index into ``nodes``.
    Returns (sum, number of searches)."""
    import networkx as nx
    total, searches = 0, 0
    n = min(max_slices, len(slices))

    def dist(a, b):
        try:
            return nx.shortest_path_length(network, nodes[a], nodes[b], weight="weight")
        except (nx.NetworkXNoPath, KeyError):
            try:
                return nx.shortest_path_length(network, nodes[a], nodes[b])
            except Exception:  # noqa: BLE001 - Placer.py:75-76
                return float("inf")

    for i in range(n):
        row = part[i]
        for g in slices[i]:
            qs = gate_q[g]
            if len(qs) == 2 and not gate_mr[g] and row[qs[0]] != row[qs[1]]:
                total += dist(row[qs[0]], row[qs[1]])
                searches += 1
    for i in range(2, n):
        prev, cur = part[i - 1], part[i]
        for q in range(len(cur)):
            if cur[q] != prev[q]:
                total += dist(prev[q], cur[q])
                searches += 1
    return total, searches


class NumpyCost:
    """HQA-family cost of dense partitions with vectorised gathers on a precomputed matrix (zero diagonal)."""

    def __init__(self, slices, gate_q, gate_mr, D):
        t_idx, u, v = [], [], []
        for t, layer in enumerate(slices):
            for g in layer:
                qs = gate_q[g]
                if len(qs) == 2 and not gate_mr[g]:
                    t_idx.append(t)
                    u.append(qs[0])
                    v.append(qs[1])
        self.t = np.asarray(t_idx, np.int64)
        self.u = np.asarray(u, np.int64)
        self.v = np.asarray(v, np.int64)
        self.T = len(slices)
        self.D = np.array(D, np.float64)
        np.fill_diagonal(self.D, 0.0)

    def cost(self, A):
        """A: uint8 [Tp, Q].  Pairwise (not left-to-right) summation: agrees with the sequential sum to
        rounding, not bit for bit."""
        A = np.asarray(A)
        gates = self.D[A[self.t, self.u], A[self.t, self.v]].sum()
        moves = self.D[A[1:self.T - 1], A[2:self.T]].sum() if self.T > 2 else 0.0
        return float(gates + moves)
