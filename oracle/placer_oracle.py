"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's placement-cost path.

Nothing in the product imports this file.  It exists so that tests, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` have an independent, slow, obviously-sequential statement of
what the reference computes, on plain integer data:

* a circuit is ``gate_q`` (list of tuples of qubit indices, QASM file order; index = position in
  ``circuit.qubits``) plus ``gate_mr`` (True for Measure/Reset);
* a core graph is ``C`` nodes labelled 0..C-1 plus an edge list ``[(u, v, w)]`` in ``G.edges()``
  order (neighbour order = insertion order of that list, which is networkx's adjacency order for
  every topology the reference builds, multicore.py:6-56);
* a partition is a list of rows; a row is a list of core indices with ``None`` for "qubit not in this
  slice's dict".

Pinned against tests/golden/*.npz, which were produced by the unmodified reference placers
(oracle/gen_golden.py).  Every function names the reference lines it follows.  Floating-point
operations are done one at a time, in the reference's order, so results agree bit for bit.
"""
import heapq
import random as _random
import math
from math import inf, isfinite

NONE = None


# ------------------------------------------------------------------------------------------------
# A0  timeslices
# ------------------------------------------------------------------------------------------------

def timeslices(gate_q):
    """ASAP layering; Placer.py:14-61.

    Returns a list of slices, each a list of gate indices in the reference's within-slice order.
    ``succ`` holds Python sets of ints exactly as the reference does, because the order in which a
    set is iterated decides the order of ``next_ready`` and with it the order inside later slices.
    """
    n = len(gate_q)
    pred = [set() for _ in range(n)]
    succ = [set() for _ in range(n)]
    last = {}
    for g, qs in enumerate(gate_q):
        for q in qs:
            p = last.get(q)
            if p is not None:
                pred[g].add(p)
                succ[p].add(g)
            last[q] = g
    ready = [g for g in range(n) if not pred[g]]
    done = set()
    out = []
    while ready:
        cur, busy, nxt = [], set(), []
        for g in ready:
            qs = set(gate_q[g])
            if not (busy & qs):
                cur.append(g)
                busy |= qs
                done.add(g)
        for g in cur:
            for s in succ[g]:
                if s not in done and pred[s] <= done:
                    nxt.append(s)
        out.append(cur)
        ready = nxt
    return out


# ------------------------------------------------------------------------------------------------
# A1  distances
# ------------------------------------------------------------------------------------------------

def adjacency(C, edges):
    adj = [[] for _ in range(C)]
    for u, v, w in edges:
        adj[u].append((v, w))
        adj[v].append((u, w))
    return adj


def dijkstra(adj, src):
    """Single-source weighted distances, accumulated as ``dist[u] + w`` from the source outwards —
    what ``nx.shortest_path_length(G, s, t, weight='weight')`` evaluates (Placer.py:69).  Nodes that
    are never reached are absent from the result."""
    dist = {}
    seen = {src: 0}
    heap = [(0, 0, src)]
    tick = 1
    while heap:
        d, _, u = heapq.heappop(heap)
        if u in dist:
            continue
        dist[u] = d
        for v, w in adj[u]:
            nd = d + w
            if v in dist:
                continue
            if v not in seen or nd < seen[v]:
                seen[v] = nd
                heapq.heappush(heap, (nd, tick, v))
                tick += 1
    return dist


def hops_from(adj, src):
    dist = {src: 0}
    frontier = [src]
    while frontier:
        nxt = []
        for u in frontier:
            for v, _ in adj[u]:
                if v not in dist:
                    dist[v] = dist[u] + 1
                    nxt.append(v)
        frontier = nxt
    return dist


def distance_matrix(C, edges):
    """C x C matrix the HQA-family placers cache (HQA_placer.py:47-54): row = Dijkstra source,
    zero diagonal; unreachable pairs fall back to hop counts and then to inf (Placer.py:71-76)."""
    adj = adjacency(C, edges)
    D = [[0.0] * C for _ in range(C)]
    for i in range(C):
        di = dijkstra(adj, i)
        hi = None
        for j in range(C):
            if i == j:
                continue
            if j in di:
                v = di[j]
            else:
                if hi is None:
                    hi = hops_from(adj, i)
                v = hi.get(j, inf)
            D[i][j] = inf if v != v else float(v)
    return D


def hop_matrix(C, edges):
    """Unweighted hop counts (Qubo_placer.py:35-38)."""
    adj = adjacency(C, edges)
    H = [[0] * C for _ in range(C)]
    for i in range(C):
        hi = hops_from(adj, i)
        for j in range(C):
            H[i][j] = hi.get(j, inf)
    return H


# ------------------------------------------------------------------------------------------------
# A2  cost V3 (HQA / TRHQA / GTRHQA)
# ------------------------------------------------------------------------------------------------

def cost_v3(slices, gate_q, gate_mr, part, D):
    """HQA_placer.py:259-287 (== TRHQA_placer.py:551-585, GTRHQA_placer.py:358-383).

    ``part[i]`` is used for slice i whatever ``len(part)`` is; movement is summed for slice indices
    2..T-1 over every qubit, from the previous row's core to the current one."""
    total = 0
    for i, sl in enumerate(slices):
        row = part[i]
        for g in sl:
            qs = gate_q[g]
            if len(qs) == 2 and not gate_mr[g]:
                a, b = row[qs[0]], row[qs[1]]
                if a != b:
                    total += D[a][b]
    for i in range(2, len(slices)):
        prev, cur = part[i - 1], part[i]
        for q in range(len(cur)):
            if cur[q] != prev[q]:
                total += D[prev[q]][cur[q]]
    return total


def cost_v3_terms(slices, gate_q, gate_mr, part, D):
    """The individual addends of ``cost_v3`` in the reference's order (for error-bound checks: the
    reference's left-to-right float sum of n terms carries a relative error of up to n * 2**-53)."""
    terms = []
    for i, sl in enumerate(slices):
        row = part[i]
        for g in sl:
            qs = gate_q[g]
            if len(qs) == 2 and not gate_mr[g] and row[qs[0]] != row[qs[1]]:
                terms.append(D[row[qs[0]]][row[qs[1]]])
    for i in range(2, len(slices)):
        prev, cur = part[i - 1], part[i]
        for q in range(len(cur)):
            if cur[q] != prev[q]:
                terms.append(D[prev[q]][cur[q]])
    return terms


# ------------------------------------------------------------------------------------------------
# A3 / A4  cost V1, V2, V4
# ------------------------------------------------------------------------------------------------

class CostError(Exception):
    """The reference would raise here (KeyError / TypeError); main.py:99-102 records inf."""


def cost_v2(slices, gate_q, gate_mr, part, D):
    """BFS_placer.py:202-239 == Random_placer.py:178-213.  Rows hold None for absent qubits.

    A missing qubit of a counted two-qubit gate is a KeyError in the reference
    (``partition[i][qubits[0]]``); a qubit that was placed at its previous active slice and is absent
    now makes ``get_weighted_distance(last, None)`` return a dict and the ``+=`` a TypeError."""
    total = 0
    last = {}
    for i, sl in enumerate(slices):
        row = part[i]
        for g in sl:
            qs = gate_q[g]
            if len(qs) == 2 and not gate_mr[g]:
                a, b = row[qs[0]], row[qs[1]]
                if a is None or b is None:
                    raise CostError("KeyError")
                if a != b:
                    total += D[a][b]
            for q in qs:
                cur = row[q]
                lq = last.get(q)
                if lq is not None and cur != lq:
                    if cur is None:
                        raise CostError("TypeError")
                    total += D[lq][cur]
                last[q] = cur
    return total


def cost_v1(slices, gate_q, gate_mr, part, D):
    """BasePlacer.per_timeslice_cost, Placer.py:78-120: ``.get`` everywhere, slices beyond
    ``len(part)`` are skipped, a crossing term needs both qubits present, ``last`` only moves to a
    core that is not None."""
    total = 0
    last = {}
    for i, sl in enumerate(slices):
        if i >= len(part):
            break
        row = part[i]
        for g in sl:
            qs = gate_q[g]
            if len(qs) == 2 and not gate_mr[g]:
                a, b = row[qs[0]], row[qs[1]]
                if a != b and a is not None and b is not None:
                    total += D[a][b]
            for q in qs:
                cur = row[q]
                lq = last.get(q)
                if lq is not None and cur != lq:
                    if cur is None:
                        raise CostError("TypeError")
                    total += D[lq][cur]
                if cur is not None:
                    last[q] = cur
    return total


def core_activity(slices, gate_q, gate_mr, part, C):
    """hqa_placer.per_core_intercore_activity, HQA_placer.py:288-321 (same body in
    TRHQA_placer.py:586-619): how many inter-core events every core takes part in.  A counted gate whose
    qubits sit on different cores counts once for each of the two cores; for rows 1..T-1 (from row 1,
    not 2 as in the cost) every qubit whose core changed counts once for the old and once for the new
    core.  ``part`` rows are dense (a core per qubit)."""
    out = [0] * C
    T = len(slices)
    for t in range(T):
        row = part[t]
        for g in slices[t]:
            qs = gate_q[g]
            if len(qs) == 2 and not gate_mr[g]:
                c1, c2 = row[qs[0]], row[qs[1]]
                if c1 != c2:
                    out[c1] += 1
                    out[c2] += 1
    for t in range(1, T):
        prev, cur = part[t - 1], part[t]
        for q in range(len(cur)):
            if prev[q] != cur[q]:
                out[prev[q]] += 1
                out[cur[q]] += 1
    return out


def comm_matrix(slices, gate_q, gate_mr, part, C):
    """qubo_placer.intercorecommatrix, Qubo_placer.py:208-245 (the counts; the heat-map it saves is not
    restated): symmetric C x C event counts of the gate-ordered walk.  Rows hold None for absent qubits;
    an absent qubit of a counted gate is a KeyError there, a qubit seen before on a core and absent now
    makes ``list.index(None)`` raise ValueError."""
    M = [[0] * C for _ in range(C)]
    last = {}
    for i, sl in enumerate(slices):
        row = part[i]
        for g in sl:
            qs = gate_q[g]
            if len(qs) == 2 and not gate_mr[g]:
                a, b = row[qs[0]], row[qs[1]]
                if a is None or b is None:
                    raise CostError("KeyError")
                if a != b:
                    M[a][b] += 1
                    M[b][a] += 1
            for q in qs:
                cur = row[q]
                lq = last.get(q)
                if lq is not None and cur != lq:
                    if cur is None:
                        raise CostError("ValueError")
                    M[cur][lq] += 1
                    M[lq][cur] += 1
                last[q] = cur
    return M


def cost_v4(slices, gate_q, gate_mr, part, H):
    """Qubo_placer.py:177-207: V2 with hop counts, movement looked up as (current, last)."""
    total = 0
    last = {}
    for i, sl in enumerate(slices):
        row = part[i]
        for g in sl:
            qs = gate_q[g]
            if len(qs) == 2 and not gate_mr[g]:
                a, b = row[qs[0]], row[qs[1]]
                if a is None or b is None:
                    raise CostError("KeyError")
                if a != b:
                    total += H[a][b]
            for q in qs:
                cur = row[q]
                lq = last.get(q)
                if lq is not None and cur != lq:
                    if cur is None:
                        raise CostError("TypeError")
                    total += H[cur][lq]
                last[q] = cur
    return total


# ------------------------------------------------------------------------------------------------
# LSAP (what scipy.optimize.linear_sum_assignment computes)
# ------------------------------------------------------------------------------------------------

def lsap(cost):
    """Rectangular linear sum assignment by shortest augmenting paths (Crouse, "On implementing 2D
    rectangular assignment algorithms", IEEE TAES 52(4), 2016) as scipy implements it, including its
    tie rule (among equal tentative distances prefer an unassigned column; columns are scanned from
    a list initialised in descending order).  Returns (rows, cols) with rows ascending."""
    nr = len(cost)
    nc = len(cost[0]) if nr else 0
    if nr == 0 or nc == 0:
        return [], []
    transpose = nc < nr
    if transpose:
        cost = [[cost[i][j] for i in range(nr)] for j in range(nc)]
        nr, nc = nc, nr
    u = [0.0] * nr
    v = [0.0] * nc
    col4row = [-1] * nr
    row4col = [-1] * nc
    for cur in range(nr):
        spc = [inf] * nc
        path = [-1] * nc
        SR = [False] * nr
        SC = [False] * nc
        remaining = [nc - 1 - k for k in range(nc)]
        nrem = nc
        minval = 0.0
        i = cur
        sink = -1
        while sink == -1:
            index = -1
            lowest = inf
            SR[i] = True
            ci, ui = cost[i], u[i]
            for it in range(nrem):
                j = remaining[it]
                r = minval + ci[j] - ui - v[j]
                if r < spc[j]:
                    path[j] = i
                    spc[j] = r
                if spc[j] < lowest or (spc[j] == lowest and row4col[j] == -1):
                    lowest = spc[j]
                    index = it
            minval = lowest
            if minval == inf:
                raise ValueError("cost matrix is infeasible")
            j = remaining[index]
            if row4col[j] == -1:
                sink = j
            else:
                i = row4col[j]
            SC[j] = True
            nrem -= 1
            remaining[index] = remaining[nrem]
        u[cur] += minval
        for r_ in range(nr):
            if SR[r_] and r_ != cur:
                u[r_] += minval - spc[col4row[r_]]
        for j in range(nc):
            if SC[j]:
                v[j] -= minval - spc[j]
        j = sink
        while True:
            i = path[j]
            row4col[j] = i
            col4row[i], j = j, col4row[i]
            if i == cur:
                break
    if transpose:
        order = sorted(range(nr), key=lambda k: col4row[k])
        return [col4row[k] for k in order], order
    return list(range(nr)), col4row


# ------------------------------------------------------------------------------------------------
# A5 / A6 / A7  cost-matrix entries
# ------------------------------------------------------------------------------------------------

def _pymax(a, b):
    """Python's max(a, b): b only if b > a."""
    return b if b > a else a


def attraction(q1, q2, core, a, slices, gate_q, t):
    """Look-ahead bonus of HQA_placer.py:223-237: slices t+1..t+3, weight 2^-dt, every two-qubit
    command that touches q1 or q2, 0.9 if its other qubit already sits on ``core`` else 0.3."""
    att = 0.0
    T = len(slices)
    for tt in range(t + 1, min(T, min(t + 4, T))):
        w = 2 ** -(tt - t)
        for g in slices[tt]:
            qs = gate_q[g]
            if len(qs) == 2:
                f1, f2 = qs
                if f1 == q1 or f1 == q2 or f2 == q1 or f2 == q2:
                    other = f2 if (f1 == q1 or f1 == q2) else f1
                    if a[other] == core:
                        att += w * 0.9
                    else:
                        att += w * 0.3
    return att


def entry_hqa(q1, q2, core, a, slices, gate_q, t):
    """HQA_placer.py:208-241 then the outer clamp of :122."""
    m = 0
    if a[q1] != core:
        m += 1
    if a[q2] != core:
        m += 1
    total = m - attraction(q1, q2, core, a, slices, gate_q, t)
    return _pymax(0.1, _pymax(0.1, total))


def entry_trhqa(q1, q2, core, a, slices, gate_q, t, D):
    """TRHQA_placer.py:139-166 with :471-532.  ``here`` skips a qubit whose current core label is 0
    (``if q1_core and ...``); the helper does not.  The final value goes through
    ``here - (here - T)`` before the clamp."""
    mv = 0
    if a[q1] != core:
        mv += D[a[q1]][core]
    if a[q2] != core:
        mv += D[a[q2]][core]
    T_ = _pymax(0.1, mv - attraction(q1, q2, core, a, slices, gate_q, t))
    here = 0.0
    if a[q1] and a[q1] != core:
        here += D[a[q1]][core]
    if a[q2] and a[q2] != core:
        here += D[a[q2]][core]
    att = here - T_
    return _pymax(0.1, here - att)


def entry_gtrhqa(q1, q2, core, a, D):
    """GTRHQA_placer.py:99-112: distances only, label-0 quirk included, no clamp."""
    c = 0
    if a[q1] and a[q1] != core:
        c += D[a[q1]][core]
    if a[q2] and a[q2] != core:
        c += D[a[q2]][core]
    return float(c)


# ------------------------------------------------------------------------------------------------
# A8  HQA / TRHQA control loop
# ------------------------------------------------------------------------------------------------

def _make_space(a, occ, cap, C, D):
    """HQA_placer.py:180-204."""
    for core in range(C):
        if occ[core] > cap - 2:
            inside = [q for q in range(len(a)) if a[q] == core]
            for q in inside[:2]:
                targets = [(tc, D[core][tc], occ[tc]) for tc in range(C)
                           if occ[tc] < cap - 1 and tc != core]
                targets.sort(key=lambda x: (x[1], x[2]))
                if targets:
                    tc = targets[0][0]
                    a[q] = tc
                    occ[core] -= 1
                    occ[tc] += 1
                if occ[core] <= cap - 2:
                    break


def place_hqa_family(slices, gate_q, gate_mr, Q, C, cap, D, kind, record=None, init=None):
    """``kind`` 'hqa': HQA_placer.py:20-176; 'trhqa': TRHQA_placer.py:22-228.  Returns T rows.
    ``record`` (a list) receives (t, ops, avail, matrix, live assignment) for every LSAP call."""
    if Q > C * cap:
        raise ValueError("Not enough space")
    T = len(slices)
    part = [list(init) if init is not None else [int(i / cap) for i in range(Q)]]
    for t in range(1, T):
        prev = part[t - 1]
        a = list(prev)
        ops = []
        for g in slices[t]:
            qs = gate_q[g]
            if len(qs) == 2 and not gate_mr[g] and prev[qs[0]] != prev[qs[1]]:
                ops.append((g, qs[0], qs[1]))
        if not ops:
            part.append(a)
            continue
        occ = [0] * C
        for c in a:
            occ[c] += 1
        remaining = list(ops)
        limit = len(ops) + 1
        it = 0
        while remaining and it < limit:
            it += 1
            avail = [c for c in range(C) if occ[c] <= cap - 2]
            if not avail:
                _make_space(a, occ, cap, C, D)
                continue
            if kind == "hqa":
                M = [[entry_hqa(q1, q2, c, a, slices, gate_q, t) for c in avail] for _, q1, q2 in remaining]
            else:
                M = [[entry_trhqa(q1, q2, c, a, slices, gate_q, t, D) for c in avail] for _, q1, q2 in remaining]
            if record is not None:
                record.append((t, list(remaining), list(avail), [list(r) for r in M], list(a)))
            try:
                rows, cols = lsap(M)
            except ValueError:
                break
            taken = set()
            for r, cc in zip(rows, cols):
                if M[r][cc] != inf:
                    _, q1, q2 = remaining[r]
                    core = avail[cc]
                    occ[a[q1]] -= 1
                    occ[a[q2]] -= 1
                    a[q1] = core
                    a[q2] = core
                    occ[core] += 2
                    taken.add(r)
            remaining = [op for k, op in enumerate(remaining) if k not in taken]
        part.append(a)
    return part


# ------------------------------------------------------------------------------------------------
# A7  GTRHQA
# ------------------------------------------------------------------------------------------------

def _argmin_first(vals):
    best, bi = None, -1
    for i, v in enumerate(vals):
        if best is None or v < best:
            best, bi = v, i
    return bi


def gtrhqa_rounds(M):
    """GTRHQA_placer.py:128-254 on one cost matrix (list of rows): (op, core column) pairs in the
    order the reference applies them."""
    HUGE = 1e12
    nops, nav = len(M), len(M[0])
    core_free = [True] * nav
    op_left = [True] * nops
    core_of_op = [-1] * nops
    out = []
    rounds = 0
    while any(op_left) and any(core_free) and rounds < min(nav, nops) + 2:
        rounds += 1
        cols = [j for j in range(nav) if core_free[j]]
        rows = [k for k in range(nops) if op_left[k]]
        if not any(isfinite(M[k][j]) for k in rows for j in cols):
            break
        choice = []
        for k in rows:
            masked = [M[k][j] if isfinite(M[k][j]) else HUGE for j in cols]
            r = _argmin_first(masked)
            choice.append((cols[r], masked[r], k))
        for j in sorted(set(c for c, _, _ in choice)):
            group = [(v, k) for c, v, k in choice if c == j]
            k = group[_argmin_first([v for v, _ in group])][1]
            if core_of_op[k] != -1 or not core_free[j] or not isfinite(M[k][j]):
                continue
            core_of_op[k] = j
            core_free[j] = False
            op_left[k] = False
            out.append((k, j))
    left_cores = [j for j in range(nav) if core_free[j]]
    for k in range(nops):
        if core_of_op[k] != -1:
            continue
        if not left_cores:
            break
        vals = [M[k][j] for j in left_cores]
        if not any(isfinite(v) for v in vals):
            continue
        r = _argmin_first([v if isfinite(v) else HUGE for v in vals])
        j = left_cores[r]
        core_of_op[k] = j
        out.append((k, j))
        left_cores.remove(j)
    return out


def place_gtrhqa(slices, gate_q, gate_mr, Q, C, cap, D, record=None, init=None):
    """GTRHQA_placer.py:20-267.  Returns T+1 rows: the loop starts at slice 0, slice t reads row
    t-1 (row 0 for t = 0) and appends row t+1."""
    if Q > C * cap:
        raise ValueError("Not enough space")
    T = len(slices)
    init = list(init) if init is not None else [int(i / cap) for i in range(Q)]
    part = [init]
    for t in range(T):
        prev = part[t - 1] if t > 0 else init
        a = list(prev)
        ops = []
        for g in slices[t]:
            qs = gate_q[g]
            if len(qs) == 2 and not gate_mr[g] and prev[qs[0]] != prev[qs[1]]:
                ops.append((g, qs[0], qs[1]))
        if not ops:
            part.append(a)
            continue
        occ = [0] * C
        for c in a:
            occ[c] += 1
        avail = [c for c in range(C) if occ[c] <= cap - 2]
        if not avail:
            part.append(a)
            continue
        M = [[entry_gtrhqa(q1, q2, c, a, D) for c in avail] for _, q1, q2 in ops]
        if record is not None:
            record.append((t, list(ops), list(avail), [list(r) for r in M], list(a)))
        for k, j in gtrhqa_rounds(M):
            _, q1, q2 = ops[k]
            a[q1] = avail[j]
            a[q2] = avail[j]
        part.append(a)
    return part


# ------------------------------------------------------------------------------------------------
# BFS / Random placement (host RNG driven)
# ------------------------------------------------------------------------------------------------

def place_random(slices, gate_q, Q, C, cap, qubit_key=None, rng=_random):
    """Random_placer.py:25-59.  ``qubit_key[q]`` is the hashable the reference would put in its set
    (iteration order of ``list(set(...))`` decides who draws first); default: the index itself."""
    key = qubit_key if qubit_key is not None else list(range(Q))
    back = {id(k) if not isinstance(k, int) else k: i for i, k in enumerate(key)}
    part = []
    for sl in slices:
        row = [None] * Q
        free = [cap] * C
        members = set()
        for g in sl:
            members.update(key[q] for q in gate_q[g])
        for k in list(members):
            q = back[k if isinstance(k, int) else id(k)]
            if row[q] is None:
                open_ = [c for c in range(C) if free[c] > 0]
                if open_:
                    c = rng.choice(open_)
                    row[q] = c
                    free[c] -= 1
        part.append(row)
    return part


def place_bfs(slices, gate_q, Q, C, cap, edges, D, qubit_key=None, rng=_random):
    """BFS_placer.py:101-200 with the neighbour ordering of :7-18."""
    key = qubit_key if qubit_key is not None else list(range(Q))
    back = {id(k) if not isinstance(k, int) else k: i for i, k in enumerate(key)}
    adj = adjacency(C, edges)

    def nbrs(c):
        ns = [(v, D[c][v]) for v, _ in adj[c]]
        ns.sort(key=lambda x: x[1])
        return [v for v, _ in ns]

    part = []
    for sl in slices:
        row = [None] * Q
        free = [cap] * C
        members = set()
        for g in sl:
            members.update(key[q] for q in gate_q[g])
        members = [back[k if isinstance(k, int) else id(k)] for k in list(members)]
        placed = set()
        for g in sl:
            qs = gate_q[g]
            if len(qs) == 2:
                q1, q2 = qs
                if q1 in placed or q2 in placed:
                    continue
                found = False
                for c in range(C):
                    if free[c] >= 2:
                        row[q1] = c
                        row[q2] = c
                        free[c] -= 2
                        placed.update((q1, q2))
                        found = True
                        break
                if not found:
                    pairs = []
                    for c1 in range(C):
                        if free[c1] >= 1:
                            for c2 in nbrs(c1):
                                if free[c2] >= 1:
                                    pairs.append((c1, c2, D[c1][c2]))
                    pairs.sort(key=lambda x: x[2])
                    for c1, c2, _ in pairs:
                        if free[c1] >= 1 and free[c2] >= 1:
                            row[q1] = c1
                            row[q2] = c2
                            free[c1] -= 1
                            free[c2] -= 1
                            placed.update((q1, q2))
                            break
            if len(qs) == 1:
                q = qs[0]
                if q not in placed:
                    open_ = [c for c in range(C) if free[c] > 0]
                    if open_:
                        c = rng.choice(open_)
                        row[q] = c
                        free[c] -= 1
                        placed.add(q)
        waiting = [q for q in members if q not in placed]
        order = [(c, free[c]) for c in range(C) if free[c] > 0]
        order.sort(key=lambda x: x[1], reverse=True)
        queue = [c for c, _ in order]
        visited = set(queue)
        while queue and waiting:
            c = queue.pop(0)
            q = waiting.pop(0)
            row[q] = c
            free[c] -= 1
            for v in nbrs(c):
                if v not in visited and free[v] > 0:
                    visited.add(v)
                    queue.append(v)
        part.append(row)
    return part


# ------------------------------------------------------------------------------------------------
# A9 / A10  QUBO model, decode/repair
# ------------------------------------------------------------------------------------------------

def qubo_slices(slices, gate_q, qubit_name):
    """Per-slice active-qubit lists sorted by ``str(qubit)`` and interaction edges
    (Qubo_placer.py:46-71).  ``qubit_name[q]`` is ``str`` of qubit q."""
    act, inter = [], []
    for sl in slices:
        qs = sorted({q for g in sl for q in gate_q[g]}, key=lambda q: qubit_name[q])
        pos = {q: i for i, q in enumerate(qs)}
        es = set()
        for g in sl:
            gq = gate_q[g]
            for x in range(len(gq)):
                for y in range(x + 1, len(gq)):
                    ia, ib = pos[gq[x]], pos[gq[y]]
                    if ia != ib:
                        es.add((min(ia, ib), max(ia, ib)))
        act.append(qs)
        inter.append(sorted(es))
    return act, inter


def qubo_energy(act, inter, C, cap, H, x, y):
    """Value of the objective of Qubo_placer.py:84-133 for one 0/1 state.

    ``x[t][j][i]``: qubit i of slice t on core j; ``y[t][j][p]``: slack p of core j.  Weights:
    assignment 4, capacity 4, Laplacian 10, movement 10 (Qubo_placer.py:29-32).  All terms are
    integers when H holds hop counts."""
    E = 0
    nT = len(act)
    for t in range(nT):
        n = len(act[t])
        for i in range(n):
            s = sum(x[t][j][i] for j in range(C)) - 1
            E += 4 * s * s
        for j in range(C):
            s = sum(x[t][j]) - sum(y[t][j])
            E += 4 * s * s
        deg = [0] * n
        for (ia, ib) in inter[t]:
            deg[ia] += 1
            deg[ib] += 1
        for j in range(C):
            lap = sum(deg[i] * x[t][j][i] for i in range(n))
            lap += 2 * sum(-1 * x[t][j][ia] * x[t][j][ib] for (ia, ib) in inter[t])
            E += 10 * lap
        if t < nT - 1:
            nxt = {q: i for i, q in enumerate(act[t + 1])}
            for i, q in enumerate(act[t]):
                if q in nxt:
                    i2 = nxt[q]
                    for j in range(C):
                        if x[t][j][i]:
                            for l in range(C):
                                if x[t + 1][l][i2]:
                                    E += 10 * H[j][l]
    return E


def qubo_decode(act, C, cap, x, Q):
    """Qubo_placer.py:146-172: first core whose bit is 1 wins unless it is full; unplaced qubits
    then go to the least-loaded core (first minimum).  Returns rows with None for inactive qubits."""
    part = []
    for t, qs in enumerate(act):
        row = [None] * Q
        load = [0] * C
        pending = []
        for i, q in enumerate(qs):
            chosen = None
            for j in range(C):
                if x[t][j][i] == 1:
                    chosen = j
                    break
            if chosen is not None and load[chosen] < cap:
                row[q] = chosen
                load[chosen] += 1
            else:
                pending.append(q)
        for q in pending:
            j = min(range(C), key=lambda jj: load[jj])
            row[q] = j
            load[j] += 1
        part.append(row)
    return part


def qubo_cpu_anneal(energy_of, N, sweeps, beta_hot, beta_cold, replicas, seed):
    """TEST INFRASTRUCTURE: a plain sequential single-flip Metropolis annealer over N binary variables
    with a geometric inverse-temperature schedule, the textbook form of what
    neal.SimulatedAnnealingSampler().sample does at Qubo_placer.py:139-140.  ``energy_of(state)`` is the
    objective; its pairwise coefficients are first read off by evaluating it on basis states (so nothing
    here depends on any matrix the product builds).  Returns the best energy found over all replicas."""
    import numpy as np
    zero = np.zeros(N, np.int8)
    e0 = energy_of(zero)
    lin = np.zeros(N, np.int64)
    for i in range(N):
        s = zero.copy()
        s[i] = 1
        lin[i] = energy_of(s) - e0
    W = np.zeros((N, N), np.int64)
    for i in range(N):
        for j in range(i + 1, N):
            s = zero.copy()
            s[i] = s[j] = 1
            W[i, j] = W[j, i] = energy_of(s) - e0 - lin[i] - lin[j]
    rng = np.random.default_rng(seed)
    best = None
    for _ in range(replicas):
        s = rng.integers(0, 2, size=N).astype(np.int64)
        field = W @ s
        e = e0 + int(lin @ s) + int(s @ field) // 2
        for k in range(sweeps):
            beta = beta_hot * (beta_cold / beta_hot) ** (k / (sweeps - 1) if sweeps > 1 else 0.0)
            u = rng.random(N)
            for i in range(N):
                d = 1 - 2 * int(s[i])
                dE = d * (int(lin[i]) + int(field[i]))
                if dE <= 0 or u[i] < math.exp(-beta * dE):
                    s[i] += d
                    field += d * W[:, i]
                    e += dE
        assert e == energy_of(s.astype(np.int8))
        best = e if best is None else min(best, e)
    return best
