"""TEST INFRASTRUCTURE ONLY -- the QUBO placer's objective as a sparse matrix, and the CPU annealer on it.

``sparse_model`` writes the objective of Qubo_placer.py:84-133 out term by term for the compact variable order
x[t][j][i] (core-major inside a slice), then y[t][j][p]:
    4 (sum_j x[t,j,i] - 1)^2           -> linear -4 per x, +8 between two cores of one qubit, constant 4
    4 (sum_i x[t,j,i] - sum_p y[t,j,p])^2 -> linear +4 per x and y, +8 x-x and y-y, -8 x-y inside a core
    10 (sum_i deg_i x - 2 sum_edges x x)  -> linear 10 deg_i, -20 between interacting qubits on one core
    10 hops[j][l] x[t,j,i] x[t+1,l,i']    -> between the same qubit in consecutive slices
and returns (constant, linear int32 [N], CSR (rowptr int64, col int32, val int32) with both directions, index maps).
Nothing here reads anything the product builds; tests compare it with placer_oracle.qubo_energy state by state.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libqubo_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        L = C.CDLL(_SO)
        L.oracle_qubo_anneal.restype = C.c_int64
        L.oracle_qubo_anneal.argtypes = [C.c_int32] + [C.c_void_p] * 4 + [C.c_int32, C.c_double, C.c_double, C.c_uint64,
                                                                           C.c_void_p, C.POINTER(C.c_int64)]
        L.oracle_qubo_energy.restype = C.c_int64
        L.oracle_qubo_energy.argtypes = [C.c_int32] + [C.c_void_p] * 5
        _lib = L
    return _lib


class SparseModel:
    def __init__(self, act, inter, n_cores, cap, H):
        Cn = int(n_cores)
        H = np.asarray(H, np.int64)
        T = len(act)
        n = [len(a) for a in act]
        base = np.concatenate([[0], np.cumsum([Cn * (k + cap) for k in n])]).astype(np.int64)
        self.T, self.C, self.cap, self.n, self.base = T, Cn, cap, n, base
        self.N = int(base[-1])
        self.const = 4 * sum(n)
        lin = np.zeros(self.N, np.int64)
        rows, cols, vals = [], [], []

        def add(a, b, v):  # a, b: index arrays of equal shape, one coupling v per pair (stored once here)
            rows.append(np.asarray(a, np.int64).ravel())
            cols.append(np.asarray(b, np.int64).ravel())
            vals.append(np.broadcast_to(np.asarray(v, np.int64), np.shape(a)).ravel())

        for t in range(T):
            nt = n[t]
            X = base[t] + np.arange(Cn)[:, None] * nt + np.arange(nt)[None, :]            # [C, n]
            Y = base[t] + Cn * nt + np.arange(Cn)[:, None] * cap + np.arange(cap)[None, :]  # [C, cap]
            deg = np.zeros(nt, np.int64)
            for ia, ib in inter[t]:
                deg[ia] += 1
                deg[ib] += 1
            lin[X] += -4 + 4 + 10 * deg[None, :]
            lin[Y] += 4
            jj, ll = np.triu_indices(Cn, 1)
            if nt:
                add(X[jj, :], X[ll, :], 8)                       # one qubit on two cores
                ii, kk = np.triu_indices(nt, 1)
                add(X[:, ii], X[:, kk], 8)                       # two qubits on one core
                add(np.repeat(X[:, :, None], cap, 2), np.repeat(Y[:, None, :], nt, 1), -8)
                for ia, ib in inter[t]:
                    add(X[:, ia], X[:, ib], -20)
            pp, qq = np.triu_indices(cap, 1)
            add(Y[:, pp], Y[:, qq], 8)
            if t + 1 < T:
                nxt = {q: i for i, q in enumerate(act[t + 1])}
                nt1 = n[t + 1]
                X1 = base[t + 1] + np.arange(Cn)[:, None] * nt1 + np.arange(nt1)[None, :]
                for i, q in enumerate(act[t]):
                    if q in nxt:
                        i2 = nxt[q]
                        J, Lc = np.nonzero(H)
                        add(X[J, i], X1[Lc, i2], 10 * H[J, Lc])
        r = np.concatenate(rows) if rows else np.zeros(0, np.int64)
        c = np.concatenate(cols) if cols else np.zeros(0, np.int64)
        v = np.concatenate(vals) if vals else np.zeros(0, np.int64)
        # pairs touched by two terms (interacting qubits: +8 and -20) are merged; both directions stored
        key = r * self.N + c
        order = np.argsort(key, kind="stable")
        key, v = key[order], v[order]
        uniq, start = np.unique(key, return_index=True)
        v = np.add.reduceat(v, start) if len(v) else v
        r, c = uniq // self.N, uniq % self.N
        keep = v != 0
        r, c, v = r[keep], c[keep], v[keep]
        R2, C2, V2 = np.concatenate([r, c]), np.concatenate([c, r]), np.concatenate([v, v])
        order = np.lexsort((C2, R2))
        R2, C2, V2 = R2[order], C2[order], V2[order]
        self.rowptr = np.zeros(self.N + 1, np.int64)
        np.add.at(self.rowptr, R2 + 1, 1)
        self.rowptr = np.cumsum(self.rowptr).astype(np.int64)
        self.col = C2.astype(np.int32)
        self.val = V2.astype(np.int32)
        self.lin = lin.astype(np.int32)

    def var_x(self, t, j, i):
        return int(self.base[t] + j * self.n[t] + i)

    def var_y(self, t, j, p):
        return int(self.base[t] + self.C * self.n[t] + j * self.cap + p)

    def energy(self, state):
        s = np.ascontiguousarray(state, np.int8)
        return self.const + int(lib().oracle_qubo_energy(self.N, self.rowptr.ctypes.data, self.col.ctypes.data,
                                                         self.val.ctypes.data, self.lin.ctypes.data, s.ctypes.data))

    def anneal(self, state, sweeps, beta_hot, beta_cold, seed):
        """Anneals ``state`` (int8 [N], changed in place); returns (summed energy change, accepted flips)."""
        assert state.dtype == np.int8 and state.flags["C_CONTIGUOUS"] and state.shape == (self.N,)
        acc = C.c_int64(0)
        d = lib().oracle_qubo_anneal(self.N, self.rowptr.ctypes.data, self.col.ctypes.data, self.val.ctypes.data,
                                     self.lin.ctypes.data, int(sweeps), float(beta_hot), float(beta_cold), int(seed),
                                     state.ctypes.data, C.byref(acc))
        if d == -2 ** 63:
            raise MemoryError("oracle_qubo_anneal")
        return int(d), int(acc.value)
