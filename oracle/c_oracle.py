"""TEST INFRASTRUCTURE ONLY — ctypes loader for oracle/cost_oracle.c (built by oracle/Makefile)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcost_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        L = C.CDLL(_SO)
        L.oracle_cost_v3.restype = C.c_double
        L.oracle_cost_v3.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_int]
        L.oracle_cost_v2.restype = C.c_double
        L.oracle_cost_v2.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_int,
                                                                                    C.c_int, C.c_void_p]
        _lib = L
    return _lib


class GateTables:
    """Gates in slice order as flat arrays."""

    def __init__(self, slices, gate_q, gate_mr):
        ptr, q0, q1, cnt = [0], [], [], []
        for layer in slices:
            for g in layer:
                qs = gate_q[g]
                q0.append(qs[0])
                q1.append(qs[1] if len(qs) > 1 else -1)
                cnt.append(1 if (len(qs) == 2 and not gate_mr[g]) else 0)
                if len(qs) > 2:
                    raise ValueError("the C checker handles 1- and 2-qubit gates only")
            ptr.append(len(q0))
        self.T = len(slices)
        self.ptr = np.asarray(ptr, np.int32)
        self.q0 = np.asarray(q0, np.int32)
        self.q1 = np.asarray(q1, np.int32)
        self.cnt = np.asarray(cnt, np.uint8)


def cost_v3(gt: GateTables, Q, A: np.ndarray, D: np.ndarray):
    A = np.ascontiguousarray(A, np.uint8)
    D = np.ascontiguousarray(D, np.float64)
    return lib().oracle_cost_v3(Q, gt.T, gt.ptr.ctypes.data, gt.q0.ctypes.data, gt.q1.ctypes.data, gt.cnt.ctypes.data,
                                A.ctypes.data, A.shape[1], D.ctypes.data, D.shape[0])


def cost_v2(gt: GateTables, Q, A: np.ndarray, D: np.ndarray, mode=0):
    A = np.ascontiguousarray(A, np.uint8)
    D = np.ascontiguousarray(D, np.float64)
    last = np.zeros(Q, np.int32)
    return lib().oracle_cost_v2(Q, gt.T, A.shape[0], gt.ptr.ctypes.data, gt.q0.ctypes.data, gt.q1.ctypes.data,
                                gt.cnt.ctypes.data, A.ctypes.data, A.shape[1], D.ctypes.data, D.shape[0], mode,
                                last.ctypes.data)
