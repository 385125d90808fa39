"""ctypes binding of libqplace.so (declared in include/qplace.h).

There is no fallback: if the shared library is missing or lacks a symbol, importing this module
raises, and so does every call that needs it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# QPLACE_LIB names another build of the same library (development: instrumented kernels, launch knobs)
LIB_PATH = os.environ.get("QPLACE_LIB") or os.path.join(_HERE, "libqplace.so")

QP_OK = 0
QP_DT_F64, QP_DT_U16 = 0, 1
QP_V2, QP_V1, QP_V4 = 0, 1, 2
QP_CM_HQA, QP_CM_TRHQA, QP_CM_GTRHQA = 0, 1, 2
QP_ABSENT = 0xFF
QP_PLACE_RANDOM, QP_PLACE_BFS = 0, 1
QP_SCHED_CHUNK, QP_SCHED_TAIL = 64, 512


class QpCircuit(C.Structure):
    _fields_ = [("Q", C.c_int32), ("T", C.c_int32), ("nE", C.c_int32), ("nV", C.c_int32),
                ("max_slice_edges", C.c_int32), ("group_ldq", C.c_int32),
                ("slice_edge_ptr", C.c_void_p), ("edges", C.c_void_p), ("lookahead", C.c_void_p),
                ("slice_visit_ptr", C.c_void_p), ("visits", C.c_void_p), ("edges_sched", C.c_void_p), ("sched_ptr", C.c_void_p),
                ("group_sched", C.c_void_p), ("group_ptr", C.c_void_p),
                ("visit_mask", C.c_void_p), ("visit_mask_ldq", C.c_int32), ("reserved_", C.c_int32),
                ("visit_prev", C.c_void_p)]


class QpGridMetric(C.Structure):
    _fields_ = [("valid", C.c_int32), ("xbits", C.c_int32), ("wx", C.c_uint32), ("wy", C.c_uint32),
                ("wrap_x", C.c_int32), ("wrap_y", C.c_int32)]


class QpDTable(C.Structure):
    _fields_ = [("kind", C.c_int32), ("C", C.c_int32), ("table", C.c_void_p), ("grid", QpGridMetric),
                ("symmetric", C.c_int32), ("reserved_", C.c_int32), ("grid_scale", C.c_double)]


class QpQuboKron(C.Structure):
    _fields_ = [("T", C.c_int32), ("C", C.c_int32), ("cap", C.c_int32), ("nmax", C.c_int32), ("Npad", C.c_int64),
                ("constant", C.c_int32), ("n_tiles", C.c_int32),
                ("n", C.c_void_p), ("pitch", C.c_void_p), ("off", C.c_void_p), ("same", C.c_void_p),
                ("same_off", C.c_void_p), ("tile_slice", C.c_void_p), ("tile_first", C.c_void_p), ("linear", C.c_void_p),
                ("next", C.c_void_p), ("prev", C.c_void_p), ("hops10", C.c_void_p)]


class QpQuboStruct(C.Structure):
    _fields_ = [("T", C.c_int32), ("C", C.c_int32), ("cap", C.c_int32), ("nmax", C.c_int32), ("Npad", C.c_int64),
                ("n", C.c_void_p), ("off", C.c_void_p), ("info", C.c_void_p), ("nbr_ptr", C.c_void_p),
                ("nbr", C.c_void_p), ("hops", C.c_void_p), ("pitch", C.c_void_p)]


QP_QUBO_INIT_GIVEN = 1


class QplaceError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libqplace error {code}: {msg}")
        self.code = code


EXPORTS = {
    "qp_version": (C.c_int, []),
    "qp_last_error": (C.c_char_p, []),
    "qp_host_threads": (C.c_int32, [C.c_int32]),
    "qp_set_host_threads": (C.c_int, [C.c_int32]),
    "qp_apsp_f64": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qp_hops_f64": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qp_lsap_f64": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qp_lsap_f64_batch": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                    C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]),
    "qp_timeslices": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                C.POINTER(C.c_int32)]),
    "qp_qasm_scan": (C.c_int, [C.c_char_p, C.c_int64, C.c_int64, C.c_int32, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int64,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64),
                               C.POINTER(C.c_int64)]),
    "qp_schedule_gates": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int64]),
    "qp_group_rows": (C.c_int32, [C.c_int32]),
    "qp_schedule_gate_groups": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                          C.c_int64]),
    "qp_dtable_pack": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.POINTER(C.c_int32),
                                 C.POINTER(C.c_size_t)]),
    "qp_dtable_grid_metric": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(QpGridMetric)]),
    "qp_dtable_symmetric": (C.c_int, [C.c_void_p, C.c_int32, C.c_double]),
    "qp_dtable_scaled_grid_metric": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.POINTER(QpGridMetric), C.POINTER(C.c_double)]),
    "qp_make_space": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                C.c_int32, C.c_void_p]),
    "qp_place_sparse": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_int32, C.c_int32, C.c_void_p]),
    "qp_slice_cost_workspace": (C.c_size_t, [C.c_int32, C.c_int64]),
    "qp_slice_cost_v3": (C.c_int, [C.POINTER(QpCircuit), C.POINTER(QpDTable), C.c_void_p, C.c_int64, C.c_int32,
                                   C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "qp_static_scratch_bytes": (C.c_size_t, []),
    "qp_slice_cost_v3_static": (C.c_int, [C.POINTER(QpCircuit), C.POINTER(QpDTable), C.c_void_p, C.c_int64, C.c_int32,
                                          C.c_int32, C.c_void_p, C.c_void_p]),
    "qp_slice_cost_v2": (C.c_int, [C.POINTER(QpCircuit), C.POINTER(QpDTable), C.c_void_p, C.c_int64, C.c_int32,
                                   C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "qp_cost_matrix": (C.c_int, [C.c_int32, C.POINTER(QpCircuit), C.POINTER(QpDTable), C.c_int64, C.c_void_p,
                                 C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    "qp_hqa_place_pinned_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32]),
    "qp_hqa_place_batch": (C.c_int, [C.c_int32, C.POINTER(QpCircuit), C.POINTER(QpDTable), C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_int32, C.c_void_p]),
    "qp_hqa_place_batch_multi": (C.c_int, [C.c_int32, C.POINTER(QpCircuit), C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                           C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    "qp_cost_matrix_multi": (C.c_int, [C.c_int32, C.POINTER(QpCircuit), C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                       C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "qp_slice_cost_v3_slices": (C.c_int, [C.POINTER(QpCircuit), C.POINTER(QpDTable), C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                          C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "qp_slice_cost_v3_multi": (C.c_int, [C.POINTER(QpCircuit), C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                         C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "qp_apsp_f64_batch": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                    C.c_int32]),
    "qp_gtrhqa_workspace": (C.c_size_t, [C.c_int32]),
    "qp_gtrhqa_place": (C.c_int, [C.POINTER(QpCircuit), C.POINTER(QpDTable), C.c_void_p, C.c_int64, C.c_int32,
                                  C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "qp_core_activity": (C.c_int, [C.POINTER(QpCircuit), C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_void_p, C.c_void_p]),
    "qp_comm_matrix": (C.c_int, [C.POINTER(QpCircuit), C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    "qp_qubo_workspace": (C.c_size_t, [C.POINTER(QpQuboKron), C.c_int32]),
    "qp_qubo_fields_i32": (C.c_int, [C.POINTER(QpQuboKron), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_void_p]),
    "qp_qubo_energy_i32": (C.c_int, [C.POINTER(QpQuboKron), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_void_p]),
    "qp_qubo_anneal": (C.c_int, [C.POINTER(QpQuboStruct), C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_uint64,
                                 C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
}

_lib = None


def lib():
    """The loaded library (loads it on first use)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it first (python -c 'import __graft_entry__ as g; g.build()'). "
                "This package has no CPU fallback.")
        # torch first: its wheel brings its own copy of the HIP runtime, and whichever copy is loaded first
        # is the one both end up using.  If this library's system copy came first (host helpers called before
        # anything touched torch), torch would later load a second runtime and device calls from here would
        # fail with "no ROCm-capable device is detected".
        # (QPLACE_NO_TORCH: a host-only build of the library under a sanitizer, which must not pull in the HIP runtime.)
        if not os.environ.get("QPLACE_NO_TORCH"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def host_threads(limit, polling=False):
    """Default for every ``threads`` / ``lanes`` argument: min(limit, this process's share of the node's CPUs) --
    the CPUs it may run on divided by the ranks on the node (LOCAL_WORLD_SIZE), see qp_host_threads.  Lanes that
    spin on a stream (``polling``) take half of the share, so that the solver threads beside them get a core too."""
    share = int(lib().qp_host_threads(1 << 20))
    if polling:
        share = max(1, share // 2)
    return max(1, min(int(limit), share))


def check(rc):
    if rc != QP_OK:
        raise QplaceError(rc, lib().qp_last_error().decode(errors="replace"))


def ptr(arr):
    """Address of a C-contiguous numpy array's data."""
    if not arr.flags["C_CONTIGUOUS"]:
        raise ValueError("array must be C-contiguous")
    return arr.ctypes.data
