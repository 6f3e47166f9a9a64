"""ctypes binding of include/admm_hip.h (the C ABI of libadmm_hip.so).

This is the stub a maintainer of the reference would add to drive the GPU hot path from Python; the
C++ mirror of the reference classes (host/) binds the same symbols.  There is NO CPU fallback: if
the shared library is missing or no HIP device is visible, calls fail loudly.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
lib_path = os.path.join(HERE, "libadmm_hip.so")
# Experiments only: ADMM_HIP_LIB=<path> loads another build of the library (a same-box A/B of kernel variants) WITHOUT
# touching the in-tree one; it is announced on stderr so that a number can never silently come from a variant.
_variant = os.environ.get("ADMM_HIP_LIB")
if _variant:
    import sys as _sys
    lib_path = os.path.abspath(_variant)
    print("[admm_hip] ADMM_HIP_LIB set: loading the library VARIANT %s" % lib_path, file=_sys.stderr)


class AdmmHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("admm_hip error %d: %s" % (code, msg))
        self.code = code


c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int32)


class Desc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("device", C.c_int32),
        ("n_verts", C.c_int32), ("masses", c_double_p), ("dt", C.c_double),
        ("n_tets", C.c_int32), ("tet_idx", c_int_p), ("tet_Binv", c_double_p), ("tet_weight", c_double_p),
        ("tet_kind", c_int_p), ("tet_mu", c_double_p), ("tet_lambda", c_double_p), ("tet_k", c_double_p),
        ("n_tris", C.c_int32), ("tri_idx", c_int_p), ("tri_rest", c_double_p), ("tri_weight", c_double_p),
        ("tri_limit_min", c_double_p), ("tri_limit_max", c_double_p),
        ("n_pins", C.c_int32), ("pin_vert", c_int_p), ("pin_xyz", c_double_p), ("pin_active", c_int_p),
        ("pin_weight", C.c_double),
        ("linsolver", C.c_int32), ("constraint_w", C.c_double),
        ("pcg_max_iters", C.c_int32), ("pcg_tol", C.c_double),
        ("gs_max_iters", C.c_int32), ("gs_tol", C.c_double), ("gs_omega", C.c_double),
        ("uzawa_max_iters", C.c_int32), ("uzawa_tol", C.c_double),
        ("n_obstacles", C.c_int32), ("obstacle_kind", c_int_p), ("obstacle_params", c_double_p),
        ("gs_colors", c_int_p),
        ("rank", C.c_int32), ("world_size", C.c_int32),
        ("tet_kappa", c_double_p),
        ("vert_xyz", c_double_p),
        ("n_spline_tables", C.c_int32), ("spline_tables", c_double_p), ("tet_spline", c_int_p),
        ("n_obstacle_grids", C.c_int32), ("obstacle_grid_meta", c_double_p), ("obstacle_grid_data", c_double_p),
        ("pin_normal", c_double_p),
        ("n_bends", C.c_int32), ("bend_idx", c_int_p), ("bend_coef", c_double_p), ("bend_weight", c_double_p), ("bend_stiffness", c_double_p),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("global_ms", C.c_double), ("local_ms", C.c_double), ("collision_ms", C.c_double),
        ("inner_iters", C.c_int32), ("admm_iters", C.c_int32), ("step_ms", C.c_double),
        ("last_solve_converged", C.c_int32), ("n_constraints", C.c_int32), ("rhs_ms", C.c_double), ("unconverged_solves", C.c_int32), ("pcg_launched_iters", C.c_int32), ("pcg_iters_per_solve", C.c_int32 * 64),
        ("local_kernel_ms", C.c_double),
    ]


SPLINE_TABLE_DOUBLES = 9228     # ADMM_SPLINE_TABLE_DOUBLES
SPLINE_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_int, C.c_double)     # admm_spline_fn
OBSTACLE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double))     # admm_obstacle_fn
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64)     # admm_allreduce_fn

# every symbol include/admm_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("admm_hip_last_error", C.c_char_p, []),
    ("admm_hip_device_count", C.c_int, []),
    ("admm_hip_create", C.c_int, [C.POINTER(Desc), C.POINTER(C.c_void_p)]),
    ("admm_hip_destroy", None, [C.c_void_p]),
    ("admm_hip_set_state", C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    ("admm_hip_get_state", C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    ("admm_hip_set_pins", C.c_int, [C.c_void_p, C.c_int32, c_int_p, c_double_p]),
    ("admm_hip_set_pin_normals", C.c_int, [C.c_void_p, C.c_int32, c_int_p, c_double_p]),
    ("admm_hip_set_surface_inds", C.c_int, [C.c_void_p, C.c_int32, c_int_p]),
    ("admm_hip_set_wind", C.c_int, [C.c_void_p, C.c_int32, c_int_p, c_double_p]),
    ("admm_hip_add_dynamic_tetmesh", C.c_int, [C.c_void_p, C.c_int32, C.c_int32, c_double_p, C.c_int32, c_int_p, C.c_int32, c_int_p]),
    ("admm_hip_detect_dynamic", C.c_int, [C.c_void_p, c_double_p, C.c_int32, c_int_p, c_int_p, c_int_p, c_double_p, c_double_p, c_double_p]),
    ("admm_hip_step", C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.POINTER(Stats)]),
    ("admm_hip_local_step", C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    ("admm_hip_global_solve", C.c_int, [C.c_void_p, c_double_p, c_double_p, c_int_p]),
    ("admm_hip_num_rows", C.c_int, [C.c_void_p]),
    ("admm_hip_solve_totals", C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("admm_hip_set_solver_params", C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double]),
    ("admm_hip_get_solver_params", C.c_int, [C.c_void_p, C.c_int32, c_int_p, c_double_p, c_double_p]),
    ("admm_hip_set_soft_modes", C.c_int, [C.c_void_p, C.c_int32, c_double_p]),
    ("admm_hip_compute_soft_modes", C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    ("admm_hip_get_soft_modes", C.c_int, [C.c_void_p, c_int_p, c_double_p]),
    ("admm_hip_contact_totals", C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    ("admm_hip_persistent_launches", C.c_int, [C.c_void_p] + [C.POINTER(C.c_int64)] * 3),
    ("admm_hip_pcg_findings", C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    ("admm_hip_probe_sync", C.c_int, [C.c_void_p, C.c_int32, c_double_p, c_double_p, C.POINTER(C.c_int64)]),
    ("admm_hip_time_local_launches", C.c_int, [C.c_void_p, C.c_int32]),
    ("admm_hip_local_launch_times", C.c_int, [C.c_void_p, C.POINTER(C.c_int64), c_double_p]),
    ("admm_hip_get_matrix", C.c_int, [C.c_void_p, c_int_p, c_int_p, c_double_p, c_int_p]),
    ("admm_hip_get_colors", C.c_int, [C.c_void_p, c_int_p, c_int_p]),
    ("admm_hip_comm_unique_id", C.c_int, [C.c_char_p]),
    ("admm_hip_comm_init", C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int]),
    ("admm_hip_comm_info", C.c_int, [C.c_void_p, c_int_p, c_int_p, C.c_char_p]),
    ("admm_hip_set_rhs_allreduce", C.c_int, [C.c_void_p, ALLREDUCE_FN, C.c_void_p]),
    ("admm_host_assemble_matrix", C.c_int, [C.POINTER(Desc), c_int_p, c_int_p, c_double_p, c_int_p]),
    ("admm_host_partition", None, [C.c_int32, C.c_int, C.c_int, c_int_p, c_int_p]),
    ("admm_host_component_partition", C.c_int32, [C.POINTER(Desc), C.c_int, c_int_p]),
    ("admm_host_tabulate_spline", C.c_int, [SPLINE_FN, C.c_void_p, C.c_double, C.c_double, c_double_p]),
    ("admm_host_spline_table_eval", None, [c_double_p, C.c_int, C.c_double, c_double_p]),
    ("admm_host_tet_rest", C.c_int, [C.c_int32, c_int_p, c_double_p, c_double_p, c_double_p]),
    ("admm_host_tet_rest_positions", C.c_int, [C.c_int32, C.c_int32, c_int_p, c_double_p, c_double_p, c_double_p]),
    ("admm_hip_tet_rest_mode", C.c_int, [C.c_void_p]),
    ("admm_hip_uzawa_cache_stats", C.c_int, [C.c_void_p] + [C.POINTER(C.c_int64)] * 5),
    ("admm_hip_uzawa_unconverged_columns", C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    ("admm_hip_uzawa_column_lanes", C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("admm_host_sample_obstacle", C.c_int, [OBSTACLE_FN, C.c_void_p, c_double_p, c_double_p, c_int_p, c_double_p, c_double_p]),
    ("admm_host_bend_hinges", C.c_int32, [C.c_int32, C.c_int32, c_int_p, c_double_p, C.c_int32, c_int_p, c_double_p, c_double_p]),
    ("admm_host_tri_rest", C.c_int, [C.c_int32, c_int_p, c_double_p, c_double_p, c_double_p]),
    ("admm_host_lame", None, [C.c_double, C.c_double, c_double_p, c_double_p, c_double_p]),
    ("admm_host_greedy_coloring", C.c_int, [C.c_int32, c_int_p, c_int_p, c_int_p]),
    ("admm_host_locality_order", None, [C.c_int32, C.c_int32, C.c_int32, c_int_p, c_int_p, c_double_p, c_double_p]),
    ("admm_host_block_order", None, [C.c_int32, C.c_int32, C.c_int32, c_int_p, C.c_int32, c_int_p]),
    ("admm_host_chunk_reduce", C.c_int, [C.c_int32, C.c_int32, c_int_p, c_int_p, c_double_p, c_double_p, C.POINTER(C.c_int64)]),
    ("admm_host_gs_plan_sweeps", C.c_int, [C.POINTER(Desc), C.c_int32, c_int_p, C.c_int32, C.c_int32, c_double_p, c_double_p, C.c_int32, C.c_double, c_int_p]),
    ("admm_host_big_plan", C.c_int, [C.POINTER(Desc), C.c_int32, c_int_p, c_int_p, c_double_p, C.POINTER(C.c_float)]),
    ("admm_host_oc_plan", C.c_int, [C.POINTER(Desc), C.c_int32, C.c_int32, C.c_int32, c_int_p, c_int_p, c_double_p, C.POINTER(C.c_int64), C.POINTER(C.c_float)]),
]

_lib = None


def lib():
    """Loads libadmm_hip.so (built by build.py / __graft_entry__.build()).  Fails loudly if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(lib_path):
            raise AdmmHipError(-2, "libadmm_hip.so is not built (%s); run `python __graft_entry__.py build`. "
                                   "There is no CPU fallback for the ADMM hot path." % lib_path)
        L = C.CDLL(lib_path)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise AdmmHipError(rc, lib().admm_hip_last_error().decode("utf-8", "replace"))


def device_count():
    return lib().admm_hip_device_count()


def dptr(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


def iptr(a):
    return None if a is None else a.ctypes.data_as(c_int_p)


def f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a if shape is None else a.reshape(shape)


def i32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a if shape is None else a.reshape(shape)


# ---- host-only helpers (no GPU) ----
def tet_rest(verts, tets):
    verts = f64(verts, (-1, 3)); tets = i32(tets, (-1, 4))
    n = tets.shape[0]
    Binv = np.empty((n, 9)); vol = np.empty(n)
    check(lib().admm_host_tet_rest(n, iptr(tets), dptr(verts), dptr(Binv), dptr(vol)))
    return Binv, vol


def tet_rest_positions(n_verts, tets, Binv, candidate=None):
    """admm_host_tet_rest_positions: (mode, x0) -- mode 1 candidate accepted, 2 propagated through the tets, 0 none."""
    tets = i32(tets, (-1, 4)); Binv = f64(Binv, (-1, 9))
    x0 = np.zeros((n_verts, 3))
    cand = None if candidate is None else f64(candidate, (-1, 3))
    mode = lib().admm_host_tet_rest_positions(n_verts, tets.shape[0], iptr(tets), dptr(Binv), None if cand is None else dptr(cand), dptr(x0))
    return mode, x0


def tri_rest(verts, tris):
    verts = f64(verts, (-1, 3)); tris = i32(tris, (-1, 3))
    n = tris.shape[0]
    rest = np.empty((n, 4)); area = np.empty(n)
    check(lib().admm_host_tri_rest(n, iptr(tris), dptr(verts), dptr(rest), dptr(area)))
    return rest, area


def bend_hinges(verts, tris):
    """admm_host_bend_hinges: (hinge idx [n,4], cotangent stencil coef [n,4], rest area [n]) of the interior edges of a triangle mesh."""
    verts = f64(verts, (-1, 3)); tris = i32(tris, (-1, 3))
    n = lib().admm_host_bend_hinges(verts.shape[0], tris.shape[0], iptr(tris), dptr(verts), 0, None, None, None)
    if n < 0:
        raise AdmmHipError(-1, "bend_hinges: triangle index out of range")
    idx = np.zeros((n, 4), np.int32); coef = np.zeros((n, 4)); area = np.zeros(n)
    if n:
        lib().admm_host_bend_hinges(verts.shape[0], tris.shape[0], iptr(tris), dptr(verts), n, iptr(idx), dptr(coef), dptr(area))
    return idx, coef, area


def lame(youngs, poisson):
    mu, la, k = C.c_double(), C.c_double(), C.c_double()
    lib().admm_host_lame(youngs, poisson, C.byref(mu), C.byref(la), C.byref(k))
    return mu.value, la.value, k.value


def greedy_coloring(rowptr, col):
    rowptr = i32(rowptr); col = i32(col)
    n = rowptr.shape[0] - 1
    color = np.empty(n, dtype=np.int32)
    nc = lib().admm_host_greedy_coloring(n, iptr(rowptr), iptr(col), iptr(color))
    return color, nc


def locality_order(n_verts, elems):
    """admm_host_locality_order: (new_id[n_verts], span_before, span_after) for tets [n,4] or triangles [n,3]."""
    elems = i32(elems)
    new_id = np.zeros(n_verts, np.int32)
    sb, sa = C.c_double(0), C.c_double(0)
    lib().admm_host_locality_order(n_verts, elems.shape[0], elems.shape[1], iptr(elems), iptr(new_id), C.byref(sb), C.byref(sa))
    return new_id, sb.value, sa.value


def block_order(n_verts, elems, leaf=256):
    """admm_host_block_order: new_id[n_verts] (hierarchical block order) for tets [n,4] or triangles [n,3]."""
    elems = i32(elems)
    new_id = np.zeros(n_verts, np.int32)
    lib().admm_host_block_order(n_verts, elems.shape[0], elems.shape[1], iptr(elems), leaf, iptr(new_id))
    return new_id


def chunk_reduce(n_verts, tets, kind_begin, corner_forces):
    """admm_host_chunk_reduce: the local step's block-level reduction of corner forces [n,4,3] run on the host -> (vertex sums
    [n_verts,3], dict of plan statistics)."""
    tets = i32(tets); kb = i32(kind_begin); cf = f64(corner_forces)
    out = np.zeros((n_verts, 3)); st = (C.c_int64 * 8)()
    check(lib().admm_host_chunk_reduce(n_verts, tets.shape[0], iptr(tets), iptr(kb), dptr(cf), dptr(out), st))
    return out, dict(chunks=st[0], records=st[1], max_passes=st[2], max_list=st[3], stored=st[4])


def partition(n_items, world_size, rank):
    b, e = C.c_int32(), C.c_int32()
    lib().admm_host_partition(n_items, world_size, rank, C.byref(b), C.byref(e))
    return b.value, e.value
