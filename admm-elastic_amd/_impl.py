"""Public names of the package (see capi.py, solver.py, meshes.py, build.py)."""
from .capi import AdmmHipError, lib, lib_path, device_count  # noqa: F401
from .solver import Lame, Settings, RuntimeData, Solver, Floor, Sphere, Plane, SampledObstacle, TetMeshCollision, TET_LINEAR, TET_NEOHOOKEAN, TET_STVK, TET_SPLINE_NH, TET_SPLINE_STVK, TET_SPLINE_COROTATED, TET_SPLINE_TABLE, TET_STABLE_NH, LS_LDLT, LS_NCMCGS, LS_UZAWACG  # noqa: F401
from . import meshes  # noqa: F401
from .build import build_library  # noqa: F401

__all__ = ["AdmmHipError", "lib", "lib_path", "device_count", "Lame", "Settings", "RuntimeData", "Solver", "Floor", "Sphere", "Plane", "SampledObstacle", "TetMeshCollision",
           "TET_LINEAR", "TET_NEOHOOKEAN", "TET_STVK", "TET_SPLINE_NH", "TET_SPLINE_STVK", "TET_SPLINE_COROTATED", "TET_SPLINE_TABLE", "TET_STABLE_NH", "LS_LDLT", "LS_NCMCGS", "LS_UZAWACG", "meshes", "build_library"]
