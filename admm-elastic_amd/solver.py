"""Python mirror of the reference's Solver / EnergyTerm / Lame interface for the ADMM hot path.

Same names, argument meaning and error behaviour as the reference classes (file:line cited per
method), sitting on the C ABI in include/admm_hip.h.  All per-iteration work runs in the HIP kernels
of libadmm_hip.so; this file only flattens the scene description, like Solver::initialize does.
"""
import ctypes as C
import os

import numpy as np

from . import capi
from .capi import AdmmHipError, Desc, Stats, check, dptr, f64, i32, iptr, lib

TET_LINEAR, TET_NEOHOOKEAN, TET_STVK, TET_SPLINE_NH, TET_SPLINE_STVK, TET_SPLINE_COROTATED, TET_SPLINE_TABLE, TET_STABLE_NH = 0, 1, 2, 3, 4, 5, 6, 7
LS_LDLT, LS_NCMCGS, LS_UZAWACG = 0, 1, 2


class Lame:
    """src/EnergyTerm.hpp:34-59."""

    def __init__(self, youngs=None, poisson=None, mu=None, lambda_=None):
        if youngs is not None:
            self.mu = youngs / (2.0 * (1.0 + poisson))
            self.lambda_ = youngs * poisson / ((1.0 + poisson) * (1.0 - 2.0 * poisson))
        else:
            self.mu, self.lambda_ = mu, lambda_
        self.limit_min, self.limit_max = -100.0, 100.0

    def bulk_modulus(self):
        return self.lambda_ + (2.0 / 3.0) * self.mu

    @staticmethod
    def rubber():
        return Lame(10000000.0, 0.499)

    @staticmethod
    def soft_rubber():
        return Lame(10000000.0, 0.399)

    @staticmethod
    def very_soft_rubber():
        return Lame(1000000.0, 0.299)


class Settings:
    """Solver::Settings, src/Solver.hpp:39-50 (+ the GPU inner-solver knobs)."""

    def __init__(self, **kw):
        self.timestep_s = 1.0 / 24.0
        self.verbose = 0
        self.admm_iters = 10
        self.gravity = -9.8
        self.linsolver = 0
        self.constraint_w = -1.0
        self.pcg_max_iters = 0
        self.pcg_tol = 0.0
        self.gs_max_iters = 0
        self.gs_tol = -1.0
        self.gs_omega = 0.0
        self.uzawa_max_iters = 0
        self.uzawa_tol = 0.0
        self.soft_modes = 0          # GPU build: end projection of every PCG solve on the k lowest modes of the system matrix (admm_hip_compute_soft_modes)
        self.device = 0
        self.rank = 0
        self.world_size = 1
        for k, v in kw.items():
            if not hasattr(self, k):
                raise TypeError("unknown setting " + k)
            setattr(self, k, v)


class RuntimeData:
    """Solver::RuntimeData, src/Solver.hpp:54-61."""

    def __init__(self):
        self.global_ms = 0.0
        self.local_ms = 0.0
        self.collision_ms = 0.0
        self.local_kernel_ms = 0.0   # tet local-step kernel durations (device clock); local_ms = the phase incl. dispatch gaps
        self.inner_iters = 0
        self.step_ms = 0.0
        self.last_solve_converged = 0
        self.rhs_ms = 0.0
        self.unconverged_solves = 0
        self.pcg_launched_iters = 0
        self.pcg_iters_per_solve = []


class Floor:
    """src/PassiveObject.hpp:32-45."""

    def __init__(self, y):
        self.kind, self.params = 0, [float(y), 0.0, 0.0, 0.0]


class Sphere:
    """src/PassiveObject.hpp:48-64."""

    def __init__(self, center, radius):
        self.kind, self.params = 1, [float(center[0]), float(center[1]), float(center[2]), float(radius)]


class Plane:
    """A user-side PassiveCollision (src/Collider.hpp:66-83): the half space n.x < d is solid.  signed distance n.x - d (n normalised),
    contact point x - dx n.  Floor(y) is Plane((0, 1, 0), y)."""

    def __init__(self, normal, d):
        n = f64(normal).ravel()
        l = float(np.linalg.norm(n))
        if not l > 0.0:
            raise AdmmHipError(-1, "Plane: zero normal")
        # unit normal, offset scaled with it -- as the C++ mirror's Plane and the library do: oracle and device see the same plane
        self.kind, self.params = 2, [float(n[0] / l), float(n[1] / l), float(n[2] / l), float(d) / l]


class SampledObstacle:
    """ANY user-defined PassiveCollision on the device: `obj.signed_distance(x) -> (dx, point[3], normal[3])` (the payload a fresh
    Payload would hold after src/Collider.hpp:80-82) is sampled on a dims[0] x dims[1] x dims[2] grid over the box [lo, hi] at
    Solver::initialize (admm_host_sample_obstacle); the kernels interpolate distance and normal trilinearly.  Outside the box the
    object is never hit: make the box cover the region the scene can reach."""

    def __init__(self, obj, lo, hi, dims=(64, 64, 64)):
        self.kind, self.obj = 3, obj
        self.lo, self.hi = f64(lo).ravel().copy(), f64(hi).ravel().copy()
        self.dims = i32(dims).ravel().copy()
        self.params = [0.0, 0.0, 0.0, 0.0]     # params[0] = index of its grid, set by make_desc

    def sample(self):
        n = int(self.dims[0]) * int(self.dims[1]) * int(self.dims[2])
        meta = np.zeros(10); data = np.zeros(4 * n)
        obj = self.obj

        failed = []

        def cb(user, px, pout):
            # a Python exception must not unwind through the C frames, and a sample that was never written must not read as
            # "distance 0, normal 0": NaN makes admm_host_sample_obstacle return ADMM_HIP_ERR_ARG, the exception is re-raised below
            try:
                dx, point, normal = obj.signed_distance(np.array([px[0], px[1], px[2]]))
                vals = [float(dx)] + [float(point[a]) for a in range(3)] + [float(normal[a]) for a in range(3)]
            except Exception as e:
                if not failed:
                    failed.append(e)
                vals = [float("nan")] * 7
            for a in range(7):
                pout[a] = vals[a]
        fn = capi.OBSTACLE_FN(cb)
        rc = lib().admm_host_sample_obstacle(fn, None, dptr(self.lo), dptr(self.hi), iptr(self.dims), dptr(meta), dptr(data))
        if failed:
            raise failed[0]
        check(rc)
        return meta, data


class TetMeshCollision:
    """admm::TetMeshCollision (src/DynamicObject.hpp:31-121): self-collision proxy of one tet mesh.  verts = REST
    vertices of the mesh, tets / faces index them (faces = surface triangles; meshes.surface_faces), v_offset = index
    of the mesh's first vertex in the solver's node vector."""

    def __init__(self, verts, tets, faces, v_offset):
        self.rest = f64(verts).reshape(-1, 3).copy()
        self.tets = i32(tets, (-1, 4)).copy()
        self.faces = i32(faces, (-1, 3)).copy()
        self.vert_offset = int(v_offset)
        if self.faces.shape[0] == 0:
            raise AdmmHipError(-1, "**TetMeshCollision Error: TetMesh needs surface faces")


class Solver:
    """admm::Solver (src/Solver.hpp:32-124) on the MI355X hot path."""

    def __init__(self):
        self.m_x = np.zeros(0)
        self.m_v = np.zeros(0)
        self.m_masses = np.zeros(0)
        self._tets = []   # (idx[n,4], Binv[n,9], weight[n], kind[n], mu[n], la[n], k[n], kappa[n], spline table[n])
        self._user_splines = []   # user-defined xu::Spline objects (TET_SPLINE_TABLE), in order of first use
        self._tris = []   # (idx[n,3], rest[n,4], weight[n], lmin[n], lmax[n])
        self._pins = {}   # vertex -> xyz   (ConstraintSet::pins)
        self._slides = {}  # vertex -> (point, unit normal): slide constraints (README TODO of the reference; include/admm_hip.h: desc.pin_normal)
        self._bends = []   # (idx[n,4], coef[n,4], weight[n], stiffness[n]): bending hinges (desc.bend_*)
        self._obstacles = []
        self._dynamic = []      # TetMeshCollision objects (Collider::dynamic_objs)
        self.surface_inds = []  # Solver::surface_inds (src/Solver.hpp:70): empty = every vertex is tested
        self._ctx = None
        self._settings = Settings()
        self._runtime = RuntimeData()
        self.initialized = False
        self._gs_colors = None
        self._rank, self._world = 0, 1

    def __del__(self):
        self.close()

    def close(self):
        if getattr(self, "_ctx", None):
            lib().admm_hip_destroy(self._ctx)
            self._ctx = None

    # ---- scene construction -------------------------------------------------------------------
    def add_nodes(self, x, m):
        """Solver::add_nodes (src/Solver.hpp:127-141): x, m are [n,3] / [3n] (masses x3). Returns node count."""
        x = f64(x).ravel(); m = f64(m).ravel()
        if x.size != m.size or x.size % 3:
            raise ValueError("add_nodes: x and m must both hold 3 values per node")
        self.m_x = np.concatenate([self.m_x, x])
        self.m_v = np.concatenate([self.m_v, np.zeros_like(x)])
        self.m_masses = np.concatenate([self.m_masses, m])
        return self.m_x.size // 3

    def add_tets(self, verts, inds, lame, kind=TET_LINEAR, vertex_offset=0, spline=None, kappa=0.0):
        """create_tets_from_mesh<IN_SCALAR,TYPE> (src/TetEnergyTerm.hpp:35-51) + the TetEnergyTerm ctor
        (src/TetEnergyTerm.cpp:31-48).  verts are the REST positions the indices refer to; kind selects
        TetEnergyTerm / NeoHookeanTet / StVKTet / SplineTet (TET_SPLINE_*: xu::NeoHookean / StVK / CoRotated with
        their compression term `kappa`, src/XuSpline.hpp:43-45 -- 0 as the reference's SplineTet constructors pass;
        `spline` = a Lame holding the spline's own mu / lambda, default the tet's, as the SplineTet constructors do,
        src/TetEnergyTerm.hpp:192-204).  Raises on an inverted rest tet."""
        inds = i32(inds, (-1, 4))
        Binv, vol = capi.tet_rest(verts, inds)
        k = lame.bulk_modulus()
        n = inds.shape[0]
        w = np.sqrt(k * vol)
        table = 0
        if kind == TET_SPLINE_TABLE:
            # SplineTet(tet, verts, lame, shared_ptr<xu::Spline>) with a USER-DEFINED spline (src/TetEnergyTerm.hpp:197-204): any
            # object with f, g, h, df, dg, dh (src/XuSpline.hpp:34-46); sampled once into a device table
            if spline is None or not all(hasattr(spline, a) for a in ("f", "g", "h", "df", "dg", "dh")):
                raise AdmmHipError(-1, "TET_SPLINE_TABLE needs a spline object with f, g, h, df, dg, dh")
            if not any(spline is sp for sp in self._user_splines):
                self._user_splines.append(spline)
            table = [i for i, sp in enumerate(self._user_splines) if sp is spline][0]
            sp = lame
        else:
            sp = spline if spline is not None else lame
        self._tets.append((inds + vertex_offset, Binv, w, np.full(n, kind, np.int32), np.full(n, sp.mu),
                           np.full(n, sp.lambda_), np.full(n, k), np.full(n, float(kappa) if TET_SPLINE_NH <= kind <= TET_SPLINE_COROTATED else 0.0),
                           np.full(n, table, np.int32)))
        return n

    def add_tris(self, verts, inds, lame, vertex_offset=0):
        """create_tris_from_mesh (src/TriEnergyTerm.hpp:31-46) + TriEnergyTerm ctor (src/TriEnergyTerm.cpp:29-52)."""
        if lame.limit_min > 1.0:
            raise AdmmHipError(-1, "**TriEnergyTerm Error: Strain limit min should be -inf to 1")
        if lame.limit_max < 1.0:
            raise AdmmHipError(-1, "**TriEnergyTerm Error: Strain limit max should be 1 to inf")
        inds = i32(inds, (-1, 3))
        rest, area = capi.tri_rest(verts, inds)
        n = inds.shape[0]
        w = np.sqrt(lame.bulk_modulus() * area)
        self._tris.append((inds + vertex_offset, rest, w, np.full(n, lame.limit_min), np.full(n, lame.limit_max)))
        return n

    def add_bends(self, verts, tris, k_bend, vertex_offset=0):
        """Bending terms for a triangle mesh (README.md:23-28 TODO of the reference; no reference code): one hinge term per interior edge,
        created like create_tris_from_mesh creates the stretch terms (src/TriEnergyTerm.hpp:31-46).  Discrete quadratic bending energy
        (Bergou et al. 2006): E = k_bend * 3 / (A0 + A1) / 2 * |sum_k c_k x_k|^2 with the cotangent stencil c of the REST shape
        (admm_host_bend_hinges); weight = sqrt(stiffness), so that the prox is q / 2 (the idiom of src/TriEnergyTerm.cpp:77-83).
        Returns the number of hinges."""
        idx, coef, area = capi.bend_hinges(verts, tris)
        stiff = float(k_bend) * 3.0 / area
        self._bends.append((idx + vertex_offset, coef, np.sqrt(stiff), stiff))
        return idx.shape[0]

    def set_slide_pins(self, inds, points, normals):
        """Slide constraints (README.md:23-28 TODO of the reference; the counterpart of Solver::set_pins, src/Solver.cpp:113-157, for
        normal-only constraints): vertex inds[i] may move freely in the plane through points[i] with normal normals[i].  Replaces the
        current set of slide constraints; ordinary pins are set_pins' business.  After initialize only constraints created before it may
        be moved / re-oriented (like pins with linsolver 0 / 2, src/Solver.cpp:147-151)."""
        new = {}
        for i, idx in enumerate(inds):
            n = f64(normals[i]).ravel().copy()
            l = float(np.linalg.norm(n))
            if not l > 0.0:
                raise AdmmHipError(-1, "set_slide_pins: zero normal")
            new[int(idx)] = (f64(points[i]).ravel().copy(), n / l)
        left = [k for k in self._slides if k not in new]
        if self.initialized and left:      # a vertex that leaves the set must not keep its normal (it would slide again when pinned later)
            check(lib().admm_hip_set_pin_normals(self._ctx, len(left), iptr(i32(left)), dptr(np.zeros((len(left), 3)))))
        self._slides = new
        if self.initialized:
            self._push_pins()
            v = i32(list(self._slides.keys()))
            nn = f64(np.array([q[1] for q in self._slides.values()]).reshape(-1, 3)) if len(v) else np.zeros((0, 3))
            check(lib().admm_hip_set_pin_normals(self._ctx, len(v), iptr(v), dptr(nn)))

    def _all_pins(self):
        """ordinary pins, then slide pins: (vertex list, points [n,3], normals [n,3] -- zero for ordinary pins)"""
        v = list(self._pins.keys()) + [k for k in self._slides if k not in self._pins]
        pts = [self._pins[k] for k in self._pins] + [self._slides[k][0] for k in self._slides if k not in self._pins]
        nrm = [np.zeros(3) for _ in self._pins] + [self._slides[k][1] for k in self._slides if k not in self._pins]
        return v, (f64(np.array(pts)).reshape(-1, 3) if v else np.zeros((0, 3))), (f64(np.array(nrm)).reshape(-1, 3) if v else np.zeros((0, 3)))

    def _push_pins(self):
        v, p, _ = self._all_pins()
        vi = i32(v)
        check(lib().admm_hip_set_pins(self._ctx, len(vi), iptr(vi), dptr(p)))

    def set_pins(self, inds, points=None):
        """Solver::set_pins (src/Solver.cpp:113-157)."""
        inds = [int(i) for i in inds]
        n = len(inds)
        pin_in_place = points is None or len(points) != n
        if (self.m_x.size == 0 and pin_in_place) or (pin_in_place and points is not None and len(points) > 0):
            raise AdmmHipError(-1, "**Solver::set_pins Error: Bad input.")
        if pin_in_place and self.initialized and self._ctx:
            self.download()     # the reference pins at the CURRENT m_x: after device-resident stepping the host copy is stale
        self._pins = {}
        for i, idx in enumerate(inds):
            self._pins[idx] = self.m_x[3 * idx:3 * idx + 3].copy() if pin_in_place else f64(points[i]).copy()
        if self.initialized:
            self._push_pins()

    def add_obstacle(self, obj):
        """Solver::add_obstacle (src/Solver.cpp:159-161)."""
        self._obstacles.append(obj)

    def set_wind(self, tris, direction):
        """ext_forces.push_back(WindForce(tris)) with WindForce::direction (src/ExplicitForce.hpp:39-46), applied on the device
        at the start of every step.  tris [n,3] node indices; empty = no wind.  After initialize."""
        self._need_ctx()
        t = i32(tris, (-1, 3)) if len(tris) else np.zeros((0, 3), np.int32)
        d = f64(direction).ravel().copy()
        check(lib().admm_hip_set_wind(self._ctx, t.shape[0], iptr(t), dptr(d)))

    def add_dynamic_collider(self, obj):
        """Solver::add_dynamic_collider (src/Solver.cpp:163-165)."""
        self._dynamic.append(obj)

    def detect_dynamic(self, x=None):
        """Collider::detect for the dynamic objects at x (default m_x): list of (vert, dx, face[3], barys[3], normal[3])
        = DynamicCollision::Payload, in candidate order."""
        self._need_ctx()
        x = f64(self.m_x if x is None else x).ravel().copy()
        nv = x.size // 3
        n = C.c_int32(0)
        vert = np.zeros(nv, np.int32); face = np.zeros((nv, 3), np.int32)
        bary = np.zeros((nv, 3)); nrm = np.zeros((nv, 3)); dx = np.zeros(nv)
        check(lib().admm_hip_detect_dynamic(self._ctx, dptr(x), nv, C.byref(n), iptr(vert), iptr(face), dptr(bary), dptr(nrm), dptr(dx)))
        return [(int(vert[i]), dx[i], face[i].copy(), bary[i].copy(), nrm[i].copy()) for i in range(n.value)]

    def set_gs_colors(self, colors):
        self._gs_colors = i32(colors)

    # ---- Solver::initialize (src/Solver.cpp:167-261) ----------------------------------------------
    def flatten(self):
        """Flat arrays of every energy term, in the order tets, tris (pins are appended by the library)."""
        def cat(lst, k, shape, dt):
            return np.concatenate([t[k] for t in lst]).astype(dt) if lst else np.zeros(shape, dt)
        T, R, H = self._tets, self._tris, self._bends
        pv, pp, pn = self._all_pins()
        out = dict(
            bend_idx=cat(H, 0, (0, 4), np.int32), bend_coef=cat(H, 1, (0, 4), np.float64), bend_weight=cat(H, 2, (0,), np.float64),
            bend_stiffness=cat(H, 3, (0,), np.float64), pin_normal=pn,
            tet_idx=cat(T, 0, (0, 4), np.int32), tet_Binv=cat(T, 1, (0, 9), np.float64), tet_weight=cat(T, 2, (0,), np.float64),
            tet_kind=cat(T, 3, (0,), np.int32), tet_mu=cat(T, 4, (0,), np.float64), tet_lambda=cat(T, 5, (0,), np.float64),
            tet_k=cat(T, 6, (0,), np.float64), tet_kappa=cat(T, 7, (0,), np.float64), tet_spline=cat(T, 8, (0,), np.int32),
            tri_idx=cat(R, 0, (0, 3), np.int32), tri_rest=cat(R, 1, (0, 4), np.float64), tri_weight=cat(R, 2, (0,), np.float64),
            tri_limit_min=cat(R, 3, (0,), np.float64), tri_limit_max=cat(R, 4, (0,), np.float64),
            pin_vert=i32(pv), pin_xyz=pp,
        )
        return out

    def make_desc(self, s):
        dof = self.m_x.size
        f = self.flatten()
        self._flat = f  # keep the arrays alive while the descriptor points at them
        d = Desc()
        d.struct_size = C.sizeof(Desc)
        d.device = s.device
        d.n_verts = dof // 3
        self._masses_c = f64(self.m_masses)
        d.masses = dptr(self._masses_c)
        d.dt = s.timestep_s
        d.n_tets = f["tet_idx"].shape[0]
        d.tet_idx, d.tet_Binv, d.tet_weight = iptr(f["tet_idx"]), dptr(f["tet_Binv"]), dptr(f["tet_weight"])
        d.tet_kind, d.tet_mu, d.tet_lambda, d.tet_k = iptr(f["tet_kind"]), dptr(f["tet_mu"]), dptr(f["tet_lambda"]), dptr(f["tet_k"])
        d.tet_kappa = dptr(f["tet_kappa"]) if f["tet_kappa"].any() else None
        if self._user_splines:          # tabulate every user-defined spline (admm_host_tabulate_spline)
            nd = capi.SPLINE_TABLE_DOUBLES
            self._spline_tables = np.zeros((len(self._user_splines), nd))
            for i, sp in enumerate(self._user_splines):
                fns = (sp.f, sp.g, sp.h, sp.df, sp.dg, sp.dh)
                cb = capi.SPLINE_FN(lambda user, which, x, fns=fns: float(fns[which](x)))
                check(lib().admm_host_tabulate_spline(cb, None, float(getattr(sp, "table_min", 0.02)), float(getattr(sp, "table_max", 50.0)),
                                                      dptr(self._spline_tables[i])))
            d.n_spline_tables = len(self._user_splines)
            d.spline_tables = dptr(self._spline_tables)
            d.tet_spline = iptr(f["tet_spline"])
        self._xyz_c = f64(self.m_x).copy()           # smooth coordinates for the coarse space of the on-chip PCG (desc.vert_xyz)
        d.vert_xyz = dptr(self._xyz_c) if self._xyz_c.size == dof else None
        d.n_tris = f["tri_idx"].shape[0]
        d.tri_idx, d.tri_rest, d.tri_weight = iptr(f["tri_idx"]), dptr(f["tri_rest"]), dptr(f["tri_weight"])
        d.tri_limit_min, d.tri_limit_max = dptr(f["tri_limit_min"]), dptr(f["tri_limit_max"])
        d.n_pins = f["pin_vert"].shape[0]
        d.pin_vert, d.pin_xyz, d.pin_active, d.pin_weight = iptr(f["pin_vert"]), dptr(f["pin_xyz"]), None, 0.0
        d.pin_normal = dptr(f["pin_normal"]) if self._slides else None
        d.n_bends = f["bend_idx"].shape[0]
        if d.n_bends:
            d.bend_idx, d.bend_coef = iptr(f["bend_idx"]), dptr(f["bend_coef"])
            d.bend_weight, d.bend_stiffness = dptr(f["bend_weight"]), dptr(f["bend_stiffness"])
        d.linsolver, d.constraint_w = s.linsolver, s.constraint_w
        d.pcg_max_iters, d.pcg_tol = s.pcg_max_iters, s.pcg_tol
        d.gs_max_iters, d.gs_tol, d.gs_omega = s.gs_max_iters, s.gs_tol, s.gs_omega
        d.uzawa_max_iters, d.uzawa_tol = s.uzawa_max_iters, s.uzawa_tol
        metas, datas, nodes = [], [], 0
        for o in self._obstacles:          # user-defined obstacles: sampled now, like Solver::initialize would first call them
            if o.kind == 3:
                meta, data = o.sample()
                meta[9] = nodes; o.params = [float(len(metas)), 0.0, 0.0, 0.0]
                metas.append(meta); datas.append(data); nodes += data.size // 4
        if metas:
            self._grid_meta = np.concatenate(metas); self._grid_data = np.concatenate(datas)
            d.n_obstacle_grids = len(metas); d.obstacle_grid_meta = dptr(self._grid_meta); d.obstacle_grid_data = dptr(self._grid_data)
        self._obst_kind = i32([o.kind for o in self._obstacles])
        self._obst_par = f64([o.params for o in self._obstacles]).reshape(-1, 4) if self._obstacles else np.zeros((0, 4))
        d.n_obstacles = len(self._obstacles)
        d.obstacle_kind, d.obstacle_params = iptr(self._obst_kind), dptr(self._obst_par)
        d.gs_colors = iptr(self._gs_colors) if self._gs_colors is not None else None
        d.rank, d.world_size = s.rank, s.world_size
        return d

    def host_matrix(self, settings=None):
        """Ahat for this scene from the host-only assembly path (no GPU): (rowptr, col, val)."""
        d = self.make_desc(settings if settings is not None else self._settings)
        nnz = C.c_int32(0)
        check(lib().admm_host_assemble_matrix(C.byref(d), None, None, None, C.byref(nnz)))
        nv = self.m_x.size // 3
        rp = np.zeros(nv + 1, np.int32); ci = np.zeros(nnz.value, np.int32); va = np.zeros(nnz.value)
        check(lib().admm_host_assemble_matrix(C.byref(d), iptr(rp), iptr(ci), dptr(va), C.byref(nnz)))
        return rp, ci, va

    def host_oc_plan(self, n_blocks, slices_per_block, lds_bytes=159744, settings=None, coarse=True):
        """The plan the library makes for the on-chip PCG of this scene (admm_host_oc_plan, no GPU): dict(row_vertex,
        row_aggregate, coarse_inv, stats)."""
        d = self.make_desc(settings if settings is not None else self._settings)
        n = 64 * n_blocks * slices_per_block
        rv = np.zeros(n, np.int32); ra = np.zeros(n, np.int32)
        nc = 4 * n_blocks
        ci = np.zeros((nc, nc)) if coarse else None
        st = (C.c_int64 * 11)()
        wt = np.zeros((n, 4), np.float32)
        check(lib().admm_host_oc_plan(C.byref(d), n_blocks, slices_per_block, lds_bytes, iptr(rv), iptr(ra), dptr(ci) if coarse else None, st,
                                      wt.ctypes.data_as(C.POINTER(C.c_float))))
        keys = ("nnz", "stored", "on_chip", "block_local", "max_neighbour_blocks", "coarse_unknowns", "max_halo", "lds_cols")
        stats = dict(zip(keys, list(st)[:8]))
        stats.update(lambda_bb=st[8] * 1e-9, bank_load_by_index=st[9] * 1e-6, bank_load_placed=st[10] * 1e-6)
        return dict(row_vertex=rv, row_aggregate=ra, coarse_inv=ci, stats=stats, row_weights=wt)

    def host_big_plan(self, max_aggregates=0, settings=None):
        """The plan of the launch-path two-level PCG of this scene (admm_host_big_plan, no GPU): dict(G, ra, rows, nc, ncp, slices, row_vertex,
        row_weights [rows, 4], coarse_inv [nc, nc])."""
        d = self.make_desc(settings if settings is not None else self._settings)
        st = np.zeros(6, np.int32)
        check(lib().admm_host_big_plan(C.byref(d), int(max_aggregates), iptr(st), None, None, None))
        rows, nc = int(st[2]), int(st[3])
        rv = np.zeros(rows, np.int32); wt = np.zeros((rows, 4)); ci = np.zeros((nc, nc), np.float32)
        check(lib().admm_host_big_plan(C.byref(d), int(max_aggregates), iptr(st), iptr(rv), dptr(wt), ci.ctypes.data_as(C.POINTER(C.c_float))))
        return dict(G=int(st[0]), ra=int(st[1]), rows=rows, nc=nc, ncp=int(st[4]), slices=int(st[5]), row_vertex=rv, row_weights=wt, coarse_inv=ci)

    def initialize(self, settings=None):
        s = settings if settings is not None else Settings()
        self._settings = s
        dof = self.m_x.size
        if s.timestep_s <= 0.0:
            s.timestep_s = 1.0 / 24.0
        if not (self.m_masses.size == dof and dof >= 3):
            return False  # "Problem with node data!" (Solver.cpp:180-183)
        self.m_v = np.zeros(dof)
        self.close()
        d = self.make_desc(s)
        ctx = C.c_void_p()
        check(lib().admm_hip_create(C.byref(d), C.byref(ctx)))
        self._ctx = ctx
        if len(self.surface_inds):
            si = i32(self.surface_inds)
            check(lib().admm_hip_set_surface_inds(ctx, len(si), iptr(si)))
        for o in self._dynamic:   # Solver.cpp:249-254 (LDLT: "No collisions with LDLT solver") is checked by the library
            check(lib().admm_hip_add_dynamic_tetmesh(ctx, o.vert_offset, o.rest.shape[0], dptr(o.rest), o.tets.shape[0],
                                                     iptr(o.tets), o.faces.shape[0], iptr(o.faces)))
        if s.soft_modes > 0 and s.linsolver != 1 and not (s.world_size > 1 and os.environ.get("ADMM_HIP_DIST_SOLVE") == "1"):
            check(lib().admm_hip_compute_soft_modes(ctx, int(s.soft_modes), 0))
        self.initialized = True
        return True

    # ---- Solver::step (src/Solver.cpp:35-110) ------------------------------------------------------
    def step(self):
        """One time step on the host-visible state m_x/m_v (uploaded, stepped on the GPU, downloaded)."""
        self.upload()
        self.step_device(stats=True)
        self.download()

    def upload(self):
        self._need_ctx()
        self.m_x = f64(self.m_x); self.m_v = f64(self.m_v)
        check(lib().admm_hip_set_state(self._ctx, dptr(self.m_x), dptr(self.m_v)))

    def download(self):
        self._need_ctx()
        check(lib().admm_hip_get_state(self._ctx, dptr(self.m_x), dptr(self.m_v)))

    def step_device(self, stats=False, admm_iters=None):
        """admm_hip_step on the device-resident state (no host<->device copies)."""
        self._need_ctx()
        s = self._settings
        it = s.admm_iters if admm_iters is None else admm_iters
        if stats:
            st = Stats()
            check(lib().admm_hip_step(self._ctx, it, s.gravity, C.byref(st)))
            r = self._runtime = RuntimeData()
            r.global_ms, r.local_ms, r.collision_ms = st.global_ms, st.local_ms, st.collision_ms
            r.inner_iters, r.step_ms, r.last_solve_converged = st.inner_iters, st.step_ms, st.last_solve_converged
            r.rhs_ms, r.unconverged_solves, r.pcg_launched_iters = st.rhs_ms, st.unconverged_solves, st.pcg_launched_iters
            r.local_kernel_ms = st.local_kernel_ms
            r.pcg_iters_per_solve = list(st.pcg_iters_per_solve)[:min(it, 64)]
        else:
            check(lib().admm_hip_step(self._ctx, it, s.gravity, None))

    def comm_init(self, dist):
        """Multi-GPU: build the RCCL communicator of this rank's context.  `dist` is an initialised
        torch.distributed (any backend) used only to broadcast the 128-byte RCCL unique id."""
        self._need_ctx()
        s = self._settings
        buf = C.create_string_buffer(128)
        if s.rank == 0:
            check(lib().admm_hip_comm_unique_id(buf))
        obj = [bytes(buf.raw)]
        dist.broadcast_object_list(obj, src=0)
        check(lib().admm_hip_comm_init(self._ctx, obj[0], s.rank, s.world_size))

    def comm_info(self):
        """admm_hip_comm_info: dict(n_ranks, rank, device) -- what RCCL says about the context's communicator, and the context's device."""
        self._need_ctx()
        n = C.c_int32(0); r = C.c_int32(-1); buf = C.create_string_buffer(64)
        check(lib().admm_hip_comm_info(self._ctx, C.byref(n), C.byref(r), buf))
        return dict(n_ranks=n.value, rank=r.value, device=buf.value.decode("ascii", "replace"))

    def set_rhs_allreduce(self, fn):
        """admm_hip_set_rhs_allreduce: the multi-GPU exchange over the caller's own transport instead of RCCL.  fn(buf) must sum
        the numpy array `buf` (a view of the library's pinned host buffer, [3 n_verts]) IN PLACE over all ranks; None removes it."""
        self._need_ctx()
        if fn is None:
            self._ar_cb = capi.ALLREDUCE_FN(0)
        else:
            def cb(user, ptr, n):
                try:
                    fn(np.ctypeslib.as_array(ptr, shape=(n,)))
                    return 0
                except Exception:        # a Python exception must not unwind through the C frames
                    import traceback
                    traceback.print_exc()
                    return 1
            self._ar_cb = capi.ALLREDUCE_FN(cb)      # (kept alive as long as the context may call it)
        check(lib().admm_hip_set_rhs_allreduce(self._ctx, self._ar_cb, None))

    def component_partition(self, world_size, settings=None):
        """admm_host_component_partition: (number of connected components, owning rank of every vertex) -- the multi-GPU
        partition admm_hip_create uses when the scene has at least world_size bodies."""
        d = self.make_desc(settings if settings is not None else self._settings)
        vr = np.zeros(self.m_x.size // 3, np.int32)
        n = lib().admm_host_component_partition(C.byref(d), world_size, iptr(vr))
        return n, vr

    def solve_totals(self):
        """admm_hip_solve_totals: (solves, converged solves, inner iterations) of the on-chip PCG since initialize."""
        self._need_ctx()
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(lib().admm_hip_solve_totals(self._ctx, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def set_solver_params(self, kind, max_iters=0, tol=-1.0, omega=0.0):
        """admm_hip_set_solver_params: the LinearSolver objects' public tuning members after initialize (the reference reads them on
        every solve: src/NodalMultiColorGS.hpp:40-46,100, src/UzawaCG.hpp:44-45,92).  kind = LS_NCMCGS (max_iters, tol, omega),
        LS_UZAWACG (max_iters, tol) or LS_LDLT (the PCG standing for the prefactored solve); values <= 0 (tol < 0) keep the current one."""
        self._need_ctx()
        check(lib().admm_hip_set_solver_params(self._ctx, int(kind), int(max_iters), float(tol), float(omega)))

    def solver_params(self, kind):
        """admm_hip_get_solver_params: (max_iters, tol, omega) in effect for that solver."""
        self._need_ctx()
        it = C.c_int32(0); t = C.c_double(0.0); o = C.c_double(0.0)
        check(lib().admm_hip_get_solver_params(self._ctx, int(kind), C.byref(it), C.byref(t), C.byref(o)))
        return it.value, t.value, o.value

    def set_soft_modes(self, Z):
        """admm_hip_set_soft_modes: Z [k, n_verts] (or None): every PCG solve ends with the exact Galerkin projection of its residual on them."""
        self._need_ctx()
        if Z is None or len(Z) == 0:
            check(lib().admm_hip_set_soft_modes(self._ctx, 0, None)); return
        Zc = f64(Z).reshape(len(Z), -1)
        check(lib().admm_hip_set_soft_modes(self._ctx, Zc.shape[0], dptr(Zc)))

    def compute_soft_modes(self, k, iters=0):
        """admm_hip_compute_soft_modes: the library computes the k lowest modes itself and installs them."""
        self._need_ctx()
        check(lib().admm_hip_compute_soft_modes(self._ctx, int(k), int(iters)))

    def get_soft_modes(self):
        """admm_hip_get_soft_modes: Z [k, n_verts] in effect (k = 0: none)."""
        self._need_ctx()
        k = C.c_int32(0)
        check(lib().admm_hip_get_soft_modes(self._ctx, C.byref(k), None))
        Z = np.zeros((k.value, self.m_x.size // 3))
        if k.value:
            check(lib().admm_hip_get_soft_modes(self._ctx, C.byref(k), dptr(Z)))
        return Z

    def soft_modes(self, k, iters=8, seed=0):
        """The k lowest eigenvectors of K = diag(m) + Ahat by inverse subspace iteration on the context's own solver (three right-hand sides
        per global solve) with Rayleigh-Ritz steps on the host; returns (eigenvalues, Z [k, n_verts])."""
        import scipy.sparse as sp
        self._need_ctx()
        rp, ci, va = self.system_matrix()
        nv = self.m_x.size // 3
        K = (sp.csr_matrix((va, ci, rp), shape=(nv, nv)) + sp.diags(f64(self.m_masses)[0::3])).tocsr()
        k3 = 3 * ((k + 2) // 3)
        X = np.random.default_rng(seed).standard_normal((nv, k3))
        ew = np.zeros(k3)
        for it in range(iters):
            X, _ = np.linalg.qr(X)
            Y = np.empty_like(X)
            for c0 in range(0, k3, 3):
                y, _ = self.global_solve(np.ascontiguousarray(X[:, c0:c0 + 3]).ravel(), np.zeros(3 * nv))
                Y[:, c0:c0 + 3] = y.reshape(-1, 3)
            Q, _ = np.linalg.qr(Y)
            H = Q.T @ (K @ Q)
            ew, V = np.linalg.eigh(0.5 * (H + H.T))
            X = Q @ V
        return ew[:k], np.ascontiguousarray(X[:, :k].T)

    def contact_totals(self):
        """admm_hip_contact_totals: rows of C over all UzawaCG solves / rows projected onto an obstacle over all GS sweeps, since initialize."""
        self._need_ctx()
        n = C.c_int64(0)
        check(lib().admm_hip_contact_totals(self._ctx, C.byref(n)))
        return n.value

    def persistent_launches(self):
        """admm_hip_persistent_launches: dict(pcg, gs, schur) -- launches of the persistent solver kernels since initialize."""
        self._need_ctx()
        a = [C.c_int64(0) for _ in range(3)]
        check(lib().admm_hip_persistent_launches(self._ctx, *[C.byref(x) for x in a]))
        return dict(zip(("pcg", "gs", "schur"), (x.value for x in a)))

    def pcg_findings(self):
        """admm_hip_pcg_findings: dict(smoother_given_up, trust_revoked, failed_checks) -- what the on-chip PCG holds against this context."""
        self._need_ctx()
        a = C.c_int32(0); b = C.c_int32(0); n = C.c_int64(0)
        check(lib().admm_hip_pcg_findings(self._ctx, C.byref(a), C.byref(b), C.byref(n)))
        return dict(smoother_given_up=bool(a.value), trust_revoked=bool(b.value), failed_checks=n.value)

    def probe_sync(self, n=200):
        """admm_hip_probe_sync: (us per all-to-all, us per vector exchange, plan statistics dict) of the on-chip PCG."""
        self._need_ctx()
        a = C.c_double(0.0); b = C.c_double(0.0); st = (C.c_int64 * 6)()
        check(lib().admm_hip_probe_sync(self._ctx, n, C.byref(a), C.byref(b), st))
        keys = ("nnz", "stored", "on_chip", "block_local", "max_neighbour_blocks", "coarse_unknowns")
        return a.value, b.value, dict(zip(keys, list(st)))

    def time_local_launches(self, on=True):
        """admm_hip_time_local_launches: event pairs around the local-step launches of the steps issued without statistics
        (on = 2: attached to the dominant local-step kernel's own dispatch)."""
        self._need_ctx()
        check(lib().admm_hip_time_local_launches(self._ctx, int(on) if on in (0, 1, 2) else (1 if on else 0)))

    def local_launch_times(self):
        """admm_hip_local_launch_times: (pairs recorded since the last call, sum of their intervals in ms)."""
        self._need_ctx()
        n = C.c_int64(0); ms = C.c_double(0.0)
        check(lib().admm_hip_local_launch_times(self._ctx, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def uzawa_cache_stats(self):
        """admm_hip_uzawa_cache_stats: dict(columns, column_solves, schur_from_columns, schur_by_pcg, evicted)."""
        self._need_ctx()
        a = [C.c_int64(0) for _ in range(5)]
        check(lib().admm_hip_uzawa_cache_stats(self._ctx, *[C.byref(x) for x in a]))
        d = dict(zip(("columns", "column_solves", "schur_from_columns", "schur_by_pcg", "evicted"), (x.value for x in a)))
        u = C.c_int64(0)
        check(lib().admm_hip_uzawa_unconverged_columns(self._ctx, C.byref(u)))
        d["unconverged_columns"] = u.value
        nb, nl, na, nw = C.c_int64(0), C.c_int(0), C.c_int64(0), C.c_int64(0)
        check(lib().admm_hip_uzawa_column_lanes(self._ctx, C.byref(nb), C.byref(nl), C.byref(na), C.byref(nw)))
        d["lane_batches"], d["lanes"], d["ahead_columns"], d["ahead_waits"] = nb.value, nl.value, na.value, nw.value
        return d

    def tet_rest_mode(self):
        """admm_hip_tet_rest_mode: 0 = the local step streams Binv, 1 / 2 = it recomputes Binv from gathered rest positions."""
        self._need_ctx()
        return lib().admm_hip_tet_rest_mode(self._ctx)

    def runtime_data(self):
        return self._runtime

    def settings(self):
        return self._settings

    # ---- kernel-level entry points (parity tests) -------------------------------------------------
    def num_rows(self):
        self._need_ctx()
        return lib().admm_hip_num_rows(self._ctx)

    def local_step(self, x, u, Mxbar=None):
        """EnergyTerm::update over all terms. Returns (z, u_new[, b])."""
        self._need_ctx()
        R = self.num_rows()
        x = f64(x).ravel().copy(); u = f64(u).ravel().copy()
        assert u.size == R
        z = np.zeros(R)
        b = np.zeros(x.size) if Mxbar is not None else None
        mx = f64(Mxbar).ravel().copy() if Mxbar is not None else None
        check(lib().admm_hip_local_step(self._ctx, dptr(x), dptr(u), dptr(z), dptr(mx), dptr(b)))
        return (z, u, b) if Mxbar is not None else (z, u)

    def global_solve(self, b, x0):
        """LinearSolver::solve: returns (x, inner_iters)."""
        self._need_ctx()
        b = f64(b).ravel().copy(); x = f64(x0).ravel().copy()
        it = C.c_int32(0)
        check(lib().admm_hip_global_solve(self._ctx, dptr(b), dptr(x), C.byref(it)))
        return x, it.value

    def system_matrix(self):
        """Ahat as (rowptr, col, val): A = diag(m) + Ahat (x) I3."""
        self._need_ctx()
        nnz = C.c_int32(0)
        check(lib().admm_hip_get_matrix(self._ctx, None, None, None, C.byref(nnz)))
        nv = self.m_x.size // 3
        rp = np.zeros(nv + 1, np.int32); ci = np.zeros(nnz.value, np.int32); va = np.zeros(nnz.value)
        check(lib().admm_hip_get_matrix(self._ctx, iptr(rp), iptr(ci), dptr(va), C.byref(nnz)))
        return rp, ci, va

    def gs_colors(self):
        self._need_ctx()
        nv = self.m_x.size // 3
        col = np.zeros(nv, np.int32); nc = C.c_int32(0)
        check(lib().admm_hip_get_colors(self._ctx, iptr(col), C.byref(nc)))
        return col, nc.value

    def _need_ctx(self):
        if not self._ctx:
            raise AdmmHipError(-4, "Solver is not initialized")
