"""Scene builders shared by the tests: the same description goes to the product Solver
(admm_elastic_amd, HIP) and to the CPU oracle (oracle/oracle.py)."""
import numpy as np

import admm_elastic_amd as pkg
from admm_elastic_amd import meshes
from admm_elastic_amd.solver import Floor, Lame, Settings, Solver, Sphere, TetMeshCollision
from oracle import oracle as orc


class Scene:
    def __init__(self):
        self.x = np.zeros((0, 3)); self.m = np.zeros(0)
        self.tets = []   # (verts_rest, idx, lame, kind)
        self.tris = []   # (verts_rest, idx, lame)
        self.pins = {}
        self.slides = {}     # vertex -> (point, normal): slide constraints
        self.bends = []      # (verts_rest, tris, k_bend, offset): bending terms of a triangle mesh
        self.obstacles = []  # (kind, params)
        self.dynamic = []    # dict(offset, rest, tets, faces): TetMeshCollision per mesh
        self.surface_inds = []
        self.settings = dict(timestep_s=1.0 / 24.0, admm_iters=10, gravity=-9.8, linsolver=0, constraint_w=-1.0)

    def add_tet_mesh(self, verts, tets, lame, kind, density=1522.0):
        off = self.x.shape[0]
        self.x = np.concatenate([self.x, verts])
        self.m = np.concatenate([self.m, meshes.lumped_masses_tets(verts, tets, density)])
        self.tets.append((verts, tets, lame, kind, off))
        return off

    def add_self_collision(self, verts, tets, off):
        """binding::add_tetmesh without NOSELFCOLLISION (samples/utils/AddMeshes.hpp:124-138)."""
        faces = meshes.surface_faces(tets)
        self.dynamic.append(dict(offset=off, rest=np.array(verts, dtype=np.float64), tets=np.asarray(tets, np.int32), faces=faces))
        self.surface_inds.extend(int(i) + off for i in np.unique(faces))

    def add_tri_mesh(self, verts, tris, lame, density=1.0):
        off = self.x.shape[0]
        self.x = np.concatenate([self.x, verts])
        self.m = np.concatenate([self.m, meshes.lumped_masses_tris(verts, tris, density)])
        self.tris.append((verts, tris, lame, off))
        return off

    def masses3(self):
        return np.repeat(self.m, 3)

    # ---- product ----
    def make_solver(self, init=True, **gpu_kw):
        s = Solver()
        s.add_nodes(self.x, self.masses3())
        for verts, tets, lame, kind, off in self.tets:
            s.add_tets(verts, tets, lame, kind, vertex_offset=off)
        for verts, tris, lame, off in self.tris:
            s.add_tris(verts, tris, lame, vertex_offset=off)
        for verts, tris, k_bend, off in self.bends:
            s.add_bends(verts, tris, k_bend, vertex_offset=off)
        if self.pins:
            s.set_pins(list(self.pins.keys()), [self.pins[k] for k in self.pins])
        if self.slides:
            s.set_slide_pins(list(self.slides.keys()), [self.slides[k][0] for k in self.slides], [self.slides[k][1] for k in self.slides])
        for kind, par in self.obstacles:
            s.add_obstacle(Floor(par[0]) if kind == 0 else Sphere(par[:3], par[3]))
        for d in self.dynamic:
            s.add_dynamic_collider(TetMeshCollision(d["rest"], d["tets"], d["faces"], d["offset"]))
        s.surface_inds = list(self.surface_inds)
        st = Settings(**self.settings)
        for k, v in gpu_kw.items():
            setattr(st, k, v)
        self.product_settings = st
        if init:
            assert s.initialize(st)
        else:
            s._settings = st
        return s

    # ---- oracle ----
    def make_oracle(self, mode=1, gs_colors=None, big=False, **kw):
        tets = tris = None
        if self.tets:
            idx = np.concatenate([t[1] + t[4] for t in self.tets]).astype(np.int32)
            kind = np.concatenate([np.full(len(t[1]), t[3], np.int32) for t in self.tets])
            mu = np.concatenate([np.full(len(t[1]), t[2].mu) for t in self.tets])
            la = np.concatenate([np.full(len(t[1]), t[2].lambda_) for t in self.tets])
            tets = dict(idx=idx, verts=self.x, kind=kind, mu=mu, la=la)
        if self.tris:
            idx = np.concatenate([t[1] + t[3] for t in self.tris]).astype(np.int32)
            mu = np.concatenate([np.full(len(t[1]), t[2].mu) for t in self.tris])
            la = np.concatenate([np.full(len(t[1]), t[2].lambda_) for t in self.tris])
            lmin = np.concatenate([np.full(len(t[1]), t[2].limit_min) for t in self.tris])
            lmax = np.concatenate([np.full(len(t[1]), t[2].limit_max) for t in self.tris])
            tris = dict(idx=idx, verts=self.x, mu=mu, la=la, limit_min=lmin, limit_max=lmax)
        bends = None
        if self.bends:      # hinges by the oracle's own (numpy) restatement of the stencil
            hs = [orc.bend_hinges(verts, tris) + (k_bend, off) for verts, tris, k_bend, off in self.bends]
            stiff = np.concatenate([k_bend * 3.0 / h[2] for h in hs for k_bend in [h[3]]])
            bends = dict(idx=np.concatenate([h[0] + h[4] for h in hs]), coef=np.concatenate([h[1] for h in hs]), weight=np.sqrt(stiff), stiffness=stiff)
            kw = dict(kw, bends=bends)
        if self.slides:
            kw = dict(kw, slides=self.slides)
        st = self.settings
        return orc.OracleSolver(self.x, self.masses3(), dt=st["timestep_s"], gravity=st["gravity"],
                                admm_iters=st["admm_iters"], linsolver=st["linsolver"], constraint_w=st["constraint_w"],
                                tets=tets, tris=tris, pins=self.pins, obstacles=self.obstacles, mode=mode,
                                gs_colors=gs_colors, big=big, dynamic=self.dynamic, surface_inds=self.surface_inds, **kw)


def cube_scene(n, kind, lame=None, pin_face=True, size=1.0, **settings):
    """Kuhn cube, x=0 face pinned (SURVEY 8d config 2/3 shape)."""
    sc = Scene()
    verts, tets = meshes.kuhn_cube(n, size)
    sc.add_tet_mesh(verts, tets, lame or Lame.soft_rubber(), kind)
    if pin_face:
        for i in np.nonzero(verts[:, 0] < 1e-9)[0]:
            sc.pins[int(i)] = verts[i].copy()
    sc.settings.update(settings)
    return sc


def mixed_cube_scene(n, **settings):
    """StVK / Neo-Hookean alternating by z-slab + a linear slab (config 3 shape)."""
    sc = Scene()
    verts, tets = meshes.kuhn_cube(n)
    cz = verts[tets].mean(axis=1)[:, 2]
    slab = np.minimum((cz * 3).astype(int), 2)
    sc.x = verts; sc.m = meshes.lumped_masses_tets(verts, tets)
    lame = Lame.soft_rubber()
    for s, kind in ((0, pkg.TET_NEOHOOKEAN), (1, pkg.TET_STVK), (2, pkg.TET_LINEAR)):
        sel = tets[slab == s]
        if len(sel):
            sc.tets.append((verts, sel, lame, kind, 0))
    for i in np.nonzero(verts[:, 0] < 1e-9)[0]:
        sc.pins[int(i)] = verts[i].copy()
    sc.settings.update(settings)
    return sc


def cloth_scene(m, limits=(0.95, 1.05), floor=None, **settings):
    """config 5 shape: m x m cloth, Lame(100, 0.1) with strain limits, two corner pins."""
    sc = Scene()
    verts, tris = meshes.cloth_grid(m, size=1.0, y=0.5)
    lame = Lame(100.0, 0.1)
    if limits:
        lame.limit_min, lame.limit_max = limits
    sc.add_tri_mesh(verts, tris, lame)
    for i in (0, m):  # two corners of the x=0 edge
        sc.pins[int(i)] = verts[i].copy()
    if floor is not None:
        sc.obstacles.append((0, [floor, 0.0, 0.0, 0.0]))
    sc.settings.update(settings)
    return sc


def two_blocks_scene(n=3, overlap=0.2, floor=-0.4, kind=None, jitter=0.02, **settings):
    """Two n-cell cubes with self-collision proxies, the upper one pushed `overlap` into the lower one (so the very
    first detect finds dynamic hits) and shifted sideways; optional floor under the lower one.  UzawaCG (the
    reference's torus.cpp set-up: dynamic mesh + Floor, linsolver 2).  Rest vertices are jittered so that no query
    point is equidistant from two surface triangles (the index tie rules would otherwise be decided by round-off)."""
    sc = Scene()
    rng = np.random.default_rng(5)
    lame = Lame(1.0e6, 0.3)
    for b, shift in enumerate(((0.0, 0.0, 0.0), (0.13, 1.0 - overlap, 0.07))):
        verts, tets = meshes.kuhn_cube(n)
        verts = verts + (jitter / n) * rng.uniform(-1.0, 1.0, verts.shape) + np.asarray(shift)
        off = sc.add_tet_mesh(verts, tets, lame, pkg.TET_LINEAR if kind is None else kind)
        sc.add_self_collision(verts, tets, off)
    if floor is not None:
        sc.obstacles.append((0, [floor, 0.0, 0.0, 0.0]))
    sc.settings.update(linsolver=2, admm_iters=5)
    sc.settings.update(settings)
    return sc


def perturb(x, amp, seed=0):
    rng = np.random.default_rng(seed)
    return x + amp * rng.standard_normal(x.shape)


def rel_err(a, b, x_ref=None):
    """max_v |a_v - b_v| / bbox_diag (SURVEY 8d parity metric)."""
    a = np.asarray(a).reshape(-1, 3); b = np.asarray(b).reshape(-1, 3)
    ref = b if x_ref is None else np.asarray(x_ref).reshape(-1, 3)
    diag = np.linalg.norm(ref.max(axis=0) - ref.min(axis=0))
    return np.linalg.norm(a - b, axis=1).max() / max(diag, 1e-300)


def bodies_scene(cells, kinds=None, gap=1.6, **settings):
    """Several SEPARATE bodies in one Solver (the reference's boxes.cpp / beams.cpp shape): Kuhn cubes of `cells[i]` cells per edge
    side by side along x, each with its own constitutive model, each pinned on its x-min face.  The system matrix is block
    diagonal by body -- the scene the component-aware multi-GPU partition cuts for free."""
    sc = Scene()
    kinds = kinds or [pkg.TET_NEOHOOKEAN, pkg.TET_STVK, pkg.TET_LINEAR]
    for i, n in enumerate(cells):
        verts, tets = meshes.kuhn_cube(n)
        verts = verts + np.array([gap * i, 0.0, 0.0])
        off = sc.add_tet_mesh(verts, tets, Lame.soft_rubber(), kinds[i % len(kinds)])
        for v in np.nonzero(verts[:, 0] < gap * i + 1e-9)[0]:
            sc.pins[int(v) + off] = verts[v].copy()
    sc.settings.update(settings)
    return sc


def blob_scene(n, jitter=0.15, seed=0, order="rcm", **settings):
    """BASELINE configs[2] on an UNSTRUCTURED body: meshes.unstructured_blob (valences 3..26, not 2-colourable, no exact
    zeros in Ahat), numbered randomly like a mesh file and renumbered for locality as the samples do; Neo-Hookean / StVK
    by z-slab, soft rubber, the feet (y < 0.1) pinned."""
    sc = Scene()
    verts, tets = meshes.unstructured_blob(n, jitter=jitter, seed=seed)
    verts, tets, _ = meshes.renumber_for_locality(verts, tets, force=True, method=order)
    cz = verts[tets].mean(axis=1)[:, 2]
    slab = (cz * 8).astype(int) % 2
    sc.x = verts; sc.m = meshes.lumped_masses_tets(verts, tets)
    lame = Lame.soft_rubber()
    sc.tets.append((verts, tets[slab == 0], lame, pkg.TET_NEOHOOKEAN, 0))
    sc.tets.append((verts, tets[slab == 1], lame, pkg.TET_STVK, 0))
    for i in np.nonzero(verts[:, 1] < 0.1)[0]:
        sc.pins[int(i)] = verts[i].copy()
    sc.settings.update(settings)
    return sc
