"""Synthetic mesh generators for the BASELINE configs and lumped masses.

The reference's generators (mcl::factory::make_tet_blocks / make_plane, TetMesh::weighted_masses)
live in the absent mclscene submodule; these are written from their described behaviour.
"""
import itertools

import numpy as np


def tet_blocks(nx, ny, nz, size=None, origin=(0.0, 0.0, 0.0)):
    """Kuhn triangulation: 6 tets per cell of an nx x ny x nz grid, all positively oriented
    (the reference throws on negative rest volume, src/TetEnergyTerm.cpp:42-44).
    Returns verts [nv,3] float64, tets [nt,4] int32.  size: (sx,sy,sz) extents (default = cell counts)."""
    gx, gy, gz = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1), np.arange(nz + 1), indexing="ij")
    verts = np.stack([gx, gy, gz], axis=-1).reshape(-1, 3).astype(np.float64)

    def vid(i, j, k):
        return (i * (ny + 1) + j) * (nz + 1) + k

    ci, cj, ck = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    ci, cj, ck = ci.ravel(), cj.ravel(), ck.ravel()
    tets = []
    for perm in itertools.permutations(range(3)):
        p = [np.stack([ci, cj, ck], axis=1)]
        for ax in perm:
            q = p[-1].copy()
            q[:, ax] += 1
            p.append(q)
        ids = [vid(a[:, 0], a[:, 1], a[:, 2]) for a in p]
        # orientation of (v1-v0, v2-v0, v3-v0) is the parity of the permutation
        e = np.zeros((3, 3))
        cum = np.zeros(3)
        for c, ax in enumerate(perm):
            cum = cum.copy(); cum[ax] += 1
            e[:, c] = cum
        if np.linalg.det(e) < 0:
            ids[2], ids[3] = ids[3], ids[2]
        tets.append(np.stack(ids, axis=1))
    # interleave so the 6 tets of a cell are contiguous (cell-major order = good locality)
    tets = np.stack(tets, axis=1).reshape(-1, 4).astype(np.int32)
    if size is not None:
        verts = verts * (np.asarray(size, dtype=np.float64) / np.array([nx, ny, nz], dtype=np.float64))
    verts = verts + np.asarray(origin, dtype=np.float64)
    return verts, tets


def kuhn_cube(n, size=1.0):
    """Unit cube, n cells per edge: nt = 6 n^3, nv = (n+1)^3 (n=26 -> 105 456 tets, n=55 -> 998 250)."""
    return tet_blocks(n, n, n, size=(size, size, size))


def cloth_grid(m, size=1.0, y=0.0):
    """m x m cells in the xz-plane at height y, two triangles (a,b,c),(a,c,d) per cell."""
    g = np.arange(m + 1)
    gx, gz = np.meshgrid(g, g, indexing="ij")
    verts = np.stack([gx * (size / m), np.full_like(gx, y, dtype=np.float64), gz * (size / m)], axis=-1).reshape(-1, 3)
    verts = verts.astype(np.float64)
    ci, ck = np.meshgrid(np.arange(m), np.arange(m), indexing="ij")
    ci, ck = ci.ravel(), ck.ravel()
    a = ci * (m + 1) + ck; b = (ci + 1) * (m + 1) + ck; c = (ci + 1) * (m + 1) + ck + 1; d = ci * (m + 1) + ck + 1
    tris = np.stack([np.stack([a, b, c], 1), np.stack([a, c, d], 1)], axis=1).reshape(-1, 3).astype(np.int32)
    return verts, tris


def tet_volumes(verts, tets):
    v0, v1, v2, v3 = (verts[tets[:, i]] for i in range(4))
    return np.einsum("ij,ij->i", np.cross(v1 - v0, v2 - v0), v3 - v0) / 6.0


def surface_faces(tets):
    """Triangles that belong to exactly one tet, outward for positively oriented tets (mcl::TetMesh::need_faces,
    used by TetMeshCollision, src/DynamicObject.hpp:48-57).  Returns [nf,3] int32, sorted by their vertex triple."""
    tets = np.asarray(tets, dtype=np.int64).reshape(-1, 4)
    loc = np.array([[0, 2, 1], [0, 1, 3], [1, 2, 3], [0, 3, 2]])
    tri = tets[:, loc].reshape(-1, 3)
    key = np.sort(tri, axis=1)
    _, inv, cnt = np.unique(key, axis=0, return_inverse=True, return_counts=True)
    keep = cnt[inv.ravel()] == 1
    out = tri[keep]
    order = np.lexsort(np.sort(out, axis=1).T[::-1])
    return out[order].astype(np.int32)


def surface_inds(tets):
    """Vertices on the surface (mcl::TetMesh::surface_inds, samples/utils/AddMeshes.hpp:131-137)."""
    return np.unique(surface_faces(tets)).astype(np.int32)


def lumped_masses_tets(verts, tets, density=1522.0):
    """rho * vol / 4 to every corner (AddMeshes.hpp:113-122 uses density 1522 for tets). Returns [nv]."""
    vol = tet_volumes(verts, tets)
    m = np.zeros(verts.shape[0])
    np.add.at(m, tets.ravel(), np.repeat(density * vol / 4.0, 4))
    return m


def lumped_masses_tris(verts, tris, density=1.0):
    v0, v1, v2 = (verts[tris[:, i]] for i in range(3))
    area = 0.5 * np.linalg.norm(np.cross(v1 - v0, v2 - v0), axis=1)
    m = np.zeros(verts.shape[0])
    np.add.at(m, tris.ravel(), np.repeat(density * area / 3.0, 3))
    return m


def renumber_for_locality(verts, elems, force=False, method="rcm", leaf=256):
    """Vertex numbering with locality for a mesh whose numbering has none: method "rcm" = capi.locality_order (reverse
    Cuthill-McKee), "blocks" = capi.block_order (compact leaves from a recursive graph bisection: the smaller active window
    for the gathers on unstructured meshes).  Returns (verts, elems, new_id) renumbered when the mean edge span shrinks by more
    than 2x (or force), else unchanged with new_id = identity.  Everything indexed by vertex (pins, surface lists) goes through
    new_id."""
    from . import capi
    verts = np.asarray(verts); elems = np.asarray(elems, dtype=np.int32)
    new_id, before, after = capi.locality_order(len(verts), elems)
    if method == "blocks":
        new_id = capi.block_order(len(verts), elems, leaf)
        after = 0.0 if force else after
    if not force and not (after < 0.5 * before):
        return verts, elems, np.arange(len(verts), dtype=np.int32)
    out = np.empty_like(verts); out[new_id] = verts
    return out, new_id[elems].astype(np.int32), new_id


def load_tetgen(node_path, ele_path):
    """TetGen .node/.ele reader (the reference's samples/data format, 0- or 1-indexed)."""
    with open(node_path) as f:
        rows = [ln.split() for ln in f if ln.strip() and not ln.lstrip().startswith("#")]
    nv = int(rows[0][0])
    body = rows[1:1 + nv]
    first = int(body[0][0])
    verts = np.array([[float(r[1]), float(r[2]), float(r[3])] for r in body])
    with open(ele_path) as f:
        rows = [ln.split() for ln in f if ln.strip() and not ln.lstrip().startswith("#")]
    nt = int(rows[0][0])
    tets = np.array([[int(r[1]), int(r[2]), int(r[3]), int(r[4])] for r in rows[1:1 + nt]], dtype=np.int32) - first
    return verts, tets


# ---- unstructured synthetic body (BASELINE configs[2]: "1M-tet synthetic bunny/dragon") -------------------------------
# corner c of a lattice cell = (c & 1, (c >> 1) & 1, (c >> 2) & 1); for every corner the three cell faces that do NOT
# contain it, each as its four corners in cyclic order
def _pull_faces():
    faces = np.zeros((8, 3, 4), dtype=np.int64)
    for c in range(8):
        for a in range(3):
            side = 1 - ((c >> a) & 1)
            b, d = (a + 1) % 3, (a + 2) % 3
            cyc = []
            for (ub, ud) in ((0, 0), (1, 0), (1, 1), (0, 1)):
                cyc.append((side << a) | (ub << b) | (ud << d))
            faces[c, a] = cyc
    return faces


def _blob_inside(p):
    """Implicit bunny-like body in the unit box: union of ellipsoids (body, head, two ears, tail, two feet)."""
    parts = [  # centre, radii
        ((0.45, 0.36, 0.50), (0.30, 0.24, 0.26)),   # body
        ((0.70, 0.60, 0.50), (0.15, 0.15, 0.15)),   # head
        ((0.72, 0.82, 0.41), (0.045, 0.17, 0.06)),  # ears
        ((0.72, 0.82, 0.59), (0.045, 0.17, 0.06)),
        ((0.14, 0.36, 0.50), (0.08, 0.08, 0.08)),   # tail
        ((0.60, 0.14, 0.36), (0.16, 0.07, 0.08)),   # feet
        ((0.60, 0.14, 0.64), (0.16, 0.07, 0.08)),
    ]
    inside = np.zeros(len(p), dtype=bool)
    for c, r in parts:
        q = (p - np.asarray(c)) / np.asarray(r)
        inside |= np.einsum("ij,ij->i", q, q) <= 1.0
    return inside


def unstructured_blob(n, jitter=0.15, seed=0, shuffle=True):
    """Deterministic unstructured tet mesh: an n^3 lattice over the unit box, vertices jittered by up to `jitter` cells,
    every cell split into 6 tets by a *pulling* triangulation on a random global vertex priority (the cell's
    lowest-priority corner is coned to the three faces that do not contain it, every face is split along the diagonal
    through ITS lowest-priority corner -- a face's split depends only on the global priorities, so neighbouring cells
    agree and the mesh conforms), and only the cells inside an implicit bunny-like body kept.  Vertex valences then range
    from 4 to 26 like a TetGen mesh (a Kuhn lattice has 14 everywhere, 8 of them exact-zero couplings), element shapes
    vary, nothing in the system matrix cancels and the graph is not 2-colourable.  shuffle: vertices and tets are
    randomly renumbered (a mesh file's order carries no locality; callers run renumber_for_locality like the samples do).
    n = 118 -> ~1.0 M tets.  Returns verts [nv,3], tets [nt,4] int32 (positively oriented)."""
    rng = np.random.default_rng(seed)
    g = np.arange(n + 1)
    gx, gy, gz = np.meshgrid(g, g, g, indexing="ij")
    lat = np.stack([gx, gy, gz], axis=-1).reshape(-1, 3).astype(np.float64)
    pos = (lat + jitter * (2.0 * rng.random(lat.shape) - 1.0)) / n
    prio = rng.permutation(len(lat))

    def vid(i, j, k):
        return (i * (n + 1) + j) * (n + 1) + k

    c = np.arange(n)
    ci, cj, ck = (a.ravel() for a in np.meshgrid(c, c, c, indexing="ij"))
    centre = (np.stack([ci, cj, ck], axis=1) + 0.5) / n
    keep = _blob_inside(centre)
    ci, cj, ck = ci[keep], cj[keep], ck[keep]
    corners = np.stack([vid(ci + (q & 1), cj + ((q >> 1) & 1), ck + ((q >> 2) & 1)) for q in range(8)], axis=1)  # [nc,8]
    cp = prio[corners]
    vstar = np.argmin(cp, axis=1)                                   # lowest-priority corner of the cell
    fl = _pull_faces()[vstar]                                       # [nc,3,4] local corner ids of the opposite faces
    fg = np.take_along_axis(corners[:, None, :].repeat(3, 1), fl, axis=2)   # global ids
    fstar = np.argmin(prio[fg], axis=2)                             # lowest-priority corner of every face
    roll = (fstar[:, :, None] + np.arange(4)[None, None, :]) % 4
    fg = np.take_along_axis(fg, roll, axis=2)                       # that corner first, cyclic order kept
    apex = np.take_along_axis(corners, vstar[:, None], axis=1)[:, 0]
    tets = []
    for f in range(3):
        for (b, d) in ((1, 2), (2, 3)):
            tets.append(np.stack([apex, fg[:, f, 0], fg[:, f, b], fg[:, f, d]], axis=1))
    tets = np.stack(tets, axis=1).reshape(-1, 4)
    vol = tet_volumes(pos, tets)
    flip = vol < 0
    tets[flip] = tets[flip][:, [0, 1, 3, 2]]
    vol = np.abs(vol)
    if not (vol.min() > 1e-3 / (6.0 * n ** 3)):
        raise ValueError("unstructured_blob: jitter too large (degenerate tet)")
    used = np.unique(tets)
    new = np.full(len(lat), -1, dtype=np.int64)
    if shuffle:
        new[used] = rng.permutation(len(used))
        tets = tets[rng.permutation(len(tets))]
    else:
        new[used] = np.arange(len(used))
    verts = np.empty((len(used), 3)); verts[new[used]] = pos[used]
    return verts, new[tets].astype(np.int32)


def valence_stats(nv, elems):
    """Vertex valences of a mesh (edges per vertex; the row of Ahat has valence + 1 entries): dict(min, mean, max)."""
    e = np.asarray(elems, dtype=np.int64)
    k = e.shape[1]
    pairs = np.concatenate([np.stack([e[:, a], e[:, b]], 1) for a in range(k) for b in range(a + 1, k)])
    pairs.sort(axis=1)
    pairs = np.unique(pairs, axis=0)
    deg = np.bincount(pairs.ravel(), minlength=nv)
    return dict(min=int(deg.min()), mean=float(deg.mean()), max=int(deg.max()))
