"""CPU-only checks of the product's host side: the C-ABI library loads and exports every symbol
include/admm_hip.h declares, the set-up arithmetic (rest poses, Lame, A assembly, colouring,
partition) matches the oracle, and the hot path fails loudly when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

import admm_elastic_amd as pkg
from admm_elastic_amd import capi, meshes
from admm_elastic_amd.solver import Lame, Settings, Solver
from oracle import oracle as orc
import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "admm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(admm_(?:hip|host)_[a-z_0-9]+)\s*\(", hdr))
    bound = {s[0] for s in capi.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    L = C.CDLL(capi.lib_path)
    for name in declared:
        assert hasattr(L, name), name
    assert C.sizeof(capi.Desc) > 0 and capi.lib().admm_hip_device_count() >= 0


def test_no_gpu_means_loud_failure():
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    sc = scenes.cube_scene(2, pkg.TET_LINEAR)
    s = sc.make_solver(init=False)
    with pytest.raises(pkg.AdmmHipError) as e:
        s.initialize(sc.product_settings)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)


def test_rest_pose_and_lame_match_oracle():
    verts, tets = meshes.kuhn_cube(3, 0.7)
    verts = scenes.perturb(verts, 0.01, 1)
    B, vol = capi.tet_rest(verts, tets)
    Bo, volo = orc.tet_rest(verts, tets)
    assert np.allclose(B, Bo, rtol=1e-12, atol=1e-12) and np.allclose(vol, volo, rtol=1e-13)
    assert np.all(vol > 0)
    v2, tris = meshes.cloth_grid(4)
    v2 = scenes.perturb(v2, 0.01, 2)
    R, area = capi.tri_rest(v2, tris)
    Ro, areao = orc.tri_rest(v2, tris)
    assert np.allclose(R, Ro, rtol=1e-12, atol=1e-12) and np.allclose(area, areao, rtol=1e-13)
    assert np.allclose(capi.lame(1e7, 0.399), orc.lame(1e7, 0.399), rtol=1e-15)
    bad = tets.copy(); bad[0, [2, 3]] = bad[0, [3, 2]]
    with pytest.raises(pkg.AdmmHipError) as e:
        capi.tet_rest(verts, bad)
    assert e.value.code == -3 and "Inverted initial tet" in str(e.value)


@pytest.mark.parametrize("ls", [0, 1])
def test_assembled_matrix_matches_oracle(ls):
    sc = scenes.mixed_cube_scene(3, linsolver=ls)
    v2, tris = meshes.cloth_grid(3, 1.0, 1.5)
    sc.add_tri_mesh(v2, tris, Lame(100.0, 0.1))
    s = sc.make_solver(init=False)
    rp, ci, va = s.host_matrix(sc.product_settings)
    nv = sc.x.shape[0]
    Ah = sp.csr_matrix((va, ci, rp), shape=(nv, nv))
    o = sc.make_oracle()
    A = (sp.kron(Ah, sp.identity(3)) + sp.diags(sc.masses3())).tocsr()
    assert abs(A - o.A).max() <= 1e-12 * abs(o.A).max()
    assert o.R == 9 * o.nt + 6 * o.ntri + 6 * o.npin
    assert o.D.nnz == 36 * o.nt + 18 * o.ntri + 3 * o.npin
    assert np.all(np.diff(ci.reshape(-1)[rp[0]:rp[1]]) > 0)     # sorted columns


def test_greedy_coloring_is_valid():
    sc = scenes.cube_scene(4, pkg.TET_NEOHOOKEAN, linsolver=1)
    s = sc.make_solver(init=False)
    rp, ci, va = s.host_matrix(sc.product_settings)
    col, nc = capi.greedy_coloring(rp, ci)
    assert col.min() == 0 and col.max() == nc - 1
    for i in range(len(rp) - 1):
        nb = ci[rp[i]:rp[i + 1]]
        assert np.all(col[nb[nb != i]] != col[i])


def test_partition_covers_everything():
    for n in (1, 7, 1000, 998250):
        for w in (1, 2, 4, 8):
            spans = [capi.partition(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1


def test_bad_descriptions_are_rejected():
    sc = scenes.cube_scene(2, pkg.TET_LINEAR)
    sc.obstacles.append((0, [0.0, 0, 0, 0]))
    s = sc.make_solver(init=False)
    with pytest.raises(pkg.AdmmHipError) as e:      # Solver.cpp:249-254
        s.host_matrix(sc.product_settings)
    assert "No collisions with LDLT solver" in str(e.value)
    lame = Lame(100.0, 0.1); lame.limit_min = 1.5
    with pytest.raises(pkg.AdmmHipError):            # TriEnergyTerm.cpp:32
        Solver().add_tris(*meshes.cloth_grid(2), lame)
    assert Solver().initialize(Settings()) is False  # Solver.cpp:180-183


def test_locality_order_is_a_permutation_that_shrinks_the_span():
    """admm_host_locality_order (reverse Cuthill-McKee): a randomly numbered cube gets a numbering at least as local as the
    generator's; meshes.renumber_for_locality renumbers consistently and leaves a good numbering alone."""
    verts, tets = meshes.kuhn_cube(12)
    nv = len(verts)
    _, lex, _ = capi.locality_order(nv, tets)
    p = np.random.default_rng(0).permutation(nv)
    inv = np.empty(nv, np.int64); inv[p] = np.arange(nv)
    v2, t2 = verts[inv], p[tets].astype(np.int32)
    new_id, before, after = capi.locality_order(nv, t2)
    assert sorted(int(i) for i in new_id) == list(range(nv)) and before > 3 * lex and after <= 1.05 * lex, (lex, before, after)
    v3, t3, nid = meshes.renumber_for_locality(v2, t2)
    assert np.array_equal(v3[t3], v2[t2]) and np.array_equal(nid, new_id)
    v4, t4, nid4 = meshes.renumber_for_locality(verts, tets)
    assert v4 is verts or np.array_equal(v4, verts)
    assert np.array_equal(nid4, np.arange(nv))
    tris = meshes.cloth_grid(10)[1]
    nid_t, _, _ = capi.locality_order(121, tris)                     # triangles too
    assert sorted(nid_t) == list(range(121))
    # the hierarchical block order (admm_host_block_order): a permutation; consecutive runs of `leaf` vertices are compact
    # (few distinct neighbour leaves), which is what bounds the active window of the per-vertex gathers
    nid_b = capi.block_order(nv, t2, 64)
    assert sorted(int(i) for i in nid_b) == list(range(nv))
    leaf_of = nid_b // 64
    e = np.concatenate([t2[:, [a, b]] for a in range(4) for b in range(a + 1, 4)])
    pairs = np.unique(np.sort(leaf_of[e], axis=1), axis=0)
    nbr = np.bincount(pairs[pairs[:, 0] != pairs[:, 1]].ravel(), minlength=leaf_of.max() + 1)
    assert nbr.max() <= 26 and nbr.mean() < 14, (nbr.max(), nbr.mean())
    v5, t5, nid5 = meshes.renumber_for_locality(v2, t2, force=True, method="blocks", leaf=64)
    assert np.array_equal(v5[t5], v2[t2]) and np.array_equal(nid5, nid_b)


@pytest.mark.parametrize("case", ["cube", "blob", "shuffled", "tiny", "kinds"])
def test_chunk_reduction_plan_sums_every_corner_force_once(case):
    """The host-built plan of the local step's block-level reduction (chunks of 256 tets -> records of <= 8 corner forces ->
    per-vertex lists of records), run on the host as the kernels run it, equals the plain scatter-add of the corner forces:
    every (tet, corner) is counted exactly once, in chunks that straddle nothing, with padding that adds nothing."""
    rng = np.random.default_rng(3)
    if case == "cube":
        verts, tets = meshes.kuhn_cube(9)
    elif case == "blob":
        verts, tets = meshes.unstructured_blob(14)
    elif case == "shuffled":        # vertex-incoherent tets: a chunk sees ~1000 distinct vertices -> several 256-record passes
        verts, tets = meshes.kuhn_cube(9)
        tets = rng.integers(0, len(verts), tets.shape).astype(np.int32)
    elif case == "tiny":
        verts, tets = meshes.kuhn_cube(1)
    else:
        verts, tets = meshes.kuhn_cube(7)
    nt, nv = len(tets), len(verts)
    if case in ("cube", "blob"):    # the order the library puts the tets of one model in: by lowest vertex, then index sum
        tets = tets[np.lexsort((tets.sum(1), tets.min(1)))]
    if case == "kinds":             # five constitutive-model groups of uneven sizes, one of them empty
        cuts = sorted(rng.choice(np.arange(1, nt), 3, replace=False).tolist())
        kb = [0, cuts[0], cuts[1], cuts[1], cuts[2], nt]
    else:
        kb = [0, 0, nt, nt, nt, nt]
    cf = rng.standard_normal((nt, 4, 3))
    got, st = capi.chunk_reduce(nv, tets, kb, cf)
    want = np.zeros((nv, 3))
    np.add.at(want, tets.reshape(-1), cf.reshape(-1, 3))
    assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
    n_chunks = sum((kb[k + 1] - kb[k] + 255) // 256 for k in range(5))
    assert st["chunks"] == n_chunks and st["records"] >= len(np.unique(tets))
    if case in ("cube", "blob"):    # vertex-coherent tets: one or two passes per chunk, a handful of records per vertex
        assert st["max_passes"] <= 2 and st["records"] < 0.3 * 4 * nt and st["max_list"] <= 16, st
    if case == "shuffled":
        assert st["max_passes"] > 1, st


def test_spline_tables_reproduce_the_spline():
    """admm_host_tabulate_spline / admm_host_spline_table_eval (the device evaluates the same interpolant, csrc/device_math.hpp:
    spline_table_eval): a user-defined xu::Spline (here xu::StVK with a compression term, re-stated) is reproduced by its tables --
    value to 1e-11, first derivative to 1e-9, second (from central differences of the first) to 1e-5, relative to the spline's
    stiffness, over the whole table range; a quadratic continues the function outside; bad ranges and non-finite samples refuse."""
    import ctypes as C
    import pytest
    L = capi.lib()
    mu, la, kap = 3.0e5, 7.0e5, 2.0e5
    fns = [lambda s: la * (s ** 4 - 6 * s * s + 5) / 8 + mu * (s * s - 1) ** 2 / 4, lambda p: la * (p * p - 1) / 4, lambda J: kap * ((1 - J) / 6) ** 3 / 12,
           lambda s: la * (s ** 3 - 3 * s) / 2 + mu * s * (s * s - 1), lambda p: la * p / 2, lambda J: -kap * ((1 - J) / 6) ** 2 / 24]
    d2 = [lambda s: la * (3 * s * s - 3) / 2 + mu * (3 * s * s - 1), lambda p: la / 2, lambda J: kap * ((1 - J) / 6) / 72]
    cb = capi.SPLINE_FN(lambda u, w, x: float(fns[w](x)))
    tab = np.zeros(capi.SPLINE_TABLE_DOUBLES)
    capi.check(L.admm_host_tabulate_spline(cb, None, 0.02, 50.0, capi.dptr(tab)))
    out = np.zeros(3); worst = np.zeros(3)
    rng = np.random.default_rng(0)
    for which in range(3):
        for x in np.exp(rng.uniform(np.log(0.02 ** (which + 1)), np.log(50.0 ** (which + 1)), 1500)):
            L.admm_host_spline_table_eval(capi.dptr(tab), which, float(x), capi.dptr(out))
            ref = np.array([fns[which](x), fns[which + 3](x), d2[which](x)])
            worst = np.maximum(worst, np.abs(out - ref) / (np.abs(ref) + mu))
    assert worst[0] < 1e-11 and worst[1] < 1e-9 and worst[2] < 1e-5, worst
    L.admm_host_spline_table_eval(capi.dptr(tab), 0, 60.0, capi.dptr(out))          # beyond the table: C1 continuation, finite
    assert np.isfinite(out).all() and abs(out[1] - fns[3](60.0)) < 0.2 * abs(fns[3](60.0))
    with pytest.raises(capi.AdmmHipError):
        capi.check(L.admm_host_tabulate_spline(cb, None, 2.0, 1.0, capi.dptr(tab)))
    bad = capi.SPLINE_FN(lambda u, w, x: float("nan"))
    with pytest.raises(capi.AdmmHipError):
        capi.check(L.admm_host_tabulate_spline(bad, None, 0.02, 50.0, capi.dptr(tab)))


def test_rest_positions_behind_the_tets_binv():
    """admm_host_tet_rest_positions (what admm_hip_create decides the local step's Binv source with): the mesh's own positions
    are accepted as they are; without them (or with a deformed candidate) positions propagated from tet to tet reproduce every
    Binv; tets that do not come from one set of positions are refused."""
    verts, tets = meshes.unstructured_blob(14)[:2]
    Binv, _ = capi.tet_rest(verts, tets)
    mode, x0 = capi.tet_rest_positions(len(verts), tets, Binv, verts)
    assert mode == 1 and np.array_equal(x0, verts)
    for cand in (None, verts * 1.05 + 0.01 * np.random.default_rng(0).standard_normal(verts.shape)):
        mode, x0 = capi.tet_rest_positions(len(verts), tets, Binv, cand)
        assert mode == 2
        B2, _ = capi.tet_rest(x0, tets)
        assert np.abs(B2 - Binv).max() <= 1e-12 * np.abs(Binv).max()
        used = np.unique(tets)
        d = x0[used] - verts[used]                       # one translation for the (single) component
        assert np.abs(d - d[0]).max() < 1e-12
    # two separate bodies: a translation each
    v2 = np.vstack([verts, verts + 3.0]); t2 = np.vstack([tets, tets + len(verts)]).astype(np.int32)
    B2, _ = capi.tet_rest(v2, t2)
    mode, x0 = capi.tet_rest_positions(len(v2), t2, B2, None)
    assert mode == 2 and np.abs(capi.tet_rest(x0, t2)[0] - B2).max() <= 1e-12 * np.abs(B2).max()
    # one tet built from other rest positions than its neighbours (a pre-strained element): no common rest state
    Bb = Binv.copy(); Bb[7] *= 1.001
    assert capi.tet_rest_positions(len(verts), tets, Bb, verts)[0] == 0
    assert capi.tet_rest_positions(len(verts), tets, Bb, None)[0] == 0
