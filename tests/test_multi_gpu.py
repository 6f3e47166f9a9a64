"""The N>1 path: element-block partition, partial right-hand sides, one sum all-reduce per
ADMM iteration, replicated global solve.

  * CPU, world_size 2, gloo: the sharded algorithm (product partition + oracle arithmetic) reproduces
    the single-rank oracle step bit-for-bit up to summation order.
  * GPU (one device): two rank-contexts of a world of 2 produce partial RHS / disjoint z,u rows whose
    union equals the single-context result (everything but the RCCL call itself, which needs >1 GPU).
"""
import os
import socket

import numpy as np
import pytest

import admm_elastic_amd as pkg
from admm_elastic_amd import capi
import scenes


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, n, frames, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scenes.mixed_cube_scene(n, admm_iters=6, linsolver=0)
    o = sc.make_oracle(mode=1)            # full scene: used for the replicated global solve + matrix
    tb, te = capi.partition(o.nt, world, rank)   # the product's partition rule
    rows = slice(9 * tb, 9 * te)
    Dt = o.DtWtW[:, rows]
    for _ in range(frames):
        # Solver::step with the local step restricted to the owned element block
        o.v[1::3] += o.dt * o.gravity
        x_bar = o.x + o.dt * o.v
        Mxbar = o.m * x_bar
        curr = x_bar.copy()
        z = np.zeros(o.R); u = np.zeros(o.R)
        for _it in range(o.admm_iters):
            zz = np.ascontiguousarray(z[rows]); uu = np.ascontiguousarray(u[rows])
            from oracle import oracle as orc
            orc.lib().orc_local_tets(te - tb, orc._i(np.ascontiguousarray(o.t_idx[tb:te])), orc._p(np.ascontiguousarray(o.t_Binv[tb:te])),
                                     orc._i(np.ascontiguousarray(o.t_kind[tb:te])), orc._p(np.ascontiguousarray(o.t_mu[tb:te])),
                                     orc._p(np.ascontiguousarray(o.t_la[tb:te])), orc._p(np.ascontiguousarray(o.t_k[tb:te])),
                                     orc._p(curr), orc._p(zz), orc._p(uu), 1)
            z[rows] = zz; u[rows] = uu
            part = Dt @ (zz - uu)                       # partial dt^2 D^T W^2 (z - u) of this block
            # pin terms + M x_bar live on rank 0 only
            if rank == 0:
                pr = slice(9 * o.nt, o.R)
                zp = np.ascontiguousarray(z[pr]); up = np.ascontiguousarray(u[pr])
                orc.lib().orc_local_pins(o.npin, orc._i(o.p_vert), orc._p(o.p_xyz), orc._i(o.p_active), orc._p(curr), orc._p(zp), orc._p(up))
                z[pr] = zp; u[pr] = up
                part = part + o.DtWtW[:, pr] @ (zp - up) + Mxbar
            t = torch.from_numpy(part.copy())
            dist.all_reduce(t)                           # the one exchange step per ADMM iteration
            curr = o.solve_ldlt(t.numpy())               # replicated global solve
        o.v = (curr - o.x) / o.dt
        o.x = curr
    if rank == 0:
        q.put(o.x.copy())
    dist.destroy_process_group()


def test_sharded_step_matches_single_rank_gloo():
    import torch.multiprocessing as mp
    n, frames, world = 4, 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    x_sharded = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sc = scenes.mixed_cube_scene(n, admm_iters=6, linsolver=0)
    o = sc.make_oracle(mode=1)
    for _ in range(frames):
        o.step()
    assert scenes.rel_err(x_sharded, o.x) < 1e-9   # only the summation order of the partial RHS differs


@pytest.mark.gpu
def test_rank_contexts_partition_the_local_step():
    sc = scenes.mixed_cube_scene(5, linsolver=0)
    v2, tris = pkg.meshes.cloth_grid(4, 1.0, 1.3)
    sc.add_tri_mesh(v2, tris, pkg.Lame(100.0, 0.1))
    single = sc.make_solver()
    R = single.num_rows()
    rng = np.random.default_rng(3)
    x = scenes.perturb(sc.x, 0.02, 4).ravel()
    u0 = 0.02 * rng.standard_normal(R); u0[9 * len(single.flatten()["tet_idx"]) + 6 * len(tris):] = 0.0
    Mx = rng.standard_normal(x.size)
    z1, u1, b1 = single.local_step(x, u0, Mx)
    world = 3
    zs = np.zeros(R); us = u0.copy(); bs = np.zeros(x.size)
    touched = np.zeros(R, bool)
    for r in range(world):
        s = sc.make_solver(rank=r, world_size=world)     # no comm_init: local_step returns the PARTIAL b
        z, u, b = s.local_step(x, u0, Mx)
        own = np.abs(z) > 0
        assert not (touched & own)[: 9 * single._flat["tet_idx"].shape[0]].any()
        touched |= own
        zs += z
        us = np.where(u != u0, u, us)
        bs += b
        with pytest.raises(pkg.AdmmHipError):             # stepping a rank context needs the communicator
            s.upload(); s.step_device()
        s.close()
    nrow_el = 9 * single._flat["tet_idx"].shape[0] + 6 * len(tris)
    assert np.abs(zs[:nrow_el] - z1[:nrow_el]).max() < 1e-12
    assert np.abs(us[:nrow_el] - u1[:nrow_el]).max() < 1e-12
    assert np.abs(bs - b1).max() <= 1e-9 * np.abs(b1).max()


@pytest.mark.gpu
def test_rccl_allreduce_path_on_one_gpu():
    """The RCCL leg of the multi-GPU step on a single GPU: a communicator of world size 1 (ADMM_HIP_FORCE_COMM=1)
    makes every ADMM iteration run the in-place ncclAllReduce of the right-hand side on the context's stream,
    between the gather kernel and the persistent PCG kernel.  The all-reduce over one rank is the identity, so
    the trajectory must equal the plain single-GPU one bit for bit."""
    import ctypes as C
    from admm_elastic_amd import capi
    sc = scenes.mixed_cube_scene(6, admm_iters=6, linsolver=0)
    ref = sc.make_solver(pcg_tol=1e-10, pcg_max_iters=500)
    for _ in range(3):
        ref.step()
    s = sc.make_solver(pcg_tol=1e-10, pcg_max_iters=500)
    os.environ["ADMM_HIP_FORCE_COMM"] = "1"
    try:
        buf = C.create_string_buffer(128)
        capi.check(capi.lib().admm_hip_comm_unique_id(buf))
        capi.check(capi.lib().admm_hip_comm_init(s._ctx, bytes(buf.raw), 0, 1))
    finally:
        os.environ.pop("ADMM_HIP_FORCE_COMM", None)
    for _ in range(3):
        s.step()
    assert np.array_equal(s.m_x, ref.m_x)
    assert s.runtime_data().unconverged_solves == 0


def _rccl_worker(rank, world, port, n, frames, q):
    """One rank of a real multi-GPU run: own device, RCCL communicator, product contexts."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)     # only carries the 128-byte RCCL id
    sc = scenes.mixed_cube_scene(n, admm_iters=6, linsolver=0)
    s = sc.make_solver(device=rank, rank=rank, world_size=world, pcg_tol=1e-10, pcg_max_iters=500)
    s.comm_init(dist)
    for _ in range(frames):
        s.step()
    q.put((rank, s.m_x.copy(), s.runtime_data().unconverged_solves))
    dist.barrier()
    s.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_two_gpus_match_single_context():
    """A real world-2 step (element-block partition, ncclAllReduce of the partial right-hand side over xGMI between the
    gather kernel and the persistent PCG kernel, replicated solve) on two devices reproduces the single-context
    trajectory.  Needs two GPUs: skipped on the 1-GPU test boxes, run wherever the driver has a multi-GPU node."""
    if pkg.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import torch.multiprocessing as mp
    n, frames, world = 6, 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, n, frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sc = scenes.mixed_cube_scene(n, admm_iters=6, linsolver=0)
    ref = sc.make_solver(pcg_tol=1e-10, pcg_max_iters=500)
    for _ in range(frames):
        ref.step()
    for rank, x, unconv in res:
        assert unconv == 0
        assert scenes.rel_err(x, ref.m_x) < 1e-9, (rank, scenes.rel_err(x, ref.m_x))   # summation order of the partial RHS differs
    assert np.array_equal(res[0][1], res[1][1])      # the replicated solves see the same all-reduced right-hand side


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus N` ignored the flag (one rank, "n_gpus": 1).  It now starts N ranks itself
    when no launcher did (the driver's `python -m torch.distributed.run --nproc-per-node N` line), and a mismatch between
    --gpus and WORLD_SIZE is an error.  Without GPUs every rank fails loudly: the hot path has no CPU fallback."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if pkg.device_count() > 0:
        pytest.skip("CPU-only check of the launcher")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode != 0
    assert "rank 0 needs HIP device 0" in out and "rank 1 needs HIP device 1" in out, out[-2000:]     # two ranks were started
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in (r.stdout + r.stderr)
