"""Two ranks on the ONE GPU of the test box (collectives over gloo through host copies): the real
N > 1 path of bench.py / ShardedTwoPoint -- maps dealt by cost, map2alm writing straight into the gather buffer,
all-gather, tiled pair partition, the HIP alm2cl kernel on every rank, gather of the Cl blocks -- must return
exactly the single-process result.  RCCL itself needs one GPU per rank and is exercised by the driver's
multi-GPU run only."""

import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SPINS = [0, 0, 0, 2, 2, 2, 2]  # the strong-scaling shape in small: a fixed job dealt to the ranks
NSIDE, LMAX = 32, 64


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _map_of(g, spin):
    rng = np.random.default_rng(300 + g)
    return rng.standard_normal(((2,) if spin else ()) + (12 * NSIDE**2,))


def _transform_local(work, plan):
    import torch

    a0, a2 = work.local_alm_views("cuda")
    m0 = [_map_of(g, 0) for g in work.local_maps if work.spins[g] == 0]
    m2 = [_map_of(g, 2) for g in work.local_maps if work.spins[g] == 2]
    if m0:
        plan.map2alm(torch.as_tensor(np.stack(m0)).cuda(), 0, out=a0)
    if m2:
        plan.map2alm(torch.as_tensor(np.stack(m2)).cuda().view(2 * len(m2), -1), 2, out=a2.view(2 * len(m2), -1))


def _worker(rank, world, port, outdir):
    import torch
    import torch.distributed as dist

    import heracles_amd as hx
    from heracles_amd.distributed import ShardedTwoPoint

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    hx.init(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = hx.Plan(NSIDE, LMAX)
    work = ShardedTwoPoint(SPINS, world, rank, plan.nlm, LMAX)
    for _ in range(2):  # the second step reuses the gather buffer
        _transform_local(work, plan)
        res = work.all_pairs_cl()
    if rank == 0:
        np.save(os.path.join(outdir, "sharded.npy"), res)
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()
    plan.close()


def test_two_ranks_one_gpu_equal_single_process(tmp_path):
    import torch.multiprocessing as mp

    import heracles_amd as hx
    from heracles_amd.distributed import ShardedTwoPoint

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "sharded.npy")
    plan = hx.get_plan(NSIDE, LMAX)
    one = ShardedTwoPoint(SPINS, 1, 0, plan.nlm, LMAX)
    _transform_local(one, plan)
    ref = one.all_pairs_cl()
    assert got.shape == ref.shape == (3 * 4 // 2 + 3 * 4 * 2 + 4 * 5 // 2 * 4, LMAX + 1)
    # a map's sweep differs with the number of maps a rank holds (other kernel variant, other summation order): not bitwise
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max())


# ---- m-sharded route: hx_ring_modes -> (all-to-all) -> hx_legendre_from_modes on an m-range -> partial Cl -> all-reduce ----
def _all_maps():
    import torch

    m0 = [_map_of(g, 0) for g, s in enumerate(SPINS) if s == 0]
    m2 = [_map_of(g, 2) for g, s in enumerate(SPINS) if s == 2]
    return torch.as_tensor(np.stack(m0)).cuda(), torch.as_tensor(np.stack(m2)).cuda()


@pytest.mark.parametrize("nside,lmax", [(32, 64), (256, 400)])
def test_modes_then_legendre_on_order_sets_equals_map2alm(nside, lmax):
    """The two halves of the transform as the m-sharded route uses them, in ONE process: the mode blocks of hx_ring_modes for the
    orders dealt cyclically to 5 virtual ranks (and for uneven contiguous ranges, step 1), then hx_legendre_from_modes set by set
    into one alm buffer, give the alms of hx_map2alm (same kernels, same operands: bit for bit).  At nside 256 polar pruning is
    active and the scratch budget cuts every set into several m-chunks."""
    import torch

    import heracles_amd as hx
    from heracles_amd.distributed import HipStages, order_sets

    rng = np.random.default_rng(11)
    npix = 12 * nside**2
    plan = hx.get_plan(nside, lmax)
    st = HipStages(plan)
    pw = torch.as_tensor(1.0 + 1e-2 * rng.standard_normal(npix)).cuda()
    rw = torch.as_tensor(1.0 + 1e-2 * rng.standard_normal(2 * nside)).cuda()
    bounds = [0, 1, 7, lmax // 3, lmax - 2, lmax + 1]
    contiguous = [(bounds[q], bounds[q + 1] - bounds[q], 1) for q in range(5)]
    for sets in (order_sets(lmax, 5), contiguous, order_sets(lmax, 8)[:3] + order_sets(lmax, 8)[3:]):
        for spin, ncomp in ((0, 5), (2, 6), (0, 1), (2, 2), (0, 10)):
            maps = torch.as_tensor(rng.standard_normal((ncomp, npix))).cuda()
            ref = plan.map2alm(maps, spin, pix_weights=pw, ring_weights=rw)
            blocks = st.ring_modes(maps, sets, pix_weights=pw, ring_weights=rw)
            alm = st.zeros_alm(ncomp, plan.nlm)
            if nside >= 256:
                hx._lib.set_scratch_budget(4e6)
            try:
                for q, orders in enumerate(sets):
                    size = st.modes_size(orders[1])
                    st.legendre(spin, [blocks[q][c * size : (c + 1) * size] for c in range(ncomp)], orders, alm)
            finally:
                hx._lib.set_scratch_budget(0)
            np.testing.assert_array_equal(alm.cpu().numpy(), ref.cpu().numpy())


def _ms_worker(rank, world, port, outdir):
    import torch
    import torch.distributed as dist

    import heracles_amd as hx
    from heracles_amd.distributed import HipStages, MShardedTwoPoint

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    hx.init(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = hx.Plan(NSIDE, LMAX)
    work = MShardedTwoPoint(SPINS, world, rank, plan.nlm, LMAX, HipStages(plan))
    m0 = [_map_of(g, 0) for g in work.local_maps if SPINS[g] == 0]
    m2 = [_map_of(g, 2) for g in work.local_maps if SPINS[g] == 2]
    t0 = torch.as_tensor(np.stack(m0)).cuda() if m0 else torch.empty((0, 12 * NSIDE**2), dtype=torch.float64, device="cuda")
    t2 = torch.as_tensor(np.stack(m2)).cuda() if m2 else torch.empty((0, 2, 12 * NSIDE**2), dtype=torch.float64, device="cuda")
    for _ in range(2):
        res = work.run(t0, t2)
    np.save(os.path.join(outdir, f"ms_{rank}.npy"), res)
    dist.barrier()
    dist.destroy_process_group()
    plan.close()


@pytest.mark.parametrize("world", [2, 3])
def test_m_sharded_ranks_on_one_gpu_equal_single_process(tmp_path, world):
    """MShardedTwoPoint with the HIP stages on 2 / 3 ranks that share the one GPU (all-to-all and all-reduce over gloo through
    host copies): every rank ends with the spectra of the single-process job."""
    import torch.multiprocessing as mp

    import heracles_amd as hx
    from heracles_amd.distributed import ShardedTwoPoint

    mp.spawn(_ms_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    plan = hx.get_plan(NSIDE, LMAX)
    one = ShardedTwoPoint(SPINS, 1, 0, plan.nlm, LMAX)
    _transform_local(one, plan)
    ref = one.all_pairs_cl()
    for r in range(world):
        got = np.load(tmp_path / f"ms_{r}.npy")
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max())


def test_bench_rehearsal_two_ranks_verifies_itself(tmp_path):
    """``python3 bench.py --gpus 2`` WITHOUT a launcher, as the driver's N > 1 command may be shaped: the parent starts its own two
    ranks before anything touches the GPU (bench.self_launch); here they SHARE the one GPU over gloo (HX_BENCH_SHARE_GPU=1), at a small
    size.  The line must be the one a scaling curve can be built from -- ``value`` on the SAME job as N = 1 (pairs = nmaps (nmaps + 1) / 2
    of ONE set of maps, ``scaling: strong``), both fixed-job routes timed, the growing job beside it as ``value_weak`` -- and it must
    carry ``verify_multi``: the m-sharded route's spectra against the all-gather route's rows for the same seeded maps (1e-10) and map
    pairs against direct sums over the gathered alms, ``verified: true``."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HX_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--nside", "128", "--lmax", "160", "--nbins", "3", "--no-cpu-baseline", "--no-mixmat", "--no-single", "--no-host-leg"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "strong"
    assert out["config"]["pairs"] == 6 * 7 // 2 and out["config"]["maps_total"] == 6      # the N = 1 job, not 12 maps / 78 pairs
    assert out["routes"]["m_sharded"]["value"] > 0 and out["routes"]["all_gather"]["value"] > 0
    assert out["routes"]["m_sharded"]["pairs"] == out["routes"]["all_gather"]["pairs"] == 21
    assert out["value"] == max(out["routes"]["m_sharded"]["value"], out["routes"]["all_gather"]["value"]) == out["value_strong"]
    assert out["strong_scaling"]["route"] in ("m_sharded", "all_gather")
    assert out["weak_scaling"]["pairs"] == 12 * 13 // 2 and out["value_weak"] > 0 and out["weak_scaling"]["cl_vs_direct_sum"]["ok"]
    vm = out["verify_multi"]
    assert vm["ok"], vm
    assert vm["routes"]["m_sharded_vs_all_gather_max_err_over_max"] <= 1e-10
    assert vm["cl_vs_direct_sum"]["max_err_over_max"] <= 1e-11
    assert out["verified"] is True


def test_bench_under_torchrun_still_works(tmp_path):
    """The driver's documented N > 1 command (torch.distributed.run around bench.py) takes the same path: WORLD_SIZE is set, so no
    self-launch happens."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HX_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--nside", "64", "--lmax", "96", "--nbins", "2", "--no-cpu-baseline", "--no-mixmat", "--no-single", "--no-host-leg"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["config"]["pairs"] == 10 and out["verified"] is True


@pytest.mark.gpu
def test_bench_line_survives_a_failure_of_the_growing_job(tmp_path):
    """The job that grows with N is the side figure of an N > 1 line.  If it fails on every rank alike (HX_BENCH_FAIL_WEAK=1 stands in for a
    collective that has never run over RCCL), the fixed job -- ``value`` -- must still be timed, verified and printed; the line says what
    happened to the other one."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HX_BENCH_SHARE_GPU="1", HX_BENCH_FAIL_WEAK="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--nside", "64", "--lmax", "96", "--nbins", "2", "--no-cpu-baseline", "--no-mixmat", "--no-single", "--no-host-leg"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["config"]["pairs"] == 10 and out["value"] > 0
    assert out["value_weak"] is None and "HX_BENCH_FAIL_WEAK" in out["weak_scaling"]["error"]
    assert out["verify_multi"]["ok"] and out["verified"] is True
    assert out["roofline"]["launches"] > 0  # the per-GPU kernel figures come from the local transforms


@pytest.mark.gpu
def test_allgather_alms_runs_over_rccl_with_a_one_rank_communicator():
    """``hx_allgather_alms`` (the all-gather of alm shards as a C entry point, for hosts without torch.distributed): the
    only thing one GPU can show is that the entry point binds RCCL, takes a communicator the HOST created and runs its grouped
    broadcasts on the library's stream -- a one-rank communicator, whose gather leaves the buffer as it is.  More ranks need more GPUs
    (RCCL refuses two ranks on one device): unmeasured on hardware, like every N > 1 path here."""
    import ctypes as C

    import torch

    import heracles_amd as hx
    from heracles_amd import _lib

    hx.init(0)
    try:
        rccl = C.CDLL("librccl.so.1")
    except OSError:
        rccl = C.CDLL("librccl.so")
    comm = C.c_void_p()
    assert rccl.ncclCommInitAll(C.byref(comm), 1, (C.c_int * 1)(0)) == 0
    try:
        buf = torch.randn(12345, dtype=torch.complex128, device="cuda")
        ref = buf.clone()
        L = _lib.load()
        _lib.check(L.hx_allgather_alms(comm, 1, (C.c_int64 * 1)(buf.numel()), _lib.ptr(buf)))
        torch.cuda.synchronize()
        assert torch.equal(buf, ref)
        # an empty shard is skipped; a host buffer and a null communicator are refused
        _lib.check(L.hx_allgather_alms(comm, 1, (C.c_int64 * 1)(0), _lib.ptr(buf)))
        with pytest.raises(hx.HxError):
            _lib.check(L.hx_allgather_alms(comm, 1, (C.c_int64 * 1)(4), _lib.ptr(np.zeros(4, dtype=np.complex128))))
        with pytest.raises(hx.HxError):
            _lib.check(L.hx_allgather_alms(None, 1, (C.c_int64 * 1)(4), _lib.ptr(buf)))
    finally:
        rccl.ncclCommDestroy(comm)
