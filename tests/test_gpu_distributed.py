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
