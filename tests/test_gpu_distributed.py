"""Two ranks on the ONE GPU of the test box (collectives over gloo through host copies): the real
N > 1 path of bench.py / PairWork -- device-resident alm shards, all-gather, pair partition, the HIP
alm2cl kernel on every rank, gather of the Cl blocks -- must return exactly the single-process result.
RCCL itself needs one GPU per rank and is exercised by the driver's multi-GPU run only."""

import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make(rank, nbins, lmax, device):
    import torch

    nlm = (lmax + 1) * (lmax + 2) // 2
    g = torch.Generator().manual_seed(300 + rank)
    a0 = torch.view_as_complex(torch.randn((nbins, nlm, 2), dtype=torch.float64, generator=g)).contiguous()
    a2 = torch.view_as_complex(torch.randn((nbins, 2, nlm, 2), dtype=torch.float64, generator=g)).contiguous()
    return a0.to(device), a2.to(device)


def _worker(rank, world, port, nbins, lmax, outdir):
    import torch
    import torch.distributed as dist

    import heracles_amd as hx
    from heracles_amd.distributed import PairWork

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    hx.init(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nlm = (lmax + 1) * (lmax + 2) // 2
    a0, a2 = _make(rank, nbins, lmax, "cuda")
    work = PairWork(world, rank, nbins, nlm, lmax)
    for _ in range(2):  # the second call reuses the gather buffer
        res = work.all_pairs_cl(a0, a2)
    if rank == 0:
        np.save(os.path.join(outdir, "sharded.npy"), res)
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_one_gpu_equal_single_process(tmp_path):
    import torch.multiprocessing as mp

    import heracles_amd as hx
    from heracles_amd.distributed import comps_of_map, map_pairs

    world, nbins, lmax = 2, 3, 200
    nlm = (lmax + 1) * (lmax + 2) // 2
    mp.spawn(_worker, args=(world, _free_port(), nbins, lmax, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "sharded.npy")
    comps = []
    for r in range(world):
        a0, a2 = _make(r, nbins, lmax, "cpu")
        comps += [a0[k].numpy() for k in range(nbins)] + [a2.reshape(2 * nbins, nlm)[k].numpy() for k in range(2 * nbins)]
    nmaps = 2 * nbins * world
    cpairs = [(a, b) for (i, j) in map_pairs(nmaps) for a in comps_of_map(i, nbins) for b in comps_of_map(j, nbins)]
    ref = hx.twopoint.alm2cl_pairs(comps, cpairs, lmax)
    assert got.shape == ref.shape == (len(cpairs), lmax + 1)
    np.testing.assert_array_equal(got, ref)  # same kernel, same tiles per pair: bit-identical
