"""world_size-2 gloo test of the multi-GPU sharding logic on CPU: all-gather of alm shards,
pair partition, gather of the Cl blocks.  The arithmetic kernel is the oracle here (the HIP
kernel needs a GPU); what is tested is that the sharded job returns exactly the spectra of
the single-process job over all maps."""

import os
import socket

import numpy as np
import pytest


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _kernel(comps, plist, lmax):
    from oracle import hxoracle as ho

    out = np.zeros((len(plist), lmax + 1))
    for n, (i, j) in enumerate(plist):
        out[n] = ho.alm2cl(np.asarray(comps[i]), np.asarray(comps[j]), lmax=lmax)
    return out


def _make(rank, nbins, lmax):
    import torch

    nlm = (lmax + 1) * (lmax + 2) // 2
    g = torch.Generator().manual_seed(100 + rank)
    a0 = torch.randn((nbins, nlm, 2), dtype=torch.float64, generator=g)
    a2 = torch.randn((nbins, 2, nlm, 2), dtype=torch.float64, generator=g)
    return torch.view_as_complex(a0).contiguous(), torch.view_as_complex(a2).contiguous()


def _worker(rank, world, port, nbins, lmax, outdir):
    import torch.distributed as dist

    from heracles_amd.distributed import PairWork

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nlm = (lmax + 1) * (lmax + 2) // 2
    a0, a2 = _make(rank, nbins, lmax)
    work = PairWork(world, rank, nbins, nlm, lmax, kernel=_kernel)
    res = work.all_pairs_cl(a0, a2)
    if rank == 0:
        np.save(os.path.join(outdir, "sharded.npy"), res)
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_equals_single(tmp_path, world):
    import torch
    import torch.multiprocessing as mp

    from heracles_amd.distributed import PairWork, comps_of_map, map_pairs

    nbins, lmax = 2, 12
    nlm = (lmax + 1) * (lmax + 2) // 2
    mp.spawn(_worker, args=(world, _free_port(), nbins, lmax, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "sharded.npy")
    # single-process job over the same maps
    comps = []
    for r in range(world):
        a0, a2 = _make(r, nbins, lmax)
        comps += [a0[k].numpy() for k in range(nbins)] + [a2.reshape(2 * nbins, nlm)[k].numpy() for k in range(2 * nbins)]
    nmaps = 2 * nbins * world
    cpairs = [(a, b) for (i, j) in map_pairs(nmaps) for a in comps_of_map(i, nbins) for b in comps_of_map(j, nbins)]
    ref = _kernel(comps, cpairs, lmax)
    assert got.shape == ref.shape
    np.testing.assert_array_equal(got, ref)
    # world == 1 path: no communication, same answer for rank 0's maps
    a0, a2 = _make(0, nbins, lmax)
    one = PairWork(1, 0, nbins, nlm, lmax, kernel=_kernel).all_pairs_cl(a0, a2)
    n1 = 2 * nbins
    assert one.shape[0] == n1 * (n1 + 1) // 2 - nbins * (nbins + 1) // 2 + 4 * (nbins * (nbins + 1) // 2) + nbins * nbins
    assert torch.is_tensor(a0)


def test_partition_covers_all_pairs():
    from heracles_amd.distributed import PairWork

    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            w = PairWork(world, r, 3, 10, 3, kernel=_kernel)
            seen += w.my_pairs
            assert sum(w.counts) == sum(len(PairWork(world, q, 3, 10, 3, kernel=_kernel).my_cpairs) for q in range(world))
        assert sorted(seen) == sorted(PairWork(world, 0, 3, 10, 3, kernel=_kernel).pairs)
        assert len(seen) == len(set(seen))
