"""Multi-GPU sharding of the map -> Cl path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm).

Each rank owns the alms of its own (field, bin) maps.  The one exchange step of the path
is an all-gather of the alm shards (every cross pair needs both partners); the list of map
pairs is then cut into contiguous slices, one per rank, and the small Cl blocks are
collected on rank 0.  With world == 1 nothing is copied or communicated.
"""

from __future__ import annotations

import numpy as np


def map_pairs(nmaps_total):
    return [(i, j) for i in range(nmaps_total) for j in range(i, nmaps_total)]


def comps_of_map(g, nbins):
    """Global component indices of global map g.  Per rank the component order is
    [spin-0 bin 0..nbins-1, (E,B) of spin-2 bin 0..nbins-1]; maps are ordered
    [spin-0 bins, spin-2 bins]."""
    r, k = divmod(g, 2 * nbins)
    base = r * 3 * nbins
    if k < nbins:
        return [base + k]
    k -= nbins
    return [base + nbins + 2 * k, base + nbins + 2 * k + 1]


def slice_of_rank(n, world, rank):
    lo = n * rank // world
    hi = n * (rank + 1) // world
    return lo, hi


class PairWork:
    """All auto/cross spectra of the maps held by all ranks."""

    def __init__(self, world, rank, nbins, nlm, lmax, kernel=None, group=None):
        self.world, self.rank, self.nbins, self.nlm, self.lmax = world, rank, nbins, nlm, lmax
        self.group = group
        if kernel is None:
            from .twopoint import alm2cl_pairs as kernel
        self.kernel = kernel
        self.nmaps_total = 2 * nbins * world
        self.pairs = map_pairs(self.nmaps_total)
        lo, hi = slice_of_rank(len(self.pairs), world, rank)
        self.my_pairs = self.pairs[lo:hi]
        # component pairs of my slice, and how many each rank produces (for the gather)
        self.my_cpairs = [(a, b) for (i, j) in self.my_pairs
                          for a in comps_of_map(i, nbins) for b in comps_of_map(j, nbins)]
        self.counts = []
        for r in range(world):
            l, h = slice_of_rank(len(self.pairs), world, r)
            self.counts.append(sum(len(comps_of_map(i, nbins)) * len(comps_of_map(j, nbins)) for i, j in self.pairs[l:h]))
        self._gather_buf = None

    def _all_gather(self, out, inp):
        """all_gather_into_tensor; over gloo (CPU tests, one-GPU rehearsals) device tensors go through
        host copies, because gloo gathers host memory only.  RCCL ("nccl") takes the device path."""
        import torch.distributed as dist

        if inp.is_cuda and dist.get_backend(self.group) == "gloo":
            host = out.new_empty(out.shape, device="cpu")
            dist.all_gather_into_tensor(host, inp.cpu(), group=self.group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, inp, group=self.group)

    def gathered_components(self, alm0, alm2):
        """List of all 3*nbins*world component arrays (views into the gather buffer)."""
        import torch

        nb, nlm = self.nbins, self.nlm
        if self.world == 1:
            a2 = alm2.reshape(2 * nb, nlm)
            return [alm0[k] for k in range(nb)] + [a2[k] for k in range(2 * nb)]
        import torch.distributed as dist

        local = torch.cat([alm0.reshape(nb, nlm), alm2.reshape(2 * nb, nlm)], dim=0).contiguous()
        if self._gather_buf is None or self._gather_buf.device != local.device:
            self._gather_buf = torch.empty((self.world * 3 * nb, nlm), dtype=local.dtype, device=local.device)
        # complex dtypes are gathered through their real view (same bytes)
        self._all_gather(torch.view_as_real(self._gather_buf), torch.view_as_real(local))
        if local.is_cuda:
            # libhxsht launches on its own stream: the gathered shards must have landed first
            torch.cuda.current_stream(local.device).synchronize()
        flat = self._gather_buf
        return [flat[k] for k in range(flat.shape[0])]

    def all_pairs_cl(self, alm0, alm2):
        """Returns on rank 0 the array (n_component_pairs_total, lmax+1) ordered by map pair
        (combinations_with_replacement order) then component block; None elsewhere."""
        import torch

        comps = self.gathered_components(alm0, alm2)
        mine = self.kernel(comps, self.my_cpairs, self.lmax)
        mine = np.ascontiguousarray(mine, dtype=np.float64)
        if self.world == 1:
            return mine
        import torch.distributed as dist

        dev = comps[0].device
        nmax = max(self.counts)
        send = torch.zeros((nmax, self.lmax + 1), dtype=torch.float64, device=dev)
        if mine.shape[0]:
            send[: mine.shape[0]] = torch.from_numpy(mine).to(dev)
        recv = torch.empty((self.world * nmax, self.lmax + 1), dtype=torch.float64, device=dev)
        self._all_gather(recv, send)
        if recv.is_cuda:
            torch.cuda.current_stream(recv.device).synchronize()
        recv = recv.reshape(self.world, nmax, self.lmax + 1)
        if self.rank != 0:
            return None
        recv = recv.cpu().numpy()
        return np.concatenate([recv[r, : self.counts[r]] for r in range(self.world)], axis=0)
