"""Multi-GPU sharding of the map -> Cl path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm).

The (field, bin) maps of a job are dealt to the ranks by descending cost (a spin-2 map costs three
spin-0 transforms).  Every rank transforms its own maps straight into its slice of ONE buffer, the
one exchange step of the path is an in-place all-gather of that buffer (every cross pair needs both
partners), and the upper triangle of map pairs is cut into tiles that are dealt to the ranks by
descending size, so that a rank reads ~ nmaps / sqrt(world) distinct alms instead of all of them.
The small Cl blocks are collected on rank 0.  With world == 1 nothing is copied or communicated.
"""

from __future__ import annotations

import math

import numpy as np


def map_cost(spin):
    """Relative cost of one analysis transform: F2 = 3 F0 (two Wigner functions, twice the columns per map)."""
    return 3 if spin else 1


def assign_maps(spins, world):
    """Owner rank of every map: longest-processing-time-first on the transform cost; ties go to the
    lowest rank, equal costs keep their order.  Returns a list of ranks."""
    order = sorted(range(len(spins)), key=lambda g: (-map_cost(spins[g]), g))
    load = [0] * world
    owner = [0] * len(spins)
    for g in order:
        r = min(range(world), key=lambda q: (load[q], q))
        owner[g] = r
        load[r] += map_cost(spins[g])
    return owner


def map_pairs(nmaps):
    """(i, j), i <= j, in the order of itertools.combinations_with_replacement."""
    return [(i, j) for i in range(nmaps) for j in range(i, nmaps)]


def tile_pairs(nmaps, world):
    """Block-cyclic cut of the upper triangle of map pairs: the maps are grouped into ~ 2 sqrt(world) tiles,
    every (tile, tile') block with tile <= tile' is one unit of work.  Returns [(tile_i, tile_j, [map pairs])]."""
    ntile = min(nmaps, max(1, math.ceil(2.0 * math.sqrt(world)))) if world > 1 else 1
    edges = [nmaps * t // ntile for t in range(ntile + 1)]
    out = []
    for a in range(ntile):
        for b in range(a, ntile):
            ps = [(i, j) for i in range(edges[a], edges[a + 1]) for j in range(edges[b], edges[b + 1]) if i <= j]
            if ps:
                out.append((a, b, ps))
    return out


class ShardedTwoPoint:
    """All auto/cross spectra of a set of maps whose transforms are sharded over the ranks.

    spins: spin (0 or 2) of every map of the job, in the job's global order -- identical on all ranks."""

    def __init__(self, spins, world, rank, nlm, lmax, kernel=None, group=None):
        self.spins = [int(s) for s in spins]
        self.world, self.rank, self.nlm, self.lmax, self.group = world, rank, nlm, lmax, group
        if kernel is None:
            from .twopoint import alm2cl_pairs as kernel
        self.kernel = kernel
        nmaps = len(self.spins)
        self.owner = assign_maps(self.spins, world)
        ncomp = lambda g: 2 if self.spins[g] else 1  # noqa: E731
        # per rank: its maps, spin-0 first (one batched transform per spin), and their component slots
        self.maps_of = [[g for g in range(nmaps) if self.owner[g] == r and self.spins[g] == 0] +
                        [g for g in range(nmaps) if self.owner[g] == r and self.spins[g] != 0] for r in range(world)]
        self.ncomp_of = [sum(ncomp(g) for g in ms) for ms in self.maps_of]
        self.ncomp_max = max(self.ncomp_of) if self.ncomp_of else 0
        self.slot = {}  # map -> first global component index in the gather buffer
        for r, ms in enumerate(self.maps_of):
            c = r * self.ncomp_max
            for g in ms:
                self.slot[g] = c
                c += ncomp(g)
        self.comps_of_map = {g: list(range(self.slot[g], self.slot[g] + ncomp(g))) for g in range(nmaps)}
        # global output rows: map pairs in combinations_with_replacement order, component block row-major
        self.pairs = map_pairs(nmaps)
        self.row0 = {}
        n = 0
        for (i, j) in self.pairs:
            self.row0[i, j] = n
            n += ncomp(i) * ncomp(j)
        self.nrows = n
        # tiles of map pairs, dealt by descending size
        tiles = tile_pairs(nmaps, world)
        size = lambda t: sum(ncomp(i) * ncomp(j) for i, j in t[2])  # noqa: E731
        order = sorted(range(len(tiles)), key=lambda k: (-size(tiles[k]), k))
        load = [0] * world
        self.pairs_of = [[] for _ in range(world)]
        for k in order:
            r = min(range(world), key=lambda q: (load[q], q))
            load[r] += size(tiles[k])
            self.pairs_of[r] += tiles[k][2]
        self.rows_of = [[self.row0[p] + k for p in ps for k in range(ncomp(p[0]) * ncomp(p[1]))] for ps in self.pairs_of]
        self.my_pairs = self.pairs_of[rank]
        self.my_cpairs = [(a, b) for (i, j) in self.my_pairs for a in self.comps_of_map[i] for b in self.comps_of_map[j]]
        self._buf = None

    # -- local transforms write here -------------------------------------------------
    @property
    def local_maps(self):
        """Global indices of this rank's maps: spin-0 maps first, then spin-2 maps."""
        return self.maps_of[self.rank]

    def buffer(self, device=None):
        """(world * ncomp_max, nlm) complex128: slice r * ncomp_max ... holds the alms of rank r's maps."""
        import torch

        if self._buf is not None and device is not None:
            want = torch.device(device)
            if want.type != self._buf.device.type or (want.index is not None and want.index != self._buf.device.index):
                self._buf = None  # asked for another device
        if self._buf is None:
            self._buf = torch.zeros((self.world * self.ncomp_max, self.nlm), dtype=torch.complex128, device=device)
        return self._buf

    def local_alm_views(self, device=None):
        """(alm0, alm2): views into this rank's slice of the gather buffer with shapes (n0, nlm) and (n2, 2, nlm):
        map2alm writes its result straight into the buffer that is then all-gathered."""
        buf = self.buffer(device)
        n0 = sum(1 for g in self.local_maps if self.spins[g] == 0)
        n2 = len(self.local_maps) - n0
        base = self.rank * self.ncomp_max
        a0 = buf[base : base + n0]
        a2 = buf[base + n0 : base + n0 + 2 * n2].view(n2, 2, self.nlm)
        return a0, a2

    # -- the exchange step -------------------------------------------------------------
    def exchange(self):
        """In-place all-gather of the alm shards (RCCL over xGMI; over gloo -- CPU tests, one-GPU rehearsals --
        device tensors go through host copies, because gloo gathers host memory only)."""
        if self.world == 1:
            return
        import torch
        import torch.distributed as dist

        buf = self.buffer()
        if buf.is_cuda:
            # the transforms ran on libhxsht's own stream (possibly asynchronously, hx_set_async): they must have written
            # this rank's slice before the collective reads it
            from . import _lib

            _lib.synchronize()
        flat = torch.view_as_real(buf)  # complex dtypes are gathered through their real view (same bytes)
        mine = flat[self.rank * self.ncomp_max : (self.rank + 1) * self.ncomp_max]
        if dist.get_backend(self.group) == "gloo":
            host = flat.new_empty(flat.shape, device="cpu")
            dist.all_gather_into_tensor(host, mine.cpu().contiguous(), group=self.group)
            flat.copy_(host)
        else:
            if buf.is_cuda:
                torch.cuda.current_stream(buf.device).synchronize()
            dist.all_gather_into_tensor(flat, mine, group=self.group)  # in place: mine is flat's own slice
        if buf.is_cuda:
            # libhxsht launches on its own stream: the gathered shards must have landed first
            torch.cuda.current_stream(buf.device).synchronize()

    # -- all pairs -----------------------------------------------------------------------
    def all_pairs_cl(self):
        """After the local transforms: exchange, compute this rank's tiles of map pairs, collect on rank 0.
        Returns on rank 0 the array (n_component_pairs_total, lmax+1) ordered by map pair
        (combinations_with_replacement order) then component block; None elsewhere."""
        import torch

        self.exchange()
        buf = self.buffer()
        comps = [buf[k] for k in range(buf.shape[0])]
        mine = self.kernel(comps, self.my_cpairs, self.lmax) if self.my_cpairs else np.zeros((0, self.lmax + 1))
        mine = np.ascontiguousarray(mine, dtype=np.float64)
        if self.world == 1:
            out = np.empty((self.nrows, self.lmax + 1))
            out[self.rows_of[0]] = mine
            return out
        import torch.distributed as dist

        nmax = max(len(r) for r in self.rows_of)
        gl = dist.get_backend(self.group) == "gloo"
        dev = "cpu" if gl else buf.device
        send = torch.zeros((nmax, self.lmax + 1), dtype=torch.float64, device=dev)
        if mine.shape[0]:
            send[: mine.shape[0]] = torch.from_numpy(mine).to(dev)
        recv = torch.empty((self.world * nmax, self.lmax + 1), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(recv, send, group=self.group)
        if self.rank != 0:
            return None
        recv = recv.cpu().numpy().reshape(self.world, nmax, self.lmax + 1)
        out = np.empty((self.nrows, self.lmax + 1))
        for r in range(self.world):
            out[self.rows_of[r]] = recv[r, : len(self.rows_of[r])]
        return out
