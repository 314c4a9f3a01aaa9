"""Multi-GPU sharding of the map -> Cl path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm).

The (field, bin) maps of a job are dealt to the ranks by descending cost (a spin-2 map costs three
spin-0 transforms).  Every rank transforms its own maps straight into its slices of ONE buffer, the
one exchange step of the path is an in-place all-gather of that buffer (every cross pair needs both
partners), and the upper triangle of map pairs is cut into tiles that are dealt to the ranks by
descending size, so that a rank reads ~ nmaps / sqrt(world) distinct alms instead of all of them.
The small Cl blocks are collected on rank 0.  With world == 1 nothing is copied or communicated.

The buffer is laid out spin by spin -- [spin-2 shards of all ranks | spin-0 shards of all ranks] -- so that
the exchange can be issued in two parts (round 4): ``exchange_begin(2)`` right after the spin-2 transform
(asynchronous: the spin-0 transform runs under it), ``exchange_begin(0)`` after the spin-0 transform, and
``all_pairs_cl`` computes the spin-2 x spin-2 blocks of its tiles as soon as the first part has landed, the
blocks with a spin-0 partner after the second.  ``exchange()`` (both parts, blocking) is the round-3 behaviour;
the gathered bytes are the same either way.
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
        self.n0_of = [sum(1 for g in ms if self.spins[g] == 0) for ms in self.maps_of]
        self.c2_of = [nc - n0 for nc, n0 in zip(self.ncomp_of, self.n0_of)]
        self.n0_max, self.c2_max = max(self.n0_of, default=0), max(self.c2_of, default=0)
        # rows of the gather buffer: region of spin 2 = rows [0, world c2_max), rank r's shard at r c2_max; region of spin 0 behind it
        self.region = {2: (0, self.c2_max), 0: (world * self.c2_max, self.n0_max)}
        self.nbuf_rows = world * (self.c2_max + self.n0_max)
        self.slot = {}  # map -> first global component index in the gather buffer
        for r, ms in enumerate(self.maps_of):
            c0, c2 = self.region[0][0] + r * self.n0_max, self.region[2][0] + r * self.c2_max
            for g in ms:
                if self.spins[g] == 0:
                    self.slot[g] = c0
                    c0 += 1
                else:
                    self.slot[g] = c2
                    c2 += 2
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
        if world == 1:
            self.pairs_of = [list(self.pairs)]  # one rank: the job's own order, so that its rows are the result as it stands
        self.rows_of = [[self.row0[p] + k for p in ps for k in range(ncomp(p[0]) * ncomp(p[1]))] for ps in self.pairs_of]
        self._rows_in_order = world == 1 and self.rows_of[0] == list(range(len(self.rows_of[0])))
        try:
            import inspect

            self._kernel_takes_out = "out" in inspect.signature(self.kernel).parameters
        except (TypeError, ValueError):
            self._kernel_takes_out = False
        self.my_pairs = self.pairs_of[rank]
        self.my_cpairs = [(a, b) for (i, j) in self.my_pairs for a in self.comps_of_map[i] for b in self.comps_of_map[j]]
        # the component pairs of this rank whose partners are both spin-2 (ready after the first part of the exchange), and the rest
        n2rows = world * self.c2_max
        self._first = [k for k, (a, b) in enumerate(self.my_cpairs) if a < n2rows and b < n2rows]
        self._second = [k for k, (a, b) in enumerate(self.my_cpairs) if not (a < n2rows and b < n2rows)]
        self._buf = None
        self._pending = {}

    # -- local transforms write here -------------------------------------------------
    @property
    def local_maps(self):
        """Global indices of this rank's maps: spin-0 maps first, then spin-2 maps."""
        return self.maps_of[self.rank]

    def buffer(self, device=None):
        """(world * (c2_max + n0_max), nlm) complex128: the spin-2 shards of all ranks (rank r at row r c2_max), then the spin-0 shards
        (rank r at row world c2_max + r n0_max); ``comps_of_map`` gives the rows of a map."""
        import torch

        if self._buf is not None and device is not None:
            want = torch.device(device)
            if want.type != self._buf.device.type or (want.index is not None and want.index != self._buf.device.index):
                self._buf = None  # asked for another device
        if self._buf is None:
            self._buf = torch.zeros((self.nbuf_rows, self.nlm), dtype=torch.complex128, device=device)
        return self._buf

    def local_alm_views(self, device=None):
        """(alm0, alm2): views into this rank's slice of the gather buffer with shapes (n0, nlm) and (n2, 2, nlm):
        map2alm writes its result straight into the buffer that is then all-gathered."""
        buf = self.buffer(device)
        n0, c2 = self.n0_of[self.rank], self.c2_of[self.rank]
        b0, b2 = self.region[0][0] + self.rank * self.n0_max, self.region[2][0] + self.rank * self.c2_max
        a0 = buf[b0 : b0 + n0]
        a2 = buf[b2 : b2 + c2].view(c2 // 2, 2, self.nlm)
        return a0, a2

    # -- the exchange step -------------------------------------------------------------
    def exchange_begin(self, spin):
        """Start the in-place all-gather of one spin's shards (RCCL over xGMI: asynchronous, on the communicator's stream -- the
        caller goes on to queue its next transform, which runs under the transfer; over gloo -- CPU tests, one-GPU rehearsals --
        device tensors go through host copies and the part is complete on return).  Call after that spin's transform was queued."""
        if self.world == 1 or spin in self._pending:
            return
        import torch
        import torch.distributed as dist

        row0, per = self.region[spin]
        if per == 0:
            self._pending[spin] = None
            return
        buf = self.buffer()
        if buf.is_cuda:
            # the transform ran on libhxsht's own stream (possibly asynchronously, hx_set_async): it must have written this rank's
            # shard before the collective reads it
            from . import _lib

            _lib.synchronize()
        flat = torch.view_as_real(buf[row0 : row0 + self.world * per])  # complex dtypes are gathered through their real view (same bytes)
        mine = flat[self.rank * per : (self.rank + 1) * per]
        if dist.get_backend(self.group) == "gloo":
            if buf.is_cuda:
                host = flat.new_empty(flat.shape, device="cpu")
                dist.all_gather_into_tensor(host, mine.cpu().contiguous(), group=self.group)
                flat.copy_(host)
                self._pending[spin] = None
            else:
                self._pending[spin] = dist.all_gather_into_tensor(flat, mine, group=self.group, async_op=True)
        else:
            if buf.is_cuda:
                torch.cuda.current_stream(buf.device).synchronize()
            self._pending[spin] = dist.all_gather_into_tensor(flat, mine, group=self.group, async_op=True)  # in place: mine is flat's own slice

    def exchange_wait(self, spin):
        """Block until the part started by ``exchange_begin(spin)`` has landed (starting it now if it was not)."""
        if self.world == 1:
            return
        import torch

        self.exchange_begin(spin)
        h = self._pending.pop(spin)
        if h is not None:
            h.wait()
        buf = self.buffer()
        if buf.is_cuda:
            # libhxsht launches on its own stream: the gathered shards must have landed first
            torch.cuda.current_stream(buf.device).synchronize()

    def exchange(self):
        """In-place all-gather of all alm shards, complete on return (both parts of the exchange)."""
        for spin in (2, 0):
            self.exchange_begin(spin)
        for spin in (2, 0):
            self.exchange_wait(spin)

    # -- all pairs -----------------------------------------------------------------------
    def all_pairs_cl(self, out=None):
        """After the local transforms: exchange, compute this rank's tiles of map pairs, collect on rank 0.
        Returns on rank 0 the array (n_component_pairs_total, lmax+1) ordered by map pair
        (combinations_with_replacement order) then component block; None elsewhere.
        out (one process, library kernel only): the destination, e.g. one page-locked array for every step of a loop."""
        import torch

        buf = self.buffer()
        comps = [buf[k] for k in range(buf.shape[0])]
        if self.world == 1 and self._rows_in_order and self._kernel_takes_out:
            self.exchange()
            return self.kernel(comps, self.my_cpairs, self.lmax, out=out)  # (the rows of the one rank are the result as it stands)
        if self.world == 1 or not self._first or not self._second:
            self.exchange()
            mine = self.kernel(comps, self.my_cpairs, self.lmax) if self.my_cpairs else np.zeros((0, self.lmax + 1))
            mine = np.ascontiguousarray(mine, dtype=np.float64)
        else:
            # the spin-2 x spin-2 blocks of this rank's tiles as soon as the spin-2 shards are here, the rest after the spin-0 shards
            self.exchange_begin(2)
            self.exchange_begin(0)
            mine = np.empty((len(self.my_cpairs), self.lmax + 1))
            self.exchange_wait(2)
            mine[self._first] = self.kernel(comps, [self.my_cpairs[k] for k in self._first], self.lmax)
            self.exchange_wait(0)
            mine[self._second] = self.kernel(comps, [self.my_cpairs[k] for k in self._second], self.lmax)
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


# ======================================================================================================================
# m-sharded route for a FIXED job (strong scaling): SURVEY.md 8e's "lower-traffic alternative"
# ======================================================================================================================
def order_sets(lmax, world):
    """The orders of every rank: rank q owns m = q, q + world, ...  Returns [(first, count, step)] -- the cost of an order falls
    steadily with m, so a cyclic deal balances the work to a fraction of a percent AND gives every rank work-groups of every
    length, as many as one GPU's launch / world (contiguous ranges of equal cost left the rank of the lowest orders with 317
    work-groups for 256 compute units: 108 ms against 78 on the other ranks of 8)."""
    return [(q, max(0, (lmax - q) // world + 1) if q <= lmax else 0, world) for q in range(world)]


def assign_maps_by_components(spins, world):
    """Owner rank of every map for the ring Fourier stage, whose cost is one unit per component: largest first, ties to the
    lowest rank."""
    ncomp = [2 if s else 1 for s in spins]
    order = sorted(range(len(spins)), key=lambda g: (-ncomp[g], g))
    load = [0] * world
    owner = [0] * len(spins)
    for g in order:
        r = min(range(world), key=lambda q: (load[q], q))
        owner[g] = r
        load[r] += ncomp[g]
    return owner


class HipStages:
    """The two halves of the transform on the GPU (hx_ring_modes / hx_legendre_from_modes of libhxsht)."""

    def __init__(self, plan, device="cuda"):
        self.plan, self.device = plan, device

    def modes_size(self, count):
        from . import _lib

        return int(_lib.load().hx_ring_modes_size(self.plan._h, int(count)))

    def send_buffer(self, ncomp, sets):
        """One flat float64 device tensor that holds the blocks of ``ring_modes(maps of ncomp components, sets)`` side by side, set after
        set: the send buffer of the exchange (``ring_modes(..., out=)`` writes into it, no copy in between)."""
        import torch

        return torch.empty(ncomp * sum(self.modes_size(c) for (_, c, _) in sets), dtype=torch.float64, device=self.device)

    def ring_modes(self, maps, sets, pix_weights=None, ring_weights=None, out=None):
        """maps (ncomp, npix) -> one float64 device tensor per set of orders (first, count, step): (ncomp * modes_size(count),).
        ``out`` (``send_buffer(ncomp, sets)``): the blocks are views of it, written in place by hx_ring_modes (per-set output pointers)."""
        import ctypes as C

        import torch

        from . import _lib

        ncomp = maps.shape[0]
        ns = len(sets)
        if out is None:
            outs = [torch.empty(ncomp * self.modes_size(c), dtype=torch.float64, device=self.device) for (_, c, _) in sets]
        else:
            sizes = [ncomp * self.modes_size(c) for (_, c, _) in sets]
            if out.numel() != sum(sizes) or out.dtype != torch.float64 or not out.is_contiguous():
                raise ValueError("ring_modes: out is not the send buffer of these sets")
            outs, o = [], 0
            for n_ in sizes:
                outs.append(out[o : o + n_])
                o += n_
        if ncomp == 0:
            return outs
        step = sets[0][2]
        assert all(s[2] == step for s in sets)
        first = (C.c_int * ns)(*[int(s[0]) for s in sets])
        count = (C.c_int * ns)(*[int(s[1]) for s in sets])
        ptrs = (C.c_void_p * ns)(*[o.data_ptr() for o in outs])
        maps = maps.contiguous() if hasattr(maps, "data_ptr") else np.ascontiguousarray(maps, dtype=np.float64)
        _lib.check(_lib.load().hx_ring_modes(self.plan._h, ncomp, _lib.ptr(maps), _lib.ptr(pix_weights), _lib.ptr(ring_weights), ns, first, count,
                                             int(step), ptrs))
        return outs

    def legendre(self, spin, blocks, orders, alm_out):
        """blocks: one float64 device tensor per component ([count][nrp_pad][4]); alm_out (ncomp, nlm) complex device tensor,
        written for the orders (first, count, step) only."""
        import ctypes as C

        from . import _lib

        if not blocks or orders[1] <= 0:
            return
        ptrs = (C.c_void_p * len(blocks))(*[b.data_ptr() for b in blocks])
        _lib.check(_lib.load().hx_legendre_from_modes(self.plan._h, int(spin), len(blocks), ptrs, int(orders[0]), int(orders[1]), int(orders[2]),
                                                      _lib.ptr(alm_out), None))

    def zeros_alm(self, ncomp, nlm):
        import torch

        return torch.zeros((ncomp, nlm), dtype=torch.complex128, device=self.device)

    def synchronize(self):
        from . import _lib

        _lib.synchronize()


class MShardedTwoPoint:
    """All auto/cross spectra of a FIXED set of maps on `world` ranks, sharded by the order m:

        ring Fourier stage of the rank's own maps  ->  all-to-all of the ring modes by owner of the order  ->  Legendre stage of
        EVERY component on the rank's orders (the full-batch kernels, 1/world of the orders)  ->  all-pairs Cl over the rank's
        orders  ->  all-reduce of the Cl blocks.

    Dealing the MAPS of a 20-map job to 8 ranks instead (ShardedTwoPoint) leaves two or three maps per rank: a batch shape the
    matrix kernels cannot fill.  Replaces the loops of heracles/mapping.py:151-172 and heracles/twopoint.py:198-215.

    spins: spin of every map of the job in its global order (identical on all ranks); stages: HipStages(plan) -- or any object
    with the same methods (the CPU tests pass an oracle-backed one)."""

    def __init__(self, spins, world, rank, nlm, lmax, stages, kernel=None, group=None):
        self.spins = [int(s) for s in spins]
        self.world, self.rank, self.nlm, self.lmax, self.group, self.stages = world, rank, nlm, lmax, group, stages
        if kernel is None:
            from .twopoint import alm2cl_pairs as kernel
        self.kernel = kernel
        nmaps = len(self.spins)
        self.owner = assign_maps_by_components(self.spins, world)
        # per rank: its maps, spin-0 first; the modes of a rank travel in that order, one block per component
        self.maps_of = [[g for g in range(nmaps) if self.owner[g] == r and self.spins[g] == 0] +
                        [g for g in range(nmaps) if self.owner[g] == r and self.spins[g] != 0] for r in range(world)]
        self.n0_of = [sum(1 for g in ms if self.spins[g] == 0) for ms in self.maps_of]
        self.n2_of = [len(ms) - n0 for ms, n0 in zip(self.maps_of, self.n0_of)]
        self.ncomp_of = [n0 + 2 * n2 for n0, n2 in zip(self.n0_of, self.n2_of)]
        # alm buffer: all spin-0 components (rank by rank), then all spin-2 components
        self.nc0, self.nc2 = sum(self.n0_of), 2 * sum(self.n2_of)
        self.comps_of_map = {}
        c0, c2 = 0, self.nc0
        for r, ms in enumerate(self.maps_of):
            for g in ms:
                if self.spins[g] == 0:
                    self.comps_of_map[g] = [c0]
                    c0 += 1
                else:
                    self.comps_of_map[g] = [c2, c2 + 1]
                    c2 += 2
        ncomp = lambda g: 2 if self.spins[g] else 1  # noqa: E731
        self.pairs = map_pairs(nmaps)
        self.row0, n = {}, 0
        for (i, j) in self.pairs:
            self.row0[i, j] = n
            n += ncomp(i) * ncomp(j)
        self.nrows = n
        self.cpairs = [(a, b) for (i, j) in self.pairs for a in self.comps_of_map[i] for b in self.comps_of_map[j]]
        self.sets = order_sets(lmax, world)      # (first, count, step) of every rank: the same for both spins
        self.orders = self.sets[rank]
        self._alm = None
        self._send = {}  # per part (spin 0 / spin 2): (ncomp, the send buffer of the exchange)
        import inspect

        try:  # (a kernel without the keyword sums over all orders: the alms are zero outside this rank's)
            self._kernel_takes_m_range = "m_range" in inspect.signature(self.kernel).parameters
        except (TypeError, ValueError):
            self._kernel_takes_m_range = False

    @property
    def local_maps(self):
        """Global indices of this rank's maps: spin-0 maps first, then spin-2 maps."""
        return self.maps_of[self.rank]

    def buffer(self):
        """(nc0 + nc2, nlm) alms of ALL components; only the orders of this rank are non-zero."""
        if self._alm is None:
            self._alm = self.stages.zeros_alm(self.nc0 + self.nc2, self.nlm)
        return self._alm

    def _all_to_all_begin(self, send_blocks, ncomp_of, flat=None):
        """Start the all-to-all of one part (the blocks of ``ncomp_of[s]`` components from every rank s); returns a token for
        ``_all_to_all_end``.  Over RCCL the transfer runs asynchronously on the communicator's stream; over gloo device blocks go through
        host copies (complete on return).  ``flat``: the one buffer the blocks are views of, side by side (``HipStages.send_buffer``) -- it is
        sent as it is; without it the blocks are concatenated first (stage objects that return separate blocks: the CPU tests').
        A part that is empty on EVERY rank (no map of that spin in the job) is not a collective at all.  ``HX_MSHARD_BLOCKING=1``: the
        blocking exchange of round 4 instead of the asynchronous one (an A/B switch for the first run over RCCL: the asynchronous path
        has only run over gloo and in one-GPU rehearsals -- RCCL parity of it is unpinned)."""
        import os

        import torch
        import torch.distributed as dist

        size = self.stages.modes_size(self.orders[1])
        sizes_out = [ncomp_of[s] * size for s in range(self.world)]
        if self.world == 1:
            return (send_blocks[0], None, sizes_out, None)
        if sum(ncomp_of) == 0:  # (identical on all ranks: everybody skips)
            return (torch.empty(0, dtype=torch.float64, device=send_blocks[0].device), None, sizes_out, None)
        gloo = dist.get_backend(self.group) == "gloo"
        on_dev = send_blocks[0].is_cuda
        send = flat if flat is not None else torch.cat([b.reshape(-1) for b in send_blocks])
        if gloo and on_dev:
            send = send.cpu()
        recv = torch.empty(sum(sizes_out), dtype=torch.float64, device=send.device)
        sizes_in = [int(b.numel()) for b in send_blocks]
        if gloo and on_dev:
            dist.all_to_all_single(recv, send, output_split_sizes=sizes_out, input_split_sizes=sizes_in, group=self.group)
            return (recv.to(send_blocks[0].device), None, sizes_out, None)
        if os.environ.get("HX_MSHARD_BLOCKING") == "1":
            dist.all_to_all_single(recv, send, output_split_sizes=sizes_out, input_split_sizes=sizes_in, group=self.group)
            if recv.is_cuda:
                torch.cuda.current_stream(recv.device).synchronize()
            return (recv, None, sizes_out, None)
        h = dist.all_to_all_single(recv, send, output_split_sizes=sizes_out, input_split_sizes=sizes_in, group=self.group, async_op=True)
        return (recv, h, sizes_out, send)  # (send is kept alive until the transfer has finished)

    def _all_to_all_end(self, token):
        """Block until the part has landed; returns the list recv[s]: what rank s sent to this rank."""
        import torch

        recv, h, sizes_out, _keep = token
        if h is not None:
            h.wait()
            if recv.is_cuda:
                torch.cuda.current_stream(recv.device).synchronize()  # libhxsht reads the blocks on its own stream
        out, o = [], 0
        for s in range(self.world):
            out.append(recv[o : o + sizes_out[s]])
            o += sizes_out[s]
        return out

    def _agree(self, err, guard):
        """End of a local phase: with ``guard`` every rank learns whether ANY rank failed in it (one tiny all-reduce) and all raise
        together -- a rank that raised alone would leave the others blocked in the next collective."""
        if not guard or self.world == 1:
            if err is not None:
                raise err
            return
        import torch
        import torch.distributed as dist

        gloo = dist.get_backend(self.group) == "gloo"
        flag = torch.tensor([0 if err is None else 1], dtype=torch.int32, device="cpu" if gloo else self.stages.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
        if int(flag.item()):
            raise err if err is not None else RuntimeError("MShardedTwoPoint: another rank failed in this phase")

    def run(self, maps0, maps2, pix_weights=None, ring_weights=None, guard=False):
        """maps0 (n0_local, npix), maps2 (n2_local, 2, npix): this rank's maps.  Returns on EVERY rank the array
        (n_component_pairs_total, lmax + 1) ordered by map pair (combinations_with_replacement order) then component block.
        ``guard``: agree on success across the ranks before each collective (see ``_agree``).

        The exchange goes out in two parts that overlap the compute (round 5): ring modes of the spin-0 maps -> all-to-all #1 (asynchronous)
        -> ring modes of the spin-2 maps, under #1 -> all-to-all #2 (asynchronous) -> Legendre stage of the spin-0 components, under #2 ->
        Legendre stage of the spin-2 components -> partial Cl -> all-reduce.  At N = 8 the two transfers (1.8 + 3.5 GB per rank) sit under
        ~9 and ~12 ms of kernels; one blocking exchange of everything was exposed in full.  The sums are those of the single exchange in the
        same order: the spectra are the same bit for bit."""
        n0, n2 = self.n0_of[self.rank], self.n2_of[self.rank]
        c2_of = [2 * v for v in self.n2_of]

        def modes_of(maps, ncomp, npix, slot):
            """the blocks of this rank's maps of one spin, one per destination rank, and the flat buffer they are views of (or None)"""
            if ncomp == 0:  # (a rank without maps of this spin still takes part in the exchange, with empty blocks)
                import torch

                return [torch.empty(0, dtype=torch.float64, device=getattr(self.stages, "device", "cpu")) for _ in self.sets], None
            flat = None
            if hasattr(self.stages, "send_buffer"):
                # ONE send buffer per part, kept between steps: hx_ring_modes writes the blocks of all destinations into it side by side
                # and the all-to-all sends it as it is (round 5 concatenated them first: 5.3 GB per rank read and written once more at N = 8)
                if self._send.get(slot) is None or self._send[slot][0] != ncomp:
                    self._send[slot] = (ncomp, self.stages.send_buffer(ncomp, self.sets))
                flat = self._send[slot][1]
                return self.stages.ring_modes(maps.reshape(ncomp, npix), self.sets, pix_weights=pix_weights, ring_weights=ring_weights, out=flat), flat
            return self.stages.ring_modes(maps.reshape(ncomp, npix), self.sets, pix_weights=pix_weights, ring_weights=ring_weights), None

        npix = maps0.shape[-1] if n0 else maps2.shape[-1]
        # ---- part 1: spin 0 ----
        err, send0, flat0 = None, None, None
        try:
            send0, flat0 = modes_of(maps0, n0, npix, 0)
        except Exception as exc:  # noqa: BLE001
            err = exc
        self._agree(err, guard)
        tok0 = self._all_to_all_begin(send0, self.n0_of, flat0)
        # ---- part 2: spin 2 (its ring Fourier stage runs under the first transfer) ----
        err, send2, flat2 = None, None, None
        try:
            send2, flat2 = modes_of(maps2, 2 * n2, npix, 2)
        except Exception as exc:  # noqa: BLE001
            err = exc
        self._agree(err, guard)
        tok2 = self._all_to_all_begin(send2, c2_of, flat2)
        size = self.stages.modes_size(self.orders[1])
        alm = self.buffer()
        # ---- Legendre stage of the spin-0 components on this rank's orders (under the second transfer) ----
        recv0 = self._all_to_all_end(tok0)
        err = None
        try:
            blocks0 = [recv0[s][c * size : (c + 1) * size] for s in range(self.world) for c in range(self.n0_of[s])]
            self.stages.legendre(0, blocks0, self.orders, alm[: self.nc0])
        except Exception as exc:  # noqa: BLE001
            err = exc
        self._agree(err, guard)
        # ---- spin 2, then the partial spectra ----
        recv2 = self._all_to_all_end(tok2)
        err, part = None, None
        try:
            blocks2 = [recv2[s][c * size : (c + 1) * size] for s in range(self.world) for c in range(c2_of[s])]
            self.stages.legendre(2, blocks2, self.orders, alm[self.nc0 :])
            self.stages.synchronize()
            comps = [alm[k] for k in range(alm.shape[0])]
            # this rank's orders only: the alms are zero elsewhere; a kernel that takes the set does not even read them there
            first, count, step = self.orders
            if self._kernel_takes_m_range:
                part = self.kernel(comps, self.cpairs, self.lmax, m_range=(first, first + max(count - 1, 0) * step + (1 if count else 0), step))
            else:
                part = self.kernel(comps, self.cpairs, self.lmax)
            part = np.ascontiguousarray(part, dtype=np.float64)
        except Exception as exc:  # noqa: BLE001
            err = exc
        self._agree(err, guard)
        if self.world > 1:
            import torch
            import torch.distributed as dist

            gloo = dist.get_backend(self.group) == "gloo"
            t = torch.from_numpy(part)
            if not gloo:
                t = t.to(alm.device)
            dist.all_reduce(t, group=self.group)  # alm2cl is a sum over m: the partial spectra add up
            part = t.cpu().numpy()
        return part


# ---- agreement between ranks WITHOUT a collective -----------------------------------------------------------------------------------
def unanimous(ok, tag, timeout=120.0):
    """How many ranks of the default process group report failure at the point ``tag`` -- through the rendezvous key-value store, not
    through a collective: a rank that has FAILED inside a phase must not enter collectives that the others are not in (collectives of
    different type and size would pair up over RCCL: a hang or undefined results).  Every rank calls this with the same tag once;
    returns the number of failed ranks (0 .. world).  A rank that does not arrive within ``timeout`` seconds counts as failed.
    Callers act on the verdict the same way everywhere: all ok -> go on; all failed -> the common fallback; mixed -> leave the job
    (``sys.exit`` non-zero, so that the launcher tears it down)."""
    import time

    import torch.distributed as dist

    world = dist.get_world_size()
    if world == 1:
        return 0 if ok else 1
    store = dist.distributed_c10d._get_default_store()
    store.add(f"hx/{tag}/failed", 0 if ok else 1)
    store.add(f"hx/{tag}/seen", 1)
    t0 = time.monotonic()
    while store.add(f"hx/{tag}/seen", 0) < world:
        if time.monotonic() - t0 > timeout:
            return max(1, store.add(f"hx/{tag}/failed", 0))
        time.sleep(0.01)
    return int(store.add(f"hx/{tag}/failed", 0))


def mixing_matrices_sharded(fields, cls, rank=None, world=None, **kwargs):
    """``heracles_amd.mixing_matrices`` on the ranks of the default process group: the request list of heracles/twopoint.py:354-397 is
    dealt to the ranks by cost (``twopoint.split_requests``), every rank builds its own keys -- replicas only, no collective (SURVEY
    section 8e, last bullet).  Returns this rank's share; the union over the ranks is the single-process result."""
    import torch.distributed as dist

    from .twopoint import mixing_matrices

    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    return mixing_matrices(fields, cls, rank=rank, world=world, **kwargs)
