"""world_size-2/3 gloo tests of the multi-GPU sharding logic on CPU: cost-balanced map assignment, in-place
all-gather of the alm shards, tiled pair partition, gather of the Cl blocks.  The arithmetic kernel is the
oracle here (the HIP kernel needs a GPU); what is tested is that the sharded job returns exactly the spectra
of the single-process job over all maps."""

import os
import socket

import numpy as np
import pytest

SPINS = [0, 2, 0, 2, 2, 0, 0, 2, 0]  # a job of 9 maps in a mixed order


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


def _alm_of_map(g, spin, lmax):
    """Seeded alms of global map g: (nlm,) for spin 0, (2, nlm) for spin 2 -- what a rank's map2alm would produce."""
    import torch

    nlm = (lmax + 1) * (lmax + 2) // 2
    gen = torch.Generator().manual_seed(1000 + g)
    shape = (nlm, 2) if spin == 0 else (2, nlm, 2)
    return torch.view_as_complex(torch.randn(shape, dtype=torch.float64, generator=gen)).contiguous()


def _fill_local(work, lmax):
    a0, a2 = work.local_alm_views("cpu")
    k0 = k2 = 0
    for g in work.local_maps:
        if work.spins[g] == 0:
            a0[k0] = _alm_of_map(g, 0, lmax)
            k0 += 1
        else:
            a2[k2] = _alm_of_map(g, 2, lmax)
            k2 += 1
    assert k0 == a0.shape[0] and k2 == a2.shape[0]


def _worker(rank, world, port, lmax, outdir):
    import torch.distributed as dist

    from heracles_amd.distributed import ShardedTwoPoint

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nlm = (lmax + 1) * (lmax + 2) // 2
    work = ShardedTwoPoint(SPINS, world, rank, nlm, lmax, kernel=_kernel)
    for _ in range(2):  # the second step reuses the buffer
        _fill_local(work, lmax)
        res = work.all_pairs_cl()
    if rank == 0:
        np.save(os.path.join(outdir, "sharded.npy"), res)
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def _worker_split(rank, world, port, lmax, outdir):
    """The exchange issued in two parts, as bench.py's step does it: spin-2 shards right after "their transform", the spin-0
    transform "under" the transfer, spin-0 shards, then the pairs -- against the single blocking exchange on a second object."""
    import torch
    import torch.distributed as dist

    from heracles_amd.distributed import ShardedTwoPoint

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nlm = (lmax + 1) * (lmax + 2) // 2
    split = ShardedTwoPoint(SPINS, world, rank, nlm, lmax, kernel=_kernel)
    whole = ShardedTwoPoint(SPINS, world, rank, nlm, lmax, kernel=_kernel)
    for _ in range(2):
        a0, a2 = split.local_alm_views("cpu")
        k2 = 0
        for g in split.local_maps:  # "spin-2 transform"
            if split.spins[g] == 2:
                a2[k2] = _alm_of_map(g, 2, lmax)
                k2 += 1
        split.exchange_begin(2)
        k0 = 0
        for g in split.local_maps:  # "spin-0 transform", while the first part is in flight
            if split.spins[g] == 0:
                a0[k0] = _alm_of_map(g, 0, lmax)
                k0 += 1
        split.exchange_begin(0)
        res = split.all_pairs_cl()
        _fill_local(whole, lmax)
        whole.exchange()
        assert not whole._pending
        ref = whole.all_pairs_cl()
        assert torch.equal(split.buffer(), whole.buffer())  # the gathered bytes are the same either way
        if rank == 0:
            np.testing.assert_array_equal(res, ref)
    if rank == 0:
        np.save(os.path.join(outdir, "split.npy"), res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_split_async_exchange_equals_single_exchange(tmp_path, world):
    """the all-gather issued as two asynchronous parts (spin 2 first, the spin-0 transform under it) gives bitwise
    the buffer and the spectra of the single blocking gather."""
    import torch.multiprocessing as mp

    lmax = 12
    mp.spawn(_worker_split, args=(world, _free_port(), lmax, str(tmp_path)), nprocs=world, join=True)
    np.testing.assert_array_equal(np.load(tmp_path / "split.npy"), _reference(lmax))


def _reference(lmax):
    """All spectra from the definition, map pairs in combinations_with_replacement order."""
    rows = []
    alms = [np.atleast_2d(_alm_of_map(g, s, lmax).numpy()) for g, s in enumerate(SPINS)]
    for i in range(len(SPINS)):
        for j in range(i, len(SPINS)):
            for a in alms[i]:
                for b in alms[j]:
                    rows.append(_kernel([a, b], [(0, 1)], lmax)[0])
    return np.array(rows)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_equals_single(tmp_path, world):
    import torch.multiprocessing as mp

    from heracles_amd.distributed import ShardedTwoPoint

    lmax = 12
    nlm = (lmax + 1) * (lmax + 2) // 2
    mp.spawn(_worker, args=(world, _free_port(), lmax, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "sharded.npy")
    ref = _reference(lmax)
    assert got.shape == ref.shape
    np.testing.assert_array_equal(got, ref)
    # world == 1 path: no communication, same answer
    one = ShardedTwoPoint(SPINS, 1, 0, nlm, lmax, kernel=_kernel)
    _fill_local(one, lmax)
    np.testing.assert_array_equal(one.all_pairs_cl(), ref)


def test_assignment_is_cost_balanced():
    from heracles_amd.distributed import assign_maps, map_cost

    # the north-star job: 10 spin-0 + 10 spin-2 maps = 40 cost units
    spins = [0] * 10 + [2] * 10
    for world, worst in ((1, 40), (2, 20), (4, 10), (8, 6)):
        owner = assign_maps(spins, world)
        load = [sum(map_cost(spins[g]) for g in range(20) if owner[g] == r) for r in range(world)]
        assert sum(load) == 40 and max(load) == worst, (world, load)
        assert min(load) >= worst - 2  # (8 ranks: two spin-2 maps on two of them is the best there is: 6 against a mean of 5)
    # deterministic and identical on every rank
    assert assign_maps(SPINS, 3) == assign_maps(list(SPINS), 3)


def test_partition_covers_all_pairs():
    from heracles_amd.distributed import ShardedTwoPoint

    for spins in ([0] * 10 + [2] * 10, SPINS, [2], [0, 0, 0]):
        for world in (1, 2, 4, 8):
            ws = [ShardedTwoPoint(spins, world, r, 10, 3, kernel=_kernel) for r in range(world)]
            seen = [p for w in ws for p in w.my_pairs]
            assert sorted(seen) == sorted(ws[0].pairs) and len(seen) == len(set(seen))
            rows = sorted(k for w in ws for k in w.rows_of[w.rank])
            assert rows == list(range(ws[0].nrows))
            # every component slot is owned by exactly one map, shards do not overlap
            slots = sorted(c for g in range(len(spins)) for c in ws[0].comps_of_map[g])
            assert len(slots) == len(set(slots)) and max(slots) < ws[0].nbuf_rows
            if world == 8 and len(spins) == 20:
                # tiles: a rank reads far fewer than all 30 components
                touched = [len({c for pr in w.my_cpairs for c in pr}) for w in ws]
                assert max(touched) < 30 and sum(touched) / world <= 22, touched
                counts = [len(w.my_cpairs) for w in ws]
                assert max(counts) <= 1.5 * sum(counts) / world, counts


# ---- m-sharded route (MShardedTwoPoint): ring modes -> all-to-all -> Legendre on an m-range -> partial Cl -> all-reduce --------
MS_NSIDE, MS_LMAX = 4, 9


class OracleStages:
    """The two halves of the transform from the oracle (hxo_fourier_analysis / hxo_legendre_analysis), CPU tensors: stands in for
    HipStages so that the sharding logic -- order sets, block order, all-to-all splits, all-reduce -- runs under gloo without a GPU."""

    def __init__(self, nside, lmax):
        self.nside, self.lmax = nside, lmax
        self.nr = 4 * nside - 1

    def modes_size(self, count):
        return self.nr * count * 2

    @staticmethod
    def _ms(orders):
        first, count, step = orders
        return np.arange(first, first + count * step, step)

    def ring_modes(self, maps, sets, pix_weights=None, ring_weights=None):
        import torch

        from oracle import hxoracle as ho

        assert ring_weights is None
        F = ho.fourier_analysis(np.asarray(maps), self.nside, self.lmax, pix_weights=pix_weights)  # [comp][ring][m]
        return [torch.from_numpy(np.ascontiguousarray(F[:, :, self._ms(o)]).view(np.float64).reshape(-1)) for o in sets]

    def legendre(self, spin, blocks, orders, alm_out):
        import torch

        from oracle import hxoracle as ho

        ms = self._ms(orders)
        if not blocks or ms.size == 0:
            return
        F = np.zeros((len(blocks), self.nr, self.lmax + 1), dtype=np.complex128)
        for c, b in enumerate(blocks):
            F[c][:, ms] = b.numpy().view(np.complex128).reshape(self.nr, ms.size)
        alm = ho.legendre_analysis(F, self.nside, self.lmax, spin=spin)
        for m in ms:  # only the orders of the set are written
            base = m * (2 * self.lmax + 1 - m) // 2
            alm_out[:, base + m : base + self.lmax + 1] = torch.from_numpy(alm[:, base + m : base + self.lmax + 1])

    def zeros_alm(self, ncomp, nlm):
        import torch

        return torch.zeros((ncomp, nlm), dtype=torch.complex128)

    def synchronize(self):
        pass


def _ms_map(g, spin):
    rng = np.random.default_rng(7000 + g)
    return rng.standard_normal(((2,) if spin else ()) + (12 * MS_NSIDE**2,))


def _ms_worker(rank, world, port, outdir, spins=None):
    import torch
    import torch.distributed as dist

    from heracles_amd.distributed import MShardedTwoPoint

    SPINS = list(spins) if spins is not None else globals()["SPINS"]

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nlm = (MS_LMAX + 1) * (MS_LMAX + 2) // 2
    work = MShardedTwoPoint(SPINS, world, rank, nlm, MS_LMAX, OracleStages(MS_NSIDE, MS_LMAX), kernel=_kernel)
    mine = work.local_maps
    m0 = np.array([_ms_map(g, 0) for g in mine if SPINS[g] == 0]).reshape(-1, 12 * MS_NSIDE**2)
    m2 = np.array([_ms_map(g, 2) for g in mine if SPINS[g] == 2]).reshape(-1, 2, 12 * MS_NSIDE**2)
    pw = 1.0 + 0.01 * np.cos(np.arange(12 * MS_NSIDE**2))
    for _ in range(2):  # the second step reuses the alm buffer
        res = work.run(torch.from_numpy(m0), torch.from_numpy(m2), pix_weights=pw)
    np.save(os.path.join(outdir, f"msharded_{rank}.npy"), res)
    np.save(os.path.join(outdir, f"orders_{rank}.npy"), np.array(work.sets))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_m_sharded_equals_single(tmp_path, world):
    """The m-sharded job returns, on every rank, the spectra of the single-process job over all maps (the sum over m is split
    between the ranks: equal to rounding, 1e-12)."""
    import torch
    import torch.multiprocessing as mp

    from heracles_amd.distributed import MShardedTwoPoint
    from oracle import hxoracle as ho

    mp.spawn(_ms_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    pw = 1.0 + 0.01 * np.cos(np.arange(12 * MS_NSIDE**2))
    alms = [np.atleast_2d(ho.map2alm(_ms_map(g, s), MS_NSIDE, MS_LMAX, spin=s, pix_weights=pw)) for g, s in enumerate(SPINS)]
    rows = []
    for i in range(len(SPINS)):
        for j in range(i, len(SPINS)):
            for a in alms[i]:
                for b in alms[j]:
                    rows.append(ho.alm2cl(a, b, lmax=MS_LMAX))
    ref = np.array(rows)
    for r in range(world):
        got = np.load(tmp_path / f"msharded_{r}.npy")
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-13 * np.abs(ref).max())
        sets = np.load(tmp_path / f"orders_{r}.npy")
        ms = sorted(m for (f, c, st) in sets for m in range(f, f + c * st, st))
        assert ms == list(range(MS_LMAX + 1)) and len(sets) == world  # every order owned exactly once
    # world == 1: no communication, same answer
    nlm = (MS_LMAX + 1) * (MS_LMAX + 2) // 2
    one = MShardedTwoPoint(SPINS, 1, 0, nlm, MS_LMAX, OracleStages(MS_NSIDE, MS_LMAX), kernel=_kernel)
    m0 = np.array([_ms_map(g, 0) for g in one.local_maps if SPINS[g] == 0])
    m2 = np.array([_ms_map(g, 2) for g in one.local_maps if SPINS[g] == 2])
    np.testing.assert_allclose(one.run(torch.from_numpy(m0), torch.from_numpy(m2), pix_weights=pw), ref, rtol=1e-12, atol=1e-13 * np.abs(ref).max())


def _spectra_of(spins):
    from oracle import hxoracle as ho

    pw = 1.0 + 0.01 * np.cos(np.arange(12 * MS_NSIDE**2))
    alms = [np.atleast_2d(ho.map2alm(_ms_map(g, s), MS_NSIDE, MS_LMAX, spin=s, pix_weights=pw)) for g, s in enumerate(spins)]
    return np.array([ho.alm2cl(a, b, lmax=MS_LMAX) for i in range(len(spins)) for j in range(i, len(spins)) for a in alms[i] for b in alms[j]])


@pytest.mark.parametrize("spins", [(2, 0, 0), (2, 2, 0), (0, 0), (2, 2, 2)])
def test_m_sharded_rank_without_a_spin(tmp_path, spins):
    """The exchange goes out in a spin-0 and a spin-2 part: a rank that holds maps of one spin only (or a job without one of the spins)
    sends empty blocks in the other part and still receives its share of it."""
    import torch.multiprocessing as mp

    from heracles_amd.distributed import MShardedTwoPoint

    world = 2
    nlm = (MS_LMAX + 1) * (MS_LMAX + 2) // 2
    works = [MShardedTwoPoint(list(spins), world, r, nlm, MS_LMAX, OracleStages(MS_NSIDE, MS_LMAX), kernel=_kernel) for r in range(world)]
    if len(set(spins)) == 2:
        assert any(w.n0_of[w.rank] == 0 or w.n2_of[w.rank] == 0 for w in works)  # the case this test is for
    mp.spawn(_ms_worker, args=(world, _free_port(), str(tmp_path), tuple(spins)), nprocs=world, join=True)
    ref = _spectra_of(spins)
    for r in range(world):
        got = np.load(tmp_path / f"msharded_{r}.npy")
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-13 * np.abs(ref).max())


def test_order_sets_cover_and_balance():
    from heracles_amd.distributed import assign_maps_by_components, order_sets

    lmax = 6144
    cost = (lmax + 1.0 - np.arange(lmax + 1)) * np.minimum(1.0, 0.2 + np.arange(lmax + 1)[::-1] / lmax)
    for world in (1, 2, 4, 8):
        sets = order_sets(lmax, world)
        ms = [np.arange(f, f + c * st, st) for (f, c, st) in sets]
        assert sorted(np.concatenate(ms).tolist()) == list(range(lmax + 1))
        loads = [cost[m].sum() for m in ms]
        assert max(loads) <= 1.005 * sum(loads) / world, (world, loads)
        assert max(len(m) for m in ms) - min(len(m) for m in ms) <= 1
    assert [c for (_, c, _) in order_sets(2, 8)] == [1, 1, 1, 0, 0, 0, 0, 0]  # more ranks than orders: empty sets, nothing lost
    # ring Fourier stage: components dealt evenly (30 components of the north-star job: 4, 4, 4, 4, 4, 4, 3, 3 on 8 ranks)
    spins = [0] * 10 + [2] * 10
    owner = assign_maps_by_components(spins, 8)
    loads = [sum((2 if spins[g] else 1) for g in range(20) if owner[g] == r) for r in range(8)]
    assert sum(loads) == 30 and max(loads) - min(loads) <= 1, loads


def test_bench_job_is_the_same_at_every_n():
    """bench.py's `value` is quoted on ONE job at every N: the pair count of both fixed-job routes does not depend on
    the number of ranks, and equals the N = 1 job's."""
    import importlib.util

    from heracles_amd.distributed import MShardedTwoPoint, ShardedTwoPoint

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)                      # (imports neither torch nor the library at module level)
    assert "torch" not in bench.__dict__
    per_set = bench.fixed_job("north_star", 10)
    assert per_set == [0] * 10 + [2] * 10
    lmax = 6144
    nlm = (lmax + 1) * (lmax + 2) // 2
    one = ShardedTwoPoint(per_set, 1, 0, nlm, lmax, kernel=lambda *a: None)
    assert len(one.pairs) == 210
    for world in (2, 4, 8):
        for rank in (0, world - 1):
            ag = ShardedTwoPoint(per_set, world, rank, nlm, lmax, kernel=lambda *a: None)
            ms = MShardedTwoPoint(per_set, world, rank, nlm, lmax, stages=None, kernel=lambda *a: None)
            assert len(ag.pairs) == len(ms.pairs) == len(one.pairs) == 210
            assert ag.nrows == ms.nrows == one.nrows
        # every pair is some rank's, exactly once
        ag = ShardedTwoPoint(per_set, world, 0, nlm, lmax, kernel=lambda *a: None)
        assert sorted(p for ps in ag.pairs_of for p in ps) == sorted(one.pairs)
    assert len(bench.fixed_job("euclid", 10)) == 39


def test_bench_self_launch_starts_n_ranks_and_propagates_failure():
    """`python3 bench.py --gpus N` without a launcher starts N children with the rendezvous variables (no GPU involved with
    --launch-check), prints ONE line, and exits non-zero when any rank does."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--launch-check"], env=env, capture_output=True,
                         text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-1000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 4 and json.loads(lines[0])["master"].startswith("127.0.0.1:")
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--launch-check"], env=dict(env, HX_LAUNCH_CHECK_FAIL="2"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode == 3


# ---- mixing-matrix request list split across ranks (SURVEY 8e last bullet; heracles/twopoint.py:354-397) ----------------------------
def _mm_job():
    """config-5-like: 2 scalar fields + 1 spin-2 field on two masks, 3 bins"""
    import types

    fields = {"POS": types.SimpleNamespace(mask="VIS", spin=0), "CON": types.SimpleNamespace(mask="VIS", spin=0),
              "SHE": types.SimpleNamespace(mask="WHT", spin=2), "NOM": types.SimpleNamespace(mask=None, spin=0)}
    l = np.arange(25)
    cls = {}
    for a, b in (("VIS", "VIS"), ("VIS", "WHT"), ("WHT", "WHT")):
        for i in range(3):
            for j in range(i if a == b else 0, 3):
                cls[a, b, i, j] = 1.0 / (1.0 + l + i + 2 * j) ** 2
    return fields, cls


def _mm_context(cl, l1max, l2max, l3max, spin):
    from oracle import hxoracle as ho

    return (ho.mixmat_eb if all(spin) else ho.mixmat)(cl, l1max=l1max, l2max=l2max, l3max=l3max, spin=spin)


def _mm_worker(rank, world, port, outdir, bins):
    import torch.distributed as dist

    from heracles_amd.distributed import mixing_matrices_sharded, unanimous

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fields, cls = _mm_job()
    kw = dict(l1max=12, l2max=16, l3max=24, context=_mm_context)
    if bins:
        kw.update(bins=np.array([2, 4, 8, 13]), weights="2l+1")
    mine = mixing_matrices_sharded(fields, cls, **kw)  # (rank / world from the process group; no collective inside)
    np.savez(os.path.join(outdir, f"mm_{rank}.npz"), **{"|".join(map(str, k)): np.asarray(v.array) for k, v in mine.items()})
    # agreement without a collective: everybody ok -> 0; rank 1 reports a failure -> every rank sees exactly one
    assert unanimous(True, "mm-ok") == 0
    assert unanimous(rank != 1, "mm-one-failed") == 1
    dist.destroy_process_group()


@pytest.mark.parametrize("world,bins", [(2, False), (2, True), (3, True), (8, True)])
def test_mixing_matrix_requests_split_across_ranks(tmp_path, world, bins):
    import torch.multiprocessing as mp

    import heracles_amd as hx

    mp.spawn(_mm_worker, args=(world, _free_port(), str(tmp_path), bins), nprocs=world, join=True)
    fields, cls = _mm_job()
    kw = dict(l1max=12, l2max=16, l3max=24, context=_mm_context)
    if bins:
        kw.update(bins=np.array([2, 4, 8, 13]), weights="2l+1")
    single = hx.mixing_matrices(fields, cls, **kw)
    got = {}
    counts = []
    for r in range(world):
        with np.load(os.path.join(tmp_path, f"mm_{r}.npz")) as z:
            counts.append(len(z.files))
            for k in z.files:
                assert k not in got, f"{k} computed twice"
                got[k] = z[k]
    assert set(got) == {"|".join(map(str, k)) for k in single}  # the union is the single-process dictionary
    for k, v in single.items():
        np.testing.assert_array_equal(got["|".join(map(str, k))], np.asarray(v.array))
    assert max(counts) - min(counts) <= 2 and min(counts) > 0, counts
