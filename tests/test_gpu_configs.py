"""One parity case per configuration of BASELINE.json ("configs"), through the drop-in surface
(HipHealpixMapper.transform / angular_power_spectra / mixmat_eb) against the oracle.

  0  1 spin-0 map, nside 64, lmax 128: auto-Cl
  1  1 spin-0 + 1 spin-2 map, nside 1024, lmax 2048: auto + cross Cl          (oracle in full)
  2  10 bins x (spin 0, spin 2), nside 2048, lmax 3072: one bin here, oracle on every 96th m;
     the all-pairs sharding over ranks is covered on CPU (tests/test_distributed_cpu.py)
  3  mixing matrix at lmax 4096: identities that hold at any size + the 3j oracle on blocks anywhere in the matrices
     (low corner here; high-l corner, mid-diagonal and off-diagonal strips, all spins: tests/test_gpu_mixmat.py::
     test_mixmat_blocks_at_high_l_vs_3j, ::test_mixmat_eb_full_sky_identity_lmax4096)
  4  nside 4096, lmax 6144: tests/test_gpu_fullsize.py
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fields_spectra(hx, oracle, nside, lmax, seed):
    """maps -> alms through the mapper (niter = 0, no deconvolution), all spectra on the GPU, the same
    through the oracle."""
    rng = np.random.default_rng(seed)
    npix = 12 * nside * nside
    t = rng.standard_normal(npix)
    qu = rng.standard_normal((2, npix))
    mapper = hx.HipHealpixMapper(nside, lmax, deconvolve=False, niter=0)
    hx.update_metadata(t, spin=0, nside=nside)
    hx.update_metadata(qu, spin=2, nside=nside)
    alms = {("POS", 0): mapper.transform(t, spin=0), ("SHE", 0): mapper.transform(qu, spin=2)}
    cls = hx.angular_power_spectra(alms, debias=False)
    a0 = oracle.map2alm(t[None], nside, lmax, spin=0)[0]
    a2 = oracle.map2alm(qu, nside, lmax, spin=2)
    return alms, cls, a0, a2


def test_config0_nside64_lmax128(oracle):
    import heracles_amd as hx

    alms, cls, a0, _ = _fields_spectra(hx, oracle, 64, 128, 50)
    scale = np.abs(a0).max()
    assert np.abs(alms[("POS", 0)] - a0).max() <= 1e-12 * scale
    np.testing.assert_allclose(np.asarray(cls[("POS", "POS", 0, 0)]), oracle.alm2cl(a0, a0), rtol=1e-11, atol=1e-16)


def test_config1_nside1024_lmax2048(oracle):
    import heracles_amd as hx

    alms, cls, a0, a2 = _fields_spectra(hx, oracle, 1024, 2048, 51)
    assert np.abs(alms[("POS", 0)] - a0).max() <= 1e-11 * np.abs(a0).max()
    assert np.abs(alms[("SHE", 0)] - a2).max() <= 1e-11 * np.abs(a2).max()
    assert list(cls.keys()) == [("POS", "POS", 0, 0), ("POS", "SHE", 0, 0), ("SHE", "SHE", 0, 0)]
    for key, ref in ((("POS", "POS", 0, 0), oracle.alm2cl(a0, a0)), (("POS", "SHE", 0, 0), oracle.alm2cl(a0, a2)),
                     (("SHE", "SHE", 0, 0), oracle.alm2cl(a2, a2))):
        got = np.asarray(cls[key])
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-13 * np.abs(ref).max())


def test_config2_nside2048_lmax3072_sampled_m(oracle):
    import heracles_amd as hx

    nside, lmax, stride = 2048, 3072, 96
    rng = np.random.default_rng(52)
    maps = rng.standard_normal((3, 12 * nside * nside))
    plan = hx.get_plan(nside, lmax)
    got0 = plan.map2alm(maps[:1], 0)
    got2 = plan.map2alm(maps[1:], 2)
    oracle.set_mstride(stride)
    try:
        ref0 = oracle.map2alm(maps[:1], nside, lmax, spin=0)
        ref2 = oracle.map2alm(maps[1:], nside, lmax, spin=2)
    finally:
        oracle.set_mstride(1)
    for got, ref in ((got0, ref0), (got2, ref2)):
        scale = np.abs(got).max()
        for m in range(0, lmax + 1, stride):
            base = m * (2 * lmax + 1 - m) // 2
            sl = slice(base + m, base + lmax + 1)
            assert np.abs(got[:, sl] - ref[:, sl]).max() <= 1e-10 * scale, m


def test_config3_mixing_matrix_lmax4096(oracle):
    import heracles_amd as hx

    L = 4096
    ell = np.arange(L + 1)
    wl = 4 * np.pi * 0.35 * np.exp(-ell * (ell + 1) / 3000.0) + 1e-3 / (1.0 + ell) ** 2
    eb = hx.mixmat_eb(wl)                       # l1max = l2max = l3max = 4096
    assert eb.shape == (3, L + 1, L + 1)
    np.testing.assert_allclose(eb[2], eb[0] - eb[1], atol=1e-12 * np.abs(eb[0]).max())
    assert not eb[:, :2].any() and not eb[:, :, :2].any()      # spin 2 starts at l = 2
    m00 = hx.mixmat(wl, spin=(0, 0))
    s = m00 * (2 * ell + 1)[:, None]            # detailed balance of the coupling
    np.testing.assert_allclose(s, s.T, atol=1e-11 * np.abs(s).max())
    # the oracle's 3j-recursion route on a corner of the matrix (rows < 24)
    ref = oracle.mixmat_eb(wl, l1max=23, l2max=200)
    np.testing.assert_allclose(eb[:, :24, :201], ref, atol=1e-12 * np.abs(ref).max())
    # full-sky mask: identity at this size too
    one = np.zeros(L + 1)
    one[0] = 4 * np.pi
    ident = hx.mixmat(one, spin=(0, 0))
    np.testing.assert_allclose(np.diag(ident), 1.0, atol=1e-11)
    assert np.abs(ident - np.diag(np.diag(ident))).max() <= 1e-11


def test_config2_full_job_dealt_to_8_ranks(oracle):
    """BASELINE configs[2] in full: 10 bins x (Positions, Shears) at nside 2048 / lmax 3072, dealt to 8 ranks the way
    bench.py --scaling strong deals them (cost-balanced maps, tiled pair split).  The 8 ranks run one after the other on
    the one GPU of the test box and write into ONE buffer (which is what the all-gather produces); their Cl rows
    together must equal the single-rank job, and the alms of a spin-0 and a spin-2 map agree with the oracle on every
    96th m whatever sweep shape their owner ran them in."""
    import torch
    import heracles_amd as hx
    from heracles_amd.distributed import ShardedTwoPoint
    from heracles_amd.twopoint import alm2cl_pairs

    nside, lmax, nbins, stride = 2048, 3072, 10, 96
    npix = 12 * nside * nside
    plan = hx.get_plan(nside, lmax)
    nlm = plan.nlm
    spins = [0] * nbins + [2] * nbins

    def map_of(g):
        gen = torch.Generator(device="cuda").manual_seed(5200 + g)
        return torch.randn(((2,) if spins[g] else ()) + (npix,), dtype=torch.float64, device="cuda", generator=gen)

    def transform_local(work):
        a0, a2 = work.local_alm_views("cuda")
        m0 = [map_of(g) for g in work.local_maps if spins[g] == 0]
        m2 = [map_of(g) for g in work.local_maps if spins[g] == 2]
        if m0:
            plan.map2alm(torch.stack(m0), 0, out=a0)
        if m2:
            plan.map2alm(torch.stack(m2).view(2 * len(m2), npix), 2, out=a2.view(2 * len(m2), nlm))

    single = ShardedTwoPoint(spins, 1, 0, nlm, lmax)
    transform_local(single)
    ref_rows = single.all_pairs_cl()
    assert ref_rows.shape == (10 * 11 // 2 + 4 * 10 * 11 // 2 + 2 * 100, lmax + 1)
    ref_alm = {g: single.buffer()[single.comps_of_map[g]].cpu().numpy() for g in (0, 13)}
    del single
    torch.cuda.empty_cache()

    world = 8
    ranks = [ShardedTwoPoint(spins, world, r, nlm, lmax) for r in range(world)]
    shared = ranks[0].buffer("cuda")
    rows = np.full_like(ref_rows, np.nan)
    for w in ranks:
        w._buf = shared
        transform_local(w)
    comps = [shared[k] for k in range(shared.shape[0])]
    for w in ranks:
        assert 2 <= len(w.local_maps) <= 3
        rows[w.rows_of[w.rank]] = alm2cl_pairs(comps, w.my_cpairs, lmax)
    assert not np.isnan(rows).any()
    # other sweep shapes per owner (2-3 maps instead of 10): rounding differs, nothing else
    np.testing.assert_allclose(rows, ref_rows, rtol=1e-9, atol=1e-13 * np.abs(ref_rows).max())
    for g in (0, 13):
        got = shared[ranks[0].comps_of_map[g]].cpu().numpy()
        assert np.abs(got - ref_alm[g]).max() <= 1e-11 * np.abs(got).max()
        m_host = map_of(g).cpu().numpy().reshape(-1, npix)
        oracle.set_mstride(stride)
        try:
            ref = oracle.map2alm(m_host, nside, lmax, spin=spins[g])
        finally:
            oracle.set_mstride(1)
        scale = np.abs(got).max()
        for m in range(0, lmax + 1, stride):
            base = m * (2 * lmax + 1 - m) // 2
            sl = slice(base + m, base + lmax + 1)
            assert np.abs(got[:, sl] - ref[:, sl]).max() <= 1e-10 * scale, (g, m)
