"""GPU parity: hx_map2alm / hx_alm2map through the C ABI against the CPU oracle.

Tolerance: both sides evaluate lambda_lm with relative error O(m eps) (the ring
co-latitudes are only known to eps); differences are bounded normwise,
max|delta| <= TOL * max|alm|, TOL = 1e-11 (bit-exactness is not defined for this path)."""

import numpy as np
import pytest

from helpers import band_limited_maps, idx, random_alm

pytestmark = pytest.mark.gpu
TOL = 1e-11


def close(a, b, tol=TOL):
    scale = max(np.abs(b).max(), 1e-300)
    err = np.abs(np.asarray(a) - np.asarray(b)).max()
    assert err <= tol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("nside,lmax", [(1, 2), (2, 5), (4, 8), (8, 16), (8, 31), (12, 20), (16, 24), (32, 48), (64, 100)])
@pytest.mark.parametrize("spin", [0, 2])
def test_map2alm_random_maps(oracle, nside, lmax, spin):
    import heracles_amd as hx

    rng = np.random.default_rng(100 * nside + lmax + spin)
    ncomp = 2 if spin == 2 else 3
    maps = rng.standard_normal((ncomp, 12 * nside**2))
    plan = hx.Plan(nside, lmax)
    out = plan.map2alm(maps, spin)
    ref = oracle.map2alm(maps, nside, lmax, spin=spin)
    assert out.shape == ref.shape
    close(out, ref)
    plan.close()


@pytest.mark.parametrize("spin,ncomp", [(0, 1), (0, 2), (0, 3), (0, 4), (0, 5), (0, 8), (0, 9), (0, 19),
                                        (2, 2), (2, 4), (2, 6), (2, 8), (2, 10), (2, 16), (2, 18), (2, 22)])
def test_map2alm_batching(oracle, spin, ncomp):
    """Launch shapes: 4-column MFMA path (<= 2 maps / 1 field), 8-column path, one and two
    16-column groups, and several launches per call."""
    import heracles_amd as hx

    rng = np.random.default_rng(7 + ncomp)
    nside, lmax = 16, 32
    maps = rng.standard_normal((ncomp, 12 * nside**2))
    out = hx.get_plan(nside, lmax).map2alm(maps, spin)
    close(out, oracle.map2alm(maps, nside, lmax, spin=spin))


@pytest.mark.parametrize("spin", [0, 2])
def test_map2alm_weights_filter_iterations(oracle, spin):
    import heracles_amd as hx

    rng = np.random.default_rng(17)
    nside, lmax = 16, 24
    ncomp = 2
    _, maps = band_limited_maps(oracle, rng, nside, lmax, spin, 1 if spin == 2 else 2)
    rw = rng.uniform(0.9, 1.1, 2 * nside)
    pw = rng.uniform(0.9, 1.1, 12 * nside**2)
    fl = rng.uniform(0.5, 1.5, lmax + 1)
    plan = hx.get_plan(nside, lmax)
    close(plan.map2alm(maps, spin, ring_weights=rw), oracle.map2alm(maps, nside, lmax, spin=spin, ring_weights=rw))
    close(plan.map2alm(maps, spin, pix_weights=pw), oracle.map2alm(maps, nside, lmax, spin=spin, pix_weights=pw))
    ref = oracle.map2alm(maps, nside, lmax, spin=spin)
    out = plan.map2alm(maps, spin, fl=fl)
    for m in range(lmax + 1):
        s = idx(lmax, m, m)
        close(out[..., s : s + lmax - m + 1], ref[..., s : s + lmax - m + 1] * fl[m:])
    for niter in (1, 3):
        close(plan.map2alm(maps, spin, niter=niter, fl=fl),
              _apply_fl(oracle.map2alm(maps, nside, lmax, spin=spin, niter=niter), fl, lmax), 1e-10)
    assert ncomp == maps.shape[0]


def _apply_fl(alm, fl, lmax):
    out = np.array(alm)
    for m in range(lmax + 1):
        s = idx(lmax, m, m)
        out[..., s : s + lmax - m + 1] *= fl[m:]
    return out


@pytest.mark.parametrize("spin,ncomp", [(0, 11), (2, 10)])
def test_iterations_on_batches(oracle, spin, ncomp):
    """Jacobi iterations of a batch: the analysis passes take the batch in the sweeps of the plain transform (one matrix sweep of
    11 maps / 5 fields), the syntheses in sweeps of 4 maps / 2 fields that share the recursion, residual maps in plan scratch
    sized for the whole sweep -- against the oracle's iterations map by map."""
    import heracles_amd as hx

    rng = np.random.default_rng(29 + spin)
    nside, lmax = 32, 64
    maps = rng.standard_normal((ncomp, 12 * nside**2))
    plan = hx.get_plan(nside, lmax)
    got = plan.map2alm(maps, spin, niter=2)
    unit = 1 if spin == 0 else 2
    for c in (0, ncomp - unit):
        close(got[c : c + unit], oracle.map2alm(maps[c : c + unit], nside, lmax, spin=spin, niter=2), 1e-10)
    close(got, np.concatenate([plan.map2alm(maps[c : c + unit], spin, niter=2) for c in range(0, ncomp, unit)]), 1e-11)


@pytest.mark.parametrize("nside,lmax", [(4, 8), (8, 20), (12, 20), (32, 48)])
@pytest.mark.parametrize("spin", [0, 2])
def test_alm2map(oracle, nside, lmax, spin):
    import heracles_amd as hx

    rng = np.random.default_rng(3 * nside + spin)
    alm = random_alm(rng, lmax, spin, (2,))
    out = hx.get_plan(nside, lmax).alm2map(alm, spin)
    close(out, oracle.alm2map(alm, nside, lmax, spin=spin), 1e-11)


@pytest.mark.parametrize("nside,lmax", [(128, 200), (256, 300)])
@pytest.mark.parametrize("spin", [0, 2])
def test_alm2map_medium_scaled_chains(oracle, nside, lmax, spin):
    """Synthesis at sizes where most recursion chains start far below 2^-300 (every phase of the vector-unit synthesis kernel:
    scaled-only blocks, blocks whose late chains are dropped at the next check, steady blocks), three maps / two fields, against
    the oracle's direct sums."""
    import heracles_amd as hx

    rng = np.random.default_rng(7 * nside + spin)
    alm = random_alm(rng, lmax, spin, (3,) if spin == 0 else (4,))
    out = hx.get_plan(nside, lmax).alm2map(alm, spin)
    close(out, oracle.alm2map(alm, nside, lmax, spin=spin), 1e-11)


@pytest.mark.parametrize("nside,lmax", [(16, 24), (64, 100), (128, 200)])
@pytest.mark.parametrize("spin,ncomp", [(0, 5), (0, 8), (0, 10), (2, 6), (2, 16), (2, 20), (0, 13), (2, 26)])
def test_alm2map_batches(oracle, nside, lmax, spin, ncomp):
    """Batches of 5 .. 13 maps / 3 .. 13 fields (sweeps of four maps / two fields that share the recursion, then the rest) against the
    oracle's direct sums."""
    import heracles_amd as hx

    rng = np.random.default_rng(11 * nside + ncomp + spin)
    alm = random_alm(rng, lmax, spin, (ncomp,))
    out = hx.get_plan(nside, lmax).alm2map(alm, spin)
    if nside <= 64 or ncomp in (5, 6, 10, 20):
        close(out, oracle.alm2map(alm, nside, lmax, spin=spin), 1e-11)
    else:  # (the oracle takes a minute per ten maps at this size: first and last unit)
        unit = 1 if spin == 0 else 2
        for c in (0, ncomp - unit):
            close(out[c : c + unit], oracle.alm2map(alm[c : c + unit], nside, lmax, spin=spin), 1e-11)


@pytest.mark.parametrize("spin", [0, 2])
def test_alm2map_sweeps_of_several_maps_match_single_sweeps(spin):
    """Vector-unit synthesis (batches of <= 4 maps / <= 2 fields): four spin-0 maps / two spin-2 fields per sweep share the recursion
    (other ring slots per lane, other block lengths): every lane still adds the same terms in the same order, so a batch is bitwise
    what its maps give one by one."""
    import heracles_amd as hx

    rng = np.random.default_rng(91 + spin)
    nside, lmax = 128, 200
    alm = random_alm(rng, lmax, spin, (4,))  # one sweep of 4 maps / 2 fields
    plan = hx.get_plan(nside, lmax)
    batch = plan.alm2map(alm, spin)
    unit = 1 if spin == 0 else 2
    for c in range(0, alm.shape[0], unit):
        np.testing.assert_array_equal(batch[c : c + unit], plan.alm2map(alm[c : c + unit], spin))
    three = plan.alm2map(alm[: 3 * unit] if spin == 0 else alm[:2], spin)   # 2 + 1 maps / one field
    np.testing.assert_array_equal(three, batch[: three.shape[0]])


# units = maps (spin 0) / fields (spin 2): every shape of k_synth_duo -- (1,0) (1,1) (1,2) (2,0) (2,1) (2,2) -- with and without
# padding columns, a second sweep and an evenly split tail
@pytest.mark.parametrize("spin,units", [(0, 5), (0, 8), (0, 9), (0, 10), (0, 11), (0, 12), (0, 16), (0, 17), (0, 19), (0, 20), (0, 23), (0, 41),
                                        (2, 3), (2, 4), (2, 5), (2, 6), (2, 7), (2, 8), (2, 9), (2, 10), (2, 11), (2, 13), (2, 21)])
def test_alm2map_batches_on_the_matrix_unit(oracle, spin, units):
    """Batched synthesis on the matrix unit (hx_synth_duo.hip: >= 5 maps / >= 3 fields per call) against the oracle's direct sums, and
    against the same maps / fields synthesised one by one on the vector-unit kernel (1e-12: other orders of summation).  Sizes at which
    the polar chains start far below 2^-300 (scaled-only blocks, mixed blocks, live blocks), odd l ranges, m = 0, 1 (spin 2: off = 1)."""
    import heracles_amd as hx

    nside, lmax = (64, 150) if units % 2 else (32, 77)
    unit = 1 if spin == 0 else 2
    rng = np.random.default_rng(100 * units + spin)
    alm = random_alm(rng, lmax, spin, (units * unit,))
    plan = hx.get_plan(nside, lmax)
    out = plan.alm2map(alm, spin)
    assert out.shape == (units * unit, 12 * nside * nside)
    scale = np.abs(out).max()
    for u in sorted({0, 1, units // 2, units - 2, units - 1}):
        sl = slice(u * unit, (u + 1) * unit)
        close(out[sl], oracle.alm2map(alm[sl], nside, lmax, spin=spin), 1e-11)
        one = plan.alm2map(alm[sl], spin)
        assert np.abs(out[sl] - one).max() <= 1e-12 * scale
    # run-to-run: the kernel has no atomics and no order that depends on timing
    np.testing.assert_array_equal(out, plan.alm2map(alm, spin))


@pytest.mark.parametrize("spin,units", [(0, 10), (2, 5), (2, 10)])
def test_alm2map_matrix_unit_medium(oracle, spin, units):
    """The same at nside 256 / lmax 400 (tasks of several ring groups per m, pruned polar rings, chains that become live late): first and
    last unit against the oracle, every unit against the vector-unit kernel."""
    import heracles_amd as hx

    nside, lmax = 256, 400
    unit = 1 if spin == 0 else 2
    rng = np.random.default_rng(7 * units + spin)
    alm = random_alm(rng, lmax, spin, (units * unit,))
    plan = hx.get_plan(nside, lmax)
    out = plan.alm2map(alm, spin)
    scale = np.abs(out).max()
    for u in (0, units - 1):
        sl = slice(u * unit, (u + 1) * unit)
        close(out[sl], oracle.alm2map(alm[sl], nside, lmax, spin=spin), 1e-11)
    for u in range(units):
        sl = slice(u * unit, (u + 1) * unit)
        assert np.abs(out[sl] - plan.alm2map(alm[sl], spin)).max() <= 1e-12 * scale


@pytest.mark.parametrize("spin", [0, 2])
def test_roundtrip_medium(oracle, spin):
    """nside=256, lmax=384: alm -> map (GPU) -> alm (GPU, niter=3) returns the input."""
    import heracles_amd as hx

    rng = np.random.default_rng(23)
    nside, lmax = 256, 384
    alm = random_alm(rng, lmax, spin, (2,))
    plan = hx.get_plan(nside, lmax)
    maps = plan.alm2map(alm, spin)
    back0 = plan.map2alm(maps, spin)
    back3 = plan.map2alm(maps, spin, niter=3)
    e0 = np.abs(back0 - alm).max()
    e3 = np.abs(back3 - alm).max()
    assert e0 < 2e-2 and e3 < 1e-5, (e0, e3)
    # one component against the oracle at this size (seconds on CPU)
    close(back0[0:2], oracle.map2alm(maps, nside, lmax, spin=spin)[0:2], 2e-11)


def test_linearity_and_repeatability():
    import heracles_amd as hx

    rng = np.random.default_rng(31)
    nside, lmax = 64, 96
    a = rng.standard_normal((1, 12 * nside**2))
    b = rng.standard_normal((1, 12 * nside**2))
    plan = hx.get_plan(nside, lmax)
    A, B = plan.map2alm(a, 0), plan.map2alm(b, 0)
    AB = plan.map2alm(2 * a - 3 * b, 0)
    close(AB, 2 * A - 3 * B, 1e-12)
    np.testing.assert_array_equal(plan.map2alm(a, 0), A)  # bitwise run-to-run


def test_constant_map_gives_monopole(oracle):
    """The reference's own fixtures use constant maps (tests/conftest.py:25-79)."""
    import heracles_amd as hx

    nside, lmax = 32, 8
    plan = hx.get_plan(nside, lmax)
    m = 4 * np.ones((1, 12 * nside**2))
    alm = plan.map2alm(m, 0)[0]
    assert abs(alm[0] - 4 * np.sqrt(4 * np.pi)) < 1e-13  # the pixel areas sum to 4 pi exactly
    alm3 = plan.map2alm(m, 0, niter=3)[0]
    close(alm3, oracle.map2alm(m, nside, lmax, niter=3)[0])
    assert abs(alm3[0] - 4 * np.sqrt(4 * np.pi)) < 1e-9
    assert np.abs(alm3[1:]).max() < 1e-6


def test_errors():
    import heracles_amd as hx

    plan = hx.get_plan(4, 8)
    with pytest.raises(hx.HxError):
        plan.map2alm(np.zeros((1, 12 * 16)), 1)
    with pytest.raises(hx.HxError):
        plan.map2alm(np.zeros((3, 12 * 16)), 2)
    with pytest.raises(ValueError):
        plan.map2alm(np.zeros((1, 10)), 0)


@pytest.mark.parametrize("case", range(12))
def test_map2alm_randomised_shapes(oracle, case):
    """Seeded random (nside, lmax, spin, batch, weights, filter): odd NSIDE, lmax up to 3 nside - 1 and
    beyond 2 nside (m aliasing on the polar rings), batches that hit every sweep variant, and l ranges of
    several flush spans (the 4x4x4 variants sweep 64-128 l per flush)."""
    import heracles_amd as hx

    rng = np.random.default_rng(9000 + case)
    nside = int(rng.choice([3, 5, 6, 8, 16, 24, 32, 64]))
    lmax = int(rng.integers(max(2, nside), 3 * nside))
    spin = int(rng.choice([0, 2]))
    ncomp = int(rng.choice([1, 2, 3, 4, 7, 8, 10, 12])) if spin == 0 else int(rng.choice([2, 4, 6, 8, 10, 16]))
    maps = rng.standard_normal((ncomp, 12 * nside**2))
    kw = {}
    if rng.random() < 0.5:
        kw["ring_weights"] = rng.uniform(0.8, 1.2, 2 * nside)
    if rng.random() < 0.3:
        kw["pix_weights"] = rng.uniform(0.9, 1.1, 12 * nside**2)
    fl = rng.uniform(0.5, 2.0, lmax + 1) if rng.random() < 0.5 else None
    plan = hx.get_plan(nside, lmax)
    out = plan.map2alm(maps, spin, fl=fl, **kw)
    ref = oracle.map2alm(maps, nside, lmax, spin=spin, **kw)
    if fl is not None:
        for m in range(lmax + 1):
            base = m * (2 * lmax + 1 - m) // 2
            ref[:, base + m : base + lmax + 1] *= fl[m:]
    close(out, ref)


def test_long_l_ranges_small_batches(oracle):
    """nside 64, lmax 191: six 32-l blocks per m, i.e. several flush spans for every NSUB in use."""
    import heracles_amd as hx

    rng = np.random.default_rng(77)
    nside, lmax = 64, 191
    plan = hx.get_plan(nside, lmax)
    for spin, ncomps in ((0, (1, 2, 4, 8)), (2, (2, 4, 8))):
        for ncomp in ncomps:
            maps = rng.standard_normal((ncomp, 12 * nside**2))
            close(plan.map2alm(maps, spin), oracle.map2alm(maps, nside, lmax, spin=spin))


@pytest.mark.parametrize("spin,ncomp", [(0, 10), (2, 20), (0, 16), (2, 8), (0, 3), (2, 2)])
def test_map2alm_m_chunked_batches_medium(oracle, spin, ncomp):
    """The sweep shapes bench.py times -- 10 spin-0 maps in one sweep (one 16-column group + one 4-column block),
    20 spin-2 components in one 40-column sweep, full two-group sweeps -- at a size where the sweep is cut into
    several m-chunks (scratch budget lowered through hx_set_scratch_budget) and polar pruning is active
    (nside 256: rings beyond m > lmax sin(theta) + 100 are skipped), against the oracle on every 16th m."""
    import heracles_amd as hx

    rng = np.random.default_rng(400 + 10 * spin + ncomp)
    nside, lmax = 256, 511
    maps = rng.standard_normal((ncomp, 12 * nside**2))
    import torch

    plan = hx.get_plan(nside, lmax)
    dev = torch.as_tensor(maps).cuda()  # (resident maps, as in the bench: host maps are cut into the upload pipeline's sweeps)
    whole = plan.map2alm(dev, spin).cpu().numpy()
    nchunk_default = plan.last_chunks
    # (<= 4 components go through one vector-unit sweep per map / field, whose operands and rows are a fraction of a 16-column sweep's)
    hx._lib.set_scratch_budget(2.5e6 * min(ncomp, 10) if ncomp > 4 else 2.0e6)
    try:
        out = plan.map2alm(dev, spin).cpu().numpy()
        assert plan.last_chunks >= 3, plan.last_chunks
    finally:
        hx._lib.set_scratch_budget(0)
    assert nchunk_default == 1
    np.testing.assert_array_equal(out, whole)  # the chunking does not change a bit
    host = plan.map2alm(maps, spin)  # the same from host arrays: other sweeps, same alms to rounding
    assert np.abs(host - out).max() <= 1e-12 * np.abs(out).max()
    stride = 16
    oracle.set_mstride(stride)
    try:
        ref = oracle.map2alm(maps, nside, lmax, spin=spin)
    finally:
        oracle.set_mstride(1)
    scale = np.abs(ref).max()
    for m in range(0, lmax + 1, stride):
        base = m * (2 * lmax + 1 - m) // 2
        err = np.abs(out[:, base + m : base + lmax + 1] - ref[:, base + m : base + lmax + 1]).max()
        assert err <= TOL * scale, (m, err / scale)


@pytest.mark.parametrize("nside,lmax", [(16, 32), (16, 47), (64, 100), (256, 511)])
@pytest.mark.parametrize("ncomp", [18, 20, 22])
def test_one_ring_set_kernel_device_batches(oracle, nside, lmax, ncomp):
    """9 - 10 spin-2 fields that are ALREADY IN HBM go through the Legendre kernel with one ring set per wave in one sweep of
    36 / 40 columns (host arrays are uploaded and transformed as two overlapped sweeps of the two-set kernel instead, which is
    what the other tests of this file exercise): against the oracle, with odd and even numbers of l-blocks, tasks of fewer
    than 4 ring blocks, 22 components = one 40-column sweep + one field on the vector-unit kernel (the split that costs least), and
    -- at nside 256 -- cut into m-chunks without changing a bit."""
    import torch
    import heracles_amd as hx

    rng = np.random.default_rng(900 + nside + ncomp)
    maps = rng.standard_normal((ncomp, 12 * nside**2))
    dev = torch.as_tensor(maps).cuda()
    plan = hx.get_plan(nside, lmax)
    out = plan.map2alm(dev, 2)
    got = out.cpu().numpy() if hasattr(out, "cpu") else np.asarray(out)
    if nside >= 256:
        hx._lib.set_scratch_budget(2.5e7)
        try:
            again = plan.map2alm(dev, 2)
            if ncomp != 22:  # (the last sweep of 22 components is the single field: it needs less scratch)
                assert plan.last_chunks >= 3, plan.last_chunks
        finally:
            hx._lib.set_scratch_budget(0)
        np.testing.assert_array_equal(again.cpu().numpy() if hasattr(again, "cpu") else np.asarray(again), got)
        stride = 16
        oracle.set_mstride(stride)
        try:
            ref = oracle.map2alm(maps, nside, lmax, spin=2)
        finally:
            oracle.set_mstride(1)
        scale = np.abs(got).max()
        for m in range(0, lmax + 1, stride):
            base = m * (2 * lmax + 1 - m) // 2
            sl = slice(base + m, base + lmax + 1)
            assert np.abs(got[:, sl] - ref[:, sl]).max() <= 1e-11 * scale, m
    else:
        close(got, oracle.map2alm(maps, nside, lmax, spin=2))
    # the host route (two sweeps of the two-set kernel) agrees to rounding
    host = plan.map2alm(maps, 2)
    assert np.abs(host - got).max() <= 1e-12 * np.abs(got).max()


@pytest.mark.parametrize("spin,ncomp", [(0, 10), (2, 10), (2, 16)])
def test_ring_groups_summed_in_fixed_order(spin, ncomp):
    """The pipelined Legendre kernel adds the ring groups of an m in place with f64 atomics: one work-group per m, program
    order -- so repeated runs agree bit for bit although several groups (nside 1024: 4 for spin 0, 8 for spin 2) and
    hundreds of work-groups on all XCDs add into the same buffer; and the sum over groups equals the transform of the
    northern-group and southern-group rings taken separately to rounding (linearity across the ring groups)."""
    import heracles_amd as hx

    rng = np.random.default_rng(77 + spin + ncomp)
    nside, lmax = 1024, 1535
    maps = rng.standard_normal((ncomp, 12 * nside**2))
    plan = hx.get_plan(nside, lmax)
    a = plan.map2alm(maps, spin)
    for _ in range(3):
        np.testing.assert_array_equal(plan.map2alm(maps, spin), a)
    # rings of the polar caps only / of the equatorial belt only: different ring groups carry the signal
    ncap = 2 * nside * (nside - 1)
    cap = np.zeros_like(maps)
    cap[:, :ncap] = maps[:, :ncap]
    cap[:, -ncap:] = maps[:, -ncap:]
    b = plan.map2alm(cap, spin) + plan.map2alm(maps - cap, spin)
    assert np.abs(a - b).max() <= 1e-12 * np.abs(a).max()


@pytest.mark.parametrize("nside,cap", [(64, 64), (32, 32), (48, 64)])
def test_split_bluestein_rings(oracle, nside, cap):
    """Rings whose Bluestein convolution exceeds the in-LDS FFT limit (nside 8192: 4096 < n < 8192 needs 16384 points)
    run as an even and an odd half-length pass.  The limit is lowered (hx_set_max_lds_fft) so that the cap rings of a
    small map take that path: analysis and synthesis against the oracle, and bit-equality of nothing -- only rounding
    differs from the unsplit plan."""
    import heracles_amd as hx

    rng = np.random.default_rng(nside + cap)
    lmax = 3 * nside // 2 + 5
    maps0 = rng.standard_normal((3, 12 * nside**2))
    maps2 = rng.standard_normal((4, 12 * nside**2))
    ref_plan = hx.Plan(nside, lmax)
    a0_ref, a2_ref = ref_plan.map2alm(maps0, 0), ref_plan.map2alm(maps2, 2)
    ref_plan.close()
    L = hx._lib.load()
    hx._lib.check(L.hx_set_max_lds_fft(cap))
    try:
        plan = hx.Plan(nside, lmax)
        a0, a2 = plan.map2alm(maps0, 0), plan.map2alm(maps2, 2)
        back = plan.alm2map(a0, 0)
        plan.close()
    finally:
        hx._lib.check(L.hx_set_max_lds_fft(8192))
    close(a0, oracle.map2alm(maps0, nside, lmax, spin=0))
    close(a2, oracle.map2alm(maps2, nside, lmax, spin=2))
    close(a0, a0_ref, 1e-12)
    close(a2, a2_ref, 1e-12)
    close(back, oracle.alm2map(a0, nside, lmax), 1e-11)
    with pytest.raises(hx.HxError):
        hx._lib.check(L.hx_set_max_lds_fft(100))


def test_map2alm_multi_matches_separate_calls(oracle):
    """hx_map2alm_multi (the loop of heracles/mapping.py:151-172 as one call with one upload pipeline across jobs): device
    inputs give bit-identical results to separate hx_map2alm calls; host inputs (cut into sweeps of <= 5 fields / 8 maps, the
    last sweep halved down to <= 2 units) agree with them to rounding and with the oracle; pixel weights and fl are applied."""
    import torch

    import heracles_amd as hx

    rng = np.random.default_rng(2024)
    nside, lmax = 32, 64
    npix = 12 * nside**2
    plan = hx.get_plan(nside, lmax)
    m2 = rng.standard_normal((14, npix))   # 7 spin-2 fields: two sweeps (4 + 3 fields)
    m0 = rng.standard_normal((11, npix))   # 11 spin-0 maps: 8 + 3, the last sweep halved: 2 + 1
    pw = 1.0 + 1e-2 * rng.standard_normal(npix)
    fl = 1.0 / (1.0 + 0.01 * np.arange(lmax + 1))
    sep2 = plan.map2alm(m2, 2, pix_weights=pw)
    sep0 = plan.map2alm(m0, 0, pix_weights=pw, fl=fl)
    got2, got0 = plan.map2alm_multi([(m2, 2, None), (m0, 0, None, fl)], pix_weights=pw)
    for got, sep in ((got2, sep2), (got0, sep0)):
        assert np.abs(got - sep).max() <= 1e-13 * np.abs(sep).max()
    close(got0[:3], oracle.map2alm(m0[:3], nside, lmax, spin=0, pix_weights=pw) * np.concatenate([fl[m:] for m in range(lmax + 1)]))
    close(got2[-2:], oracle.map2alm(m2[-2:], nside, lmax, spin=2, pix_weights=pw))
    # device-resident jobs: the sweeps of hx_map2alm itself, bit for bit; outputs written in place
    d2, d0, dpw = torch.as_tensor(m2).cuda(), torch.as_tensor(m0).cuda(), torch.as_tensor(pw).cuda()
    o2 = torch.empty((14, plan.nlm), dtype=torch.complex128, device="cuda")
    o0 = torch.empty((11, plan.nlm), dtype=torch.complex128, device="cuda")
    r2, r0 = plan.map2alm_multi([(d2, 2, o2), (d0, 0, o0)], pix_weights=dpw)
    assert r2 is o2 and r0 is o0
    np.testing.assert_array_equal(o2.cpu().numpy(), plan.map2alm(d2, 2, pix_weights=dpw).cpu().numpy())
    np.testing.assert_array_equal(o0.cpu().numpy(), plan.map2alm(d0, 0, pix_weights=dpw).cpu().numpy())
    # a mixed call: one host job, one device job
    h2, e0 = plan.map2alm_multi([(m2[:2], 2, None), (d0, 0, None)], pix_weights=pw)
    assert np.abs(h2 - sep2[:2]).max() <= 1e-13 * np.abs(sep2).max()
    np.testing.assert_array_equal(e0.cpu().numpy(), o0.cpu().numpy())


@pytest.mark.parametrize("device", [False, True])
def test_map2alm_list_of_separate_arrays(oracle, device):
    """hx_map2alm_list: one array per map as heracles holds them ((npix,) / (2, npix)), mixed spins in any order, gathered into
    the sweeps of the multi-transform pipeline without a stacked copy: against the oracle and against the per-spin batch."""
    import heracles_amd as hx

    rng = np.random.default_rng(77)
    nside, lmax = 32, 50
    npix = 12 * nside**2
    spins = [0, 2, 0, 0, 2, 2, 0, 2, 0, 0, 0, 2, 0, 0, 2, 0, 0, 0, 2]  # 12 spin-0 maps, 7 spin-2 fields
    maps = [rng.standard_normal((npix,) if s == 0 else (2, npix)) for s in spins]
    pw = rng.uniform(0.9, 1.1, npix)
    fl0, fl2 = rng.uniform(0.5, 1.5, lmax + 1), rng.uniform(0.5, 1.5, lmax + 1)
    plan = hx.get_plan(nside, lmax)
    if device:
        import torch

        args = [torch.as_tensor(m).cuda() for m in maps]
    else:
        args = maps
    out = plan.map2alm_list(args, spins, pix_weights=pw, fl0=fl0, fl2=fl2)
    assert len(out) == len(maps)
    for i in (0, 1, 5, 17, 18):
        got = out[i].cpu().numpy() if hasattr(out[i], "cpu") else np.asarray(out[i])
        ref = _apply_fl(oracle.map2alm(maps[i].reshape(-1, npix), nside, lmax, spin=spins[i], pix_weights=pw), fl0 if spins[i] == 0 else fl2, lmax)
        close(got.reshape(ref.shape), ref, 2e-11)
    s0 = np.stack([m for m, s in zip(maps, spins) if s == 0])
    b0 = plan.map2alm(s0, 0, pix_weights=pw, fl=fl0)
    k = 0
    for i, s in enumerate(spins):
        if s == 0:
            got = out[i].cpu().numpy() if hasattr(out[i], "cpu") else np.asarray(out[i])
            assert np.abs(got - b0[k]).max() <= 1e-12 * np.abs(b0).max()
            k += 1
    with pytest.raises(ValueError):
        plan.map2alm_list([maps[0]], [2])
    # with Jacobi iterations: the maps of a spin are gathered into one device array and iterated as a batch
    it = plan.map2alm_list(args[:6], spins[:6], niter=2)
    for i in (0, 1, 5):
        got = it[i].cpu().numpy() if hasattr(it[i], "cpu") else np.asarray(it[i])
        ref = oracle.map2alm(maps[i].reshape(-1, npix), nside, lmax, spin=spins[i], niter=2)
        close(got.reshape(ref.shape), ref, 1e-10)


@pytest.mark.parametrize("nside,lmax", [(64, 128), (128, 200)])
def test_host_maps_streamed_in_slabs_of_rings_are_bit_identical(nside, lmax):
    """Host maps go through StreamSweep (hx_sht_common.h): the rings of a sweep are uploaded slab by slab and every slab's ring FFTs,
    operand rows and completed ring groups run behind it.  The ring groups of an order are added in the order they always are, so the
    alms equal those of the same sweeps over device-resident maps bit for bit -- for one job (hx_map2alm), several jobs
    (hx_map2alm_multi, spin 2 then spin 0, with a batch that needs two sweeps) and separate arrays (hx_map2alm_list)."""
    import os

    import torch

    import heracles_amd as hx

    rng = np.random.default_rng(5)
    npix = 12 * nside**2
    plan = hx.get_plan(nside, lmax)
    pw = 1.0 + 1e-2 * rng.standard_normal(npix)
    dpw = torch.as_tensor(pw).cuda()
    for spin, ncomp in ((2, 20), (2, 14), (0, 10), (0, 16), (2, 26), (0, 5)):
        m = rng.standard_normal((ncomp, npix))
        ref = plan.map2alm(torch.as_tensor(m).cuda(), spin, pix_weights=dpw).cpu().numpy()
        got = plan.map2alm(m, spin, pix_weights=pw)
        np.testing.assert_array_equal(got, ref)
    m2, m0 = rng.standard_normal((20, npix)), rng.standard_normal((10, npix))
    fl = 1.0 / (1.0 + 0.01 * np.arange(lmax + 1))
    r2 = plan.map2alm(torch.as_tensor(m2).cuda(), 2, pix_weights=dpw).cpu().numpy()
    r0 = plan.map2alm(torch.as_tensor(m0).cuda(), 0, pix_weights=dpw, fl=torch.as_tensor(fl).cuda()).cpu().numpy()
    g2, g0 = plan.map2alm_multi([(m2, 2, None), (m0, 0, None, fl)], pix_weights=pw)
    np.testing.assert_array_equal(g2, r2)
    np.testing.assert_array_equal(g0, r0)
    # one array per map, spins interleaved
    spins = [0, 2] * 10
    maps = [m0[i // 2] if s == 0 else m2[2 * (i // 2):2 * (i // 2) + 2] for i, s in enumerate(spins)]
    out = plan.map2alm_list(maps, spins, pix_weights=pw, fl0=fl)
    for i, s in enumerate(spins):
        want = r0[i // 2] if s == 0 else r2[2 * (i // 2):2 * (i // 2) + 2]
        np.testing.assert_array_equal(np.asarray(out[i]).reshape(want.shape), want)


def test_host_maps_fall_back_to_whole_map_sweeps_when_a_streamed_sweep_does_not_fit():
    """a streamed sweep holds F and the accumulation rows of all orders, Y of the whole sweep and two staging buffers.  When
    that does not fit (here: a scratch budget of 1 MB; on a device: the HBM that is free) the call must not fail -- host maps then go
    through the capped sweeps of whole maps, whose sums differ from the resident sweep's by rounding only."""
    import torch

    import heracles_amd as hx
    from heracles_amd import _lib

    nside, lmax = 64, 100
    rng = np.random.default_rng(77)
    npix = 12 * nside**2
    plan = hx.get_plan(nside, lmax)
    m = rng.standard_normal((20, npix))
    ref = plan.map2alm(torch.as_tensor(m).cuda(), 2).cpu().numpy()
    import os

    streamed = plan.map2alm(m, 2)
    np.testing.assert_array_equal(streamed, ref)
    _lib.set_scratch_budget(1e6)
    try:
        small = plan.map2alm(m, 2)                     # (m-chunked, whole maps in sweeps of five fields)
    finally:
        _lib.set_scratch_budget(0)
    assert np.abs(small - ref).max() <= 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize("spin", [0, 2])
def test_matrix_unit_synthesis_sweeps_shrink_to_the_memory_there_is(oracle, spin):
    """A sweep of the batched synthesis holds ring modes, ring spectra, Y and its operand table (213 + 129 GB for ten fields at nside
    8192): it is cut to what fits (here: the scratch budget), and below the matrix kernel's smallest batch the vector-unit kernel takes
    the rest -- never an allocation failure.  Results: the full sweep's to rounding."""
    import heracles_amd as hx
    from heracles_amd import _lib

    nside, lmax = 64, 150
    unit = 1 if spin == 0 else 2
    units = 12 if spin == 0 else 10
    rng = np.random.default_rng(400 + spin)
    alm = random_alm(rng, lmax, spin, (units * unit,))
    plan = hx.get_plan(nside, lmax)
    full = plan.alm2map(alm, spin)
    close(full[:unit], oracle.alm2map(alm[:unit], nside, lmax, spin=spin), 1e-11)
    scale = np.abs(full).max()
    # bytes of a sweep of u units at this size: ring modes (lmax + 1) nrp_pad 4 nc 8 + two arrays of ny nc 16 + the table
    for budget in (6e6, 2.5e6, 1e5):            # ~ 6 units, ~ 2-3 units, nothing: vector-unit kernel
        _lib.set_scratch_budget(budget)
        try:
            got = plan.alm2map(alm, spin)
        finally:
            _lib.set_scratch_budget(0)
        assert np.abs(got - full).max() <= 1e-12 * scale, budget


def test_plan_release_scratch_and_regrow():
    """hx_plan_release_scratch frees the transient HBM scratch of a plan (operands, rows, ring spectra, staging, synthesis tables);
    the next call allocates what it needs again and gives the same bits."""
    import heracles_amd as hx

    nside, lmax = 64, 100
    rng = np.random.default_rng(12)
    m = rng.standard_normal((12, 12 * nside**2))
    plan = hx.Plan(nside, lmax)
    a = plan.map2alm(m, 2, niter=1)
    back = plan.alm2map(a, 2)
    held = plan.scratch_bytes
    plan.release_scratch()
    assert plan.scratch_bytes < held
    np.testing.assert_array_equal(plan.map2alm(m, 2, niter=1), a)
    np.testing.assert_array_equal(plan.alm2map(a, 2), back)
    plan.release_scratch()
    plan.close()
