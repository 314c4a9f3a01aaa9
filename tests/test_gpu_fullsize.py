"""Parity at BASELINE.json's full size (nside 4096, lmax 6144), where the oracle cannot run a whole
transform in test time: size-independent identities that tie the kernels to each other and to
closed forms, plus the oracle itself on a SAMPLE of m (its Legendre stage restricted to every
512th m finishes in seconds on the GPU box's host cores)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NSIDE, LMAX = 4096, 6144
NPIX = 12 * NSIDE * NSIDE
NLM = (LMAX + 1) * (LMAX + 2) // 2


@pytest.fixture(scope="module")
def plan():
    import heracles_amd as hx

    return hx.get_plan(NSIDE, LMAX)


def _random_alm(torch, n, seed, lmin=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randn((n, NLM, 2), dtype=torch.float64, device="cuda", generator=g)
    a[:, : LMAX + 1, 1] = 0.0  # m = 0 is real
    a = torch.view_as_complex(a)
    if lmin:  # spin 2: l < 2 carries nothing
        for m in range(lmin):
            base = m * (2 * LMAX + 1 - m) // 2
            a[:, base + m : base + lmin] = 0.0
    return a


def _mweights(torch):
    w = torch.full((NLM,), 2.0, dtype=torch.float64, device="cuda")
    w[: LMAX + 1] = 1.0
    return w


@pytest.mark.parametrize("spin", [0, 2])
def test_analysis_is_the_adjoint_of_synthesis(plan, spin):
    """<alm2map(a), x> 4pi/npix == sum_{m>=0} (2 - delta_m0) Re(conj(a) . map2alm(x)) with unit
    weights and no iterations -- exact up to rounding for ANY a and x, so every (l, m, ring) term
    of the two MFMA kernels and both directions of the ring FFT are checked against each other."""
    import torch

    nc = 1 if spin == 0 else 2
    g = torch.Generator(device="cuda").manual_seed(17 + spin)
    x = torch.randn((nc, NPIX), dtype=torch.float64, device="cuda", generator=g)
    a = _random_alm(torch, nc, 23 + spin, lmin=spin)
    y = torch.empty_like(x)
    plan.alm2map(a, spin, out=y)
    b = torch.empty_like(a)
    plan.map2alm(x, spin, out=b, niter=0)
    w = _mweights(torch)
    lhs = float((x * y).sum()) * 4.0 * np.pi / NPIX
    rhs = float((w * (a.real * b.real + a.imag * b.imag)).sum())
    scale = float(torch.linalg.vector_norm(x) * torch.linalg.vector_norm(y)) * 4.0 * np.pi / NPIX
    assert abs(lhs - rhs) <= 1e-10 * scale, (lhs, rhs, scale)
    # bit-reproducible: the reductions have a fixed association
    b2 = torch.empty_like(b)
    plan.map2alm(x, spin, out=b2, niter=0)
    assert torch.equal(b, b2)


@pytest.mark.parametrize("spin", [0, 2])
def test_synthesis_sweeps_of_several_maps_full_size(plan, spin):
    """The multi-map shapes of the vector-unit synthesis kernel at nside 4096 (four spin-0 maps / two spin-2 fields share one
    recursion): bitwise equal to the single-map sweeps, which the adjointness test above ties to the analysis."""
    import torch

    nc = 4  # one sweep of 4 maps / 2 fields (larger batches run on the matrix unit: next test)
    a = _random_alm(torch, nc, 31 + spin, lmin=spin)
    y = torch.empty((nc, NPIX), dtype=torch.float64, device="cuda")
    plan.alm2map(a, spin, out=y)
    unit = 1 if spin == 0 else 2
    one = torch.empty((unit, NPIX), dtype=torch.float64, device="cuda")
    for c in (0, nc - unit):
        plan.alm2map(a[c : c + unit], spin, out=one)
        assert torch.equal(one, y[c : c + unit])
    assert bool(torch.isfinite(y).all())


@pytest.mark.parametrize("spin,units", [(0, 10), (2, 10), (2, 5), (0, 20)])
def test_matrix_unit_synthesis_full_size(plan, spin, units):
    """Batched synthesis on the matrix unit (k_synth_duo, round 5) at nside 4096 / lmax 6144 in the shapes the Jacobi iterations of the
    bench's job use (ten maps, ten fields) and two more: first, middle and last unit against the single-unit sweeps of the vector-unit
    kernel (other order of summation: 1e-12 of the largest pixel), which the adjointness and closed-form tests of this file tie to the
    analysis and to an independent recursion; bitwise repeatable (no atomics, no order that depends on timing)."""
    import torch

    unit = 1 if spin == 0 else 2
    nc = units * unit
    a = _random_alm(torch, nc, 57 + spin + units, lmin=spin)
    y = torch.empty((nc, NPIX), dtype=torch.float64, device="cuda")
    plan.alm2map(a, spin, out=y)
    scale = float(y.abs().max())
    one = torch.empty((unit, NPIX), dtype=torch.float64, device="cuda")
    worst = 0.0
    for u in (0, units // 2, units - 1):
        plan.alm2map(a[u * unit : (u + 1) * unit], spin, out=one)
        worst = max(worst, float((one - y[u * unit : (u + 1) * unit]).abs().max()))
    assert worst <= 1e-12 * scale, worst / scale
    y2 = torch.empty_like(y)
    plan.alm2map(a, spin, out=y2)
    assert torch.equal(y, y2)
    del y2
    assert bool(torch.isfinite(y).all())


@pytest.mark.parametrize("spin", [0, 2])
def test_linearity_full_size(plan, spin):
    import torch

    nc = 2
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((nc, NPIX), dtype=torch.float64, device="cuda", generator=g)
    z = torch.randn((nc, NPIX), dtype=torch.float64, device="cuda", generator=g)
    ax, az, am = (torch.empty((nc, NLM), dtype=torch.complex128, device="cuda") for _ in range(3))
    plan.map2alm(x, spin, out=ax)
    plan.map2alm(z, spin, out=az)
    plan.map2alm(0.75 * x - 2.5 * z, spin, out=am)
    ref = 0.75 * ax - 2.5 * az
    assert float((am - ref).abs().max()) <= 1e-12 * float(ref.abs().max())


def test_synthesis_against_closed_form_on_rings(plan, oracle):
    """alm2map of a few m only, compared on selected rings with sum_l a_lm lambda_lm(theta) e^{i m phi},
    lambda from the extended-precision textbook recursion of tests/helpers.py (independent of the
    oracle's and the kernels' recursions); rings from both caps, the belt and the poles, m up to 6000."""
    import torch
    from helpers import lambda_lm_column

    rng = np.random.default_rng(99)
    ms = [0, 1, 2, 37, 700, 3001, 6000]
    alm = np.zeros(NLM, dtype=np.complex128)
    for m in ms:
        base = m * (2 * LMAX + 1 - m) // 2
        v = rng.standard_normal(LMAX + 1 - m) + 1j * rng.standard_normal(LMAX + 1 - m) * (m > 0)
        alm[base + m : base + LMAX + 1] = v / np.sqrt(1.0 + np.arange(m, LMAX + 1))
    y = torch.empty((1, NPIX), dtype=torch.float64, device="cuda")
    plan.alm2map(torch.as_tensor(alm[None]).cuda(), 0, out=y)
    rings = (1, 3, 1000, 4095, 4096, 6001, 8192, 12000, 16383 - 2, 16383)
    info = [oracle.ring_info(NSIDE, r) for r in rings]
    zs, sths = np.array([i[2] for i in info]), np.array([i[3] for i in info])
    fm = {}
    for m in ms:
        lam = lambda_lm_column(m, LMAX, zs, sths)                      # (l, ring), longdouble
        base = m * (2 * LMAX + 1 - m) // 2
        a = alm[base + m : base + LMAX + 1]
        fm[m] = (np.sum(a.real[:, None] * lam, axis=0) + 1j * np.sum(a.imag[:, None] * lam, axis=0)).astype(np.complex128)
    for k, (sp, nphi, z, sth, phi0) in enumerate(info):
        got = y[0, sp : sp + nphi].cpu().numpy()
        phi = phi0 + 2 * np.pi * np.arange(nphi) / nphi
        exp = np.zeros(nphi)
        for m in ms:
            exp += (1.0 if m == 0 else 2.0) * np.real(fm[m][k] * np.exp(1j * m * phi))
        assert np.abs(got - exp).max() <= 1e-9 * max(np.abs(exp).max(), 1.0), rings[k]


@pytest.mark.parametrize("spin", [0, 2])
def test_map2alm_against_oracle_on_sampled_m(plan, oracle, spin):
    """The oracle's own map2alm at full size with its Legendre stage restricted to every 512th m
    (all rings, all l): the GPU result must agree on those m."""
    import torch

    nc = 1 if spin == 0 else 2
    g = torch.Generator(device="cuda").manual_seed(3 + spin)
    x = torch.randn((nc, NPIX), dtype=torch.float64, device="cuda", generator=g)
    b = torch.empty((nc, NLM), dtype=torch.complex128, device="cuda")
    plan.map2alm(x, spin, out=b, niter=0)
    stride = 512
    oracle.set_mstride(stride)
    try:
        ref = oracle.map2alm(x.cpu().numpy(), NSIDE, LMAX, spin=spin)
    finally:
        oracle.set_mstride(1)
    got = b.cpu().numpy()
    scale = np.abs(got).max()
    for m in range(0, LMAX + 1, stride):
        base = m * (2 * LMAX + 1 - m) // 2
        sl = slice(base + m, base + LMAX + 1)
        # both sides evaluate lambda_lm with relative error O(m eps) (ring co-latitudes are known to eps):
        # measured 1.5e-11 of max|alm| at m = 3584; the bound leaves a factor 6
        assert np.abs(got[:, sl] - ref[:, sl]).max() <= 1e-10 * scale, m


@pytest.mark.parametrize("spin,ncomp,check", [(0, 10, (0, 7, 8, 9)), (2, 20, (0, 1, 8, 9, 10, 11, 18, 19))])
def test_bench_batch_shapes_against_oracle_on_sampled_m(plan, oracle, spin, ncomp, check):
    """The batches bench.py times, at its size: 10 spin-0 maps = ONE sweep of the pipelined kernel (a 16-column group + a
    4-column block: components 8, 9 sit in the block), 20 spin-2 components = ONE sweep of 40 columns on the kernel with one
    ring set per wave (fields 0 and 4 in the first and second 16-column group, 5 in the second group, 9 in the second
    4-column block).  Components of every column group / block against the oracle on every 512th m (all rings, all l);
    one work-group per m adds its 16 (spin 0) / 64 (spin 2) ring groups in place there."""
    import torch

    g = torch.Generator(device="cuda").manual_seed(30 + spin)
    x = torch.randn((ncomp, NPIX), dtype=torch.float64, device="cuda", generator=g)
    b = torch.empty((ncomp, NLM), dtype=torch.complex128, device="cuda")
    plan.map2alm(x, spin, out=b, niter=0)
    assert plan.last_chunks == 1  # F and the rows of a whole sweep fit the default budget (spin 2: 64 + 7 GB of 80)
    stride = 512
    step = 1 if spin == 0 else 2
    for c0 in check[::step]:
        oracle.set_mstride(stride)
        try:
            ref = oracle.map2alm(x[c0 : c0 + step].cpu().numpy(), NSIDE, LMAX, spin=spin)
        finally:
            oracle.set_mstride(1)
        got = b[c0 : c0 + step].cpu().numpy()
        scale = np.abs(got).max()
        for m in range(0, LMAX + 1, stride):
            base = m * (2 * LMAX + 1 - m) // 2
            sl = slice(base + m, base + LMAX + 1)
            assert np.abs(got[:, sl] - ref[:, sl]).max() <= 1e-10 * scale, (c0, m)
    del x, b
    torch.cuda.empty_cache()


def test_alm2cl_full_size_against_numpy(plan):
    import torch
    import heracles_amd as hx

    a = _random_alm(torch, 3, 41)
    w = _mweights(torch)
    idx_l = torch.cat([torch.arange(m, LMAX + 1, device="cuda") for m in range(LMAX + 1)])
    comps = [a[i] for i in range(3)]
    pairs = [(0, 0), (0, 1), (1, 2), (2, 2)]
    cls = hx.twopoint.alm2cl_pairs(comps, pairs, LMAX)
    for k, (i, j) in enumerate(pairs):
        prod = w * (a[i].real * a[j].real + a[i].imag * a[j].imag)
        ref = torch.zeros(LMAX + 1, dtype=torch.float64, device="cuda").index_add_(0, idx_l, prod)
        ref = (ref / (2.0 * torch.arange(LMAX + 1, device="cuda") + 1.0)).cpu().numpy()
        np.testing.assert_allclose(np.asarray(cls[k]), ref, rtol=1e-11, atol=1e-14)


def test_nside_8192_against_oracle_on_sampled_m(oracle):
    """/root/reference/examples/heracles.cfg configures maps up to nside 8192 (lmax 8000): the cap rings with
    4096 < n < 8192 take the split Bluestein path (two 8192-point halves of the 16384-point convolution).  The oracle's
    own map2alm with its Legendre stage restricted to every 1000th m must agree on those m, spin 0 and spin 2; the
    synthesis of a few m agrees with the closed form on rings of the split class."""
    import torch
    import heracles_amd as hx
    from helpers import lambda_lm_column

    nside, lmax = 8192, 8000
    npix, nlm = 12 * nside * nside, (lmax + 1) * (lmax + 2) // 2
    hx.sht.clear_plans()  # the nside-4096 tables are not needed any more
    torch.cuda.empty_cache()
    plan = hx.Plan(nside, lmax)
    try:
        for spin in (0, 2):
            nc = 1 if spin == 0 else 2
            g = torch.Generator(device="cuda").manual_seed(81 + spin)
            x = torch.randn((nc, npix), dtype=torch.float64, device="cuda", generator=g)
            b = torch.empty((nc, nlm), dtype=torch.complex128, device="cuda")
            plan.map2alm(x, spin, out=b, niter=0)
            stride = 1000
            oracle.set_mstride(stride)
            try:
                ref = oracle.map2alm(x.cpu().numpy(), nside, lmax, spin=spin)
            finally:
                oracle.set_mstride(1)
            got = b.cpu().numpy()
            scale = np.abs(got).max()
            for m in range(0, lmax + 1, stride):
                base = m * (2 * lmax + 1 - m) // 2
                sl = slice(base + m, base + lmax + 1)
                assert np.abs(got[:, sl] - ref[:, sl]).max() <= 2e-10 * scale, (spin, m)
            del x, b, ref, got
            torch.cuda.empty_cache()
        # synthesis: rings 5000 (n = 5000, split class), 8191 and the equator against sum_l a_lm lambda_lm e^{i m phi}
        rng = np.random.default_rng(7)
        ms = [0, 3, 1500, 7000]
        alm = np.zeros(nlm, dtype=np.complex128)
        for m in ms:
            base = m * (2 * lmax + 1 - m) // 2
            v = rng.standard_normal(lmax + 1 - m) + 1j * rng.standard_normal(lmax + 1 - m) * (m > 0)
            alm[base + m : base + lmax + 1] = v / np.sqrt(1.0 + np.arange(m, lmax + 1))
        y = torch.empty((1, npix), dtype=torch.float64, device="cuda")
        plan.alm2map(torch.as_tensor(alm[None]).cuda(), 0, out=y)
        rings = (4100, 5000, 8191, 16384, 4 * nside - 5000)
        info = [oracle.ring_info(nside, r) for r in rings]
        zs, sths = np.array([i[2] for i in info]), np.array([i[3] for i in info])
        for k, (sp, nphi, z, sth, phi0) in enumerate(info):
            got = y[0, sp : sp + nphi].cpu().numpy()
            phi = phi0 + 2 * np.pi * np.arange(nphi) / nphi
            exp = np.zeros(nphi)
            for m in ms:
                lam = lambda_lm_column(m, lmax, zs[k : k + 1], sths[k : k + 1])[:, 0]
                base = m * (2 * lmax + 1 - m) // 2
                a = alm[base + m : base + lmax + 1]
                fm = complex(np.sum(a.real * lam), np.sum(a.imag * lam))
                exp += (1.0 if m == 0 else 2.0) * np.real(fm * np.exp(1j * m * phi))
            assert np.abs(got - exp).max() <= 1e-9 * max(np.abs(exp).max(), 1.0), rings[k]
    finally:
        plan.close()


def test_nside_8192_batched_sweeps_on_the_matrix_unit(oracle):
    """nside 8192 / lmax 8000 (examples/heracles.cfg:56-62) through the BATCHED kernels: three spin-2 fields and five spin-0 maps per
    call, so that k_legendre_duo<2,*> / <0,*> and k_synth_duo run beyond the lmax 6144 their dead-block margins were first calibrated
    at (VERDICT r5 Next #3; tools/calibrate_dead_blocks.py shows the margins needed saturate: profiles/r06_dead_block_calibration.txt).
    map2alm: the last unit of each batch against the oracle on every 1000th m (2e-10 of the largest alm), every unit
    against its own single-unit call (the vector-unit kernels: no dead-block rule, oracle-checked above) at 1e-12; alm2map: every unit
    of the batch against its single-unit sweep at 1e-12 of the largest pixel."""
    import torch
    import heracles_amd as hx

    nside, lmax = 8192, 8000
    npix, nlm = 12 * nside * nside, (lmax + 1) * (lmax + 2) // 2
    hx.sht.clear_plans()
    torch.cuda.empty_cache()
    plan = hx.Plan(nside, lmax)
    stride = 1000
    try:
        for spin, units in ((2, 3), (0, 5)):
            nc = 1 if spin == 0 else 2
            g = torch.Generator(device="cuda").manual_seed(91 + spin)
            x = torch.randn((units * nc, npix), dtype=torch.float64, device="cuda", generator=g)
            b = torch.empty((units * nc, nlm), dtype=torch.complex128, device="cuda")
            hx._lib.executed_flops(reset=True)
            plan.map2alm(x, spin, out=b, niter=0)
            assert hx._lib.executed_flops(reset=True)[0] > 0, "the batch did not run on the matrix unit"
            scale = float(b.abs().max())
            one = torch.empty((nc, nlm), dtype=torch.complex128, device="cuda")
            for u in range(units):
                plan.map2alm(x[u * nc : (u + 1) * nc], spin, out=one, niter=0)
                assert float((one - b[u * nc : (u + 1) * nc]).abs().max()) <= 1e-12 * scale, (spin, u)
            del one
            for u in (units - 1,):  # (the oracle's ring stage of an nside-8192 component takes ~10 s on the box's cores: one unit per batch)
                oracle.set_mstride(stride)
                try:
                    ref = oracle.map2alm(x[u * nc : (u + 1) * nc].cpu().numpy(), nside, lmax, spin=spin)
                finally:
                    oracle.set_mstride(1)
                got = b[u * nc : (u + 1) * nc].cpu().numpy()
                for m in range(0, lmax + 1, stride):
                    base = m * (2 * lmax + 1 - m) // 2
                    sl = slice(base + m, base + lmax + 1)
                    assert np.abs(got[:, sl] - ref[:, sl]).max() <= 2e-10 * scale, (spin, u, m)
                del ref, got
            # synthesis of the batch's own alms (band-limited input of the right symmetry) through k_synth_duo vs single-unit sweeps
            b[:, : lmax + 1] = b[:, : lmax + 1].real.to(torch.complex128)
            y = x  # (the maps are not needed any more: their buffer takes the synthesis)
            plan.alm2map(b, spin, out=y)
            ymax = max(float(y[i].abs().max()) for i in range(y.shape[0]))
            back = torch.empty((nc, npix), dtype=torch.float64, device="cuda")
            for u in range(units):
                plan.alm2map(b[u * nc : (u + 1) * nc], spin, out=back)
                back -= y[u * nc : (u + 1) * nc]
                assert float(back.abs_().max()) <= 1e-12 * ymax, (spin, u)
            del x, y, b, back
            torch.cuda.empty_cache()
    finally:
        plan.close()


def test_config5_euclid_job_on_one_gpu(oracle):
    """BASELINE configs[4] in its own shape on ONE GPU: 13 bins x (2 spin-0 + 1 spin-2) at nside 4096 / lmax 6144 = 39 maps / 52
    components / 780 map pairs (sizes: heracles/examples/heracles.cfg:28-62; ~100 GB of maps and alms, 288 GB on the device).
    Its sweeps -- 26 spin-0 maps = 16 + 10 (two full 16-column groups; one group + one 4-column block), 13 spin-2 fields = 8 + 5, the
    split that costs least -- are not the bench's: components of different sweeps and column groups against the oracle on every 512th m, spectra of map pairs
    against a direct device-side sum, and the L = 6144 mixing matrices against the 3j oracle on a corner."""
    import torch

    import heracles_amd as hx
    from heracles_amd import distributed as hxd

    torch.cuda.empty_cache()
    plan = hx.get_plan(NSIDE, LMAX)  # (the nside-8192 tests above release the module's plan to make room)
    nb = 13
    spins = [0] * (2 * nb) + [2] * nb
    work = hxd.ShardedTwoPoint(spins, 1, 0, NLM, LMAX)
    alm0, alm2 = work.local_alm_views("cuda")
    g = torch.Generator(device="cuda").manual_seed(555)
    maps0 = torch.randn((2 * nb, NPIX), dtype=torch.float64, device="cuda", generator=g)
    maps2 = torch.randn((nb, 2, NPIX), dtype=torch.float64, device="cuda", generator=g)
    plan.map2alm(maps0, 0, out=alm0)
    plan.map2alm(maps2.view(2 * nb, NPIX), 2, out=alm2.view(2 * nb, NLM))
    stride = 512

    def check(got, x, spin, tag):
        oracle.set_mstride(stride)
        try:
            ref = oracle.map2alm(x.cpu().numpy(), NSIDE, LMAX, spin=spin)
        finally:
            oracle.set_mstride(1)
        got = got.cpu().numpy()
        scale = np.abs(got).max()
        for m in range(0, LMAX + 1, stride):
            base = m * (2 * LMAX + 1 - m) // 2
            sl = slice(base + m, base + LMAX + 1)
            assert np.abs(got[:, sl] - ref[:, sl]).max() <= 1e-10 * scale, (tag, m)

    # spin 0: map 9 (first sweep, second column group), map 25 (second sweep, its 4-column block); spin 2: field 6 (first sweep,
    # second group), field 12 (second sweep, the field of its 4-column block)
    check(alm0[9:10], maps0[9:10], 0, "spin0 map 9")
    check(alm0[25:26], maps0[25:26], 0, "spin0 map 25")
    check(alm2[6], maps2[6], 2, "spin2 field 6")
    check(alm2[12], maps2[12], 2, "spin2 field 12")
    del maps0, maps2
    torch.cuda.empty_cache()
    cls = work.all_pairs_cl()
    assert len(work.pairs) == 780 and cls.shape == (work.nrows, LMAX + 1) and work.nrows == 26 * 27 // 2 + 26 * 13 * 2 + 4 * (13 * 14 // 2)
    w = _mweights(torch)
    idx_l = torch.cat([torch.arange(m, LMAX + 1, device="cuda") for m in range(LMAX + 1)])
    buf = work.buffer()
    nchk = 0
    for (i, j) in ((0, 0), (3, 25), (25, 26), (12, 38), (30, 38), (38, 38)):
        row = work.row0[i, j]
        for ka, ca in enumerate(work.comps_of_map[i]):
            for kb, cb in enumerate(work.comps_of_map[j]):
                a, b_ = buf[ca], buf[cb]
                ref = torch.zeros(LMAX + 1, dtype=torch.float64, device="cuda").index_add_(0, idx_l, w * (a.real * b_.real + a.imag * b_.imag))
                ref = (ref / (2.0 * torch.arange(LMAX + 1, device="cuda") + 1.0)).cpu().numpy()
                got = cls[row + ka * len(work.comps_of_map[j]) + kb]
                assert np.abs(got - ref).max() <= 1e-11 * np.abs(ref).max(), (i, j, ka, kb)
                nchk += 1
    assert nchk >= 8
    del buf, w, idx_l, work, alm0, alm2
    torch.cuda.empty_cache()
    # mixing matrices of the job: L = 6144
    L = LMAX
    ell = np.arange(L + 1)
    wl = 4 * np.pi * 0.35 * np.exp(-ell * (ell + 1) / 3000.0) + 1e-3 / (1.0 + ell) ** 2
    eb = hx.mixmat_eb(wl)
    assert eb.shape == (3, L + 1, L + 1)
    np.testing.assert_allclose(eb[2], eb[0] - eb[1], atol=1e-12 * np.abs(eb[0]).max())
    ref = oracle.mixmat_eb(wl, l1max=23, l2max=200)
    np.testing.assert_allclose(eb[:, :24, :201], ref, atol=1e-12 * np.abs(ref).max())
