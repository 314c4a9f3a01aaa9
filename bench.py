#!/usr/bin/env python3
"""bench.py -- map -> Cl pairs/s (+ mixing-matrix build seconds) on MI355X.

Metric (BASELINE.json): "map->Cl pairs/sec + mixing-matrix build sec, nside=4096 lmax=6144".
Workload at N = 1 (the north_star target, which fits one GPU): 10 spin-0 + 10 spin-2 maps
(30 components) at nside=4096, lmax=6144, synthetic Gaussian pixels, resident in HBM when
the timed region starts.  One "step" = batched map2alm of all maps (niter=0, unit ring
weights, a synthetic full-sky pixel-weight array) + all auto/cross Cl of every map pair + D2H of the Cl blocks.
`value` is that device-resident rate (the task contract: inputs resident in HBM when the timed region starts);
`value_host_to_host` is SURVEY 8d's definition (pageable host maps in, Cl blocks on the host out; median of --host-steps).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1, --scaling weak (default): every rank brings its own 20 maps, alms are all-gathered over RCCL/xGMI, the
tiled pair list over all 20*N maps is dealt to the ranks (per-GPU SHT work is fixed; pairs grow as N^2).
N > 1, --scaling strong: the SAME 20-map job is dealt to the N ranks by cost (a spin-2 map = 3 units).
Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6  # MI355X datasheet FP64 matrix (= vector) peak
HBM_PEAK_GBS = 8000.0
PROFILE_ROUND = "r04"    # profiles/<round>_traffic.json holds the committed PMC passes of THIS round's kernels


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--nside", type=int, default=4096)
    p.add_argument("--lmax", type=int, default=6144)
    p.add_argument("--nbins", type=int, default=10, help="tomographic bins: nbins x (spin-0, spin-2) maps per GPU (weak) / in all (strong)")
    p.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    p.add_argument("--workload", choices=("north_star", "euclid"), default="north_star",
                   help="north_star: nbins x (spin-0, spin-2) maps (the default line); euclid: BASELINE configs[4] on ONE GPU -- "
                        "13 bins x (2 spin-0 + 1 spin-2) = 39 maps / 52 components / 780 pairs (not the driver's line)")
    p.add_argument("--host-steps", type=int, default=5, help="steps of the host -> host leg (median reported)")
    p.add_argument("--mixmat-lmax", type=int, default=None, help="L of the timed mixmat_eb (default: lmax)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-mixmat", action="store_true")
    p.add_argument("--no-host-leg", action="store_true", help="skip the host -> host (PCIe-inclusive) measurement")
    p.add_argument("--no-verify", action="store_true")
    p.add_argument("--no-single", action="store_true", help="skip the single-map transform timings")
    p.add_argument("--generic-weights", action="store_true", help="pixel weights without the symmetry of healpy's files (generic path of the ring kernels)")
    return p.parse_args()


def cpu_engines():
    """The CPU engines SURVEY 8d names, from a real import attempt on this box."""
    out = {}
    for name in ("ducc0", "healpy"):
        try:
            mod = __import__(name)
            out[name] = getattr(mod, "__version__", "available")
        except Exception as exc:  # noqa: BLE001
            out[name] = f"unavailable ({type(exc).__name__})"
    return out


def oracle_sample(nside, lmax, t_map, qu_map, stride, pix_weights=None):
    """Oracle (CPU restatement, kind "port") map2alm of one spin-0 map and one spin-2 map at full size: all rings,
    every `stride`-th m of the Legendre stage.  Returns (alm0, alm2, timings) -- used as the cpu_baseline sample AND as
    the checker of the GPU alms on those m."""
    from oracle import hxoracle as ho

    ho.set_mstride(stride)
    try:
        a0 = ho.map2alm(t_map, nside, lmax, spin=0, pix_weights=pix_weights)
        f0, l0 = ho.last_timings()
        a2 = ho.map2alm(qu_map, nside, lmax, spin=2, pix_weights=pix_weights)
        f2, l2 = ho.last_timings()
    finally:
        ho.set_mstride(1)
    return a0, a2, (f0, l0, f2, l2)


def cpu_baseline(nside, lmax, nmaps0, nmaps2, a0, a2, tim, stride):
    from oracle import hxoracle as ho

    f0, l0, f2, l2 = tim
    t2 = time.perf_counter()
    ho.alm2cl(a0, a0), ho.alm2cl(a0, a2), ho.alm2cl(a2, a2)
    t3 = time.perf_counter()
    nmaps = nmaps0 + nmaps2
    npairs = nmaps * (nmaps + 1) // 2
    ncs = nmaps0 * (nmaps0 + 1) // 2 + 4 * (nmaps2 * (nmaps2 + 1) // 2) + 2 * nmaps0 * nmaps2
    s0, s2 = f0 + l0 * stride, f2 + l2 * stride  # full-transform estimates
    total = nmaps0 * s0 + nmaps2 * s2 + (t3 - t2) * ncs / 6.0
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next(line.split(":", 1)[1].strip() for line in f if line.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return {
        "value": npairs / total, "unit": "map->Cl pairs/s", "cores": ho.num_threads(), "cpu_model": cpu_model,
        "host_logical_cpus": os.cpu_count(), "kind": "port",
        "engines": cpu_engines(),
        "note": "a scalar C restatement of the algorithm (oracle/), NOT ducc0 / healpy: those engines are absent from this "
                "image (see engines); the GPU/CPU ratio says nothing about kernel quality -- roofline.frac does",
        "sample": f"oracle map2alm of 1 spin-0 + 1 spin-2 map of the bench's own input at nside={nside} lmax={lmax}, all rings, "
                  f"every {stride}th m (measured fourier+legendre {f0:.2f}+{l0:.2f}s / {f2:.2f}+{l2:.2f}s; full-transform "
                  f"estimate {s0:.1f}s / {s2:.1f}s), 6 component spectra {t3 - t2:.2f}s; scaled to {nmaps} maps / {npairs} pairs",
    }


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    # HX_BENCH_SHARE_GPU=1 (rehearsal on a one-GPU box only): every rank uses device 0 and the
    # collectives run over gloo on host copies, so the N > 1 code path can be exercised without N GPUs
    share = os.environ.get("HX_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        # the communicator is set up by the first collective: do one now, so that a run with --warmup 0 does not time it
        tok = torch.ones(1, dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(tok)
        assert float(tok.item()) == world

    import heracles_amd as hx
    from heracles_amd import distributed as hxd

    hx.init(local)
    nside, lmax, nbins = args.nside, args.lmax, args.nbins
    npix, nlm = 12 * nside * nside, (lmax + 1) * (lmax + 2) // 2
    plan = hx.Plan(nside, lmax)

    # ---- the job: maps ordered (spin-0 bins, spin-2 bins) per brought-in set ---------------------------------------
    nsets = world if args.scaling == "weak" else 1
    if args.workload == "euclid":
        nbins = 13
        per_set = [0] * (2 * nbins) + [2] * nbins  # 13 bins x (2 spin-0 fields + 1 spin-2 field): BASELINE configs[4]
    else:
        per_set = [0] * nbins + [2] * nbins
    spins = per_set * nsets
    nmaps_total = len(spins)
    work = hxd.ShardedTwoPoint(spins, world, rank, nlm, lmax)
    mine = work.local_maps
    n0 = sum(1 for g in mine if spins[g] == 0)
    n2 = len(mine) - n0
    # synthetic inputs, resident in HBM (seeded by rank; which seed a map has does not matter to the metric)
    gen = torch.Generator(device=dev)
    gen.manual_seed(50 + rank)
    maps0 = torch.randn((n0, npix), dtype=torch.float64, device=dev, generator=gen)
    maps2 = torch.randn((n2, 2, npix), dtype=torch.float64, device=dev, generator=gen)
    alm0, alm2 = work.local_alm_views(dev)
    # Pixel weights in the timed path: the reference always transforms with use_pixel_weights=True (heracles/healpy.py:186).
    # healpy's weight files are not available offline; a synthetic file of the same FORMAT stands in: random values (1e-3) in
    # healpy's compressed layout, expanded by the library exactly as a real file would be (heracles_amd.weights) -- so the full-sky
    # array has the symmetry real weights have (it repeats over the four quadrants of a ring and from north to south), which the
    # ring kernels detect per call and use (one weight per pixel pair instead of eight; an array without it takes the generic
    # path: --generic-weights).
    from heracles_amd import weights as hxw

    if args.generic_weights:
        pw = 1.0 + 1e-3 * torch.cos(torch.arange(npix, dtype=torch.float64, device=dev) * (2.0 * np.pi / 1024.0))
    else:
        pw = hxw.expand_pixel_weights(nside, 1e-3 * np.random.default_rng(7).standard_normal(hxw.compressed_size(nside)), device=dev)

    def step():
        # N > 1: the exchange goes out in two parts -- the spin-2 shards as soon as their transform is done, the spin-0 transform runs
        # under that transfer, then the spin-0 shards; all_pairs_cl starts the spin-2 x spin-2 blocks of its tiles when the first part has
        # landed (ShardedTwoPoint.exchange_begin; nothing is communicated at N = 1)
        if n2:
            plan.map2alm(maps2.view(2 * n2, npix), 2, pix_weights=pw, out=alm2.view(2 * n2, nlm))
        work.exchange_begin(2)
        if n0:
            plan.map2alm(maps0, 0, pix_weights=pw, out=alm0)
        work.exchange_begin(0)
        return work.all_pairs_cl()  # rank 0: every Cl block on the host

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    hx._lib.profile_enable(True)
    hx._lib.profile_reset()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cls = step()
    sync()
    dt = time.perf_counter() - t0
    hx._lib.profile_enable(False)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    npairs = len(work.pairs)
    value = npairs * args.steps / dt

    # ---- per-family kernel times (HIP events on the library stream) ----------------------------------------------
    prof = {}
    for k in ("ring_fft", "fourier_combine", "legendre_analysis", "legendre_analysis_s0", "legendre_analysis_s2",
              "alm_reduce", "alm2cl"):
        n_, ms_ = hx._lib.profile_get(k)
        prof[k] = {"launches": n_, "ms_per_step": ms_ / max(args.steps, 1)}

    # ---- roofline of the dominant kernel family (Legendre / Wigner-d analysis, FP64 MFMA + the vector recursion) ----
    F0 = 8.0 * 2 * nside * nlm  # ALGORITHMIC flops of one spin-0 component; a spin-2 field is 3 F0

    def pmc_traffic(prefix):
        """HBM bytes per launch from the committed PMC passes of this round's kernels (rocprofv3 cannot run inside
        this process); None when the file is missing or belongs to another workload."""
        if (nside, lmax, nbins, world, args.workload) != (4096, 6144, 10, 1, "north_star"):
            return None, None
        path = os.path.join("profiles", f"{PROFILE_ROUND}_traffic.json")
        try:
            with open(os.path.join(ROOT, path)) as f:
                tk = json.load(f)["kernels"]
        except (OSError, ValueError, KeyError):
            return None, None
        # the variant the timed (device-resident) step runs is the one with the most bytes per launch: the host -> host leg of
        # the profiled command goes through two-sweep variants of the same family (maps uploaded sweep by sweep)
        sel = [v for k, v in tk.items() if k.startswith(prefix)]
        return (max(v["hbm_bytes_per_launch"] for v in sel) if sel else None), path

    # executed work of ONE step per spin, counted by the kernels themselves (hx_executed_flops: matrix instructions actually
    # issued -- dead stages skip theirs -- and vector-unit flops); one extra call per spin, outside the timed region
    def counted(fn):
        hx._lib.executed_flops(reset=True)
        fn()
        return hx._lib.executed_flops(reset=True)

    exec2 = counted(lambda: plan.map2alm(maps2.view(2 * n2, npix), 2, pix_weights=pw, out=alm2.view(2 * n2, nlm))) if n2 else (0.0, 0.0)
    exec0 = counted(lambda: plan.map2alm(maps0, 0, pix_weights=pw, out=alm0)) if n0 else (0.0, 0.0)

    def roof(name, kernel, alg_flops_step, spin, ncomp):
        nl_, ms_ = hx._lib.profile_get(name)
        mf, vf = exec2 if spin else exec0
        mf_model = plan.mfma_flops(spin, ncomp) if ncomp else 0.0
        sec = ms_ * 1e-3
        exe = (mf + vf) * args.steps / sec / 1e12 if sec > 0 else 0.0
        traffic, tpath = pmc_traffic("hx::k_legendre_duo<%d" % spin)
        return {"kernel": kernel, "bound": "mfma", "achieved": exe, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": exe / FP64_PEAK_TFLOPS,
                "achieved_is": "EXECUTED FP64 flops, counted by the kernel itself (matrix instructions actually issued -- stages whose "
                               "rings are all below 2^-300 skip theirs -- + 4 flops per generated lambda_lm on the vector unit) / kernel time; "
                               "the two share one FP64 pipe (peak 78.6 either way); equals the SQ_INSTS_VALU_MFMA_F64-based figure of profiles/",
                "task_list_mfma_tflops": mf_model * args.steps / sec / 1e12 if sec > 0 else 0.0,
                "executed_mfma_tflops": mf * args.steps / sec / 1e12 if sec > 0 else 0.0,
                "executed_valu_tflops": vf * args.steps / sec / 1e12 if sec > 0 else 0.0,
                "algorithmic_tflops": alg_flops_step * args.steps / sec / 1e12 if sec > 0 else 0.0,
                "algorithmic_is": "SURVEY 8d's F0 = 8 * 2 nside * nlm per spin-0 component (3 F0 per spin-2 field) / kernel time; "
                                  "exceeds the executed rate because north/south symmetry, ring pruning and the shared recursion "
                                  "remove work -- not a pipe utilisation",
                "traffic": traffic, "traffic_unit": "HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE)",
                "traffic_source": f"committed PMC passes ({tpath}), not measured by this run" if tpath else None,
                "launches": nl_, "avg_launch_ms": ms_ / nl_ if nl_ else None}

    nc0, nc2 = n0, 2 * n2
    roofline = roof("legendre_analysis_s2", "hx::k_legendre_duo<2,*>", n2 * 3 * F0, 2, nc2)
    roofline_s0 = roof("legendre_analysis_s0", "hx::k_legendre_duo<0,*>", n0 * F0, 0, nc0)

    # ---- N > 1: the FIXED job (one set of maps) on all N GPUs, sharded by m (MShardedTwoPoint): ring modes of the rank's maps ->
    # all-to-all by m-range -> full-batch Legendre on the rank's range -> partial Cl -> all-reduce.  Reported as value_strong. ----
    strong, verify_multi = None, None
    if world > 1:
        del maps0, maps2  # (the weak-scaling maps are not needed any more at N > 1: room for the fixed job's buffers)
        torch.cuda.empty_cache()
        cdev = "cpu" if share else dev

        def seeded_maps(gs):
            """maps of the fixed job's global indices gs (spin-0 first, as the routes hold them), seeded by the index: the same
            map on whichever rank and route it lands"""
            m0 = [g for g in gs if per_set[g] == 0]
            m2 = [g for g in gs if per_set[g] != 0]
            t0 = torch.empty((len(m0), npix), dtype=torch.float64, device=dev)
            t2 = torch.empty((len(m2), 2, npix), dtype=torch.float64, device=dev)
            for k, g in enumerate(m0):
                t0[k] = torch.randn(npix, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(7000 + g))
            for k, g in enumerate(m2):
                t2[k] = torch.randn((2, npix), dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(7000 + g))
            return t0, t2

        # (a failure of this leg -- it has never run over RCCL, only over gloo and in a one-GPU rehearsal -- must not cost the line
        # its weak-scaling value: the ranks agree on success before every collective (guard=True) and an exception is raised by
        # all of them alike and reported instead of the figure)
        cls_strong = None
        try:
            ms = hxd.MShardedTwoPoint(per_set, world, rank, nlm, lmax, hxd.HipStages(plan, dev))
            s0, s2 = seeded_maps(ms.local_maps)
            for _ in range(max(args.warmup, 1)):
                ms.run(s0, s2, pix_weights=pw, guard=True)
            sync()
            ts = time.perf_counter()
            for _ in range(args.steps):
                cls_strong = ms.run(s0, s2, pix_weights=pw, guard=True)
            sync()
            dts = time.perf_counter() - ts
            tt = torch.tensor([dts], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dts = float(tt.item())
            strong = {"value": len(ms.pairs) * args.steps / dts, "unit": "map->Cl pairs/s", "ms_per_step": dts / args.steps * 1e3,
                      "maps_total": len(per_set), "pairs": len(ms.pairs), "orders_first_count_step": ms.sets,
                      "checksum": float(np.abs(cls_strong).sum()),
                      "what": "the SAME job at every N (one set of maps): ring Fourier stage of the rank's maps, all-to-all of the ring modes by "
                              "owner of the order m (rank q: m = q, q + N, ...) over RCCL, Legendre stage of ALL components on the rank's orders, partial Cl, all-reduce"}
            del s0, s2, ms
            torch.cuda.empty_cache()
        except Exception as exc:  # noqa: BLE001
            strong = {"value": None, "error": f"{type(exc).__name__}: {exc}"[:400]}
            cls_strong = None

        # ---- verification at N > 1 (outside every timed region): the SAME seeded job through the all-gather route, whose rows on
        # rank 0 are compared (i) with the m-sharded route's spectra and (ii) with direct sums over the gathered alms ----
        if not args.no_verify:
            try:
                wk = hxd.ShardedTwoPoint(per_set, world, rank, nlm, lmax)
                v0, v2 = seeded_maps(wk.local_maps)
                a0, a2 = wk.local_alm_views(dev)
                if v2.shape[0]:
                    plan.map2alm(v2.view(-1, npix), 2, pix_weights=pw, out=a2.view(-1, nlm))
                wk.exchange_begin(2)
                if v0.shape[0]:
                    plan.map2alm(v0, 0, pix_weights=pw, out=a0)
                wk.exchange_begin(0)
                rows = wk.all_pairs_cl()
                if rank == 0:
                    verify_multi = {"routes": None, "cl_vs_direct_sum": None}
                    if cls_strong is not None:
                        e = float(np.abs(cls_strong - rows).max() / np.abs(rows).max())
                        verify_multi["routes"] = {"m_sharded_vs_all_gather_max_err_over_max": e, "spectra": int(rows.shape[0]), "tolerance": 1e-10}
                    wv = torch.full((nlm,), 2.0, dtype=torch.float64, device=dev)
                    wv[: lmax + 1] = 1.0
                    idx_l = torch.cat([torch.arange(m, lmax + 1, device=dev) for m in range(lmax + 1)])
                    bufv, ecl, nchk = wk.buffer(), 0.0, 0
                    nm = len(per_set)
                    for (i, j) in [(0, 0), (0, nm - 1), (nm - 1, nm - 1), (nm // 2, nm // 2 + 1)]:
                        for ka, ca in enumerate(wk.comps_of_map[i]):
                            for kb, cb in enumerate(wk.comps_of_map[j]):
                                a, b_ = bufv[ca], bufv[cb]
                                ref = torch.zeros(lmax + 1, dtype=torch.float64, device=dev).index_add_(0, idx_l, wv * (a.real * b_.real + a.imag * b_.imag))
                                ref = (ref / (2.0 * torch.arange(lmax + 1, device=dev) + 1.0)).cpu().numpy()
                                got = rows[wk.row0[i, j] + ka * len(wk.comps_of_map[j]) + kb]
                                ecl = max(ecl, float(np.abs(got - ref).max() / np.abs(ref).max()))
                                nchk += 1
                    verify_multi["cl_vs_direct_sum"] = {"max_err_over_max": ecl, "spectra_checked": nchk, "tolerance": 1e-11}
                    verify_multi["ok"] = bool(ecl <= 1e-11 and (verify_multi["routes"] is None or verify_multi["routes"]["m_sharded_vs_all_gather_max_err_over_max"] <= 1e-10))
                    verify_multi["what"] = ("the fixed job (maps seeded by their global index) through both routes: spectra of the m-sharded route (all-to-all "
                                            "+ all-reduce) against the rows of the all-gather route on rank 0, and map pairs of the latter against direct sums "
                                            "over the gathered alms")
                    del wv, idx_l
                del v0, v2, wk
            except Exception as exc:  # noqa: BLE001
                verify_multi = {"ok": False, "error": f"{type(exc).__name__}: {exc}"[:400]}
            torch.cuda.empty_cache()

    out = None
    if rank == 0:
        assert cls is not None and cls.shape[0] == work.nrows
        # ---- host -> host leg (SURVEY 8d's metric definition: pageable numpy maps in, Cl blocks on the host out) -------
        # (before the CPU legs, so that their working set on the host is not in its way)
        host_leg = None
        if world == 1 and not args.no_host_leg:
            h0 = maps0.cpu().numpy()
            h2 = maps2.cpu().numpy().reshape(2 * n2, npix)

            def host_step():
                # ONE call for all transforms: the uploads of the next sweep overlap the transform of the current one across the
                # two jobs (hx_map2alm_multi); large job first, the last sweep is a small one.  alms stay in HBM.
                plan.map2alm_multi([(h2, 2, alm2.view(2 * n2, nlm)), (h0, 0, alm0)], pix_weights=pw)
                return work.all_pairs_cl()          # numpy array on the host

            host_step()
            nh = max(args.host_steps, 1)
            times = []
            for _ in range(nh):
                th = time.perf_counter()
                host_step()
                times.append(time.perf_counter() - th)
            dth = float(np.median(times))
            host_leg = {"value": npairs / dth, "unit": "map->Cl pairs/s", "ms_per_step": dth * 1e3, "steps": nh,
                        "ms_per_step_all": [t * 1e3 for t in times], "statistic": "median",
                        "pcie_floor_ms": (n0 + 2 * n2) * npix * 8 / 55e9 * 1e3,
                        "what": "pageable numpy maps on the host -> ONE hx_map2alm_multi call (H2D through the library's pinned "
                                "staging, uploads overlapped with the transforms across jobs, pixel weights applied) -> all-pairs Cl -> "
                                "numpy Cl blocks on the host; alms never leave HBM.  pcie_floor_ms = the maps' bytes at the 55 GB/s the "
                                "staging sustains"}
            del h0, h2

        # ---- verification of what was timed (outside the timed region) --------------------------------------------
        verify, cpu = None, None
        osample = None
        if world == 1 and not (args.no_verify and args.no_cpu_baseline):
            stride = (8 if nside >= 2048 else 1) if not args.no_cpu_baseline else (512 if nside >= 2048 else 4)
            t_host = maps0[:1].cpu().numpy()
            qu_host = maps2[0].cpu().numpy()
            oa0, oa2, tim = oracle_sample(nside, lmax, t_host, qu_host, stride, pix_weights=pw.cpu().numpy())
            osample = (oa0, oa2, tim, stride)
            del t_host, qu_host
        if not args.no_verify:
            verify = {}
            if osample is not None:
                oa0, oa2, tim, stride = osample
                g0 = alm0[0].cpu().numpy()
                g2 = alm2[0].cpu().numpy()
                scale0, scale2 = np.abs(g0).max(), np.abs(g2).max()
                e0 = e2 = 0.0
                ms_checked = list(range(0, lmax + 1, stride))
                for m in ms_checked:
                    b = m * (2 * lmax + 1 - m) // 2
                    sl = slice(b + m, b + lmax + 1)
                    e0 = max(e0, float(np.abs(g0[sl] - oa0[0, sl]).max()))
                    e2 = max(e2, float(np.abs(g2[:, sl] - oa2[:, sl]).max()))
                verify.update(alm_vs_oracle={"spin0_max_err_over_max": e0 / scale0, "spin2_max_err_over_max": e2 / scale2,
                                             "m_checked": len(ms_checked), "m_stride": stride, "tolerance": 1e-10})
            # Cl rows against a direct sum over the device alms (independent of the all-pairs kernel)
            w = torch.full((nlm,), 2.0, dtype=torch.float64, device=dev)
            w[: lmax + 1] = 1.0
            idx_l = torch.cat([torch.arange(m, lmax + 1, device=dev) for m in range(lmax + 1)])
            buf = work.buffer()
            ecl = 0.0
            nchk = 0
            checks = [(0, 0), (0, min(1, nmaps_total - 1)), (nmaps_total - 1, nmaps_total - 1)]
            if nmaps_total > nbins:
                checks.append((0, nbins))
            for (i, j) in checks:
                row = work.row0[min(i, j), max(i, j)]
                for ka, ca in enumerate(work.comps_of_map[min(i, j)]):
                    for kb, cb in enumerate(work.comps_of_map[max(i, j)]):
                        a, b_ = buf[ca], buf[cb]
                        ref = torch.zeros(lmax + 1, dtype=torch.float64, device=dev).index_add_(0, idx_l, w * (a.real * b_.real + a.imag * b_.imag))
                        ref = (ref / (2.0 * torch.arange(lmax + 1, device=dev) + 1.0)).cpu().numpy()
                        got = cls[row + ka * len(work.comps_of_map[max(i, j)]) + kb]
                        ecl = max(ecl, float(np.abs(got - ref).max() / np.abs(ref).max()))
                        nchk += 1
            verify.update(cl_vs_direct_sum={"max_err_over_max": ecl, "spectra_checked": nchk, "tolerance": 1e-11})
            ok = ecl <= 1e-11
            if "alm_vs_oracle" in verify:
                ok = ok and verify["alm_vs_oracle"]["spin0_max_err_over_max"] <= 1e-10 and verify["alm_vs_oracle"]["spin2_max_err_over_max"] <= 1e-10
            verify["ok"] = bool(ok)
            del w, idx_l
        if not args.no_cpu_baseline and world == 1 and osample is not None:  # reported on rank 0 at N = 1 only
            oa0, oa2, tim, stride = osample
            cpu = cpu_baseline(nside, lmax, per_set.count(0), per_set.count(2), oa0, oa2, tim, stride)

        # ---- the reference's own call shape: ONE map / ONE field per transform (heracles/mapping.py:171), resident inputs; with
        # niter = 0 (weights supplied) and with healpy's default three Jacobi iterations; one alm2map.  Informational: not `value`.
        single = None
        if world == 1 and not args.no_single:
            single = {}
            for spin, src in ((0, maps0[:1]), (2, maps2[0])):
                a1 = torch.empty((src.shape[0], nlm), dtype=torch.complex128, device=dev)
                back = torch.empty_like(src)
                for name, fn in (("map2alm_niter0_ms", lambda: plan.map2alm(src, spin, out=a1, pix_weights=pw)),
                                 ("map2alm_niter3_ms", lambda: plan.map2alm(src, spin, out=a1, niter=3)),
                                 ("alm2map_ms", lambda: plan.alm2map(a1, spin, out=back))):
                    fn()
                    torch.cuda.synchronize(dev)
                    t1 = time.perf_counter()
                    fn()
                    torch.cuda.synchronize(dev)
                    single[f"spin{spin}_{name}"] = (time.perf_counter() - t1) * 1e3
                del a1, back
            single["what"] = ("one spin-0 map / one spin-2 (Q, U) field per call, device-resident: the vector-unit kernels "
                              "(k_legendre_valu, k_legendre_synth_valu)")

        mix = None
        if not args.no_mixmat:
            L = args.mixmat_lmax or lmax
            ell = np.arange(L + 1)
            wl = 4 * np.pi * 0.35 * np.exp(-ell * (ell + 1) / 3000.0) + 1e-3 / (1.0 + ell) ** 2
            hx.mixmat_eb(wl[:65], l1max=64, l2max=64)  # warm-up (module load)
            # median of five builds, host -> host each (a 0.9 GB numpy array per call, as the reference returns one: the first touch of fresh pages
            # is part of the figure and varies with the state of the host's page cache; all three are listed)
            mix_all = []
            for _ in range(5):
                hx._lib.profile_enable(True)
                hx._lib.profile_reset()
                tm = time.perf_counter()
                mm = hx.mixmat_eb(wl)
                mix_all.append(time.perf_counter() - tm)
                hx._lib.profile_enable(False)
            mix_s = float(np.median(mix_all))
            ng, gms = hx._lib.profile_get("mixmat_gemm")
            _, gkms = hx._lib.profile_get("mixmat_gemm_kernel")
            gkms = gkms or gms
            N = (3 * L) // 2 + 1
            gflop = 2.0 * (L + 1) ** 2 * N * 2  # two products
            nb_, kpad_ = (L + 1 + 127) // 128, (N + 31) // 32 * 32
            xflop = 2.0 * 128 * 128 * kpad_ * (nb_ * (nb_ + 1) // 2) * 2  # what k_mixmat_gemm executes: the upper triangle of 128 x 128 tiles, padded nodes
            mix = {"L": L, "seconds": mix_s, "seconds_all": mix_all, "statistic": "median of 5", "gemm_ms": gms, "gemm_kernel_ms": gkms,
                   "seconds_is": "hx.mixmat_eb host -> host, five builds in a row, each result dropped when the next one has arrived: the first "
                                 "two fill fresh pages (first-touch page faults of a 0.9 GB numpy array), the later ones a block of the "
                                 "library's host result pool that an earlier result has released (heracles_amd/_lib.py: _HostPool; HX_HOST_POOL_MB=0 turns it off)",
                   "gemm_ms_is": "gemm_kernel_ms: the two launches of k_mixmat_gemm_dma; gemm_ms: the same with the two passes that form T diag(s) "
                                 "for them (k_scale_table: the k tiles go from HBM into LDS without passing through registers)",
                   "gemm_tflops_executed": xflop / (gkms * 1e-3) / 1e12 if gkms else None,
                   "gemm_frac_of_fp64_mfma_peak": xflop / (gkms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if gkms else None,
                   "gemm_tflops_executed_incl_scaling_pass": xflop / (gms * 1e-3) / 1e12 if gms else None,
                   "gemm_tflops_algorithmic": gflop / (gms * 1e-3) / 1e12 if gms else None,
                   "gemm_tflops_algorithmic_is": "SURVEY 8d's 2 (l1max + 1)(l2max + 1) N per product / kernel time: the symmetric product is computed for the "
                                                 "upper triangle of tiles only, so this exceeds what the matrix pipe executes (gemm_tflops_executed) -- not a utilisation",
                   "checksum": float(np.abs(mm[2] - (mm[0] - mm[1])).max())}
        peaks = hx._lib.measure_peaks()
        out = {
            "metric": "map->Cl pairs/sec + mixing-matrix build sec, nside=%d lmax=%d" % (nside, lmax),
            "value": value, "unit": "map->Cl pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{nbins} bins x ({'2 spin-0 + 1 spin-2' if args.workload == 'euclid' else 'spin-0, spin-2'}) maps "
                                   f"{'per GPU' if args.scaling == 'weak' else 'in all'} = "
                                   f"{nmaps_total} maps / {sum(2 if s else 1 for s in spins)} components over {world} GPU(s), nside={nside}, lmax={lmax}, "
                                   f"niter=0, ring weights 1, pix_weights: synthetic file in healpy's format (its files are not available offline); "
                                   f"{npairs} auto+cross map pairs; inputs resident in HBM",
                       "nside": nside, "lmax": lmax, "maps_total": nmaps_total, "maps_this_rank": len(mine), "pairs": npairs,
                       "pix_weights": ("synthetic array without symmetry (generic path)" if args.generic_weights else "synthetic weights in healpy's compressed format, expanded to the full sky (symmetric like healpy's)") + ", applied in the timed path",
                       "parallelism": (f"maps dealt to {world} GPUs by cost, RCCL all-gather of alms, tiled pair split"
                                       if world > 1 else "1 GPU")},
            "verified": (None if verify is None and verify_multi is None
                         else bool((verify is None or verify.get("ok")) and (world == 1 or (verify_multi or {}).get("ok")))),
            "verify": verify,
            "verify_multi": verify_multi,
            "value_strong": strong.get("value") if strong else None,
            "strong_scaling": strong,
            "value_host_to_host": host_leg["value"] if host_leg else None,
            "host_to_host": host_leg,
            "single_map_transforms": single,
            "mixmat_build_sec": mix["seconds"] if mix else None,
            "mixmat": mix,
            "roofline": roofline,
            "roofline_spin0_kernel": roofline_s0,
            "cpu_baseline": cpu,
            "measured_peaks": peaks,
            "kernels": prof,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    plan.close()


if __name__ == "__main__":
    main()
