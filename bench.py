#!/usr/bin/env python3
"""bench.py -- map -> Cl pairs/s (+ mixing-matrix build seconds) on MI355X.

Metric (BASELINE.json): "map->Cl pairs/sec + mixing-matrix build sec, nside=4096 lmax=6144".
Workload per GPU (the north_star target that fits one GPU): 10 spin-0 + 10 spin-2 maps
(30 components) at nside=4096, lmax=6144, synthetic Gaussian pixels, resident in HBM when
the timed region starts.  One "step" = batched map2alm of all maps (niter=0, unit ring
weights) + all auto/cross Cl of every map pair + D2H of the Cl blocks.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: every rank transforms its own 20 maps, alms are all-gathered over RCCL/xGMI, the
pair list over all 20*N maps is partitioned over ranks ("scaling": "weak": per-GPU SHT
work is fixed).  Rank 0 prints ONE JSON line.
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

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X datasheet FP64 matrix (= vector) peak; measured MFMA loop: 47.8
HBM_PEAK_GBS = 8000.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--nside", type=int, default=4096)
    p.add_argument("--lmax", type=int, default=6144)
    p.add_argument("--nbins", type=int, default=10, help="tomographic bins per GPU: nbins x (spin-0, spin-2) maps")
    p.add_argument("--mixmat-lmax", type=int, default=None, help="L of the timed mixmat_eb (default: lmax)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-mixmat", action="store_true")
    return p.parse_args()


def pair_list(nmaps_total):
    """(i, j) map pairs, i <= j, in the order of combinations_with_replacement."""
    return [(i, j) for i in range(nmaps_total) for j in range(i, nmaps_total)]


def comp_pairs_of(map_pairs, comps_of_map):
    out, owner = [], []
    for n, (i, j) in enumerate(map_pairs):
        for a in comps_of_map[i]:
            for b in comps_of_map[j]:
                out.append((a, b))
                owner.append(n)
    return out, owner


def cpu_baseline(nside, lmax, nbins):
    """Oracle (CPU restatement, kind "port") on a bounded sample of the same workload:
    one spin-0 map and one spin-2 map at full size, their 3 map pairs; scaled to the
    nbins x (spin-0, spin-2) job.  Never used by the GPU path."""
    from oracle import hxoracle as ho

    rng = np.random.default_rng(50)
    npix = 12 * nside * nside
    t = rng.standard_normal((1, npix))
    qu = rng.standard_normal((2, npix))
    # bounded sample: full ring-FFT stage, every `stride`-th m of the Legendre stage
    stride = 8 if nside >= 2048 else 1
    ho.set_mstride(stride)
    a0 = ho.map2alm(t, nside, lmax, spin=0)
    f0, l0 = ho.last_timings()
    a2 = ho.map2alm(qu, nside, lmax, spin=2)
    f2, l2 = ho.last_timings()
    ho.set_mstride(1)
    t2 = time.perf_counter()
    ho.alm2cl(a0, a0), ho.alm2cl(a0, a2), ho.alm2cl(a2, a2)
    t3 = time.perf_counter()
    nmaps = 2 * nbins
    npairs = nmaps * (nmaps + 1) // 2
    # alm2cl: 6 component spectra took (t3-t2); the job has nbins(nbins+1)/2*(1+4) + nbins^2*2
    ncs = nbins * (nbins + 1) // 2 * 5 + nbins * nbins * 2
    s0, s2 = f0 + l0 * stride, f2 + l2 * stride  # full-transform estimates
    total = nbins * s0 + nbins * s2 + (t3 - t2) * ncs / 6.0
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next(line.split(":", 1)[1].strip() for line in f if line.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return {
        "value": npairs / total,
        "unit": "map->Cl pairs/s",
        "cores": ho.num_threads(),
        "cpu_model": cpu_model,
        "host_logical_cpus": os.cpu_count(),
        "kind": "port",
        "sample": f"oracle map2alm of 1 spin-0 + 1 spin-2 map at nside={nside} lmax={lmax}, all rings, every "
                  f"{stride}th m (measured fourier+legendre {f0:.2f}+{l0:.2f}s / {f2:.2f}+{l2:.2f}s; full-transform "
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

    import heracles_amd as hx
    from heracles_amd import distributed as hxd

    hx.init(local)
    nside, lmax, nbins = args.nside, args.lmax, args.nbins
    npix, nlm = 12 * nside * nside, (lmax + 1) * (lmax + 2) // 2
    plan = hx.Plan(nside, lmax)

    # ---- synthetic inputs, resident in HBM ------------------------------------------
    gen = torch.Generator(device=dev)
    gen.manual_seed(50 + rank)
    maps0 = torch.randn((nbins, npix), dtype=torch.float64, device=dev, generator=gen)
    maps2 = torch.randn((nbins, 2, npix), dtype=torch.float64, device=dev, generator=gen)
    alm0 = torch.empty((nbins, nlm), dtype=torch.complex128, device=dev)
    alm2 = torch.empty((nbins, 2, nlm), dtype=torch.complex128, device=dev)

    # maps are ordered (spin-0 bin 0..nbins-1, spin-2 bin 0..nbins-1) per rank
    nmaps_local = 2 * nbins
    nmaps_total = nmaps_local * world
    map_pairs = pair_list(nmaps_total)
    work = hxd.PairWork(world, rank, nbins, nlm, lmax)

    def step():
        plan.map2alm(maps0, 0, out=alm0)
        plan.map2alm(maps2.view(2 * nbins, npix), 2, out=alm2.view(2 * nbins, nlm))
        return work.all_pairs_cl(alm0, alm2)  # rank 0: every Cl block on the host

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
    npairs = len(map_pairs)
    value = npairs * args.steps / dt

    # ---- roofline of the dominant kernel (Legendre analysis, FP64 MFMA) ---------------
    F0 = 8.0 * 2 * nside * nlm  # algorithmic flops of one spin-0 component
    flops_step = nbins * F0 + nbins * 3 * F0
    prof = {}
    for k in ("ring_fft", "fourier_combine", "legendre_analysis", "legendre_analysis_s0", "legendre_analysis_s2",
              "alm_reduce", "alm2cl"):
        n_, ms_ = hx._lib.profile_get(k)
        prof[k] = {"launches": n_, "ms_per_step": ms_ / max(args.steps, 1)}

    # HBM traffic per launch from the committed PMC passes (rocprofv3 cannot run inside this process):
    # launch-weighted mean over the instantiations of the kernel family, or None without the file
    def pmc_traffic(prefix):
        if (nside, lmax, nbins) != (4096, 6144, 10):
            return None  # the committed counters belong to the default workload
        try:
            with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
                tk = json.load(f)["kernels"]
        except (OSError, ValueError, KeyError):
            return None
        sel = [v for k, v in tk.items() if k.startswith(prefix)]
        n = sum(v["launches"] for v in sel)
        return sum(v["hbm_bytes_per_launch"] * v["launches"] for v in sel) / n if n else None

    def roof(name, kernel, flops_per_step, executed_per_step):
        nl_, ms_ = hx._lib.profile_get(name)
        ach = flops_per_step * args.steps / (ms_ * 1e-3) / 1e12 if ms_ > 0 else 0.0
        exe = executed_per_step * args.steps / (ms_ * 1e-3) / 1e12 if ms_ > 0 else 0.0
        return {"kernel": kernel, "bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / FP64_MFMA_PEAK_TFLOPS,
                "traffic": pmc_traffic(kernel.split("|")[0].rstrip(">") if "|" not in kernel else "hx::k_legendre_analysis<"),
                "traffic_unit": "HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, profiles/r01_traffic.json, full-size PMC passes)",
                "launches": nl_,
                "avg_launch_ms": ms_ / nl_ if nl_ else None,
                "algorithmic_flops_per_launch": flops_per_step * args.steps / nl_ if nl_ else None,
                # what the matrix pipe actually ran: north/south symmetry halves the algorithmic
                # work, column padding and the second (spin-2) function add to it
                "executed_mfma_tflops": exe, "executed_mfma_frac_of_peak": exe / FP64_MFMA_PEAK_TFLOPS}

    # dominant kernel: the spin-2 Legendre/Wigner-d analysis (3 F0 per (Q,U) field)
    ex0, ex2 = plan.mfma_flops(0, nbins), plan.mfma_flops(2, 2 * nbins)
    roofline = roof("legendre_analysis_s2", "hx::k_legendre_analysis<2>", nbins * 3 * F0, ex2)
    roofline_s0 = roof("legendre_analysis_s0", "hx::k_legendre_analysis<0>", nbins * F0, ex0)
    roofline_all = roof("legendre_analysis", "hx::k_legendre_analysis<0|2>", flops_step, ex0 + ex2)

    out = None
    if rank == 0:
        assert cls is not None and len(cls) > 0
        mix = None
        if not args.no_mixmat:
            L = args.mixmat_lmax or lmax
            ell = np.arange(L + 1)
            wl = 4 * np.pi * 0.35 * np.exp(-ell * (ell + 1) / 3000.0) + 1e-3 / (1.0 + ell) ** 2
            hx.mixmat_eb(wl[:65], l1max=64, l2max=64)  # warm-up (module load)
            hx._lib.profile_enable(True)
            hx._lib.profile_reset()
            tm = time.perf_counter()
            mm = hx.mixmat_eb(wl)
            mix_s = time.perf_counter() - tm
            hx._lib.profile_enable(False)
            ng, gms = hx._lib.profile_get("mixmat_gemm")
            N = (3 * L) // 2 + 1
            gflop = 2.0 * (L + 1) ** 2 * N * 2  # two products
            mix = {"L": L, "seconds": mix_s, "gemm_ms": gms, "gemm_tflops_algorithmic": gflop / (gms * 1e-3) / 1e12 if gms else None,
                   "checksum": float(np.abs(mm[2] - (mm[0] - mm[1])).max())}
        # what this box sustains (micro-kernels, after the timed region): the roofline against the
        # datasheet peak is `frac`; the same against the measured MFMA rate is reported beside it
        peaks = hx._lib.measure_peaks()
        for rf in (roofline, roofline_s0, roofline_all):
            rf["frac_of_measured_mfma_rate"] = rf["achieved"] / peaks["fp64_mfma_tflops"]
            rf["executed_mfma_frac_of_measured_rate"] = rf["executed_mfma_tflops"] / peaks["fp64_mfma_tflops"]
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # reported on rank 0 at N = 1 only
            cpu = cpu_baseline(nside, lmax, nbins)
        out = {
            "metric": "map->Cl pairs/sec + mixing-matrix build sec, nside=%d lmax=%d" % (nside, lmax),
            "value": value, "unit": "map->Cl pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{nbins} bins x (spin-0, spin-2) maps per GPU = {nmaps_local} maps / "
                                   f"{3 * nbins} components per GPU, nside={nside}, lmax={lmax}, niter=0; "
                                   f"{npairs} auto+cross map pairs over {nmaps_total} maps",
                       "nside": nside, "lmax": lmax, "maps_per_gpu": nmaps_local, "pairs": npairs,
                       "parallelism": f"maps sharded over {world} GPU(s), RCCL all-gather of alms" if world > 1 else "1 GPU"},
            "mixmat_build_sec": mix["seconds"] if mix else None,
            "mixmat": mix,
            "roofline": roofline,
            "roofline_spin0_kernel": roofline_s0,
            "roofline_both_kernels": roofline_all,
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
