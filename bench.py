#!/usr/bin/env python3
"""bench.py -- map -> Cl pairs/s (+ mixing-matrix build seconds) on MI355X.

Metric (BASELINE.json): "map->Cl pairs/sec + mixing-matrix build sec, nside=4096 lmax=6144".
Workload at N = 1 (the north_star target, which fits one GPU): 10 spin-0 + 10 spin-2 maps
(30 components) at nside=4096, lmax=6144, synthetic Gaussian pixels, resident in HBM when
the timed region starts.  One "step" = batched map2alm of all maps (niter=0, unit ring
weights, a synthetic full-sky pixel-weight array) + all auto/cross Cl of every map pair + D2H of the Cl blocks.
`value` is that device-resident rate (the task contract: inputs resident in HBM when the timed region starts);
`value_host_to_host` is SURVEY 8d's definition (pageable host maps in, Cl blocks on the host out; median of --host-steps).

  python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: starts its own N ranks, see self_launch)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`value` at EVERY N is pairs/s of the SAME 20-map job (210 map pairs): value(N) / value(1) is a speed-up <= N ("scaling": "strong").
At N > 1 that job runs through both fixed-job routes of heracles_amd.distributed, each timed for exactly K steps --
  m-sharded:  ring Fourier stage of the rank's maps -> all-to-all of ring modes by owner of the order m -> Legendre stage of ALL
              components on the rank's orders -> partial Cl -> all-reduce (MShardedTwoPoint);
  all-gather: the 20 maps dealt to the ranks by cost -> in-place all-gather of the alms -> tiled pair split (ShardedTwoPoint)
-- `value` is the better of the two, named in config.parallelism; both are in "routes".  The job that GROWS with N (every rank
brings its own 20 maps, pairs grow as N^2) is reported beside it as `value_weak` with its pair count and a per-GPU transform rate;
`--scaling weak` makes that figure `value` instead.  Rank 0 prints ONE JSON line.
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
PROFILE_ROUND = "r06"    # profiles/<round>_traffic.json holds the committed PMC passes of THIS round's kernels


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--nside", type=int, default=4096)
    p.add_argument("--lmax", type=int, default=6144)
    p.add_argument("--nbins", type=int, default=10, help="tomographic bins: nbins x (spin-0, spin-2) maps per GPU (weak) / in all (strong)")
    p.add_argument("--scaling", choices=("weak", "strong"), default="strong",
                   help="which figure is `value` at N > 1: strong = the SAME job at every N (default), weak = a job that grows with N")
    p.add_argument("--weak-steps", type=int, default=None, help="timed steps of the weak-scaling leg at N > 1 when it is not `value` (default min(steps, 5))")
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
    p.add_argument("--no-niter3", action="store_true", help="skip the niter = 3 leg (the 20-map job with healpy's default iterations)")
    p.add_argument("--launch-check", action="store_true", help="every rank checks its rendezvous variables and exits (rank 0 prints them); no GPU is touched")
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


def host_fp64_peak(threads):
    """(GFLOP/s, GHz, how) -- threads x clock x 32 flop per cycle (two 512-bit FMA pipes), clock = the highest the kernel reports"""
    ghz, how = None, None
    try:
        with open("/sys/devices/system/cpu/cpu0/cpufreq/cpuinfo_max_freq") as f:
            ghz, how = int(f.read()) / 1e6, "cpuinfo_max_freq"
    except (OSError, ValueError):
        pass
    if ghz is None:
        try:
            with open("/proc/cpuinfo") as f:
                mhz = [float(line.split(":")[1]) for line in f if line.startswith("cpu MHz")]
            ghz, how = max(mhz) / 1e3, "max of /proc/cpuinfo cpu MHz at the time of the run"
        except (OSError, ValueError):
            return None, None, None
    return threads * ghz * 32.0, ghz, how


def cpu_baseline_vectorised(nside, lmax, nmaps0, nmaps2, t_map, qu_map, pix_weights):
    """cpu_baseline, kind "port-vectorised": oracle/hx_cpu_fast.c (AVX-512, OpenMP over m, scaled recursions with ring pruning and
    north/south symmetry; checked by the scalar oracle in tests/test_oracle_fast.py) on ALL host threads, in FULL (every m) on one
    spin-0 map and one spin-2 field of the bench's own input, scaled to the job.  Returns (record, alm0, alm2) -- the alms are a
    second, all-m check of the GPU's."""
    from oracle import hxfast as hf

    if not hf.supported():
        return None, None, None
    # threads = the CPUs this process may really use: a one-GPU lease sees all 256 logical CPUs of the host but is throttled to its
    # share by the cgroup (cpu.max: 16 here); with 128 threads on a 16-CPU quota the ring stage took 13 s instead of 0.3
    # (profiles/r06_cpu_baseline_threads.txt)
    threads, quota = hf.cpu_quota()
    hf.set_threads(threads)
    a0, tim0 = hf.map2alm(t_map, nside, lmax, spin=0, pix_weights=pix_weights)
    a2, tim2 = hf.map2alm(qu_map, nside, lmax, spin=2, pix_weights=pix_weights)
    t2 = time.perf_counter()
    for x_, y_ in ((a0[0], a0[0]), (a0[0], a2[0]), (a0[0], a2[1]), (a2[0], a2[0]), (a2[0], a2[1]), (a2[1], a2[1])):
        hf.alm2cl(x_, y_)
    tcl = time.perf_counter() - t2
    nmaps = nmaps0 + nmaps2
    npairs = nmaps * (nmaps + 1) // 2
    ncs = nmaps0 * (nmaps0 + 1) // 2 + 4 * (nmaps2 * (nmaps2 + 1) // 2) + 2 * nmaps0 * nmaps2
    s0, s2 = sum(tim0), sum(tim2)
    total = nmaps0 * s0 + nmaps2 * s2 + tcl * ncs / 6.0
    nlm = (lmax + 1) * (lmax + 2) // 2
    F0 = 8.0 * 2 * nside * nlm
    peak, ghz, how = host_fp64_peak(threads)
    gf = 4.0 * F0 / (tim0[1] + tim2[1]) / 1e9
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next(line.split(":", 1)[1].strip() for line in f if line.startswith("model name"))
    except (OSError, StopIteration):
        pass
    rec = {
        "value": npairs / total, "unit": "map->Cl pairs/s", "cores": threads, "cpu_model": cpu_model, "host_logical_cpus": os.cpu_count(),
        "cgroup_cpu_quota": quota, "kind": "port-vectorised", "engines": cpu_engines(),
        "legendre_gflops_per_core": gf / threads,
        "seconds_spin0_transform": s0, "seconds_spin2_transform": s2,
        "legendre_gflops_algorithmic": gf, "legendre_gflops_is": "SURVEY 8d's F0 (spin 0) + 3 F0 (spin 2) / the two Legendre-stage times; ring pruning and "
                                                                "north/south symmetry remove about half of it, as on the GPU",
        "host_fp64_peak_gflops": peak, "host_clock_ghz": ghz, "host_clock_from": how,
        "algorithmic_frac_of_host_fp64_peak": gf / peak if peak else None,
        "algorithmic_frac_is": "the ALGORITHMIC rate / (threads x clock x 32 flop per cycle): not a pipe utilisation -- what the cores execute is about half "
                               "of F0 (as on the GPU, whose algorithmic rate exceeds its peak for the same reason)",
        "note": "this repository's own AVX-512 / OpenMP restatement of the algorithm (oracle/hx_cpu_fast.c), NOT ducc0 / healpy: those engines are "
                "absent from this image (see engines), so the north star's '>= 10x ducc' stays unmeasured; the GPU/CPU ratio says nothing about "
                "kernel quality -- roofline.frac does",
        "sample": f"all m: hx_cpu_fast map2alm of 1 spin-0 map + 1 spin-2 field of the bench's own input at nside={nside} lmax={lmax}, pixel weights applied, "
                  f"{threads} threads (ring stage + Legendre stage {tim0[0]:.2f}+{tim0[1]:.2f}s / {tim2[0]:.2f}+{tim2[1]:.2f}s), 6 component spectra "
                  f"{tcl:.3f}s threaded over m; scaled to {nmaps} maps / {npairs} pairs / {ncs} spectra",
    }
    return rec, a0, a2


def fixed_job(workload, nbins):
    """Spins of the maps of the job `value` is quoted on (the same at every N), in the job's global order."""
    if workload == "euclid":
        return [0] * (2 * 13) + [2] * 13  # 13 bins x (2 spin-0 fields + 1 spin-2 field): BASELINE configs[4]
    return [0] * nbins + [2] * nbins


def self_launch(ngpus):
    """`python3 bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): this process -- which has not imported
    torch, initialised HIP or loaded libhxsht, and never will -- starts the N ranks as child processes of this same script with the
    rendezvous variables torch.distributed reads, relays rank 0's JSON line (the children inherit stdout; only rank 0 prints),
    waits for all of them and exits non-zero if any did.  Nothing is exec'ed and no child is started a second time."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(ngpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(ngpus), LOCAL_WORLD_SIZE=str(ngpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # HSA_ENABLE_IPC_MODE_LEGACY=0: the pool's host driver supports dmabuf IPC only; with the legacy mode RCCL (and any sharing of device
        # memory between processes) fails at hipIpcGetMemHandle with "invalid argument".  The variable is exported by the image here and on
        # the GPU boxes (the environment contract of this build: task statement, "Environment"); setdefault only carries it into ranks that are
        # started from a shell that lost it, and never overrides a value the caller chose.  Evidence: the contract, not a run of ours --
        # no multi-GPU node has run this code.
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # a rank that dies leaves the others blocked in their next collective: once one has failed, the rest get a grace period and
    # are then terminated (exactly the processes started here, by their handles)
    rc, failed_at = 0, None
    live = list(procs)
    while live:
        for pr in list(live):
            code = pr.poll()
            if code is not None:
                live.remove(pr)
                if code != 0 and rc == 0:
                    rc, failed_at = code, time.time()
        if failed_at is not None and live and time.time() - failed_at > 60.0:
            for pr in live:
                pr.terminate()
            time.sleep(5.0)
            for pr in live:
                if pr.poll() is None:
                    pr.kill()
        time.sleep(0.2)
    return rc if rc >= 0 else 1


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))  # (before anything that touches the GPU is imported)
    if args.launch_check:
        r_, w_ = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        if w_ != args.gpus or not (0 <= r_ < w_) or (w_ > 1 and not os.environ.get("MASTER_PORT")):
            raise SystemExit(4)
        if os.environ.get("HX_LAUNCH_CHECK_FAIL") == str(r_):
            raise SystemExit(3)
        if r_ == 0:
            print(json.dumps({"launch_check": True, "n_gpus": w_, "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}"}), flush=True)
        return
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # HX_BENCH_SHARE_GPU=1 (rehearsal on a one-GPU box only): every rank uses device 0 and the
    # collectives run over gloo on host copies, so the N > 1 code path can be exercised without N GPUs
    share_gpu = os.environ.get("HX_BENCH_SHARE_GPU") == "1"
    share = share_gpu
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    a2a_broken = False
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        # the communicator is set up by the first collective: do one now, so that a run with --warmup 0 does not time it
        tok = torch.ones(1, dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(tok)
        assert float(tok.item()) == world
        if not share:
            # ... and the point-to-point channels of the all-to-all / the rings of the all-gather by their first use: one tiny call of each
            # (not a step of the job: no map is touched)
            a = torch.arange(world, dtype=torch.float64, device=dev) + rank * world
            b = torch.empty_like(a)
            dist.all_to_all_single(b, a)
            g = torch.empty(world, dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(g, a[:1].contiguous())
            torch.cuda.synchronize(dev)
            bad = torch.tensor([0.0 if [int(v) for v in b.tolist()] == [q * world + rank for q in range(world)] else 1.0], dtype=torch.float64, device=dev)
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            a2a_broken = bool(bad.item())  # (every rank holds the same verdict: the m-sharded route is then reported as failed, not run)

    import heracles_amd as hx
    from heracles_amd import distributed as hxd

    backend = dist.get_backend() if world > 1 else None  # "nccl" (= RCCL on ROCm) or "gloo" (the one-GPU rehearsal): named in the line as it is
    comm = {"nccl": "RCCL", "gloo": "gloo (host copies: rehearsal on one GPU, NOT RCCL)", None: "none"}.get(backend, str(backend))

    def agree_or_leave(ok, tag):
        """after a phase that may have failed on this rank alone: the verdict of all ranks through the rendezvous store (no collective).
        0 = all fine, world = all failed (the common fallback is safe); anything else: this process leaves the job with a non-zero code
        (the launcher tears the other ranks down) instead of pairing up mismatched collectives"""
        if world == 1:
            return 0 if ok else 1
        nfail = hxd.unanimous(ok, tag)
        if 0 < nfail < world:
            print(f"[bench rank {rank}] {tag}: {nfail} of {world} ranks failed -- leaving the job", file=sys.stderr, flush=True)
            sys.stdout.flush()
            os._exit(5)
        return nfail

    hx.init(local)
    nside, lmax, nbins = args.nside, args.lmax, args.nbins
    npix, nlm = 12 * nside * nside, (lmax + 1) * (lmax + 2) // 2
    plan = hx.Plan(nside, lmax)

    # ---- the job: maps ordered (spin-0 bins, spin-2 bins) per brought-in set ---------------------------------------
    # first leg: at N = 1 the job itself; at N > 1 the job that grows with N (every rank brings one set of maps: `value_weak`) -- the
    # fixed job follows below through both of its routes
    nsets = world
    per_set = fixed_job(args.workload, nbins)
    if args.workload == "euclid":
        nbins = 13
    spins = per_set * nsets
    steps1 = args.steps if (world == 1 or args.scaling == "weak") else (args.weak_steps or min(args.steps, 5))
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

    cl_host = hx.pinned_empty((work.nrows, lmax + 1)) if world == 1 else None

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
        return work.all_pairs_cl(out=cl_host)  # rank 0: every Cl block on the host (N = 1: in ONE page-locked array, overwritten by every step)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, nsteps, nwarm):
        """nwarm untimed calls, then EXACTLY nsteps calls between (barrier + synchronize) brackets; seconds = max over the ranks"""
        res = None
        for _ in range(nwarm):
            fn()
        sync()
        t_ = time.perf_counter()
        for _ in range(nsteps):
            res = fn()
        sync()
        d_ = time.perf_counter() - t_
        if world > 1:
            tt_ = torch.tensor([d_], dtype=torch.float64, device="cpu" if share else dev)
            dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
            d_ = float(tt_.item())
        return d_, res

    def cl_direct_check(wobj, rows_, checks_):
        """rows of an all-pairs result against direct sums over the alms in wobj's buffer (independent of the all-pairs kernel):
        max |got - ref| / max |ref| over the component spectra of the map pairs in checks_, and how many were checked"""
        w_ = torch.full((nlm,), 2.0, dtype=torch.float64, device=dev)
        w_[: lmax + 1] = 1.0
        idx_ = torch.cat([torch.arange(m, lmax + 1, device=dev) for m in range(lmax + 1)])
        buf_, e_, n_ = wobj.buffer(), 0.0, 0
        for (i, j) in checks_:
            i, j = min(i, j), max(i, j)
            for ka, ca in enumerate(wobj.comps_of_map[i]):
                for kb, cb in enumerate(wobj.comps_of_map[j]):
                    a, b_ = buf_[ca], buf_[cb]
                    ref = torch.zeros(lmax + 1, dtype=torch.float64, device=dev).index_add_(0, idx_, w_ * (a.real * b_.real + a.imag * b_.imag))
                    ref = (ref / (2.0 * torch.arange(lmax + 1, device=dev) + 1.0)).cpu().numpy()
                    got = rows_[wobj.row0[i, j] + ka * len(wobj.comps_of_map[j]) + kb]
                    e_ = max(e_, float(np.abs(got - ref).max() / np.abs(ref).max()))
                    n_ += 1
        return e_, n_

    weak_error = None
    try:
        if os.environ.get("HX_BENCH_FAIL_WEAK") == "1" and world > 1:  # (test hook: the growing job fails on every rank alike)
            raise RuntimeError("HX_BENCH_FAIL_WEAK")
        for _ in range(args.warmup):
            step()
        hx._lib.profile_enable(True)
        hx._lib.profile_reset()
        dt, cls = timed(step, steps1, 0)
        cls = None if cls is None else np.array(cls)  # (a copy for the checks below: the page-locked array belongs to the steps)
        hx._lib.profile_enable(False)
    except Exception as exc:  # noqa: BLE001
        # N > 1 only: the job that grows with N is the side figure (`value_weak`); if it fails -- on every rank alike: its collectives have
        # not run over RCCL before the driver's first N > 1 run -- the fixed job below still gets its line.  The per-GPU kernel figures
        # then come from the local transforms alone.
        if world == 1:
            raise
        weak_error = f"{type(exc).__name__}: {exc}"[:400]
    # a failure on ONE rank must not go on into the collectives of the routes below while the others are elsewhere: unanimous or out
    if agree_or_leave(weak_error is None, "weak-leg"):
        weak_error = weak_error or "failed on every rank"
        hx._lib.profile_enable(True)
        hx._lib.profile_reset()
        for _ in range(steps1):
            if n2:
                plan.map2alm(maps2.view(2 * n2, npix), 2, pix_weights=pw, out=alm2.view(2 * n2, nlm))
            if n0:
                plan.map2alm(maps0, 0, pix_weights=pw, out=alm0)
        torch.cuda.synchronize()
        hx._lib.profile_enable(False)
        dt, cls = float("nan"), None
    npairs = len(work.pairs)
    value = npairs * steps1 / dt if weak_error is None else None
    ms_per_step = dt / steps1 * 1e3 if weak_error is None else None

    # ---- per-family kernel times (HIP events on the library stream) ----------------------------------------------
    prof = {}
    for k in ("ring_fft", "fourier_combine", "legendre_analysis", "legendre_analysis_s0", "legendre_analysis_s2",
              "alm_reduce", "alm2cl"):
        n_, ms_ = hx._lib.profile_get(k)
        prof[k] = {"launches": n_, "ms_per_step": ms_ / max(steps1, 1)}

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
        exe = (mf + vf) * steps1 / sec / 1e12 if sec > 0 else 0.0
        traffic, tpath = pmc_traffic("hx::k_legendre_duo<%d" % spin)
        mfx = mf * steps1 / sec / 1e12 if sec > 0 else 0.0
        return {"kernel": kernel, "bound": "mfma", "achieved": mfx, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": mfx / FP64_PEAK_TFLOPS,
                "achieved_is": "EXECUTED FP64 flops of the MATRIX instructions the kernel issued, counted by the kernel itself (stages whose rings are "
                               "all dead -- below 2^-100 for spin 2, 2^-300 for spin 0 -- skip theirs) / kernel time: the figure SQ_VALU_MFMA_BUSY_CYCLES of profiles/ gives (busy cycles x 32 flop; "
                               "0.71 of the pipe's cycles at 2.38 GHz).  The recursion's vector flops share the same FP64 pipe and are reported beside it "
                               "(executed_valu_tflops, frac_incl_vector), not added",
                "frac_incl_vector": exe / FP64_PEAK_TFLOPS,
                "task_list_mfma_tflops": mf_model * steps1 / sec / 1e12 if sec > 0 else 0.0,
                "executed_mfma_tflops": mf * steps1 / sec / 1e12 if sec > 0 else 0.0,
                "executed_valu_tflops": vf * steps1 / sec / 1e12 if sec > 0 else 0.0,
                "algorithmic_tflops": alg_flops_step * steps1 / sec / 1e12 if sec > 0 else 0.0,
                "algorithmic_is": "SURVEY 8d's F0 = 8 * 2 nside * nlm per spin-0 component (3 F0 per spin-2 field) / kernel time; "
                                  "exceeds the executed rate because north/south symmetry, ring pruning and the shared recursion "
                                  "remove work -- not a pipe utilisation",
                "traffic": traffic, "traffic_unit": "HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE)",
                "traffic_source": f"committed PMC passes ({tpath}), not measured by this run" if tpath else None,
                "launches": nl_, "avg_launch_ms": ms_ / nl_ if nl_ else None}

    nc0, nc2 = n0, 2 * n2
    # (at N > 1 the roofline blocks come from the first leg, whose per-GPU job is the N = 1 job: the same launches of the same kernels)
    roofline = roof("legendre_analysis_s2", "hx::k_legendre_duo<2,*>", n2 * 3 * F0, 2, nc2)
    roofline_s0 = roof("legendre_analysis_s0", "hx::k_legendre_duo<0,*>", n0 * F0, 0, nc0)

    # ---- N > 1: the FIXED job (one set of maps: the N = 1 job) on all N GPUs through both routes, K steps each.  `value` is the
    # better one: the same 210 pairs at every N. ----
    strong, verify_multi, routes, weak_check = None, None, None, None
    if world > 1:
        if rank == 0 and not args.no_verify and cls is not None:
            e_, n_ = cl_direct_check(work, cls, [(0, 0), (0, nmaps_total - 1), (nmaps_total - 1, nmaps_total - 1), (0, nbins)])
            weak_check = {"max_err_over_max": e_, "spectra_checked": n_, "tolerance": 1e-11, "ok": bool(e_ <= 1e-11)}
        del maps0, maps2, alm0, alm2  # (the weak-scaling job is done: room for the fixed job's buffers)
        work._buf = None
        torch.cuda.empty_cache()
        plan.release_scratch()  # (the full-size sweeps of the first leg hold ~100 GB of operands and rows; the routes below size their own)

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

        def leg_kernels(nsteps):
            kk = {}
            for k in ("ring_fft", "ring_modes", "fourier_combine", "legendre_analysis", "alm_reduce", "alm2cl"):
                n_, ms_ = hx._lib.profile_get(k)
                if n_:
                    kk[k] = {"launches": n_, "ms_per_step": ms_ / max(nsteps, 1)}
            return kk

        routes = {}
        # (a failure of a route -- neither has run over RCCL before the driver's first N > 1 run, only over gloo and in one-GPU
        # rehearsals -- must not cost the line the other route: the ranks agree on success before every collective of the m-sharded
        # route (guard=True) and an exception is raised by all of them alike and reported instead of the figure)
        cls_m = None
        try:
            if a2a_broken:
                raise RuntimeError("all_to_all_single over RCCL returned the wrong blocks in the priming call")
            ms = hxd.MShardedTwoPoint(per_set, world, rank, nlm, lmax, hxd.HipStages(plan, dev))
            s0, s2 = seeded_maps(ms.local_maps)
            for _ in range(args.warmup):
                ms.run(s0, s2, pix_weights=pw, guard=True)
            hx._lib.profile_enable(True)
            hx._lib.profile_reset()
            dts, cls_m = timed(lambda: ms.run(s0, s2, pix_weights=pw, guard=True), args.steps, 0)
            hx._lib.profile_enable(False)
            routes["m_sharded"] = {"value": len(ms.pairs) * args.steps / dts, "unit": "map->Cl pairs/s", "ms_per_step": dts / args.steps * 1e3,
                                   "maps_total": len(per_set), "pairs": len(ms.pairs), "orders_first_count_step": ms.sets,
                                   "checksum": float(np.abs(cls_m).sum()), "kernels_rank0": leg_kernels(args.steps),
                                   "what": "ring Fourier stage of the rank's maps, all-to-all of the ring modes by owner of the order m (rank q: m = q, "
                                           f"q + N, ...) over {comm}, Legendre stage of ALL components on the rank's orders, partial Cl, all-reduce",
                                   "backend": backend}
            del s0, s2, ms
        except Exception as exc:  # noqa: BLE001
            hx._lib.profile_enable(False)
            routes["m_sharded"] = {"value": None, "error": f"{type(exc).__name__}: {exc}"[:400]}
            cls_m = None
        # (inside run() the ranks agree before every collective; a failure around it -- set-up, the seeded maps -- may be this rank's alone)
        agree_or_leave(routes["m_sharded"].get("value") is not None, "route-m-sharded")
        torch.cuda.empty_cache()

        rows, wk = None, None
        try:
            wk = hxd.ShardedTwoPoint(per_set, world, rank, nlm, lmax)
            v0, v2 = seeded_maps(wk.local_maps)
            a0, a2 = wk.local_alm_views(dev)

            def ag_step():
                if v2.shape[0]:
                    plan.map2alm(v2.view(-1, npix), 2, pix_weights=pw, out=a2.view(-1, nlm))
                wk.exchange_begin(2)
                if v0.shape[0]:
                    plan.map2alm(v0, 0, pix_weights=pw, out=a0)
                wk.exchange_begin(0)
                return wk.all_pairs_cl()

            for _ in range(args.warmup):
                ag_step()
            hx._lib.profile_enable(True)
            hx._lib.profile_reset()
            dta, rows = timed(ag_step, args.steps, 0)
            hx._lib.profile_enable(False)
            routes["all_gather"] = {"value": len(wk.pairs) * args.steps / dta, "unit": "map->Cl pairs/s", "ms_per_step": dta / args.steps * 1e3,
                                    "maps_total": len(per_set), "pairs": len(wk.pairs), "maps_of_rank": [len(m_) for m_ in wk.maps_of],
                                    "kernels_rank0": leg_kernels(args.steps),
                                    "what": "the maps dealt to the ranks by cost (a spin-2 map = 3 units), transforms into the rank's shard of ONE buffer, "
                                            f"in-place all-gather per spin over {comm} (spin-2 part under the spin-0 transform), tiled pair split, Cl blocks gathered on rank 0",
                                    "backend": backend}
            del v0, v2
        except Exception as exc:  # noqa: BLE001
            hx._lib.profile_enable(False)
            routes["all_gather"] = {"value": None, "error": f"{type(exc).__name__}: {exc}"[:400]}
            rows = None
        agree_or_leave(routes["all_gather"].get("value") is not None, "route-all-gather")  # (a rank that failed alone leaves; see agree_or_leave)

        # ---- verification at N > 1 (outside every timed region), on the results of the two timed routes -- the same seeded maps went
        # through both: (i) the m-sharded route's spectra against the all-gather route's rows, (ii) map pairs of the latter against
        # direct sums over the gathered alms ----
        if not args.no_verify and rank == 0:
            try:
                verify_multi = {"routes": None, "cl_vs_direct_sum": None}
                if rows is None:
                    raise RuntimeError("the all-gather route did not produce its rows: " + str(routes["all_gather"].get("error")))
                if cls_m is not None:
                    e = float(np.abs(cls_m - rows).max() / np.abs(rows).max())
                    verify_multi["routes"] = {"m_sharded_vs_all_gather_max_err_over_max": e, "spectra": int(rows.shape[0]), "tolerance": 1e-10}
                nm = len(per_set)
                ecl, nchk = cl_direct_check(wk, rows, [(0, 0), (0, nm - 1), (nm - 1, nm - 1), (nm // 2, nm // 2 + 1)])
                verify_multi["cl_vs_direct_sum"] = {"max_err_over_max": ecl, "spectra_checked": nchk, "tolerance": 1e-11}
                verify_multi["ok"] = bool(ecl <= 1e-11 and verify_multi["routes"] is not None
                                          and verify_multi["routes"]["m_sharded_vs_all_gather_max_err_over_max"] <= 1e-10)
                verify_multi["what"] = ("the fixed job (maps seeded by their global index) through both timed routes: spectra of the m-sharded route "
                                        "(all-to-all + all-reduce) against the rows of the all-gather route on rank 0, and map pairs of the latter "
                                        "against direct sums over the gathered alms; ok needs BOTH")
            except Exception as exc:  # noqa: BLE001
                verify_multi = {"ok": False, "error": f"{type(exc).__name__}: {exc}"[:400]}
        del wk
        torch.cuda.empty_cache()
        best = max((k for k in routes if routes[k].get("value")), key=lambda k: routes[k]["value"], default=None)
        strong = dict(routes[best], route=best) if best else {"value": None, "error": "both fixed-job routes failed", "route": None}

    # ---- the mixing-matrix request list of BASELINE configs[4] (13 bins x (Positions on the visibility mask, Shears and a scalar on the
    # weight mask): 780 keys) with the configuration file's bins (32 log 2l+1, examples/heracles.cfg:4-6) -- the loop of
    # heracles/twopoint.py:354-397 as heracles/cli.py:696-716 runs it.  Independent per key: dealt to the ranks by cost, no collective
    # (SURVEY 8e last bullet); every rank builds its keys through its own context; seconds = max over the ranks. ----
    mix_list = None
    if not args.no_mixmat:
        import types as _types

        from heracles_amd.twopoint import mixing_requests, request_cost, split_requests

        Lm = args.mixmat_lmax or lmax
        ellm = np.arange(Lm + 1)
        nb13 = 13
        mfields = {"POS": _types.SimpleNamespace(mask="VIS", spin=0), "SHE": _types.SimpleNamespace(mask="WHT", spin=2),
                   "CON": _types.SimpleNamespace(mask="WHT", spin=0)}
        mcls = {}
        for a_, b_ in (("VIS", "VIS"), ("VIS", "WHT"), ("WHT", "WHT")):
            for i_ in range(nb13):
                for j_ in range(i_ if a_ == b_ else 0, nb13):
                    mcls[a_, b_, i_, j_] = 4 * np.pi * 0.35 * np.exp(-ellm * (ellm + 1) / (3000.0 + 40.0 * i_ + 7.0 * j_)) + 1e-3 / (1.0 + ellm) ** 2
        edges = np.unique(np.geomspace(2, Lm + 1, 33).astype(int))
        todo_all = mixing_requests(mfields, mcls)
        my_keys = split_requests(todo_all, rank, world)
        try:
            hx.mixing_matrices(mfields, {k: mcls[k] for k in list(mcls)[:2]}, l1max=Lm, l2max=Lm, l3max=Lm, bins=edges, weights="2l+1")  # warm-up
            sync()
            tml = time.perf_counter()
            mine_mm = hx.mixing_matrices(mfields, mcls, l1max=Lm, l2max=Lm, l3max=Lm, bins=edges, weights="2l+1", rank=rank, world=world)
            sync()
            dml = time.perf_counter() - tml
            err_ml = None if len(mine_mm) == len(my_keys) else f"{len(mine_mm)} keys built, {len(my_keys)} dealt"
        except Exception as exc:  # noqa: BLE001
            dml, err_ml, mine_mm = float("nan"), f"{type(exc).__name__}: {exc}"[:400], {}
        if agree_or_leave(err_ml is None, "mixmat-list") == 0:
            if world > 1:
                tt_ = torch.tensor([dml], dtype=torch.float64, device="cpu" if share_gpu else dev)
                dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
                dml = float(tt_.item())
            first = next(iter(mine_mm.values())) if mine_mm else None
            mix_list = {"keys": len(todo_all), "keys_this_rank": len(my_keys), "cost_this_rank": sum(request_cost(r_[2]) for r_ in my_keys),
                        "seconds": dml, "keys_per_s": len(todo_all) / dml, "L": Lm, "bins": int(edges.size - 1), "weights": "2l+1",
                        "shape_of_a_key": None if first is None else list(np.shape(first.array)),
                        "what": f"heracles_amd.mixing_matrices(fields, cls, l1max=l2max=l3max={Lm}, bins=32 log edges, weights='2l+1', rank, world): the "
                                f"{len(todo_all)} keys of 13 bins x (POS on VIS, SHE + CON on WHT) dealt to {world} rank(s) by cost, binned rows built on the "
                                "GPU (no full matrix, no collective), Result objects on the host; wall seconds, max over the ranks"}
        else:
            mix_list = {"keys": len(todo_all), "error": err_ml or "failed on every rank"}
        del mine_mm

    out = None
    if rank == 0:
        assert weak_error is not None or (cls is not None and cls.shape[0] == work.nrows)
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
            plan.release_scratch()  # (the staging buffers of the streamed sweeps are 64 GB: room for the legs that follow)

        # ---- verification of what was timed (outside the timed region) --------------------------------------------
        verify, cpu = None, None
        osample = None
        fast = None
        if world == 1 and not (args.no_verify and args.no_cpu_baseline):
            stride = (8 if nside >= 2048 else 1) if not args.no_cpu_baseline else (512 if nside >= 2048 else 4)
            t_host = maps0[:1].cpu().numpy()
            qu_host = maps2[0].cpu().numpy()
            pw_host = pw.cpu().numpy()
            try:  # (the oracle's OpenMP regions run on the CPUs the cgroup grants, not on every logical CPU of the host)
                from oracle import hxfast as _hf

                _hf.set_threads(_hf.cpu_quota()[0])
            except Exception:  # noqa: BLE001
                pass
            oa0, oa2, tim = oracle_sample(nside, lmax, t_host, qu_host, stride, pix_weights=pw_host)
            osample = (oa0, oa2, tim, stride)
            if not args.no_cpu_baseline:
                try:
                    fast = cpu_baseline_vectorised(nside, lmax, per_set.count(0), per_set.count(2), t_host, qu_host, pw_host)
                except Exception as exc:  # noqa: BLE001 -- (an optional leg must not cost the line)
                    fast = ({"value": None, "kind": "port-vectorised", "error": f"{type(exc).__name__}: {exc}"[:300]}, None, None)
            del t_host, qu_host, pw_host
        if not args.no_verify and world == 1:
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
            if fast is not None and fast[1] is not None:
                # ... and against the vectorised CPU restatement on EVERY m (it is itself checked by the oracle, tests/test_oracle_fast.py)
                f0_, f2_ = fast[1], fast[2]
                verify["alm_vs_vectorised_cpu_all_m"] = {"spin0_max_err_over_max": float(np.abs(g0 - f0_[0]).max() / scale0),
                                                         "spin2_max_err_over_max": float(np.abs(g2 - f2_).max() / scale2), "tolerance": 1e-10}
            # Cl rows against a direct sum over the device alms (independent of the all-pairs kernel)
            checks = [(0, 0), (0, min(1, nmaps_total - 1)), (nmaps_total - 1, nmaps_total - 1)]
            if nmaps_total > nbins:
                checks.append((0, nbins))
            ecl, nchk = cl_direct_check(work, cls, checks)
            verify.update(cl_vs_direct_sum={"max_err_over_max": ecl, "spectra_checked": nchk, "tolerance": 1e-11})
            ok = ecl <= 1e-11
            if "alm_vs_oracle" in verify:
                ok = ok and verify["alm_vs_oracle"]["spin0_max_err_over_max"] <= 1e-10 and verify["alm_vs_oracle"]["spin2_max_err_over_max"] <= 1e-10
            if "alm_vs_vectorised_cpu_all_m" in verify:
                va = verify["alm_vs_vectorised_cpu_all_m"]
                ok = ok and va["spin0_max_err_over_max"] <= 1e-10 and va["spin2_max_err_over_max"] <= 1e-10
            verify["ok"] = bool(ok)
        if not args.no_cpu_baseline and world == 1 and osample is not None:  # reported on rank 0 at N = 1 only
            oa0, oa2, tim, stride = osample
            scalar = cpu_baseline(nside, lmax, per_set.count(0), per_set.count(2), oa0, oa2, tim, stride)
            if fast is not None and fast[0] is not None and fast[0].get("value"):
                cpu = fast[0]
                cpu["scalar_port"] = {"value": scalar["value"], "sample": scalar["sample"],
                                      "what": "the scalar oracle (the checker) on every 8th m, scaled: round 5's cpu_baseline, kept for the record"}
            else:
                cpu = scalar  # (no AVX-512 on this host, or the vectorised leg failed: the scalar port, kind "port")
                if fast is not None and fast[0] is not None:
                    cpu["vectorised_error"] = fast[0].get("error")

        # ---- the same 20-map job with the mapper's own default, healpy's iter = 3 (HipHealpixMapper(niter=3): three Jacobi iterations =
        # three batched syntheses + three more analyses per transform; the reference passes no iter, heracles/healpy.py:183-189).
        # Informational: `value` is the niter = 0 job SURVEY 8d prescribes. ----
        def niter3_leg():
            def step3():
                if n2:
                    plan.map2alm(maps2.view(2 * n2, npix), 2, pix_weights=pw, out=alm2.view(2 * n2, nlm), niter=3)
                if n0:
                    plan.map2alm(maps0, 0, pix_weights=pw, out=alm0, niter=3)
                return work.all_pairs_cl()

            step3()
            hx._lib.profile_enable(True)
            hx._lib.profile_reset()
            d3, _ = timed(step3, 2, 0)
            hx._lib.profile_enable(False)
            k3 = {}
            for k in ("ring_fft", "fourier_combine", "legendre_analysis", "legendre_synthesis", "legendre_synth_duo", "synth_table", "alm_reduce", "alm2cl"):
                n_, ms_ = hx._lib.profile_get(k)
                k3[k] = {"launches": n_, "ms_per_step": ms_ / 2.0}
            return {"value": npairs * 2 / d3, "unit": "map->Cl pairs/s", "ms_per_step": d3 / 2 * 1e3, "steps": 2, "kernels": k3,
                    "what": "the timed job with niter = 3 (the mapper's default, healpy's iter): per transform 4 analyses + 3 syntheses; batched "
                            "synthesis on the matrix unit (k_synth_duo, round 5), residuals formed in the scatter pass"}

        niter3 = None
        if world == 1 and not args.no_niter3:
            try:  # (an optional leg must not cost the line: an exception is reported in its place)
                niter3 = niter3_leg()
            except Exception as exc:  # noqa: BLE001
                hx._lib.profile_enable(False)
                niter3 = {"value": None, "error": f"{type(exc).__name__}: {exc}"[:400]}
            plan.release_scratch()

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
            # Where the result goes decides what a build costs, so every destination is timed and named (round 4's median
            # over builds into a pool of recycled blocks hid 0.125 s outliers; that pool is gone):
            #   device      -- a device tensor: the GPU side alone (nodes, tables, two products, combine)
            #   pinned      -- out= a page-locked array the caller keeps (hx.pinned_empty / MixmatContext.result_buffer): + the DMA
            #   pageable    -- out= a pageable numpy array the caller keeps and has touched: + the staging copy
            #   fresh       -- no out=: a new 0.9 GB numpy array per build, as convolvecl returns one: + first-touch page faults
            # `mixmat_build_sec` is the pinned-destination median of five (what a loop over many masks pays per matrix when it hands each
            # result on, heracles/twopoint.py:393-397); the fresh-array figures stand beside it, first (cold) build included.
            shape = (3, L + 1, L + 1)

            def builds(n, **kw):
                ts = []
                for _ in range(n):
                    tm = time.perf_counter()
                    r_ = hx.mixmat_eb(wl, **kw)
                    if hasattr(r_, "is_cuda"):
                        torch.cuda.synchronize(dev)
                    ts.append(time.perf_counter() - tm)
                return ts, r_

            t_fresh, mm = builds(3)
            cold = t_fresh[0]
            del mm
            dev_out = torch.empty(shape, dtype=torch.float64, device=dev)
            t_dev, _ = builds(3, out=dev_out)
            del dev_out
            pin = hx.pinned_empty(shape)
            builds(1, out=pin)
            hx._lib.load().hx_mixmat_gemm_clock()      # (clear the clock samples of the builds so far)
            hx._lib.profile_enable(True)
            hx._lib.profile_reset()
            mix_all, mm = builds(5, out=pin)
            hx._lib.profile_enable(False)
            gemm_clock = hx._lib.load().hx_mixmat_gemm_clock()
            pageable = np.zeros(shape)
            t_page, _ = builds(3, out=pageable)
            del pageable
            mix_s = float(np.median(mix_all))
            ng, gms = hx._lib.profile_get("mixmat_gemm")
            _, gkms = hx._lib.profile_get("mixmat_gemm_kernel")
            gms, gkms = gms / 5.0, (gkms or gms) / 5.0   # per build (two products)
            N = (3 * L) // 2 + 1
            gflop = 2.0 * (L + 1) ** 2 * N * 2  # two products
            nb_, kpad_ = (L + 1 + 127) // 128, (N + 31) // 32 * 32
            xflop = 2.0 * 128 * 128 * kpad_ * (nb_ * (nb_ + 1) // 2) * 2  # what k_mixmat_gemm executes: the upper triangle of 128 x 128 tiles, padded nodes
            peak_at_clock = FP64_PEAK_TFLOPS * gemm_clock / 2.4 if gemm_clock else None  # 78.6 TFLOP/s is the datasheet figure at 2.4 GHz
            mix = {"L": L, "seconds": mix_s, "seconds_all": mix_all, "seconds_min": min(mix_all), "seconds_max": max(mix_all),
                   "statistic": "median of 5 builds into ONE page-locked out= buffer the caller owns", "gemm_ms": gms, "gemm_kernel_ms": gkms,
                   "seconds_is": "hx.mixmat_eb(wl, out=pinned) host-visible result, five builds in a row into the same caller-owned page-locked array "
                                 "(hx.pinned_empty): GPU work + DMA, no staging copy, no page faults, no library-side pool",
                   "seconds_by_destination": {"device_tensor": t_dev, "pinned_out": mix_all, "pageable_out_reused": t_page, "fresh_numpy_default": t_fresh},
                   "seconds_cold_first_build_fresh_numpy": cold,
                   "seconds_by_destination_is": "the same build into a device tensor (GPU side alone), the pinned buffer (+ DMA), a pageable array the caller "
                                                "re-uses (+ staging copy) and a fresh numpy array per build, the default (+ first-touch page faults; the first "
                                                "one is also the first L = 6144 build of the process)",
                   "gemm_clock_ghz": gemm_clock or None,
                   "gemm_clock_is": "shader clock measured UNDER k_mixmat_gemm_dma (every 64th tile: s_memtime / s_memrealtime around its work) during the five timed builds",
                   "gemm_frac_of_pipe_at_measured_clock": (xflop / (gkms * 1e-3) / 1e12 / peak_at_clock) if (gkms and peak_at_clock) else None,
                   "gemm_ms_is": "gemm_kernel_ms: the two launches of k_mixmat_gemm_dma; gemm_ms: the same with the two passes that form T diag(s) "
                                 "for them (k_scale_table: the k tiles go from HBM into LDS without passing through registers)",
                   "gemm_tflops_executed": xflop / (gkms * 1e-3) / 1e12 if gkms else None,
                   "gemm_frac_of_fp64_mfma_peak": xflop / (gkms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if gkms else None,
                   "gemm_tflops_executed_incl_scaling_pass": xflop / (gms * 1e-3) / 1e12 if gms else None,
                   "gemm_tflops_algorithmic": gflop / (gms * 1e-3) / 1e12 if gms else None,
                   "gemm_tflops_algorithmic_is": "SURVEY 8d's 2 (l1max + 1)(l2max + 1) N per product / kernel time: the symmetric product is computed for the "
                                                 "upper triangle of tiles only, so this exceeds what the matrix pipe executes (gemm_tflops_executed) -- not a utilisation",
                   "checksum": float(np.abs(mm[2] - (mm[0] - mm[1])).max())}
            # the production call: the rows binned (bins = 32 log 2l+1), built directly from binned Wigner-d tables -- seconds per key, host -> host
            from heracles_amd.binning import BinPlan

            edges_b = np.unique(np.geomspace(2, L + 1, 33).astype(int))
            plan_b = BinPlan(ell, edges_b, "2l+1")
            with hx.MixmatContext(L, L, L) as cx:
                tb0 = time.perf_counter()
                cx.set_bins(plan_b)
                cx.binned(wl, (2, 2))
                t_first = time.perf_counter() - tb0
                per_kind = {}
                for nm_, sp_ in (("mixmat_eb_22", (2, 2)), ("mixmat_02", (0, 2)), ("mixmat_00", (0, 0))):
                    cx.binned(wl, sp_)
                    ts_ = []
                    for _ in range(7):
                        tb = time.perf_counter()
                        rb = cx.binned(wl, sp_)
                        ts_.append(time.perf_counter() - tb)
                    per_kind[nm_] = {"seconds": float(np.median(ts_)), "seconds_all": ts_, "shape": list(rb.shape)}
                # against the host binning of the full matrices of the same context (outside every timed region)
                full_b = cx(wl, (2, 2), out=pin)
                want_b = plan_b.apply(full_b, 1)
                err_b = float(np.abs(cx.binned(wl, (2, 2)) - want_b).max() / np.abs(want_b).max())
            t_host = time.perf_counter()
            plan_b.apply(mm, 1)
            t_host = time.perf_counter() - t_host
            mix["binned"] = {"L": L, "bins": int(plan_b.nbins), "weights": "2l+1", "seconds_per_key": per_kind["mixmat_eb_22"]["seconds"],
                             "per_kind": per_kind, "seconds_first_key_incl_tables": t_first,
                             "max_err_vs_binned_full_matrix_over_max": err_b, "host_binning_of_a_full_matrix_seconds": t_host,
                             "what": "MixmatContext.set_bins + binned(cl, spin): rows of the mixing matrices binned as heracles.result.binned does "
                                     "(heracles/twopoint.py:391-397), numerators = (binned Wigner-d tables) diag(w xi) (tables)^T on the matrix unit, host -> host "
                                     "(cl on the host in, (3, bins, L + 1) numpy array out), median of 7; the reference bins the full matrix on the host "
                                     "column by column; host_binning_of_a_full_matrix_seconds is THIS repository's vectorised host rule on the same matrix"}
        peaks = hx._lib.measure_peaks()
        ncomp_set = sum(2 if s_ else 1 for s_ in per_set)
        set_name = f"{nbins} bins x ({'2 spin-0 + 1 spin-2' if args.workload == 'euclid' else 'spin-0, spin-2'}) maps"
        common = (f"nside={nside}, lmax={lmax}, niter=0, ring weights 1, pix_weights: synthetic file in healpy's format (its files are not "
                  f"available offline); inputs resident in HBM")
        weak = None
        if world > 1:
            # the job that grows with N: every rank brings one set of maps; pairs grow as N^2 while the transforms per GPU stay what they are,
            # so its pairs/s is NOT a speed-up -- the per-GPU transform rate beside it is the weak-scaling figure proper
            weak = {"value": value, "unit": "map->Cl pairs/s", "ms_per_step": ms_per_step, "steps": steps1, "maps_total": nmaps_total,
                    "pairs": npairs, "maps_per_gpu": len(per_set), "transforms_per_gpu_per_s": len(per_set) * 1e3 / ms_per_step if ms_per_step else None,
                    "error": weak_error, "cl_vs_direct_sum": weak_check, "kernels_rank0": prof,
                    "what": f"every rank brings its own {set_name} ({len(per_set)} maps): transforms, in-place all-gather of the alms in two parts, "
                            f"tiled pair split over all {nmaps_total} maps; pairs = {nmaps_total} * {nmaps_total + 1} / 2"}
        if world > 1 and args.scaling == "strong":
            # `value`: the SAME job as at N = 1, the better of its two routes (both in "routes")
            value, ms_per_step = strong.get("value"), strong.get("ms_per_step")
            cfg_maps, cfg_pairs, cfg_mine = len(per_set), len(per_set) * (len(per_set) + 1) // 2, None
            workload = (f"{set_name} in all = {len(per_set)} maps / {ncomp_set} components sharded over {world} GPU(s), {common}; "
                        f"{cfg_pairs} auto+cross map pairs: the same job at every N")
            par = {"m_sharded": f"sharded by the order m over {world} GPUs: ring modes all-to-all ({comm}), full-batch Legendre stage on m = rank mod {world}, Cl all-reduce",
                   "all_gather": f"maps dealt to {world} GPUs by cost, all-gather of alms over {comm}, tiled pair split",
                   None: "both fixed-job routes failed"}[strong.get("route")]
            prof_line = strong.get("kernels_rank0") or {}
        else:
            cfg_maps, cfg_pairs, cfg_mine = nmaps_total, npairs, len(mine)
            workload = (f"{set_name} {'per GPU' if world > 1 else 'in all'} = {nmaps_total} maps / {sum(2 if s_ else 1 for s_ in spins)} components "
                        f"over {world} GPU(s), {common}; {npairs} auto+cross map pairs")
            par = f"maps dealt to {world} GPUs by cost, all-gather of alms over {comm}, tiled pair split" if world > 1 else "1 GPU"
            prof_line = prof
        out = {
            "metric": "map->Cl pairs/sec + mixing-matrix build sec, nside=%d lmax=%d" % (nside, lmax),
            "value": value, "unit": "map->Cl pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload,
                       "nside": nside, "lmax": lmax, "maps_total": cfg_maps, "maps_this_rank": cfg_mine, "pairs": cfg_pairs,
                       "pix_weights": ("synthetic array without symmetry (generic path)" if args.generic_weights else "synthetic weights in healpy's compressed format, expanded to the full sky (symmetric like healpy's)") + ", applied in the timed path",
                       "parallelism": par, "backend": backend},
            "verified": (None if verify is None and verify_multi is None
                         else bool((verify is None or verify.get("ok")) and (world == 1 or (verify_multi or {}).get("ok"))
                                   and (weak_check is None or weak_check["ok"]))),
            "verify": verify,
            "verify_multi": verify_multi,
            "value_strong": strong.get("value") if strong else (value if world == 1 else None),
            "strong_scaling": strong,
            "routes": routes,
            "value_weak": weak["value"] if weak else None,
            "weak_scaling": weak,
            "value_host_to_host": host_leg["value"] if host_leg else None,
            "host_to_host": host_leg,
            "niter3": niter3,
            "single_map_transforms": single,
            "mixmat_build_sec": mix["seconds"] if mix else None,
            "mixmat": mix,
            "mixmat_list": mix_list,
            "roofline": roofline,
            "roofline_spin0_kernel": roofline_s0,
            "cpu_baseline": cpu,
            "measured_peaks": peaks,
            "kernels": prof_line,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    plan.close()


if __name__ == "__main__":
    main()
