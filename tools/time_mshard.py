"""Rehearsal of the m-sharded strong-scaling route on ONE GPU: the 20-map job of the bench (10 spin-0 + 10 spin-2 maps, nside 4096,
lmax 6144); for WORLD = 1, 2, 4, 8 virtual ranks the Legendre stage of every rank's m-range (all 30 components, the full-batch
kernels) is run and timed one after the other, next to the ring-mode stage of a rank's share of the maps and the partial all-pairs
Cl of its range.  Prints per-rank times and max-over-ranks against the WORLD = 1 figures.  (The all-to-all itself needs N GPUs.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, heracles_amd as hx
from heracles_amd.distributed import HipStages, MShardedTwoPoint

hx.init(0)
nside, lmax = int(os.environ.get("NSIDE", 4096)), int(os.environ.get("LMAX", 6144))
npix = 12 * nside * nside
plan = hx.Plan(nside, lmax)
st = HipStages(plan)
spins = [0] * 10 + [2] * 10
maps0 = torch.randn((10, npix), dtype=torch.float64, device="cuda")
maps2 = torch.randn((20, npix), dtype=torch.float64, device="cuda")
worlds = [int(w) for w in os.environ.get("WORLDS", "1,2,4,8").split(",")]


def timed(fn):
    torch.cuda.synchronize(); hx._lib.synchronize()
    t = time.perf_counter(); fn(); hx._lib.synchronize()
    return (time.perf_counter() - t) * 1e3


base = None
for world in worlds:
    works = [MShardedTwoPoint(spins, world, r, plan.nlm, lmax, st) for r in range(world)]
    sets = works[0].sets
    t_leg, t_modes, t_cl = [], [], []
    for r, w in enumerate(works):
        orders = sets[r]
        size = st.modes_size(orders[1])
        # mode blocks of ALL 30 components for this rank's orders (in a real run they arrive through the all-to-all)
        b0 = st.ring_modes(maps0, [orders])[0]
        b2 = st.ring_modes(maps2, [orders])[0]
        alm = st.zeros_alm(30, plan.nlm)
        blocks0 = [b0[c * size : (c + 1) * size] for c in range(10)]
        blocks2 = [b2[c * size : (c + 1) * size] for c in range(20)]
        for rep in range(2):  # the second run is the timed one (plan scratch allocated, tables uploaded)
            tl = timed(lambda: (st.legendre(0, blocks0, orders, alm[:10]), st.legendre(2, blocks2, orders, alm[10:])))
        comps = [alm[k] for k in range(30)]
        cp = [(a, b) for a in range(30) for b in range(a, 30)]
        for rep in range(2):
            tc = timed(lambda: hx.alm2cl_pairs(comps, cp, lmax, m_range=(orders[0], lmax + 1, orders[2])))
        # this rank's share of the ring-mode stage: its own maps, all ranges
        n0, n2 = w.n0_of[r], w.n2_of[r]
        mine = torch.cat([maps0[:n0], maps2[: 2 * n2]]) if n0 + n2 else maps0[:0]
        for rep in range(2):
            tm = timed(lambda: st.ring_modes(mine, sets)) if mine.shape[0] else 0.0
        t_leg.append(tl); t_cl.append(tc); t_modes.append(tm)
        del b0, b2, alm, blocks0, blocks2, comps
        torch.cuda.empty_cache()
    tot = [a + b + c for a, b, c in zip(t_leg, t_modes, t_cl)]
    if base is None:
        base = (max(t_leg), max(tot))
    print(f"world {world}: orders (first, count, step) {sets}", flush=True)
    print(f"  legendre per rank ms: {' '.join('%.1f' % x for x in t_leg)}  | max {max(t_leg):.1f} = 1/{base[0] / max(t_leg):.2f} of world 1")
    print(f"  ring modes per rank ms: {' '.join('%.1f' % x for x in t_modes)} | partial Cl ms: {' '.join('%.1f' % x for x in t_cl)}")
    print(f"  compute per rank (modes + legendre + Cl), max {max(tot):.1f} ms = 1/{base[1] / max(tot):.2f} of world 1; exchanged per rank "
          f"{30 * st.modes_size(lmax + 1) * 8 / world * (world - 1) / world / 1e9:.2f} GB out", flush=True)
