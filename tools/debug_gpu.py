"""First-light GPU bisect: runs each piece against the oracle and prints errors."""
import sys, os, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import heracles_amd as hx
from oracle import hxoracle as ho

def rel(a, b): return np.abs(a-b).max()/max(np.abs(b).max(), 1e-300)
def step(name, fn):
    t=time.time()
    try:
        r = fn(); print(f"[{name}] {r}  ({time.time()-t:.2f}s)", flush=True)
    except Exception as e:
        print(f"[{name}] EXC {e!r}", flush=True); traceback.print_exc()

hx.init(0)
rng = np.random.default_rng(1)
def t_alm2cl():
    a = rng.standard_normal((3, 153)) + 1j*rng.standard_normal((3,153))
    return rel(hx.alm2cl(a), ho.alm2cl(a))
step("alm2cl", t_alm2cl)
def t_gl():
    x,w = hx.gauss_legendre(97); xo,wo = ho.gauss_legendre(97); return np.abs(x-xo).max(), np.abs(w-wo).max()
step("gl", t_gl)
def t_mm():
    cl = 1/(1+np.arange(41))**2
    return rel(hx.mixmat(cl), ho.mixmat(cl)), rel(hx.mixmat(cl, spin=(0,2)), ho.mixmat(cl, spin=(0,2))), rel(hx.mixmat_eb(cl), ho.mixmat_eb(cl))
step("mixmat", t_mm)
for nside,lmax in [(4,8),(8,16),(12,20),(16,40),(64,96)]:
    for spin in (0,2):
        def t():
            plan = hx.Plan(nside,lmax)
            m = rng.standard_normal((2, 12*nside**2))
            r = rel(plan.map2alm(m, spin), ho.map2alm(m, nside, lmax, spin=spin))
            alm = ho.map2alm(m, nside, lmax, spin=spin)
            r2 = rel(plan.alm2map(alm, spin), ho.alm2map(alm, nside, lmax, spin=spin))
            plan.close()
            return r, r2
        step(f"sht nside={nside} lmax={lmax} spin={spin}", t)
