"""Calibration of the per-block entry thresholds of k_legendre_duo / k_synth_duo (set_mode): for every 32-l block b of an order m, the margin
E_b (bits below 2^-100) a chain must have at the entry of the block so that no value of lambda above 2^-75 is reached inside it by a ring
that is skipped.  Emulates the normalised recursions of hx_sht.hip (k_init_norm0 / k_init_norm2) in long double.  CPU only.

    python tools/calibrate_dead_blocks.py [BLK [LMAX [NSIDE]]]      BLK: 32 (k_legendre_duo) or 16 (k_synth_duo); default 32 6144

With NSIDE the rings are the plan's own: the colatitudes of the HEALPix rings of that nside that libsharp's mlim rule (ring_mlim of
hx_analysis.hip) keeps for the order m; without it a 3000-point grid of colatitudes from the pruning limit to the equator (round 5).
Round 6 (VERDICT r5 Next #3): run for lmax 8000 (examples/heracles.cfg:56-62) and 12288 with m up to lmax - 200; the record is
profiles/r06_dead_block_calibration.txt and the kernels' margins E_b follow it."""
import numpy as np, sys
ld=np.longdouble
LMAX=int(sys.argv[2]) if __name__ == "__main__" and len(sys.argv)>2 else 6144
BLK=int(sys.argv[1]) if __name__ == "__main__" and len(sys.argv)>1 else 32   # l per block: 32 (k_legendre_duo), 16 (k_synth_duo)
def ring_mlim(lmax, spin, sth, cth):
    """ring_mlim of hx_analysis.hip (libsharp's published sharp_get_mlim): rings with m > mlim are skipped"""
    ofs=max(lmax*0.01,100.0)
    b=-2*spin*np.abs(cth); t1=lmax*sth+ofs; c=float(spin)*spin-t1*t1
    discr=b*b-4*c
    res=np.where(discr>0,(-b+np.sqrt(np.maximum(discr,0)))/2,lmax)
    return np.floor(np.minimum(res,lmax)+0.5).astype(np.int64)
def healpix_rings(nside):
    """colatitudes of the northern rings and the equator (2 nside), as long doubles"""
    i=np.arange(1,2*nside+1,dtype=ld); ns=ld(nside)
    z=np.where(i<ns, 1-i*i/(3*ns*ns), ld(4)/3-2*i/(3*ns))
    s=np.sqrt((1-z)*(1+z))
    return np.arctan2(s,z), s, z
def chains0(m, th, nb):
    x=np.cos(th); s=np.sin(th)
    logc=0.5*(np.sum(np.log(np.arange(1,2*m+2,2,dtype=np.float64)))-np.sum(np.log(np.arange(2,2*m+1,2,dtype=np.float64)))-np.log(4*np.pi))
    lam0=np.exp(ld(logc)+m*np.log(s)); lam1=np.sqrt(ld(2*m+3))*x*lam0
    vals=[lam0,lam1]; L=m+BLK*nb
    a=lambda l: np.sqrt(ld(4*l*l-1)/ld(l*l-m*m))
    for l in range(m+1,L): vals.append(a(l+1)*(x*vals[-1]-vals[-2]/a(l)))
    V=np.abs(np.array(vals))[:BLK*nb]
    alpha=np.ones(BLK*nb+2,dtype=ld)
    for par in (0,1):
        al=ld(1)
        for l in range(m+par,L,2):
            alpha[l-m]=al; al=ld(0.25)*a(l+1)*a(l+2)/al
    return V, V/alpha[:BLK*nb,None]
def chains2(m, th, nb, sign):
    # d^l_{m,-2 sign}: mu_{l+1} = (p' x + sign q') mu_l - mu_{l-1}
    x=np.cos(th); s=np.sin(th); l0=max(m,2)
    nrm=np.sqrt(ld(2*l0+1)/(4*np.pi))
    # K_m: d^m_{m,2} normalisation: sqrt((2m)!/((m+2)!(m-2)!)) ; kfac2 = K 2^-(m-2)
    from math import lgamma
    logK=0.5*(lgamma(2*m+1)-lgamma(m+3)-lgamma(m-1))
    base=np.exp(ld(logK)-ld(m-2)*np.log(ld(2))+ (m-2)*np.log(s))*nrm
    seed=base*(0.25*((1-x) if sign>0 else (1+x))**2)
    mu=[seed]; am1=ld(1); a0=ld(1); alpha=[a0]
    prev=np.zeros_like(seed)
    L=l0+BLK*nb
    for l in range(l0,L-1):
        k=ld(l); lp=ld(l+1); dm=ld(m); dn=ld(-2)
        den=k*np.sqrt((lp*lp-dm*dm)*(lp*lp-dn*dn))
        r1=np.sqrt((2*k+3)/(2*k+1))
        p=r1*(2*k+1)*k*lp/den; q=-r1*(2*k+1)*dm*dn/den
        a1=ld(1)
        if l>l0:
            r2=np.sqrt((2*k+3)/(2*k-1))
            r=r2*lp*np.sqrt((k*k-dm*dm)*(k*k-dn*dn))/den
            a1=r*am1
        pp=p*a0/a1; qq=q*a0/a1
        nxt=(pp*x+sign*qq)*mu[-1]-prev
        prev=mu[-1]; mu.append(nxt); am1=a0; a0=a1; alpha.append(a0)
    MU=np.abs(np.array(mu)); AL=np.abs(np.array(alpha))
    return MU*AL[:,None], MU
def need(m, spin, blk=None, nbmax=192, nth=3000, lmax=None, nside=None):
    global BLK, LMAX
    if blk: BLK=blk
    if lmax: LMAX=lmax
    nb=min((LMAX-max(m,2))//BLK+1, nbmax)
    if nside:
        th,s_,z_=healpix_rings(nside)
        th=th[ring_mlim(LMAX,spin,s_.astype(np.float64),z_.astype(np.float64))>=m]
    else:
        smin=min(1.0,m/(1.3*LMAX)+1e-4)
        th=np.linspace(np.arcsin(smin),np.pi/2,nth).astype(ld)
    if spin==0: sets=[chains0(m,th,nb)]
    else: sets=[chains2(m,th,nb,+1), chains2(m,th,nb,-1)]
    out=[]
    for b in range(nb):
        worst=0
        for V,MU in sets:
            ent=np.maximum(MU[BLK*b],MU[BLK*b+1]) if spin==0 else MU[BLK*b]
            inside=V[BLK*b:BLK*b+BLK].max(axis=0)
            ok=ent>0
            le=np.full(ent.shape,-1e9); le[ok]=np.log2(ent[ok].astype(ld)).astype(np.float64)
            li=np.full(ent.shape,-1e9); okk=inside>0; li[okk]=np.log2(inside[okk].astype(ld)).astype(np.float64)
            # E needed: smallest E such that all rings with le < -100-E have li <= -75
            bad=(li>-75)&(le<-100)
            if bad.any(): worst=max(worst, (-100-le[bad]).max())
        out.append(worst)
    return out
def kernel_margin(b, blk):
    """E_b as set_mode of k_legendre_duo (blocks of 32 l) / k_synth_duo (16 l) takes it"""
    if blk == 32:
        return 60 if b == 0 else (34 if b == 1 else max(30 - 2 * b, 8))
    return 26 if b == 0 else (12 if b == 1 else 8)


if __name__ == "__main__":
    NSIDE=int(sys.argv[3]) if len(sys.argv)>3 else None
    ms=[m for m in (30,300,1000,3000,4500,5800,7000,7800,9000,10500,11500,12088) if m<=LMAX-200]+[LMAX-200]
    print(f"# BLK={BLK} LMAX={LMAX} rings: "+(f"HEALPix nside {NSIDE} kept by mlim" if NSIDE else "3000-point grid"),flush=True)
    for spin in (0,2):
        for m in sorted(set(ms)):
            r=need(m,spin,nbmax=40,nside=NSIDE)
            print("spin",spin,"m",m,"E_b:",[round(float(v),1) for v in r[:10]],"| b=10..:",round(float(max(r[10:] or [0])),1), "| b=30..:",round(float(max(r[30:] or [0])),1),flush=True)
