#!/usr/bin/env python3
"""Wave-execution model of the late-node loop of the threshold kernels (csrc/kernels.hpp, msh_grid) on the cfg3b batch.

Question: the counters show 74 % active lanes in cloudy_jit_sorted_n2p3_f64; would an LDS
work queue of (parcel, node) items, another ranking key or a larger ranking group fill the waves?  This script answers it
WITHOUT a GPU: it recomputes, for every parcel of the synthetic cfg3b batch, where the closed-form early group ends (J),
which algorithm each late node takes (P == 1 shortcut / power series / continued fraction, device_math.hpp) and how many
4-term convergence groups it needs -- the same recurrences the kernel runs -- and then charges a wave, per loop
iteration, the maximum over its lanes of each branch (a wave executes every branch any lane takes, for as many
groups as its slowest lane).  Costs are in VALU instructions (150 per node + 3.5 per series term + 7 per continued
-fraction step, from the ISA of the plan-time compiled kernel).

`python tools/lane_sim.py > profiles/r03_lane_simulation.txt` (CPU only: this is an analysis tool; it uses the test
oracle to invert the closures and never runs in the product path).
"""
import sys, numpy as np, math
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import cloudy_oracle as O
from scipy.special import gammaln
n=512*400
wl=bench.make_workload("cfg3b", n, seed=7)
op=bench.oracle_params("cfg3b")
prm=O.update_dist_batch(op, wl["mom"])
nn,th,k=prm[0],prm[1],prm[2]
xt=5e-10/1e-9
M=5
# node table
x_lb=min(1e-5,1e-5*xt); nb=int(math.floor(15*math.log10(xt/x_lb))); x_min=math.log(x_lb); dxl=(math.log(xt)-x_min)/nb
lx=x_min+np.arange(nb)*dxl; xs=np.exp(lx)
valid=nn>0
a_top=k+M-1
x_early=np.minimum(3.0*th, 1.5*xt/np.maximum(a_top-1,3.0))
J=np.where(x_early>=xs[0], np.minimum(np.floor((np.log(x_early)-lx[0])/dxl)+1, nb-4), 0).astype(int)
J[J<4]=0
J[~valid]=nb   # no pass
L=nb-J
print("nb",nb,"mean late nodes",L[valid].mean(), "max",L.max())
# per node iteration counts
z=(xt-xs[None,:])/th[:,None]           # [n, nb]
a=a_top[:,None]
lg_top=gammaln(a_top+1)[:,None]
lnth=np.log(th)[:,None]
E0=np.exp(a*(np.log(xt-xs)[None,:]-lnth)-z-lg_top)
series=z<=a+1
short=(~series)&(a*E0<1e-18)
cf=(~series)&(~short)
# series groups: q=a/z; loop groups of 4: q+=1/z; Nn=Nn*q+1 until Nn>=1e17
def series_groups(a,z):
    invz=1/z; q=a*invz; Nn=np.ones_like(z); g=np.zeros(z.shape,int); done=np.zeros(z.shape,bool)
    for it in range(100):
        for u in range(4):
            q=q+invz; Nn=np.where(done,Nn,Nn*q+1)
        g=np.where(done,g,g+1)
        done=done|~(Nn<1e17)
        if done.all(): break
    return g
def cf_groups(a,z):
    b=z+1-a; Ap=np.zeros_like(z); Bp=np.ones_like(z); Ac=np.ones_like(z); Bc=b.copy(); an=a-1.0; c=a-1.0
    g=np.zeros(z.shape,int); done=np.zeros(z.shape,bool)
    an=np.broadcast_to(an,z.shape).copy(); c=np.broadcast_to(c,z.shape).copy()
    for it in range(100):
        for u in range(4):
            b=b+2; An=b*Ac+an*Ap; Bn=b*Bc+an*Bp; Ap,Bp,Ac,Bc=Ac,Bc,An,Bn; c=c-2; an=an+c
        g=np.where(done,g,g+1)
        lhs=np.abs(Ac*Bp-Ap*Bc)
        done=done|~(lhs>1e-16*np.abs(Ac*Bp))
        big=np.abs(Bc)>1e150
        for arr in (Ap,Bp,Ac,Bc): arr[big]*=1e-150
        if done.all(): break
    return g
with np.errstate(all='ignore'):
    gs=np.where(series, series_groups(np.broadcast_to(a,z.shape), np.where(series,z,1.0)), 0)
    gc=np.where(cf, cf_groups(np.broadcast_to(a,z.shape), np.where(cf,z,a+2)), 0)
late=np.arange(nb)[None,:]>=J[:,None]
print("late node mix: short %.2f series %.2f cf %.2f ; mean series groups %.1f cf groups %.1f"%(short[late].mean(),series[late].mean(),cf[late].mean(),gs[late&series].mean(),gc[late&cf].mean()))
BASE=150.0; CS=3.5*4; CC=7*4; CSH=10.0
own=np.where(late, BASE+gs*CS+gc*CC+short*CSH, 0.0)
z0=xt/th
def bucket(key):
    kf=np.maximum(key.astype(np.float32),0)
    code=(kf.view(np.uint32)>>20).astype(int)-(127-8)*8
    return np.clip(code,0,510)
rng=np.random.default_rng(0)
def simulate(order_key_fn, BS=512, label=""):
    tot_cost=0.0; tot_useful=0.0
    for w0 in range(0,n,BS):
        idx=np.arange(w0,w0+BS)
        key=order_key_fn(idx)
        perm=idx[np.lexsort((rng.random(BS), key))]
        for wv in range(0,BS,64):
            lanes=perm[wv:wv+64]
            lt=late[lanes]          # [64, nb]
            act=lt.any(axis=0)      # iterations where any lane active (by node index)
            ser=(series[lanes]&lt); cfm=(cf[lanes]&lt); sh=(short[lanes]&lt)
            cost=act*BASE + ser.any(axis=0)*CS*np.where(ser,gs[lanes],0).max(axis=0) + cfm.any(axis=0)*CC*np.where(cfm,gc[lanes],0).max(axis=0)+sh.any(axis=0)*CSH
            tot_cost+=64*cost.sum(); tot_useful+=own[lanes].sum()
    print(f"{label:40s} lane-utilisation of the late loop {tot_useful/tot_cost:.3f}  cost/parcel {tot_cost/n:.0f}")
    return tot_cost/n
vb=np.where(valid, bucket(z0), 511)
simulate(lambda idx: vb[idx], label="current: rank by z0 bucket")
simulate(lambda idx: np.where(valid[idx], z0[idx], 1e30), label="exact z0 sort")
simulate(lambda idx: -L[idx], label="rank by L (late count)")
simulate(lambda idx: np.where(valid[idx], bucket(z0/(a_top+1))[idx] if False else bucket((z0/(a_top+1)))[idx], 511), label="rank by ratio bucket")
simulate(lambda idx: np.lexsort((0*idx,))*0+(-L[idx]*1000+vb[idx]), label="L primary, z0 secondary")
simulate(lambda idx: vb[idx]*1000-L[idx], label="z0 bucket primary, L secondary")
# ideal: perfect balance = useful / all
print("ideal cost/parcel (perfect packing)", own.sum()/n)
# per-round compaction: at node jj, active parcels in rank order are compacted into full waves
def simulate_compact(BS=512):
    tot_cost=0.0
    for w0 in range(0,n,BS):
        idx=np.arange(w0,w0+BS)
        perm=idx[np.lexsort((rng.random(BS), vb[idx]))]
        for jj in range(nb):
            actl=perm[late[perm,jj]]
            for wv in range(0,len(actl),64):
                lanes=actl[wv:wv+64]
                ser=series[lanes,jj]; cfm=cf[lanes,jj]; sh=short[lanes,jj]
                cost=BASE+(CS*gs[lanes,jj][ser].max() if ser.any() else 0)+(CC*gc[lanes,jj][cfm].max() if cfm.any() else 0)+(CSH if sh.any() else 0)
                tot_cost+=64*cost
    print(f"per-round compaction (no overhead) cost/parcel {tot_cost/n:.0f}  utilisation {own.sum()/tot_cost:.3f}")
simulate_compact()

print("---- phase split")
nser=(series&late).sum(axis=1); ncf=(cf&late).sum(axis=1); nsh=(short&late).sum(axis=1)
print("mean n_ser %.1f n_cf %.1f n_short %.1f"%(nser[valid].mean(),ncf[valid].mean(),nsh[valid].mean()))
def padded(mask, vals):
    # per lane: list of vals at masked nodes in order from last node down, padded with -1
    out=-np.ones((mask.shape[0], nb))
    for i in range(mask.shape[0]):
        v=vals[i][::-1][mask[i][::-1]]
        out[i,:len(v)]=v
    return out
def simulate_phase(order_key_fn, BS=512, label="", extra_cf=12.0):
    tot=0.0
    for w0 in range(0,n,BS):
        idx=np.arange(w0,w0+BS)
        perm=idx[np.lexsort((rng.random(BS), order_key_fn(idx)))]
        for wv in range(0,BS,64):
            lanes=perm[wv:wv+64]
            ps=padded(series[lanes]&late[lanes], gs[lanes]); pc=padded(cf[lanes]&late[lanes], gc[lanes])
            ms=ps.max(axis=0); mc=pc.max(axis=0)
            cost=((ms>=0)*(BASE+CS*np.maximum(ms,0))).sum()+((mc>=0)*(BASE+extra_cf+CC*np.maximum(mc,0))).sum()+nsh[lanes].max()*(BASE*0.5)
            tot+=64*cost
    own2=own.sum()-((short&late).sum()*(BASE*0.5+CSH))+0
    print(f"{label:40s} cost/parcel {tot/n:.0f}")
simulate_phase(lambda idx: vb[idx], label="phase split, rank by z0 bucket")
simulate_phase(lambda idx: -nser[idx]*100-ncf[idx], label="phase split, rank by (n_ser, n_cf)")
simulate_phase(lambda idx: vb[idx]*10000-nser[idx]*100-ncf[idx], label="phase split, z0 bucket then counts")
# ideal class-sorted queue
items_cost=own[late]
srt=np.sort(items_cost)[::-1]
pad=(-len(srt))%64
w=np.concatenate([srt,np.zeros(pad)]).reshape(-1,64)
print("class-sorted global queue (ideal) cost/parcel", 64*w.max(axis=1).sum()/n)
print("---- other keys")
tot_own=own.sum(axis=1)
simulate(lambda idx: np.where(valid[idx], -tot_own[idx], 1e30), label="rank by own total late cost")
simulate(lambda idx: np.where(valid[idx], k[idx], 1e30), label="rank by k")
kb=np.clip((k*2).astype(int),0,40)
simulate(lambda idx: np.where(valid[idx], kb[idx]*100000+z0[idx], 1e30), label="k (0.5 bins) then z0")
zb=np.clip((np.log2(np.maximum(z0,1e-30))*2).astype(int)+60,0,200)
simulate(lambda idx: np.where(valid[idx], zb[idx]*1000+k[idx], 1e30), label="z0 (sqrt2 bins) then k")
simulate(lambda idx: np.where(valid[idx], zb[idx]*1000-L[idx], 1e30), label="z0 (sqrt2 bins) then L")
for BS in (1024, 2048, 8192):
    tot_cost=0.0; tot_useful=0.0
    for w0 in range(0,n-BS+1,BS):
        idx=np.arange(w0,w0+BS); perm=idx[np.lexsort((rng.random(BS), vb[idx]))]
        for wv in range(0,BS,64):
            lanes=perm[wv:wv+64]; lt=late[lanes]; act=lt.any(axis=0)
            ser=(series[lanes]&lt); cfm=(cf[lanes]&lt); sh=(short[lanes]&lt)
            cost=act*BASE + ser.any(axis=0)*CS*np.where(ser,gs[lanes],0).max(axis=0) + cfm.any(axis=0)*CC*np.where(cfm,gc[lanes],0).max(axis=0)+sh.any(axis=0)*CSH
            tot_cost+=64*cost.sum(); tot_useful+=own[lanes].sum()
    print("sort group size",BS,"utilisation",tot_useful/tot_cost,"cost/parcel",tot_cost/(n//BS*BS))

# ---- round 5: a per-node algorithm whose cost does not depend on (z, a)? ------------------------------
# "Fixed length" = every lane runs the same number of terms, so the wave pays no divergence -- but the number must cover the
# worst node.  The counts below are the kernel's own recurrences stopped at the RELAXED tolerance (1e-10, the opt-in dtype's
# neighbourhood), per late node of the batch: the series needs ceil(groups) x 4 terms where it is used today (z <= a + 1) and
# far more beyond (it converges for every z, with ~z + 10 sqrt(z) terms); the continued fraction is the mirror image.
print("---- fixed-length schemes (round 5)")
def series_groups_tol(a, z, big):
    invz = 1 / z; q = a * invz; Nn = np.ones_like(z); g = np.zeros(z.shape, int); done = np.zeros(z.shape, bool)
    for it in range(400):
        for u in range(4):
            q = q + invz; Nn = np.where(done, Nn, Nn * q + 1)
        g = np.where(done, g, g + 1)
        done = done | ~(Nn < big)
        if done.all(): break
    return g
with np.errstate(all='ignore'):
    A2 = np.broadcast_to(a, z.shape)
    zz = np.where(late & ~short, z, 1.0)
    gs_all = series_groups_tol(A2, zz, 1e11)            # the series on EVERY late node that is not the P == 1 shortcut
    gs_own = np.where(series, series_groups_tol(A2, np.where(series, z, 1.0), 1e11), 0)
m_all = late & ~short
for name, g in (("series where it is used today (z <= a + 1), tolerance 1e-10", gs_own[late & series]),
                ("series on every late node (any z), tolerance 1e-10", gs_all[m_all])):
    print(f"{name}: mean {4 * g.mean():.0f} terms, 99th percentile {4 * np.percentile(g, 99):.0f}, 99.9th {4 * np.percentile(g, 99.9):.0f}, max {4 * g.max()}")
Lm = L[valid].mean()
for T in (40, 60, 100):
    frac_cov = (4 * gs_all[m_all] <= T).mean()
    print(f"fixed {T}-term series on every late node: {BASE + 3.5 * T:.0f} instructions per node x {Lm:.1f} late nodes = {Lm * (BASE + 3.5 * T):.0f} per parcel "
          f"at utilisation 1.0 (today {5374} at 0.58; the ideal queue 3108); covers {100 * frac_cov:.1f} % of the late nodes at 1e-10")
