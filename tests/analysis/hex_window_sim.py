"""Where the atomic rows of the HexPlane backward main kernel come from (csrc/hexplane.hip k_hexplane_bwd_agg): 2 M uniform points in 3-D Morton
order, chunks of P points, 10 x 10 spatial windows and 48-cell time windows anchored at the chunk's smallest tap.  Per scale: cell rows flushed
by the time-plane marginals (x 2 time rows), points whose time tap leaves the window, cell rows flushed by the spatial windows, (point, plane)
pairs whose spatial tap leaves the window (each four partly filled atomic instructions).  DESIGN.md section 8, round 4.
    python3 tests/analysis/hex_window_sim.py"""
import numpy as np
N=2_000_000
rng=np.random.default_rng(0)
q=rng.random((N,3)).astype(np.float32)*2-1
def spread(v):
    v=v.astype(np.uint64)
    v=(v|(v<<16))&0x030000FF
    v=(v|(v<<8))&0x0300F00F
    v=(v|(v<<4))&0x030C30C3
    v=(v|(v<<2))&0x09249249
    return v
g=np.clip(((q+1)*0.5*1023).astype(np.int64),0,1023)
code=spread(g[:,0])|(spread(g[:,1])<<1)|(spread(g[:,2])<<2)
order=np.argsort(code,kind='stable')
qs=q[order]
def taps(v,W):
    ix=np.clip((v+1)*0.5*(W-1),0,W-1)
    i0=np.floor(ix).astype(np.int64); i1=np.minimum(i0+1,W-1)
    return i0,i1
def sim(P, SW=10, TW=48):
    nb=N//P
    res={}
    for W in (64,128,256,512):
        i0=[taps(qs[:nb*P,k],W)[0].reshape(nb,P) for k in range(3)]
        anc=[a.min(axis=1,keepdims=True) for a in i0]
        span=[(a.max(axis=1)+1-an[:,0]+1) for a,an in zip(i0,anc)]   # cells incl +1 tap
        # time marginal rows per chunk: distinct cells among i0 and i0+1 within window TW
        trows=0; tmiss=0
        for k in range(3):
            rel=i0[k]-anc[k]
            inside=(rel+1)<TW
            tmiss+= (~inside).sum()
            # distinct cells
            occ=np.zeros((nb,TW+2),bool)
            r=np.clip(rel,0,TW); 
            np.put_along_axis(occ,r,True,axis=1); np.put_along_axis(occ,np.clip(r+1,0,TW+1),True,axis=1)
            trows+=occ[:,:TW].sum()
        # spatial planes (0,1),(0,2),(1,2)
        srows=0; smiss=0
        for (ax,ay) in ((0,1),(0,2),(1,2)):
            rx=i0[ax]-anc[ax]; ry=i0[ay]-anc[ay]
            inside=((rx+1)<SW)&((ry+1)<SW)
            smiss+=(~inside).sum()
            cell=np.where(inside,ry*SW+rx,SW*SW+5)
            occ=np.zeros((nb,SW*SW+SW+8),bool)
            for d in (0,1,SW,SW+1):
                np.put_along_axis(occ,np.where(inside,cell+d,SW*SW+SW+7),True,axis=1)
            srows+=occ[:,:SW*SW].sum()
        res[W]=dict(time_rows=int(trows)*2, time_miss_pts=int(tmiss), spat_rows=int(srows), spat_miss_pts=int(smiss), span=[float(s.mean()) for s in span])
    return res
for P in (256,512,1024):
    r=sim(P)
    print("P",P)
    for W,v in r.items(): print("  ",W,v)
