"""How many taps of a run of consecutive points leave the per-plane pass's window (csrc/hexplane.hip k_hexplane_bwd_plane), and how many cell
rows a run flushes, for plane orders along a Z-order curve and along a Hilbert curve: 2 M uniform points, runs of 256 / 512, a 16 x 16
window anchored at the run's smallest tap.     python3 tests/analysis/plane_order_sim.py"""
import numpy as np
N=2_000_000
rng=np.random.default_rng(0)
q=rng.random((N,2)).astype(np.float32)*2-1
def spread(v):
    v=v.astype(np.uint64)
    v=(v|(v<<8))&0x00FF00FF
    v=(v|(v<<4))&0x0F0F0F0F
    v=(v|(v<<2))&0x33333333
    return (v|(v<<1))&0x55555555
def hilbert(x,y,bits):
    # xy2d
    x=x.copy().astype(np.int64); y=y.copy().astype(np.int64)
    d=np.zeros_like(x)
    s=1<<(bits-1)
    while s>0:
        rx=((x&s)>0).astype(np.int64); ry=((y&s)>0).astype(np.int64)
        d+=s*s*((3*rx)^ry)
        # rotate
        m=(ry==0)
        f=m&(rx==1)
        x=np.where(f,s-1-x,x); y=np.where(f,s-1-y,y)
        x2=np.where(m,y,x); y2=np.where(m,x,y)
        x,y=x2,y2
        s>>=1
    return d
def sim(order,W,P=256,PW=16):
    qs=q[order]
    nb=N//P
    ix=np.clip((qs[:nb*P]+1)*0.5*(W-1),0,W-1)
    i0=np.floor(ix).astype(np.int64)
    x=i0[:,0].reshape(nb,P); y=i0[:,1].reshape(nb,P)
    ax=x.min(1,keepdims=True); ay=y.min(1,keepdims=True)
    inside=((x-ax+1)<PW)&((y-ay+1)<PW)
    cell=np.where(inside,(y-ay)*PW+(x-ax),PW*PW+PW+5)
    occ=np.zeros((nb,PW*PW+2*PW+8),bool)
    for d in (0,1,PW,PW+1): np.put_along_axis(occ,np.where(inside,cell+d,PW*PW+2*PW+7),True,axis=1)
    return (~inside).mean(), occ[:,:PW*PW].sum()/nb
g=np.clip(((q+1)*0.5*1023).astype(np.int64),0,1023)
zo=np.argsort(spread(g[:,0])|(spread(g[:,1])<<1),kind='stable')
ho=np.argsort(hilbert(g[:,0],g[:,1],10),kind='stable')
for W in (128,256,512):
    for P in (256,512):
        print(W,P,"zorder miss %.4f cells %.1f"%sim(zo,W,P),"  hilbert miss %.4f cells %.1f"%sim(ho,W,P))
