"""CPU emulation behind DESIGN.md 6.1: error of a K = 128 dot product against float64, in units of sum |terms|, for the product schemes of
the dense tail -- bf16x6 (three bf16 planes, six products), fp16x3 (two scaled fp16 planes, three products), fp32 accumulated in chunks
of 16 (an MFMA-like order) and numpy's sgemm.   python tools/fp16x3_emulation.py"""
import numpy as np
rng=np.random.default_rng(0)
n,K,N=4096,128,128
x=rng.standard_normal((n,K)).astype(np.float32)
w=(rng.uniform(-1,1,(N,K))/np.sqrt(K)).astype(np.float32)
ref=x.astype(np.float64)@w.astype(np.float64).T
den=(np.abs(x).astype(np.float64)@np.abs(w).astype(np.float64).T)
def bf16(a):
    u=a.view(np.uint32).astype(np.uint64)
    r=((u+0x7fff+((u>>16)&1))>>16)<<16
    return r.astype(np.uint32).view(np.float32)
def split3(a):
    h=bf16(a); r=a-h; m=bf16(r); r2=r-m; l=bf16(r2); return h,m,l
def mm(a,b): return (a.astype(np.float64)@b.astype(np.float64).T)  # exact products, fp64 accumulate (idealised accum)
def mm32(a,b,chunk=16):
    # fp32 accumulation in chunks of 16 along K roughly like MFMA
    acc=np.zeros((a.shape[0],b.shape[0]),np.float32)
    for k in range(0,a.shape[1],chunk):
        acc=(acc.astype(np.float64)+a[:,k:k+chunk].astype(np.float64)@b[:,k:k+chunk].astype(np.float64).T).astype(np.float32)
    return acc
xh,xm,xl=split3(x); wh,wm,wl=split3(w)
y6=np.zeros((n,N),np.float32)
for a,b in [(xl,wh),(xh,wl),(xm,wm),(xm,wh),(xh,wm),(xh,wh)]:
    y6=(y6.astype(np.float64)+mm32(a,b).astype(np.float64)).astype(np.float32)
def f16split(a,scale):
    s=a*scale
    h=s.astype(np.float16); r=s-h.astype(np.float32); l=r.astype(np.float16); return h.astype(np.float32),l.astype(np.float32)
sx=2.0**(12-np.ceil(np.log2(np.abs(x).max(1,keepdims=True)))).astype(np.float32)
sw=np.float32(2.0**(12-np.ceil(np.log2(np.abs(w).max()))))
xh2,xl2=f16split(x,sx); wh2,wl2=f16split(w,sw)
y3=np.zeros((n,N),np.float32)
for a,b in [(xl2,wh2),(xh2,wl2),(xh2,wh2)]:
    y3=(y3.astype(np.float64)+mm32(a,b).astype(np.float64)).astype(np.float32)
y3=(y3/sx/sw).astype(np.float32)
y32=mm32(x,w)  # fp32 "MFMA-like"
yt=x@w.T
for name,y in [("bf16x6",y6),("fp16x3",y3),("fp32 chunked",y32),("numpy sgemm",yt)]:
    e=(y.astype(np.float64)-ref)/den
    print(name,"rms",np.sqrt((e**2).mean()),"max",np.abs(e).max())
