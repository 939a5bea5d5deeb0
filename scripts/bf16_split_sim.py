"""Design aid (numpy, no GPU): accuracy of a windowed DFT bin computed as bf16 x bf16 products with float32 accumulation, the samples and the
window x twiddle coefficients each split into three bf16 terms -- how many of the nine cross products the f32 sample format would need on the matrix
cores.  Prints the error relative to the RMS of the exact bin for a strong carrier's bin and for two noise-only bins."""
import numpy as np
rng=np.random.default_rng(1)
N=512
def bf16(x):  # round to nearest even to bf16, return as float32
    u=x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r=((u + 0x7fff + ((u>>16)&1)) & 0xffff0000).astype(np.uint32)
    return r.view(np.float32)
def split3(x):
    a=bf16(x); r=(x-a).astype(np.float32); b=bf16(r); r2=(r-b).astype(np.float32); c=bf16(r2); return a,b,c
n=np.arange(N)
# 7-term-like window: use a Blackman-Harris stand-in
a=[0.27105140069342,0.43329793923448,0.21812299954311,0.06592544638803,0.01081174209837,0.00077658482522,0.00001388721735]
w=sum(((-1)**k)*a[k]*np.cos(2*np.pi*k*n/N) for k in range(7)).astype(np.float32)
bins=[37, 200, 411]
H=2000
x=(0.02*rng.standard_normal((H,N))+0.3*np.cos(2*np.pi*(37.3)*n/N+rng.uniform(0,6.28,(H,1)))).astype(np.float32)
xs=split3(x)
for k in bins:
    cr=(w*np.cos(2*np.pi*k*n/N)).astype(np.float32); ci=(-w*np.sin(2*np.pi*k*n/N)).astype(np.float32)
    # reference: f32 product sample*window then exact DFT in f64
    xw=(x*w).astype(np.float32).astype(np.float64)
    ref=xw@np.cos(2*np.pi*k*n/N) + 0j - 1j*(xw@np.sin(2*np.pi*k*n/N))
    crs=split3(cr); cis=split3(ci)
    for terms,label in (([(0,0)],'1'),([(0,0),(0,1),(1,0)],'3'),([(0,0),(0,1),(1,0),(1,1)],'4'),([(0,0),(0,1),(1,0),(1,1),(0,2),(2,0)],'6'),):
        accr=np.zeros(H,np.float32); acci=np.zeros(H,np.float32)
        # accumulate in f32: emulate MFMA k=16 chunks: chunk sums exact-ish (use f64 within 16) then f32 add
        for (i,j) in terms:
            for c0 in range(0,N,16):
                pr=(xs[i][:,c0:c0+16].astype(np.float64)*crs[j][c0:c0+16].astype(np.float64)).sum(axis=1)
                pi=(xs[i][:,c0:c0+16].astype(np.float64)*cis[j][c0:c0+16].astype(np.float64)).sum(axis=1)
                accr=(accr+pr.astype(np.float32)).astype(np.float32); acci=(acci+pi.astype(np.float32)).astype(np.float32)
        got=accr.astype(np.float64)+1j*acci.astype(np.float64)
        err=np.sqrt(np.mean(np.abs(got-ref)**2))/np.sqrt(np.mean(np.abs(ref)**2))
        print('bin',k,'terms',label,'rel rms err %.2e'%err)
