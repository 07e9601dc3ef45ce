"""Round 4 experiment (rejected): numpy emulation of the block + state-scan form of scipy.signal.filtfilt(b, a) on the reference filter chain -- see
r04_filtfilt_scan.hip.txt.  Prints the spectral radius of the direct-form state transition A, the largest entry of A^79, and the error against scipy: 1e126."""
import numpy as np
from scipy import signal
import sys
sys.path.insert(0,'/root/repo')
from ecg_byte_amd import preprocess_utils as pp
filters = pp.design_filters(500)
rng=np.random.default_rng(0)
n=5000
x=rng.standard_normal(n)+0.5
emax=27
B=(n+2*emax+63)//64
NG=64*B
def run(x, filters):
    buf=np.zeros(NG); buf[emax:emax+n]=x
    for (b,a) in filters:
        b=np.asarray(b)/a[0]; a=np.asarray(a)/a[0]
        nb=len(b); ns=nb-1; e=3*nb
        zi=signal.lfilter_zi(b,a)
        A=np.zeros((ns,ns))
        for r in range(ns):
            A[r,0]=-a[r+1]
            if r+1<ns: A[r,r+1]+=1
        M=np.linalg.matrix_power(A,B)
        print('filter nb',nb,'spectral radius A',max(abs(np.linalg.eigvals(A))),'|M|',np.abs(M).max())
        g0=emax-e; g1=emax+n+e
        xf=buf[emax]; xl=buf[emax+n-1]
        new=buf.copy()
        for k in range(emax):
            new[k]= 2*xf-buf[emax+e-(k-g0)] if k>=g0 else 0.0
        for k in range(NG-(emax+n)):
            new[emax+n+k]= 2*xl-buf[emax+n-2-k] if k<e else 0.0
        buf=new
        for rev in (False,True):
            gstart = g1-1 if rev else g0
            linj=gstart//B; iinj=(B-1-gstart%B) if rev else gstart%B
            S=np.zeros((64,ns))
            for l in range(64):
                z=np.zeros(ns)
                for i in range(B):
                    g = l*B+B-1-i if rev else l*B+i
                    if l==linj and i==iinj: z=zi*buf[g]
                    xi=buf[g]; y=z[0]+b[0]*xi
                    for k in range(ns-1): z[k]=z[k+1]+xi*b[k+1]-y*a[k+1]
                    z[ns-1]=xi*b[nb-1]-y*a[nb-1]
                    buf[g]=y
                S[l]=z
            # scan
            I=S.copy()
            P=M.copy()
            for j in range(6):
                d=1<<j
                newI=I.copy()
                for l in range(64):
                    src = l+d if rev else l-d
                    if 0<=src<64: newI[l]=I[l]+P@I[src]
                I=newI; P=P@P
            for l in range(64):
                src = l+1 if rev else l-1
                s = I[src].copy() if 0<=src<64 else np.zeros(ns)
                for i in range(B):
                    g = l*B+B-1-i if rev else l*B+i
                    yh=s[0]; buf[g]+=yh
                    for k in range(ns-1): s[k]=s[k+1]-yh*a[k+1]
                    s[ns-1]=-(yh*a[nb-1])
            if not rev:
                buf[g1:]=0.0
    return buf[emax:emax+n]
got=run(x,filters)
want=x
for b,a in filters: want=signal.filtfilt(b,a,want)
print('max err',np.abs(got-want).max(), 'range',np.abs(want).max())
