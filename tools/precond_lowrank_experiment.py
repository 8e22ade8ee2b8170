"""CPU experiment (numpy / scipy.sparse, small grid): what a low-rank correction of the separable density-profile preconditioner
buys on the road-like clustered stream.

In the generalized eigenbasis of the production preconditioner (K_q = X_q D_q X_q^T, X_q^T diag(t_q) X_q = I) the system
matrix is D^-1 + X^T A X and the separable model replaces X^T A X by a I.  The correction tested here keeps the exact block
(D_S^-1 + E_SS)^-1, E = X^T A X, on the r generalized modes of largest prior eigenvalue and the diagonal model elsewhere (block
Jacobi in spectral coordinates; SPD).  Streaming protocol as in bench.py (init 5 %, batches of q = 4096 m / 125000 points, warm
starts, tolerance 1e-4), E refreshed every 4th step (profile refresh) -- the iteration counts right after a refresh are what
a per-step incremental E (projection + GEMM) would give.

Result at 20^3 (python tools/precond_lowrank_experiment.py 20 clustered):
    r =   0: 5.9 iterations per step   (the production preconditioner; 6.0 measured on the GPU at 50^3)
    r =  64: 4.0    (2-3 right after a refresh, 4 with a 1-3 steps old block)
    r = 128: 3.6        r = 256: 3.6        r = 512: 3.5
Why it was not built into wiski_pcg: a CG iteration of the fused fp32 path is 4 launches = 41 us.  The correction couples
spectral coefficients across i0 slabs, i.e. between the forward and backward halves of the slab kernel: + 2 launches (~10 us)
per iteration, + a projection, a GEMM and an r x r inverse per step (~60 us at r = 64).  6.0 -> ~3.5 iterations saves ~100 us
of a ~340 us clustered step and costs ~95 us.
"""
import numpy as np, scipy.sparse as sp, scipy.linalg as sla, sys, time
sys.path.insert(0,'/root/repo')
import bench, torch
from oracle import spec
g=int(sys.argv[1]) if len(sys.argv)>1 else 24; d=3
kind=sys.argv[2] if len(sys.argv)>2 else 'clustered'
m=g**d
n0=int(3.5*m*0.05*20/20); # ~ init 5% of a stream ~ 3.5 pts per node overall
N=int(3.5*m); n0=int(0.05*N); q=max(64,int(4096*m/125000))
g0,h,gg=spec.make_grid([[-1.1,1.1]]*d,g)
X,y=bench.synth_stream(N,d,0,torch.device('cpu'),torch.float64,kind); X=X.numpy(); y=y.numpy()[:,0]
def Wsp(Xp):
    n=len(Xp); rows=[];cols=[];vals=[]
    Wd=[spec.interp_1d_dense(Xp[:,qq],g0[qq],h[qq],g) for qq in range(d)]
    W=sp.csr_matrix(Wd[0]);
    # build via kron per row: use nonzeros
    nz=[ [np.nonzero(Wd[qq][p])[0] for p in range(n)] for qq in range(d)]
    for p in range(n):
        i0,i1,i2=nz[0][p],nz[1][p],nz[2][p]
        idx=(i0[:,None,None]*g*g+i1[None,:,None]*g+i2[None,None,:]).ravel()
        v=(Wd[0][p,i0][:,None,None]*Wd[1][p,i1][None,:,None]*Wd[2][p,i2][None,None,:]).ravel()
        rows.append(np.full(len(idx),p)); cols.append(idx); vals.append(v)
    return sp.csr_matrix((np.concatenate(vals),(np.concatenate(rows),np.concatenate(cols))),shape=(n,m))
ell,osc,s2=0.6931,0.6931,0.6931
cols_=spec.toeplitz_columns('rbf',h,gg,ell,osc)
Ks=[sla.toeplitz(c) for c in cols_]
def setup(A):
    cnt=np.asarray(A.sum(1)).ravel().reshape(g,g,g)
    Xq=[];Dq=[];norm=1.0
    for qq in range(d):
        marg=cnt.sum(axis=tuple(r for r in range(d) if r!=qq)); t=np.clip(marg/marg.max(),1e-2,None); norm*=t.sum()
        rt=np.sqrt(t); w,U=np.linalg.eigh(rt[:,None]*Ks[qq]*rt[None,:]); Xq.append(U/rt[:,None]); Dq.append(np.clip(w,0,None))
    a=cnt.sum()/norm
    D=np.einsum('i,j,k->ijk',*Dq).ravel()/s2
    return Xq,D,a
def fwd(Xq,v): return np.einsum('ai,bj,ck,abc->ijk',Xq[0],Xq[1],Xq[2],v.reshape(g,g,g)).ravel()   # X^T v
def bwd(Xq,c): return np.einsum('ia,jb,kc,abc->ijk',Xq[0],Xq[1],Xq[2],c.reshape(g,g,g)).ravel()   # X c
def kmv(v):
    t=v.reshape(g,g,g); t=np.einsum('ia,ajk->ijk',Ks[0],t); t=np.einsum('jb,ibk->ijk',Ks[1],t); t=np.einsum('kc,ijc->ijk',Ks[2],t); return t.ravel()/s2
def solve(A,b,z0,Xq,D,a,corr=None,tol=1e-4,maxit=100):
    # iterate in z: u = Kt z ; system (Kt^-1 + A) u = b  <=> z + A Kt z = b ; use PCG in u-form with M^-1 = P
    def Hu(u,z): return z + A@u
    # standard PCG on u with H = Kt^-1 + A needs Kt^-1: use spectral coords instead: work in c = X^-1 u? simpler: dense CG in c-coordinates
    # c-coordinates: Hc = D^-1 + X^T A X
    Dinv=1.0/np.maximum(D,1e-14*D.max())
    def Hc(c): return Dinv*c + fwd(Xq,A@bwd(Xq,c))
    def P(r):
        out=r/(Dinv+a)
        if corr is not None:
            S,Minv=corr
            out[S]=Minv@r[S]
        return out
    bc=fwd(Xq,b); c=z0.copy(); r=bc-Hc(c); zz=P(r); p=zz.copy(); rz=r@zz; r0=np.linalg.norm(bc)
    for it in range(1,maxit+1):
        Hp=Hc(p); al=rz/(p@Hp); c+=al*p; r-=al*Hp
        if np.linalg.norm(r)<=tol*r0: return it,c
        zz=P(r); rz2=r@zz; p=zz+(rz2/rz)*p; rz=rz2
    return maxit,c
t0=time.time()
W0=Wsp(X[:n0]); A=(W0.T@W0).tocsr(); b=W0.T@y[:n0]
Xq,D,a=setup(A)
it,c=solve(A,b,np.zeros(m),Xq,D,a); print('cold iterations',it, 'setup time',time.time()-t0)
order=np.argsort(-D)
res={}
for r in (0,64,128,256,512):
    A_=A.copy(); b_=b.copy(); c_=c.copy(); its=[]
    Xq_,D_,a_=Xq,D,a
    pos=n0
    for step in range(12):
        Wn=Wsp(X[pos:pos+q]); A_=(A_+Wn.T@Wn).tocsr(); b_=b_+Wn.T@y[pos:pos+q]; pos+=q
        if step%4==0:
            Xq_,D_,a_=setup(A_)   # refresh profile occasionally (production: on doubling)
            corr=None
            if r:
                S=np.argsort(-D_)[:r]
                B=np.stack([bwd(Xq_,np.eye(m)[j]) for j in S],1)    # m x r generalized eigvecs
                E=B.T@(A_@B)
                Minv=np.linalg.inv(np.diag(1.0/D_[S])+E)
                corr=(S,Minv)
            # re-express warm start in new coords: c = X^-1 u ; here approximate by re-solving from old u
            u_old=bwd(Xq if step==0 else Xq_prev,c_)
            # X^-1 u = X^T T u  (since X^T T X = I)
            Tt=[np.linalg.inv(xq).T for xq in Xq_]  # X^-T ... need X^-1 = (X^T T) ; compute directly
            Xinv=[np.linalg.inv(xq) for xq in Xq_]
            c_=np.einsum('ia,jb,kc,abc->ijk',Xinv[0],Xinv[1],Xinv[2],u_old.reshape(g,g,g)).ravel()
            Xq_prev=Xq_
        it,c_=solve(A_,b_,c_,Xq_,D_,a_,corr)
        its.append(it)
    res[r]=its
    print('r',r,'iters per step',its, 'mean',np.mean(its[2:]))
