import sys; sys.path.insert(0,'/root/repo')
import __graft_entry__ as g
g.smoke()
import numpy as np, torch
from oracle import dataspace
from online_gp_amd.models import OnlineSKIRegression, Identity, FixedNoiseOnlineSKIGP
# 2-output batch, heteroscedastic noise
rng=np.random.default_rng(1)
dev='cuda'
X=rng.uniform(0,1,(30,1)); Y=np.stack([np.sin(3*X[:,0]),np.sin(5*X[:,0])],1); N=0.01*Y**2+0.01
Xt,Yt,Nt=[torch.as_tensor(a,device=dev) for a in (X,Y,N)]
m=FixedNoiseOnlineSKIGP(Xt[:20],Yt[:20],Nt[:20],grid_bounds=torch.tensor([[0.,1.]]),grid_size=10)
m.train(); d=m(None); print('train',d.mean.shape,d.covariance_matrix.shape)
print({k:(v.shape) for k,v in m._kernel_cache.items()})
m.eval(); nm=m.condition_on_observations(Xt[20:],Yt[20:],Nt[20:],inplace=False)
m.condition_on_observations(Xt[20:],Yt[20:],Nt[20:],inplace=True)
Xs=torch.as_tensor(rng.uniform(0,1,(5,1)),device=dev)
for mm in (m,nm):
    d=mm(Xs); print(d.mean.shape,d.covariance_matrix.shape, mm.num_data)
    for o in range(2):
        O=dataspace.DataSpaceGP([[0.,1.]],10,sigma2=1.0).fit(X,Y[:,o],N[:,o]); mo,co=O.predict(Xs.cpu().numpy(),full_cov=True)
        print(' out',o,np.abs(d.mean[o].cpu().numpy()-mo).max(), np.abs(d.covariance_matrix[o].cpu().numpy()-co).max())
# OSR wrapper
X=rng.uniform(-1,1,(300,2)); y=np.sin(3*X[:,:1])*np.cos(2*X[:,1:])
Xt,yt=torch.as_tensor(X,device=dev,dtype=torch.float32),torch.as_tensor(y,device=dev,dtype=torch.float32)
r=OnlineSKIRegression(Identity(2),Xt[:100],yt[:100],1e-2,16,1.0)
for s in range(100,300,50):
    print(r.evaluate(Xt[s:s+50],yt[s:s+50])); r.update(Xt[s:s+50],yt[s:s+50],update_gp=False)
pm,pv=r.predict(Xt[:7]); print(pm.shape,pv.shape, r.gp.prediction_cache['cg_iters'])
s2=float(r.gp.likelihood.second_noise)
O=dataspace.DataSpaceGP([[-1.1,1.1]]*2,16,sigma2=s2).fit(X,y[:,0],np.ones(300)); mo,vo=O.predict(X[:7])
print(np.abs(pm[:,0].cpu().numpy()-mo).max()/np.abs(mo).max(), np.abs(pv[:,0].cpu().numpy()-s2-vo).max()/vo.max())
# batched X (botorch style)
from online_gp_amd.models import OnlineSKIBotorchModel
bm=OnlineSKIBotorchModel(Xt[:100].double(),yt[:100].double(),None,grid_bounds=torch.tensor([[-1.1,1.1]]*2),grid_size=10,learn_additional_noise=True)
p=bm.posterior(torch.rand(4,3,2,device=dev)); print(p.mean.shape,p.variance.shape,p.mvn.covariance_matrix.shape, p.rsample(torch.Size([2])).shape)
