import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from online_gp_amd.models import FixedNoiseOnlineSKIGP
from online_gp_amd.mlls import BatchedWoodburyMarginalLogLikelihood
from online_gp_amd import settings
G=np.load('/root/repo/tests/golden/case2_mll_2d.npz')
dev='cuda'
X=torch.as_tensor(G['x'],device=dev); Y=torch.as_tensor(G['y'],device=dev); N=torch.as_tensor(G['noise'],device=dev)
for o in range(3):
    m=FixedNoiseOnlineSKIGP(X,Y[:,o:o+1],N[:,o:o+1],grid_bounds=torch.tensor([[0.,1.],[0.,1.]]),grid_size=5,learn_additional_noise=False)
    mll=BatchedWoodburyMarginalLogLikelihood(m.likelihood,m)
    m.train()
    v=mll(m(X),Y[:,o])
    v.backward()
    ls=m.covar_module.base_kernel.base_kernel; sc=m.covar_module.base_kernel
    # d/dlog(theta) = d/draw * (dtheta/draw)^-1 * theta ; softplus'(0)=0.5, theta=ln2
    th=np.log(2.0)
    g_ls=ls.raw_lengthscale.grad.cpu().numpy().reshape(-1)/0.5*th
    g_os=float(sc.raw_outputscale.grad)/0.5*th
    print(o,'mll',float(v),float(G[f'mll_{o}']),'grad',g_ls,g_os,'fd',G[f'dmll_dlog_{o}'])
