import sys, torch
sys.path.insert(0,'/root/repo')
from ir_sgmcmc_amd import ops as G
from ir_sgmcmc_amd.engine import EngineConfig, TransitionEngine
from ir_sgmcmc_amd.data_loader import synthetic_pair
DEV='cuda:0'
def run(N, eps, noise_im=0.02, sob=0):
    f,m=synthetic_pair((N,N,N),seed=0,noise=noise_im)
    to=lambda d:{k:v.unsqueeze(0).to(DEV).contiguous() for k,v in d.items() if k!='seg'}
    fixed,moving=to(f),to(m)
    def rnd(seed):
        g=torch.Generator(device=DEV).manual_seed(seed); return torch.randn(1,3,N,N,N,generator=g,device=DEV)
    def smooth(amp,seed):
        v=G.perturb_smooth(rnd(seed),G.sobolev_kernel_1d(3,0.5)); return v*(amp/float(v.abs().max()))
    def data_term(v_in,want=False):
        eng=TransitionEngine(EngineConfig(dims=(N,N,N),data_loss='SSD',virtual_decimation=False,reg_loss='RegLoss_L2',w_reg=1e-12,sobolev_s=sob,uniform_noise=0.0,lr=1e-30,seed=1),DEV)
        fd,md=eng.prepare(fixed,moving); eng.gmm_init(fd,md)
        vv=v_in.clone(); g=torch.empty_like(vv) if want else None
        eng.transition(fd,md,vv,outputs={'grad_v':g} if want else None)
        return float(eng.scalars()['data_term'][0]), g
    v=smooth(2.0,41); u=smooth(1.0,42)
    L0,g=data_term(v,True)
    rhs=float((g.double()*u.double()).sum())
    lp,_=data_term(v+eps*u); lm,_=data_term(v-eps*u)
    print(N,eps,noise_im,'L0',L0,'lhs',(lp-lm)/(2*eps),'rhs',rhs)
for N in (64,128):
    for eps in (0.0125,0.05,0.2):
        run(N,eps)
run(128,0.05,noise_im=0.0)
run(256,0.05,noise_im=0.0)
run(256,0.05)
run(256,0.2)
