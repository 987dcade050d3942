import sys, torch, time
sys.path.insert(0, '/root/repo')
from ir_sgmcmc_amd.engine import EngineConfig, TransitionEngine
from ir_sgmcmc_amd.data_loader import synthetic_pair
dev = 'cuda:0'
N = 96
f, m = synthetic_pair((N, N, N), seed=0)
fx = {k: v.unsqueeze(0).to(dev) for k, v in f.items() if k != 'seg'}
mv = {k: v.unsqueeze(0).to(dev) for k, v in m.items() if k != 'seg'}
eng = TransitionEngine(EngineConfig(dims=(N, N, N), seed=5, lr=0.1, reg_loss='RegLoss_LogNormal', reg_learnable=True), dev)
fd, md = eng.prepare(fx, mv)
eng.gmm_init(fd, md)
v = torch.zeros(1, 3, N, N, N, device=dev)
out = {'displacement': torch.empty(1, 3, N, N, N, device=dev)}
t0 = time.time()
for it in range(1, 2001):
    eng.transition(fd, md, v, outputs=out)
    if it % 250 == 0:
        sc = eng.scalars()
        print(it, 'data %.1f reg %.1f alpha %.3f |d|max %.3f |v|max %.3f' % (sc['data_term'][0], sc['reg_term'][0], sc['alpha'][0],
              float(out['displacement'].abs().max()), float(v.abs().max())), flush=True)
torch.cuda.synchronize()
print('2000 transitions in %.2f s' % (time.time() - t0), 'finite', bool(torch.isfinite(v).all()))
