"""Where does the ~1e-4 relative deviation of the HIP weight gradients from an fp64 evaluation come from?  (tests/test_fp64_adjudicator.py:
biases 1e-6, every weight tensor ~1.3e-4 at 256x256 depth 6.)  Prints, for the case of the test, the per-sample gradient norm of the
penalty term, the penalty itself and the seed coefficient from the HIP path, the fp32 oracle and the fp64 oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pggan_amd as pg
from oracle import pggan_cpu as oc
res, depth, alpha, n, C = 256, 6, 1.0, 2, 1
torch.manual_seed(1337)
shape = (1, C, res, res)
G, D = pg.Generator(shape), pg.Discriminator(shape)
gp, dp = G.reference_state_dict(), D.reference_state_dict()
G.cuda(); D.cuda()
G.depth = D.depth = depth
cfg = oc.NetCfg(res, C)
real, z_d, z_g, mix = oc.synthetic_batch(42 + depth, n, C, 4 * 2 ** depth, 512)
d_cost, rl, fl, st = pg.engine.d_loss_forward(D, G, real.cuda(), z_d.cuda(), mix.cuda(), 10.0, 0.001, 1.0)
torch.cuda.synchronize()
dbl = lambda p: {k: (v.double() if torch.is_tensor(v) else float(v)) for k, v in p.items()}
r32 = oc.d_loss_and_grads(dp, gp, cfg, real, z_d, mix, depth, alpha)
r64 = oc.d_loss_and_grads(dbl(dp), dbl(gp), cfg, real.double(), z_d.double(), mix.double(), depth, alpha)
norm = lambda gpv: 1.0 + torch.sqrt(gpv.double() / 10.0)          # |g| = 1 +- sqrt(gp / lambda): sign resolved below
print('gp   HIP ', st['gp'].cpu().double().tolist())
print('gp   fp32', r32['gp'].double().tolist())
print('gp   fp64', r64['gp'].tolist())
print('rel err of gp: HIP %.2e  fp32 %.2e' % (float(((st['gp'].cpu().double() - r64['gp']) / r64['gp']).abs().max()),
                                               float(((r32['gp'].double() - r64['gp']) / r64['gp']).abs().max())))
print('|g| - 1 (from gp, unsigned): ', torch.sqrt(r64['gp'] / 10.0).tolist())
sc = st['scores'].cpu().double()
print('scores HIP', sc.tolist())
print('scores rel err vs fp64 (real, fake): HIP %.2e %.2e | fp32 %.2e %.2e' % (
    float(((sc[:n] - r64['D_real'].reshape(-1)) / r64['D_real'].reshape(-1)).abs().max()),
    float(((sc[n:2 * n] - r64['D_fake'].reshape(-1)) / r64['D_fake'].reshape(-1)).abs().max()),
    float(((r32['D_real'].double() - r64['D_real']) / r64['D_real']).abs().max()),
    float(((r32['D_fake'].double() - r64['D_fake']) / r64['D_fake']).abs().max())))
fk = st['ctx']['x'][n:2 * n].cpu().double()
print('fake image rel L2 err vs fp64: HIP %.2e  fp32 %.2e' % (float((fk - r64['fake']).norm() / r64['fake'].norm()),
                                                              float((r32['fake'].double() - r64['fake']).norm() / r64['fake'].norm())))

# ---- the input gradient of the penalty term itself: HIP first backward vs fp64 autograd
ops, eng = pg.ops, pg.engine
sub = st['sub']
gimg, adj = eng.d_backward(D, sub, eng._ones(n, 'cuda'), full=False, want_gimg=True, save_adjoints=True)
ss = ops.row_sumsq(gimg)
torch.cuda.synchronize()
p64 = dbl(dp)
fake64 = r64['fake']
mixed = (real.double().reshape(n, -1) * (1 - mix.double()) + fake64.reshape(n, -1) * mix.double()).reshape(real.shape).requires_grad_(True)
s64 = oc.discriminator_forward(p64, cfg, mixed, depth, alpha)
g64 = torch.autograd.grad(s64.sum(), mixed)[0]
mixed32 = mixed.detach().float().requires_grad_(True)
s32 = oc.discriminator_forward(dp, cfg, mixed32, depth, alpha)
g32 = torch.autograd.grad(s32.sum(), mixed32)[0].double()
gh = gimg.cpu().double()
for i in range(n):
    print('sample %d: |g| fp64 %.9e  HIP %.9e (rel %.2e)  fp32 %.9e (rel %.2e);  sqrt(ss) HIP %.9e;  elementwise rel-L2 HIP %.2e fp32 %.2e' % (
        i, float(g64[i].norm()), float(gh[i].norm()), float(gh[i].norm() / g64[i].norm() - 1), float(g32[i].norm()), float(g32[i].norm() / g64[i].norm() - 1),
        float(ss[i].sqrt()), float((gh[i] - g64[i]).norm() / g64[i].norm()), float((g32[i] - g64[i]).norm() / g64[i].norm())))
mh = st['ctx']['x'][2 * n:].cpu().double()
print('mixed input rel-L2 err vs fp64: HIP %.2e' % float((mh - mixed.detach()).norm() / mixed.detach().norm()))
