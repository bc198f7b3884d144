"""Diagnostic (GPU box): per-parameter gradient error of the HIP path vs the fp32 and fp64 CPU oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import pggan_amd as pg
from oracle import pggan_cpu as oc
from helpers import reference_grads

res, depth, alpha, n = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
torch.set_num_threads(32)
torch.manual_seed(1337)
shape = (1, 3, res, res)
G = pg.Generator(shape); D = pg.Discriminator(shape)
gp, dp = G.reference_state_dict(), D.reference_state_dict()
G.cuda(); D.cuda()
cfg = oc.NetCfg(res, 3)
G.depth = D.depth = depth; G.alpha = D.alpha = alpha
real, z_d, z_g, mix = oc.synthetic_batch(42 + depth, n, 3, 4 * 2 ** depth, 512)
pg.wgan_gp_loss.set_mixing_factors(mix)
c, _, _ = pg.wgan_gp_D_loss(D, G, real.cuda(), z_d.cuda()); c.backward()
mine = reference_grads(D)
pg.wgan_gp_loss.set_mixing_factors(mix)
c2, _, _ = pg.wgan_gp_D_loss(D, G, real.cuda(), z_d.cuda()); c2.backward()
mine2 = reference_grads(D)
r32 = oc.d_loss_and_grads(dp, gp, cfg, real, z_d, mix, depth, alpha)
to64 = lambda p: {k: (v.double() if torch.is_tensor(v) else v) for k, v in p.items()}
r64 = oc.d_loss_and_grads(to64(dp), to64(gp), cfg, real.double(), z_d.double(), mix.double(), depth, alpha)
def errs(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a - b).abs().max() / b.abs().max()), float((a - b).norm() / b.norm())
print('D_cost hip %.7f o32 %.7f o64 %.7f' % (float(c), float(r32['D_cost']), float(r64['D_cost'])))
print('%-28s %22s %22s %22s' % ('param', 'hip vs o64 (max,l2)', 'o32 vs o64', 'hip run1 vs run2'))
for k in mine:
    print('%-28s %10.2e %10.2e  %10.2e %10.2e  %10.2e %10.2e' % ((k,) + errs(mine[k], r64['grads'][k]) + errs(r32['grads'][k], r64['grads'][k]) + errs(mine2[k], mine[k])))
