"""Diagnostic (GPU box): run one D-step + G-step with every kernel call shadowed by its CPU contract
(tests/emu_ops.py) on the SAME inputs; report the per-call error.  Pinpoints a misbehaving launch."""
import sys, os, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import pggan_amd as pg
import emu_ops as E
from oracle import pggan_cpu as oc

res, depth, alpha, n = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
thr = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-4
torch.set_num_threads(32)
real_ops = pg.ops
names = [k for k in dir(E) if not k.startswith('_') and callable(getattr(E, k)) and hasattr(real_ops, k) and k not in ('require_gpu',)]


def cpu(x):
    if torch.is_tensor(x):
        return x.detach().cpu().clone()
    if isinstance(x, (list, tuple)):
        return type(x)(cpu(v) for v in x)
    return x


def rel(a, b):
    a, b = a.detach().cpu().double().reshape(-1), b.double().reshape(-1)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


class Dual(object):
    pass


dual = Dual()
log = []
for name in names:
    def mk(name):
        hip, emu = getattr(real_ops, name), getattr(E, name)

        def f(*a, **k):
            ca, ck = cpu(a), {kk: cpu(v) for kk, v in k.items()}
            out = hip(*a, **k)
            ref = emu(*ca, **ck)
            worst = 0.0
            outs = out if isinstance(out, tuple) else (out,)
            refs = ref if isinstance(ref, tuple) else (ref,)
            for o, r in zip(outs, refs):
                if torch.is_tensor(o):
                    worst = max(worst, rel(o, r))
            for x, y in list(zip(a, ca)) + [(k[kk], ck[kk]) for kk in k]:
                if torch.is_tensor(x):
                    worst = max(worst, rel(x, y))
            shapes = [tuple(x.shape) for x in a if torch.is_tensor(x)][:3]
            log.append((name, worst, shapes))
            if worst > thr:
                print('  !! %-22s err %.2e shapes %s' % (name, worst, shapes))
            return out
        return f
    setattr(dual, name, mk(name))
dual.require_gpu = real_ops.require_gpu
dual._lib = real_ops._lib
for modname in ('engine', 'optim'):
    importlib.import_module('pggan-pytorch_amd.' + modname).ops = dual

torch.manual_seed(1337)
shape = (1, 3, res, res)
G = pg.Generator(shape).cuda(); D = pg.Discriminator(shape).cuda()
G.depth = D.depth = depth; G.alpha = D.alpha = alpha
real, z_d, z_g, mix = oc.synthetic_batch(42 + depth, n, 3, 4 * 2 ** depth, 512)
pg.wgan_gp_loss.set_mixing_factors(mix)
print('D loss fwd'); c, _, _ = pg.wgan_gp_D_loss(D, G, real.cuda(), z_d.cuda())
print('D loss bwd'); c.backward()
print('G loss fwd'); g = pg.wgan_gp_G_loss(G, D, z_g.cuda())
print('G loss bwd'); g.backward()
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for name, w, s in log:
    agg[name][0] += 1; agg[name][1] = max(agg[name][1], w)
for k, (cnt, w) in sorted(agg.items()):
    print('%-24s calls %4d worst %.2e' % (k, cnt, w))
