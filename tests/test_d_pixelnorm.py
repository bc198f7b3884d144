"""Discriminator(pixelnorm=True) (reference network.py:191-198 flag): PixelNorm after every
c1/c2 makes D non-piecewise-linear, so the WGAN-GP double backward needs a per-layer Hessian-vector term
(engine._d_backward_pn / pg_pixelnorm_tangent).  Checked against the CPU oracle (which differentiates the
reference's op sequence with torch autograd, create_graph=True) — host emulation on CPU, HIP kernels on the GPU."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from conftest import rel_err
from helpers import reference_grads
import emu_ops

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
pg = importlib.import_module('pggan-pytorch_amd')
oracle = importlib.import_module('oracle.pggan_cpu')


@pytest.fixture()
def emu(monkeypatch):
    for modname in ('engine', 'optim'):
        mod = importlib.import_module('pggan-pytorch_amd.' + modname)
        monkeypatch.setattr(mod, 'ops', emu_ops)
    monkeypatch.setattr(pg.engine, '_check_dev', lambda t, what: t.contiguous())
    yield


def _run(dev, res, depth, alpha, n, seed, C=3):
    torch.manual_seed(seed)
    shape = (1, C, res, res)
    kw = dict(fmap_base=128, fmap_max=32)
    G = pg.Generator(shape, latent_size=32, **kw)
    D = pg.Discriminator(shape, pixelnorm=True, **kw)
    gp, dp = G.reference_state_dict(), D.reference_state_dict()
    G.to(dev); D.to(dev)
    cfg = oracle.NetCfg(res, C, latent_size=32, d_pixelnorm=True, **kw)
    G.depth = D.depth = depth
    G.alpha = D.alpha = alpha
    real, z_d, z_g, mix = oracle.synthetic_batch(seed, n, C, 4 * 2 ** depth, 32)
    assert rel_err(D(real.to(dev)), oracle.discriminator_forward(dp, cfg, real, depth, alpha)) < 2e-4
    pg.wgan_gp_loss.set_mixing_factors(mix)
    d_cost, rl, fl = pg.wgan_gp_D_loss(D, G, real.to(dev), z_d.to(dev))
    d_cost.backward()
    ref = oracle.d_loss_and_grads(dp, gp, cfg, real, z_d, mix, depth, alpha)
    assert rel_err(d_cost, ref['D_cost']) < 2e-4
    assert rel_err(rl, ref['D_real_loss']) < 2e-4 and rel_err(fl, ref['D_fake_loss']) < 2e-4
    mine = reference_grads(D)
    assert sorted(mine) == sorted(ref['grads'])
    num = den = 0.0
    for k, v in ref['grads'].items():
        e = rel_err(mine[k], v)
        num += float((mine[k].cpu().double() - v.double()).pow(2).sum()); den += float(v.double().pow(2).sum())
        assert e < 2e-2, (k, e)
    assert (num / den) ** 0.5 < 2e-3, (num / den) ** 0.5
    g_cost = pg.wgan_gp_G_loss(G, D, z_g.to(dev))
    g_cost.backward()
    refg = oracle.g_loss_and_grads(gp, dp, cfg, z_g, depth, alpha)
    assert rel_err(g_cost, refg['G_cost']) < 2e-4
    for k, v in refg['grads'].items():
        assert rel_err(reference_grads(G)[k], v) < 2e-2, k


@pytest.mark.parametrize('depth,alpha,n', [(0, 1.0, 4), (1, 0.6, 3), (2, 1.0, 2)])
def test_d_pixelnorm_host(emu, depth, alpha, n):
    _run('cpu', 16, depth, alpha, n, 100 + depth)


@pytest.mark.gpu
@pytest.mark.parametrize('depth,alpha,n', [(0, 1.0, 4), (1, 0.6, 3), (2, 1.0, 5), (3, 0.3, 2)])
def test_d_pixelnorm_gpu(depth, alpha, n):
    _run('cuda', 32, depth, alpha, n, 200 + depth)


@pytest.mark.gpu
@pytest.mark.parametrize('P,C', [(64, 512), (1000, 16), (37, 4), (5, 32), (300, 256)])
def test_pixelnorm_tangent_kernel(P, C):
    g = torch.Generator().manual_seed(P + C)
    h, t, a = [torch.randn(P, C, generator=g) for _ in range(3)]
    y, r = emu_ops.pixelnorm_fwd(h)
    ty, inj = pg.ops.pixelnorm_tangent(t.cuda(), y.cuda(), r.cuda(), a.cuda())
    rty, rinj = emu_ops.pixelnorm_tangent(t, y, r, a)
    assert rel_err(ty, rty) < 1e-5 and rel_err(inj, rinj) < 1e-5
    gy = torch.randn(P, C, generator=g)
    out = pg.ops.pixelnorm_lrelu_bwd(gy.cuda(), y.cuda(), r.cuda(), 0.2, inj=inj)
    assert rel_err(out, emu_ops.pixelnorm_lrelu_bwd(gy, y, r, 0.2, inj=rinj)) < 1e-5
