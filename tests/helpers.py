"""Shared test helpers (test infrastructure)."""
import numpy as np
import torch

import pggan_amd as pg


def build_nets(meta, device='cpu'):
    c = meta['cfg']
    shape = (1, c['num_channels'], c['resolution'], c['resolution'])
    kw = dict(fmap_base=c['fmap_base'], fmap_max=c['fmap_max'], fmap_decay=c['fmap_decay'])
    G = pg.Generator(shape, latent_size=c['latent_size'], **kw)
    D = pg.Discriminator(shape, **kw)
    if device != 'cpu':
        G.to(device)
        D.to(device)
    return G, D


def load_fixture_params(net, data, prefix):
    sd = {}
    pre = prefix + '/'
    for k in data.files:
        if k.startswith(pre):
            v = data[k]
            sd[k[len(pre):]] = float(v) if k.endswith('.c') else v
    net.load_reference_state_dict(sd)


def reference_grads(net):
    """{reference parameter name: gradient in the reference layout} for parameters that have one."""
    out = {}
    for name, m in net.named_modules():
        if isinstance(m, pg.PGConv2d):
            gw, gb = m.conv.weight.grad, m.conv.bias.grad
            if gw is not None:
                if m.kind == 'conv':
                    out[name + '.conv.weight'] = gw[..., :m.ch_in].permute(2, 3, 0, 1).contiguous().cpu()
                else:
                    out[name + '.conv.weight'] = gw.reshape(m.ch_out, m.ch_in, 1, 1).cpu()
            if gb is not None:
                out[name + '.conv.bias'] = gb.cpu()
    if hasattr(net, 'linear'):
        if net.linear.weight.grad is not None:
            out['linear.weight'] = net.linear.weight.grad.cpu()
        if net.linear.bias.grad is not None:
            out['linear.bias'] = net.linear.bias.grad.cpu()
    return out


def synthetic(seed, n, C, res, latent):
    rs = np.random.RandomState(seed)
    real = rs.rand(n, C, res, res).astype(np.float32) * 2 - 1
    z_d = rs.randn(n, latent).astype(np.float32)
    z_g = rs.randn(n, latent).astype(np.float32)
    mix = rs.rand(n, 1).astype(np.float32)
    return tuple(torch.from_numpy(a) for a in (real, z_d, z_g, mix))


def build_flag_nets(meta, case, device='cpu'):
    c = meta['cfg']
    shape = (1, c['num_channels'], c['resolution'], c['resolution'])
    kw = dict(fmap_base=c['fmap_base'], fmap_max=c['fmap_max'], fmap_decay=c['fmap_decay'])
    G = pg.Generator(shape, latent_size=c['latent_size'], **dict(kw, **case['g']))
    D = pg.Discriminator(shape, **dict(kw, **case['d']))
    if device != 'cpu':
        G.to(device)
        D.to(device)
    return G, D


def grads_by_name(net):
    """{parameter name: clone of its gradient} (after a device sync) for every parameter that has one."""
    torch.cuda.synchronize()
    return {k: v.grad.detach().clone() for k, v in net.named_parameters() if v.grad is not None}


def assert_same_contributions(ga, gb, tol=0.1, total=5e-3):
    """Two runs of the same backward pass that differ only in stream placement / launch grouping: every parameter must have
    received the same contributions.  Not bit-wise: small layers take split-K forward launches whose atomic order flips a few
    LeakyReLU branches (1e-5 .. 3e-4 on a gradient tensor, measured), but a DROPPED or doubled contribution of one layer is an
    O(1) relative error of that layer's tensor, which a bound on the whole buffer would hide for a bias or an RGB layer."""
    assert sorted(ga) == sorted(gb)
    num = den = 0.0
    for k in ga:
        a, b = ga[k].double(), gb[k].double()
        # per tensor: a missing / doubled contribution is an error of ~1; a flipped branch below a small tensor (a bias of the
        # 4x4 block) has been seen at 2e-2
        assert float((a - b).norm()) <= tol * float(b.norm()) + 1e-12, (k, float((a - b).norm() / (b.norm() + 1e-30)))
        num += float((a - b).norm()) ** 2
        den += float(b.norm()) ** 2
    assert (num / max(den, 1e-300)) ** 0.5 <= total, (num / max(den, 1e-300)) ** 0.5
