"""Pin the CPU oracle (oracle/pggan_cpu.py) to golden vectors exported from the reference.

CPU-only (``-m "not gpu"``).  Fixtures: tests/golden/*.npz|json, written by
tests/golden/make_golden.py from /root/reference (which does not exist on the GPU box)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, fixture_params, load_fixture, rel_err

TOL = 2e-5          # fp32, same torch build, different op ordering only (loss algebra / mean)


def _cfg(oracle, meta):
    c = meta['cfg']
    return oracle.NetCfg(c['resolution'], c['num_channels'], fmap_base=c['fmap_base'],
                         fmap_decay=c['fmap_decay'], fmap_max=c['fmap_max'], latent_size=c['latent_size'])


@pytest.mark.parametrize('name', ['tiny32', 'tiny16c1', 'thin1024', 'trace16'])
def test_init_is_bit_exact(oracle, name):
    """Same seed + same construction order => bit-identical weights, biases and c (network.py:8-30)."""
    meta, data = load_fixture(name)
    cfg = _cfg(oracle, meta)
    torch.manual_seed(meta['init_seed'])
    gp = oracle.init_generator(cfg)
    dp = oracle.init_discriminator(cfg)
    gname, dname = ('G0', 'D0') if name == 'trace16' else ('G', 'D')
    for pre, mine in ((gname, gp), (dname, dp)):
        ref = fixture_params(data, pre)
        assert set(ref) == set(mine)
        for k, v in ref.items():
            if torch.is_tensor(v):
                assert torch.equal(v, mine[k]), k
            else:
                assert np.float32(v) == np.float32(mine[k]), k


@pytest.mark.parametrize('name', ['tiny32', 'tiny16c1', 'thin1024'])
def test_forward_and_step_gradients(oracle, name):
    meta, data = load_fixture(name)
    cfg = _cfg(oracle, meta)
    gp, dp = fixture_params(data, 'G'), fixture_params(data, 'D')
    for case in meta['cases']:
        tag, depth, alpha, n = case['tag'], case['depth'], case['alpha'], case['n']
        res = 4 * 2 ** depth
        real, z_d, z_g, mix = oracle.synthetic_batch(case['seed'], n, cfg.num_channels, res, cfg.latent_size)
        with torch.no_grad():
            g_out = oracle.generator_forward(gp, cfg, z_d, depth, alpha)
            d_real = oracle.discriminator_forward(dp, cfg, real, depth, alpha)
        if tag + '/G_out' in data.files:
            assert rel_err(g_out, data[tag + '/G_out']) < TOL
        else:
            a = g_out.double()
            cs = np.array([float(a.sum()), float(a.abs().sum()), float((a * a).sum())])
            assert np.allclose(cs[1:], data[tag + '/G_out_checksum'][1:], rtol=1e-5)
            assert rel_err(g_out[:, :, ::61, ::67], data[tag + '/G_out_sample']) < TOL
        assert rel_err(d_real, data[tag + '/D_real']) < TOL
        d = oracle.d_loss_and_grads(dp, gp, cfg, real, z_d, mix, depth, alpha)
        assert rel_err(d['D_cost'], data[tag + '/D_cost']) < TOL
        assert rel_err(d['D_real_loss'], data[tag + '/D_real_loss']) < TOL
        assert rel_err(d['D_fake_loss'], data[tag + '/D_fake_loss']) < TOL
        ref_names = sorted(k[len(tag + '/Dgrad/'):] for k in data.files if k.startswith(tag + '/Dgrad/'))
        assert ref_names == sorted(d['grads'].keys()), 'active-parameter set differs'
        for k in ref_names:
            assert rel_err(d['grads'][k], data['%s/Dgrad/%s' % (tag, k)]) < 5e-4, (tag, k)
        g = oracle.g_loss_and_grads(gp, dp, cfg, z_g, depth, alpha)
        assert rel_err(g['G_cost'], data[tag + '/G_cost']) < TOL
        ref_names = sorted(k[len(tag + '/Ggrad/'):] for k in data.files if k.startswith(tag + '/Ggrad/'))
        assert ref_names == sorted(g['grads'].keys())
        for k in ref_names:
            assert rel_err(g['grads'][k], data['%s/Ggrad/%s' % (tag, k)]) < 5e-4, (tag, k)


def test_full_width_res32(oracle):
    """Default 512-channel widths: weights re-derived from the seed (init pinned above), outputs,
    losses and gradient checksums compared with the reference's."""
    meta, data = load_fixture('full32')
    cfg = _cfg(oracle, meta)
    torch.manual_seed(meta['init_seed'])
    gp = oracle.init_generator(cfg)
    dp = oracle.init_discriminator(cfg)
    for pre, p in (('G', gp), ('D', dp)):
        cs = data[pre + '/param_checksums']
        mine = []
        for k, v in p.items():
            if torch.is_tensor(v):
                a = v.numpy().astype(np.float64)      # same numpy reduction as make_golden.checksum
                mine.append([float(a.sum()), float(np.abs(a).sum()), float((a * a).sum())])
        assert np.array_equal(np.array(mine), cs)
    for k, v in meta['c'].items():
        pre, lname = k.split('/')
        assert (gp if pre == 'G' else dp)[lname] == v
    for case in meta['cases'][:2]:          # depth 0 and depth 2 (alpha .6); depth 3 is covered on the GPU side
        tag, depth, alpha, n = case['tag'], case['depth'], case['alpha'], case['n']
        real, z_d, z_g, mix = oracle.synthetic_batch(case['seed'], n, 3, 4 * 2 ** depth, cfg.latent_size)
        d = oracle.d_loss_and_grads(dp, gp, cfg, real, z_d, mix, depth, alpha)
        assert rel_err(d['D_cost'], data[tag + '/D_cost']) < TOL
        assert rel_err(d['D_real'], data[tag + '/D_real']) < TOL
        for k in data.files:
            if k.startswith(tag + '/Dgrad/'):
                assert rel_err(d['grads'][k.split('/', 2)[2]], data[k]) < 5e-4, k
            if k.startswith(tag + '/Dgrad_sample/'):
                assert rel_err(d['grads'][k.split('/', 2)[2]].reshape(-1)[::997], data[k]) < 5e-4, k


def test_depth_schedule_bit_exact(oracle):
    with open(os.path.join(GOLDEN, 'schedule.json')) as f:
        sched = json.load(f)
    for sw in sched['sweeps']:
        kw = {k: ({int(a): b for a, b in v.items()} if isinstance(v, dict) else v) for k, v in sw['kw'].items()}
        for nimg, depth, alpha_repr, mb, tick in sw['table']:
            d, a, m, t = oracle.depth_schedule(nimg, sw['max_depth'], **kw)
            assert (d, repr(float(a)), m, t) == (depth, alpha_repr, mb, tick), nimg
    for nimg, lr_repr in sched['lr']:
        assert repr(float(0.001 * oracle.rampup(nimg))) == lr_repr


def test_trainer_trace(oracle):
    """14 Trainer.train() iterations with DepthManager (depth 0->2 incl. fades, minibatch change),
    LRScheduler and Adam, replayed through the oracle: losses per iteration and final weights."""
    meta, data = load_fixture('trace16')
    cfg = _cfg(oracle, meta)
    gp, dp = fixture_params(data, 'G0'), fixture_params(data, 'D0')
    og, od = oracle.AdamState(), oracle.AdamState()
    dm_kw = {k: ({int(a): b for a, b in v.items()} if isinstance(v, dict) else v) for k, v in meta['dm_kw'].items()}
    nimg = 0
    for it in range(meta['n_iter']):
        depth, alpha, mb, _ = oracle.depth_schedule(nimg, 2, **dm_kw)
        assert (nimg, depth, repr(float(alpha)), mb) == (meta['nimg'][it], meta['depth'][it], meta['alpha'][it], meta['mb'][it])
        lr = 0.001 * oracle.rampup(nimg, meta['lr_rampup_kimg'])
        assert repr(float(lr)) == meta['lr'][it]
        real = torch.from_numpy(data['real/%d' % it])
        z_d = torch.from_numpy(data['z/%d' % (2 * it)])
        z_g = torch.from_numpy(data['z/%d' % (2 * it + 1)])
        mix = torch.from_numpy(data['mix/%d' % it])
        d, g = oracle.train_iteration(gp, dp, cfg, og, od, real, z_d, z_g, mix, depth, alpha, lr, lr)
        nimg += real.size(0)
        assert abs(float(d['D_cost']) - meta['D_cost'][it]) < 2e-4 * max(1.0, abs(meta['D_cost'][it])), it
        assert abs(float(g['G_cost']) - meta['G_cost'][it]) < 2e-4 * max(1.0, abs(meta['G_cost'][it])), it
    assert nimg == meta['final_nimg']
    for pre, p in (('G1', gp), ('D1', dp)):
        ref = fixture_params(data, pre)
        for k, v in ref.items():
            if torch.is_tensor(v):
                assert rel_err(p[k], v) < 2e-3, k


def _flag_cfg(oracle, meta, case):
    c = meta['cfg']
    g, d = case['g'], case['d']
    return oracle.NetCfg(c['resolution'], c['num_channels'], fmap_base=c['fmap_base'], fmap_decay=c['fmap_decay'],
                         fmap_max=c['fmap_max'], latent_size=c['latent_size'],
                         normalize_latents=g.get('normalize_latents', True), wscale=g.get('wscale', True),
                         g_pixelnorm=g.get('pixelnorm', True), leakyrelu=g.get('leakyrelu', True))


def test_non_default_flags_fixture(oracle):
    """ReLU / no wscale / no PixelNorm variants (reachable from the reference CLI, network.py:76-85,191-198)."""
    meta, data = load_fixture('flags16')
    for case in meta['cases']:
        tag = case['tag']
        cfg = _flag_cfg(oracle, meta, case)
        gp, dp = fixture_params(data, tag + '/G'), fixture_params(data, tag + '/D')
        real, z_d, z_g, mix = oracle.synthetic_batch(case['seed'], case['n'], 3, 16, 32)
        d = oracle.d_loss_and_grads(dp, gp, cfg, real, z_d, mix, case['depth'], case['alpha'])
        assert rel_err(d['D_cost'], data[tag + '/D_cost']) < TOL
        for k in data.files:
            if k.startswith(tag + '/Dgrad/'):
                assert rel_err(d['grads'][k.split('/', 2)[2]], data[k]) < 5e-4, k
        g = oracle.g_loss_and_grads(gp, dp, cfg, z_g, case['depth'], case['alpha'])
        assert rel_err(g['G_cost'], data[tag + '/G_cost']) < TOL
        assert rel_err(g['fake'], oracle.generator_forward(gp, cfg, z_g, case['depth'], case['alpha'])) == 0.0
        for k in data.files:
            if k.startswith(tag + '/Ggrad/'):
                assert rel_err(g['grads'][k.split('/', 2)[2]], data[k]) < 5e-4, k
    # init with wscale=False keeps wscale on the to/fromRGB layers (they are built without layer_settings)
    torch.manual_seed(41)
    c = meta['cfg']
    cfg = oracle.NetCfg(16, 3, fmap_base=c['fmap_base'], fmap_max=c['fmap_max'], latent_size=32, wscale=False)
    gp = oracle.init_generator(cfg)
    dp = oracle.init_discriminator(cfg)
    ref_g, ref_d = fixture_params(data, 'nowscale/G'), fixture_params(data, 'nowscale/D')
    for mine, ref in ((gp, ref_g), (dp, ref_d)):
        for k, v in ref.items():
            if torch.is_tensor(v):
                assert torch.equal(v, mine[k]), k
            else:
                assert np.float32(v) == np.float32(mine[k]), k
