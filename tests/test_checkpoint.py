"""Checkpoint / resume / generate path on the new modules (reference plugins.py:142-195,
train.py:60-64,120-121, generate.py:18-30).

The trace16 fixture is the reference's own 14-iteration run (depth 0->2 with fades and a minibatch change).  Here
the run is interrupted after 7 iterations: SaverPlugin writes the whole-module snapshots (+ the trainer state
this build adds), everything is rebuilt from the files, and the resumed run must still land on the reference's
losses and final weights."""
import heapq
import importlib
import os

import numpy as np
import pytest
import torch

from conftest import load_fixture, rel_err
from helpers import build_nets, load_fixture_params
import emu_ops

pg = importlib.import_module('pggan-pytorch_amd')


def _run_trace(dev, tmp_path, stop_at):
    meta, data = load_fixture('trace16')
    cnt = dict(real=0, z=0, mix=0)
    losses = dict(G=[], D=[])

    class Data(object):
        model_depth, alpha = 0, 1.0

    def make_loader(mb):
        def gen():
            while True:
                x = torch.from_numpy(data['real/%d' % cnt['real']]); cnt['real'] += 1
                yield x
        return gen()

    def make_rlg(mb):
        def f():
            z = torch.from_numpy(data['z/%d' % cnt['z']]); cnt['z'] += 1
            return z
        return f

    def d_loss(Dm, Gm, real, z):
        pg.wgan_gp_loss.set_mixing_factors(torch.from_numpy(data['mix/%d' % cnt['mix']])); cnt['mix'] += 1
        return pg.wgan_gp_D_loss(Dm, Gm, real, z)

    class Rec(pg.Plugin):
        def __init__(self):
            super(Rec, self).__init__([(1, 'iteration')])

        def register(self, trainer):
            pass

        def iteration(self, i, g_cost, d_cost, d_real, d_fake):
            losses['G'].append(float(g_cost)); losses['D'].append(float(d_cost))

    ramp = lambda nimg: pg.utils.rampup(nimg, meta['lr_rampup_kimg'])
    dm_kw = {k: ({int(a): b for a, b in v.items()} if isinstance(v, dict) else v) for k, v in meta['dm_kw'].items()}

    def make_trainer(G, D, resume_nimg, state_from=None):
        opt_g = pg.FusedAdam(G.parameters(), 0.001, betas=(0.0, 0.99))
        opt_d = pg.FusedAdam(D.parameters(), 0.001, betas=(0.0, 0.99))
        if state_from is not None:
            resume_nimg = pg.plugins.load_trainer_state(state_from, str(tmp_path), opt_d, opt_g)
        tr = pg.Trainer(D, G, d_loss, pg.wgan_gp_G_loss, opt_d, opt_g, Data(), make_loader(4), make_rlg(4),
                        resume_nimg=resume_nimg)
        tr.register_plugin(pg.DepthManager(make_loader, make_rlg, 2, **dm_kw))
        tr.register_plugin(pg.LRScheduler(pg.RampupLR(opt_d, ramp), pg.RampupLR(opt_g, ramp)))
        tr.register_plugin(Rec())
        saver = pg.plugins.SaverPlugin(str(tmp_path), network_snapshot_ticks=10 ** 9)
        tr.register_plugin(saver)
        for q in tr.plugin_queues.values():
            heapq.heapify(q)
        return tr, saver

    G, D = build_nets(meta, dev)
    load_fixture_params(G, data, 'G0')
    load_fixture_params(D, data, 'D0')
    tr, saver = make_trainer(G, D, 0)
    for it in range(stop_at):
        tr.train()
    saver.end(1)
    kimg = '{:06}'.format(tr.cur_nimg // 1000)
    files = sorted(os.listdir(str(tmp_path)))
    assert files == ['network-snapshot-%s-%s.dat' % (n, kimg) for n in ('discriminator', 'generator', 'trainer')]
    nimg_at_stop = tr.cur_nimg
    del tr, G, D

    pattern = 'network-snapshot-{}-%s.dat' % kimg
    G2, D2 = pg.plugins.load_models(pattern, str(tmp_path))
    tr2, _ = make_trainer(G2, D2, None, state_from=pattern)
    assert tr2.cur_nimg == nimg_at_stop
    for it in range(stop_at, meta['n_iter']):
        assert (tr2.cur_nimg, G2.depth, repr(float(G2.alpha))) == (meta['nimg'][it], meta['depth'][it], meta['alpha'][it])
        tr2.train()
    for it in range(meta['n_iter']):
        assert abs(losses['D'][it] - meta['D_cost'][it]) < 5e-4 * max(1.0, abs(meta['D_cost'][it])), it
        assert abs(losses['G'][it] - meta['G_cost'][it]) < 5e-4 * max(1.0, abs(meta['G_cost'][it])), it
    for pre, net in (('G1', G2), ('D1', D2)):
        for k, v in net.reference_state_dict().items():
            if torch.is_tensor(v):
                assert rel_err(v.cpu(), data['%s/%s' % (pre, k)]) < 5e-3, k
    return G2


@pytest.fixture()
def emu(monkeypatch):
    for modname in ('engine', 'optim'):
        mod = importlib.import_module('pggan-pytorch_amd.' + modname)
        monkeypatch.setattr(mod, 'ops', emu_ops)
    monkeypatch.setattr(pg.engine, '_check_dev', lambda t, what: t.contiguous())
    monkeypatch.setattr(pg.trainer, '_to_device', lambda t: t)
    yield


def test_resume_matches_reference_trace_host(emu, tmp_path):
    _run_trace('cpu', tmp_path, stop_at=7)


@pytest.mark.gpu
def test_resume_matches_reference_trace_gpu(tmp_path):
    _run_trace('cuda', tmp_path, stop_at=7)


@pytest.mark.gpu
def test_generate_path_and_output_plugin(tmp_path):
    """generate.py:18-30 on a whole-module snapshot + OutputGenerator (plugins.py:177-195) with both a numpy
    postprocessor (reference hook signature) and the device-side saver."""
    torch.manual_seed(3)
    np.random.seed(3)
    G = pg.Generator((1, 3, 16, 16), latent_size=32, fmap_base=128, fmap_max=32).cuda()
    G.depth, G.alpha = 2, 1.0
    path = str(tmp_path / 'network-snapshot-generator-000001.dat')
    torch.save(G, path)
    got = []
    saver = pg.utils.DeviceImageSaver(str(tmp_path / 'samples'), drange=(-1, 1), resolution=64)
    np.random.seed(5)
    out = pg.utils.output_samples(path, 6, [lambda o, d: got.append((o, d)), saver], 'unit')
    np.random.seed(5)
    z = pg.utils.random_latents(6, 32).cuda()
    assert torch.equal(out, G(z))
    assert got[0][1] == 'unit' and isinstance(got[0][0], np.ndarray) and got[0][0].shape == (6, 3, 16, 16)
    assert np.array_equal(got[0][0], out.cpu().numpy())
    assert os.path.exists(str(tmp_path / 'samples' / 'fakes_unit.png'))

    class T(object):
        parallel, cur_nimg = None, 7000
    T.G = G
    calls = []
    og = pg.plugins.OutputGenerator(lambda n: pg.utils.random_latents(n, 32), [lambda o, d: calls.append((o.shape, d)), saver],
                                    samples_count=4, output_snapshot_ticks=1)
    og.register(T)
    og.epoch(1)
    assert calls == [((4, 3, 16, 16), 7)]
    assert os.path.exists(str(tmp_path / 'samples' / 'fakes_000007.png'))
