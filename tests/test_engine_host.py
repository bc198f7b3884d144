"""Host-side logic of the product (engine schedules, modules, losses, Trainer, DepthManager,
FusedAdam) checked on CPU against the golden fixtures exported from the reference.

The HIP kernels are replaced by ``tests/emu_ops.py`` (a torch-CPU statement of each kernel's
contract, test infrastructure only) through monkeypatching, so what is verified here is everything
ABOVE the C-ABI: the hand-derived gradient-penalty double backward, the batched [real|fake|mixed]
sweep with the minibatch-stddev Hessian-vector injection, active-parameter sets, optimizer and
schedules.  The kernels themselves are verified on the GPU (tests/test_kernels_gpu.py)."""
import importlib
import json

import numpy as np
import pytest
import torch

import emu_ops
from conftest import load_fixture, rel_err
from helpers import build_flag_nets, build_nets, load_fixture_params, reference_grads, synthetic

import pggan_amd as pg

OUT_TOL = 2e-4     # G/D outputs and losses: rel. max-norm (north-star bound: 1e-3)
GRAD_TOL = 1e-3    # parameter gradients: rel. max-norm per tensor


@pytest.fixture()
def emu(monkeypatch):
    for modname in ('engine', 'optim'):
        mod = importlib.import_module('pggan-pytorch_amd.' + modname)
        monkeypatch.setattr(mod, 'ops', emu_ops)
    monkeypatch.setattr(pg.engine, '_check_dev', lambda t, what: t.contiguous())
    yield


@pytest.mark.parametrize('name', ['tiny32', 'tiny16c1', 'thin1024'])
def test_schedules_match_reference(emu, name):
    meta, data = load_fixture(name)
    G, D = build_nets(meta)
    load_fixture_params(G, data, 'G')
    load_fixture_params(D, data, 'D')
    cfg = meta['cfg']
    for case in meta['cases']:
        tag, depth, alpha, n = case['tag'], case['depth'], case['alpha'], case['n']
        if name == 'thin1024' and depth > 5:
            continue                                        # CPU time; the GPU suite covers depth 7/8
        real, z_d, z_g, mix = synthetic(case['seed'], n, cfg['num_channels'], 4 * 2 ** depth, cfg['latent_size'])
        G.depth = D.depth = depth
        G.alpha = D.alpha = alpha
        if tag + '/G_out' in data.files:
            assert rel_err(G(z_d), data[tag + '/G_out']) < OUT_TOL
        assert rel_err(D(real), data[tag + '/D_real']) < OUT_TOL
        pg.wgan_gp_loss.set_mixing_factors(mix)
        d_cost, d_real_loss, d_fake_loss = pg.wgan_gp_D_loss(D, G, real, z_d)
        assert rel_err(d_cost, data[tag + '/D_cost']) < OUT_TOL
        assert rel_err(d_real_loss, data[tag + '/D_real_loss']) < OUT_TOL
        assert rel_err(d_fake_loss, data[tag + '/D_fake_loss']) < OUT_TOL
        d_cost.backward()
        mine = reference_grads(D)
        ref = sorted(k[len(tag + '/Dgrad/'):] for k in data.files if k.startswith(tag + '/Dgrad/'))
        assert sorted(mine.keys()) == ref, 'active-parameter set differs'
        for k in ref:
            assert rel_err(mine[k], data['%s/Dgrad/%s' % (tag, k)]) < GRAD_TOL, (tag, k)
        assert all(p.grad is None for p in G.parameters())
        g_cost = pg.wgan_gp_G_loss(G, D, z_g)
        assert rel_err(g_cost, data[tag + '/G_cost']) < OUT_TOL
        g_cost.backward()
        mine = reference_grads(G)
        ref = sorted(k[len(tag + '/Ggrad/'):] for k in data.files if k.startswith(tag + '/Ggrad/'))
        assert sorted(mine.keys()) == ref
        for k in ref:
            assert rel_err(mine[k], data['%s/Ggrad/%s' % (tag, k)]) < GRAD_TOL, (tag, k)


def test_non_default_flags_fixture(emu):
    """ReLU / no wscale / no PixelNorm+latent-normalisation variants against the reference's own run."""
    meta, data = load_fixture('flags16')
    for case in meta['cases']:
        tag = case['tag']
        G, D = build_flag_nets(meta, case)
        load_fixture_params(G, data, tag + '/G')
        load_fixture_params(D, data, tag + '/D')
        G.depth = D.depth = case['depth']
        G.alpha = D.alpha = case['alpha']
        real, z_d, z_g, mix = synthetic(case['seed'], case['n'], 3, 16, 32)
        assert rel_err(G(z_d), data[tag + '/G_out']) < OUT_TOL
        assert rel_err(D(real), data[tag + '/D_real']) < OUT_TOL
        pg.wgan_gp_loss.set_mixing_factors(mix)
        d_cost, _, _ = pg.wgan_gp_D_loss(D, G, real, z_d)
        assert rel_err(d_cost, data[tag + '/D_cost']) < OUT_TOL
        d_cost.backward()
        mine = reference_grads(D)
        for k in data.files:
            if k.startswith(tag + '/Dgrad/'):
                assert rel_err(mine[k.split('/', 2)[2]], data[k]) < GRAD_TOL, k
        g_cost = pg.wgan_gp_G_loss(G, D, z_g)
        assert rel_err(g_cost, data[tag + '/G_cost']) < OUT_TOL
        g_cost.backward()
        mine = reference_grads(G)
        for k in data.files:
            if k.startswith(tag + '/Ggrad/'):
                assert rel_err(mine[k.split('/', 2)[2]], data[k]) < GRAD_TOL, k
    # wscale=False construction keeps equalized lr on the to/fromRGB layers, like the reference
    torch.manual_seed(41)
    G, D = build_flag_nets(meta, meta['cases'][1])
    for pre, net in (('nowscale/G', G), ('nowscale/D', D)):
        for k, v in net.reference_state_dict().items():
            ref = data['%s/%s' % (pre, k)]
            if torch.is_tensor(v):
                assert torch.equal(v, torch.from_numpy(ref)), k
            else:
                assert np.float32(v) == np.float32(ref), k


def test_init_matches_reference_bit_exact():
    """Same seed, same construction order => same weights as the reference (network.py:8-30)."""
    for name in ('tiny32', 'tiny16c1'):
        meta, data = load_fixture(name)
        torch.manual_seed(meta['init_seed'])
        G, D = build_nets(meta)
        for pre, net in (('G', G), ('D', D)):
            sd = net.reference_state_dict()
            for k, v in sd.items():
                ref = data['%s/%s' % (pre, k)]
                if torch.is_tensor(v):
                    assert torch.equal(v, torch.from_numpy(ref)), k
                else:
                    assert np.float32(v) == np.float32(ref), k


def test_trainer_trace(emu):
    """Trainer + DepthManager + LRScheduler + FusedAdam over 14 iterations (depth 0->2 with fades and a
    minibatch change) against the reference's own run (tests/golden/trace16.*)."""
    meta, data = load_fixture('trace16')
    G, D = build_nets(meta)
    load_fixture_params(G, data, 'G0')
    load_fixture_params(D, data, 'D0')
    opt_g = pg.FusedAdam(G.parameters(), 0.001, betas=(0.0, 0.99))
    opt_d = pg.FusedAdam(D.parameters(), 0.001, betas=(0.0, 0.99))
    rk = meta['lr_rampup_kimg']
    ramp = lambda nimg: pg.utils.rampup(nimg, rk)
    lrs_d, lrs_g = pg.RampupLR(opt_d, ramp), pg.RampupLR(opt_g, ramp)
    cnt = dict(real=0, z=0, mix=0)

    class Data(object):
        model_depth = 0
        alpha = 1.0
    dataset = Data()

    def make_loader(mb):
        def gen():
            while True:
                x = torch.from_numpy(data['real/%d' % cnt['real']])
                cnt['real'] += 1
                assert x.shape[0] == mb and x.shape[-1] == 4 * 2 ** dataset.model_depth
                yield x
        return gen()

    def make_rlg(mb):
        def f():
            z = torch.from_numpy(data['z/%d' % cnt['z']])
            cnt['z'] += 1
            assert z.shape[0] == mb
            return z
        return f

    def d_loss(Dm, Gm, real, z):
        pg.wgan_gp_loss.set_mixing_factors(torch.from_numpy(data['mix/%d' % cnt['mix']]))
        cnt['mix'] += 1
        return pg.wgan_gp_D_loss(Dm, Gm, real, z)

    tr = pg.Trainer(D, G, d_loss, pg.wgan_gp_G_loss, opt_d, opt_g, dataset, make_loader(4), make_rlg(4))
    pg.trainer._to_device = lambda t: t
    dm_kw = {k: ({int(a): b for a, b in v.items()} if isinstance(v, dict) else v) for k, v in meta['dm_kw'].items()}
    tr.register_plugin(pg.DepthManager(make_loader, make_rlg, 2, **dm_kw))
    tr.register_plugin(pg.LRScheduler(lrs_d, lrs_g))
    losses = dict(G=[], D=[])

    class Rec(pg.Plugin):
        def __init__(self):
            super(Rec, self).__init__([(1, 'iteration')])

        def register(self, trainer):
            pass

        def iteration(self, i, g_cost, d_cost, d_real, d_fake):
            losses['G'].append(float(g_cost))
            losses['D'].append(float(d_cost))
    tr.register_plugin(Rec())
    import heapq
    for q in tr.plugin_queues.values():
        heapq.heapify(q)
    for it in range(meta['n_iter']):
        assert (tr.cur_nimg, G.depth, repr(float(G.alpha))) == (meta['nimg'][it], meta['depth'][it], meta['alpha'][it])
        assert repr(float(opt_d.param_groups[0]['lr'])) == meta['lr'][it]
        tr.train()
        assert abs(losses['D'][it] - meta['D_cost'][it]) < 2e-4 * max(1.0, abs(meta['D_cost'][it])), it
        assert abs(losses['G'][it] - meta['G_cost'][it]) < 2e-4 * max(1.0, abs(meta['G_cost'][it])), it
    assert tr.cur_nimg == meta['final_nimg']
    for pre, net in (('G1', G), ('D1', D)):
        sd = net.reference_state_dict()
        for k, v in sd.items():
            if torch.is_tensor(v):
                assert rel_err(v, data['%s/%s' % (pre, k)]) < 2e-3, k


def test_depth_manager_bit_exact():
    with open(__import__('os').path.join(__import__('conftest').GOLDEN, 'schedule.json')) as f:
        sched = json.load(f)

    class Net(object):
        depth, alpha = 0, 1.0

    class Tr(object):
        def __init__(self):
            self.cur_nimg = 0
            self.D, self.G, self.dataset = Net(), Net(), Net()
            self.dataset.model_depth = 0
            self.stats = {}
            self.tick_duration_nimg = 0
    for sw in sched['sweeps']:
        kw = {k: ({int(a): b for a, b in v.items()} if isinstance(v, dict) else v) for k, v in sw['kw'].items()}
        dm = pg.DepthManager(lambda mb: [mb], lambda mb: (lambda: mb), sw['max_depth'], **kw)
        tr = Tr()
        dm.register(tr)
        for nimg, depth, alpha_repr, mb, tick in sw['table']:
            tr.cur_nimg = nimg
            dm.iteration()
            assert (tr.G.depth, repr(float(tr.G.alpha)), tr.stats['minibatch_size'], tr.tick_duration_nimg) == \
                   (depth, alpha_repr, mb, tick), nimg
    for nimg, lr_repr in sched['lr']:
        assert repr(float(0.001 * pg.utils.rampup(nimg))) == lr_repr


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads and exports every symbol include/*.h declares (no compute): the product boundary
    (pggan_hip.h) and the thread-local diagnostic exports (pggan_hip_debug.h)."""
    import ctypes
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, 'include', 'pggan_hip.h')).read()
    declared = set(re.findall(r'\b(?:int|const char\*)\s+(pg_\w+)\s*\(', hdr))
    assert declared == set(pg._lib.SIGNATURES), declared ^ set(pg._lib.SIGNATURES)
    assert not any(n.startswith('pg_debug') for n in declared)          # the product header carries no debug state
    dbg = open(os.path.join(root, 'include', 'pggan_hip_debug.h')).read()
    declared_dbg = set(re.findall(r'\b(?:int|const char\*)\s+(pg_\w+)\s*\(', dbg))
    assert declared_dbg == set(pg._lib.DEBUG_SIGNATURES), declared_dbg ^ set(pg._lib.DEBUG_SIGNATURES)
    declared |= declared_dbg
    if not os.path.exists(pg.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(pg.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.pg_abi_version() == pg._lib.ABI_VERSION


def test_no_cpu_fallback():
    meta, data = load_fixture('tiny32')
    G, D = build_nets(meta)
    with pytest.raises(RuntimeError):
        G(torch.zeros(2, 16))
    with pytest.raises(RuntimeError):
        pg.wgan_gp_D_loss(D, G, torch.zeros(2, 3, 4, 4), torch.zeros(2, 16))


def test_bench_refuses_more_gpus_than_visible():
    """``python bench.py --gpus N`` launches the N ranks itself and must not silently measure fewer devices."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    want = torch.cuda.device_count() + 1 if torch.cuda.is_available() else 2
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(max(2, want))], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'GPU(s) are visible' in r.stderr, (r.returncode, r.stderr[-300:])
    env['WORLD_SIZE'] = '4'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE' in r.stderr


def test_time_monitor_stat_names():
    """plugins.TimeMonitor fills the reference's per-tick timing stats (plugins.py:114-139: 'time', 'sec.tick', 'sec.kimg') and this
    project's 'img/s' / 'd_gp_ms'; no device is needed for the host-clock half."""
    from datetime import timedelta

    class T(object):
        cur_nimg = 100
        stats = {}
        d_step_probe = None
    tr = T()
    mon = pg.TimeMonitor(base_time=5, sample_every=4)
    assert mon.trigger_interval == [(1, 'epoch')]
    mon.register(tr)
    assert tr.stats['sec'] == {'log_format': ':.1f'} and tr.d_step_probe == {'every': 4, 'pairs': []}
    tr.cur_nimg = 1100
    mon.epoch(1)
    assert isinstance(tr.stats['time'], timedelta) and tr.stats['time'].total_seconds() >= 5
    assert tr.stats['sec']['tick'] > 0 and abs(tr.stats['sec']['kimg'] - tr.stats['sec']['tick']) < 1e-9       # 1000 images in the tick
    assert abs(tr.stats['img/s']['val'] * tr.stats['sec']['tick'] - 1000) < 1e-6
    assert tr.stats['d_gp_ms']['val'] == 0.0                 # (no sampled iteration: the probe needs device events)


def test_process_group_handle_is_not_pickled():
    """SaverPlugin pickles whole modules (plugins.py:155-166).  In the exact-global stddev mode the Discriminator holds the data-parallel
    group (communicator handle, streams): it must stay out of the pickle, and a reloaded network starts in the local-shard mode."""
    import io

    class Handle(object):
        def __reduce__(self):
            raise RuntimeError('the process-group handle must not be pickled')
    D = pg.Discriminator((1, 3, 16, 16), fmap_base=64, fmap_max=16)
    D._global_stddev = Handle()
    buf = io.BytesIO()
    torch.save(D, buf)
    buf.seek(0)
    D2 = torch.load(buf, weights_only=False)
    assert D2.__dict__.get('_global_stddev') is None and isinstance(D._global_stddev, Handle)
