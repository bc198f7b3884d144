"""End-to-end parity of the HIP path on the MI355X against (a) the golden vectors exported from the
reference and (b) the CPU oracle on seeded inputs, through the product's public API
(Generator / Discriminator / wgan_gp_D_loss / wgan_gp_G_loss / Trainer / DepthManager / FusedAdam).
Tolerances (stated): outputs & losses 2e-4 rel. max-norm (north-star bound 1e-3), gradients 2e-3."""
import copy
import heapq

import numpy as np
import pytest
import torch

from conftest import fixture_params, load_fixture, rel_err
from helpers import (assert_same_contributions, build_flag_nets, build_nets, grads_by_name, load_fixture_params, reference_grads,
                     synthetic)

import pggan_amd as pg

pytestmark = pytest.mark.gpu
OUT_TOL = 2e-4       # G/D outputs and losses, rel. max-norm (north-star bound: 1e-3)
GRAD_TOL = 2e-3      # per-tensor gradient, rel. max-norm — small, well-conditioned fixtures
# Wide / high-resolution cases: LeakyReLU' is discontinuous at 0, so among >1e6 pre-activations a few
# lie within fp32 round-off of zero and land on different sides under a different (equally valid)
# fp32 summation order.  Each flip perturbs the gradient by O(1/sqrt(#elements)); the reference's own
# fp32 CPU path deviates from an fp64 evaluation by the same mechanism (tools/sweeps/diag_grad_noise.py:
# 2e-3 max-norm at 128x128).  Gradients are therefore compared in relative L2 norm per tensor, with a
# loose max-norm guard, and in global relative L2 norm.
# Round 2 tested that claim (tests/test_fp64_adjudicator.py): ON the linear piece its forward pass selects, the HIP path matches an
# fp64 evaluation to 4e-7 .. 9e-6 over all tensors (every tensor within 3x the fp32 oracle's own distance + 1.3e-7), and its piece
# differs from an fp64 pass's on 3 .. 22 of 1e7 .. 1e8 LeakyReLU branches.  ONE such branch in a low-resolution layer moves the
# gradients of everything downstream by 1e-4 .. 5e-3 (measured: 5.1e-3 on the input gradient of one sample, 3.6e-3 on all G
# gradients of the thin 1024x1024 net), for the fp32 oracle as much as for the HIP path.  Whole-pipeline comparisons therefore
# cannot be tighter than this (2e-3 global was tried and met a 2.13e-3 event at 128x128 depth 4): the arithmetic is pinned by the
# adjudicator, these bounds only have to catch real defects (wrong mask, wrong scale: O(1e-1)).
GRAD_L2_TOL, GRAD_MAX_TOL, GRAD_GLOBAL_L2_TOL = 1e-2, 3e-2, 4e-3
DEV = 'cuda'


def _l2(a, b):
    a, b = a.detach().cpu().double().reshape(-1), torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b).double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def _adam_step_bound(lr, t, beta2=0.99):
    """Largest move of one parameter in Adam step t with beta1 = 0: lr * |g| / sqrt(v_hat) <= lr * sqrt((1 - beta2^t) / (1 - beta2)) -- reached by an
    element whose gradient was ~0 until this step.  Two runs whose gradients differ only by atomic-order noise can differ by twice that on such
    an element (opposite signs)."""
    return lr * ((1.0 - beta2 ** t) / (1.0 - beta2)) ** 0.5


def _cpu_sd(sd):
    return {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in sd.items()}


def _check_grads_loose(mine, ref, what, max_tol=GRAD_MAX_TOL):
    num = den = 0.0
    worst_l2 = worst_max = 0.0
    for k, r in ref.items():
        r = torch.as_tensor(np.asarray(r) if not torch.is_tensor(r) else r)
        a = mine[k].detach().cpu().reshape(r.shape) if mine[k].numel() == r.numel() else mine[k]
        l2, mx = _l2(a, r), rel_err(a, r)
        worst_l2, worst_max = max(worst_l2, l2), max(worst_max, mx)
        assert l2 < GRAD_L2_TOL and mx < max_tol, (what, k, l2, mx)
        num += float((a.double() - r.double()).pow(2).sum())
        den += float(r.double().pow(2).sum())
    glob = (num / den) ** 0.5
    print('%s: worst per-tensor rel-L2 %.2e, worst rel-max %.2e, global rel-L2 %.2e' % (what, worst_l2, worst_max, glob))
    assert glob < GRAD_GLOBAL_L2_TOL, (what, glob)


def _check_case(G, D, data, tag, case, cfg, loose=False):
    gtol = GRAD_MAX_TOL if loose else GRAD_TOL
    depth, alpha, n = case['depth'], case['alpha'], case['n']
    real, z_d, z_g, mix = synthetic(case['seed'], n, cfg['num_channels'], 4 * 2 ** depth, cfg['latent_size'])
    G.depth = D.depth = depth
    G.alpha = D.alpha = alpha
    g_out = G(z_d.to(DEV)).cpu()
    if tag + '/G_out' in data.files:
        assert rel_err(g_out, data[tag + '/G_out']) < OUT_TOL
    elif tag + '/G_out_sample' in data.files:
        assert rel_err(g_out[:, :, ::61, ::67], data[tag + '/G_out_sample']) < OUT_TOL
        a = g_out.double()
        cs = np.array([float(a.abs().sum()), float((a * a).sum())])
        assert np.allclose(cs, data[tag + '/G_out_checksum'][1:], rtol=1e-4)
    assert rel_err(D(real.to(DEV)), data[tag + '/D_real']) < OUT_TOL
    pg.wgan_gp_loss.set_mixing_factors(mix)
    d_cost, d_real_loss, d_fake_loss = pg.wgan_gp_D_loss(D, G, real.to(DEV), z_d.to(DEV))
    assert rel_err(d_cost, data[tag + '/D_cost']) < OUT_TOL
    assert rel_err(d_real_loss, data[tag + '/D_real_loss']) < OUT_TOL
    assert rel_err(d_fake_loss, data[tag + '/D_fake_loss']) < OUT_TOL
    d_cost.backward()
    mine = reference_grads(D)
    worst = 0.0
    for k in data.files:
        if k.startswith(tag + '/Dgrad/'):
            e = rel_err(mine[k.split('/', 2)[2]], data[k]); worst = max(worst, e)
            assert e < gtol, (k, e)
        elif k.startswith(tag + '/Dgrad_sample/'):
            e = rel_err(mine[k.split('/', 2)[2]].reshape(-1)[::997], data[k]); worst = max(worst, e)
            assert e < gtol, (k, e)
    ref_names = sorted(k.split('/', 2)[2] for k in data.files if k.startswith(tag + '/Dgrad'))
    assert sorted(set(ref_names)) == sorted(mine.keys()), 'active-parameter set differs'
    g_cost = pg.wgan_gp_G_loss(G, D, z_g.to(DEV))
    assert rel_err(g_cost, data[tag + '/G_cost']) < OUT_TOL
    g_cost.backward()
    mine = reference_grads(G)
    for k in data.files:
        if k.startswith(tag + '/Ggrad/'):
            e = rel_err(mine[k.split('/', 2)[2]], data[k]); worst = max(worst, e)
            assert e < gtol, (k, e)
        elif k.startswith(tag + '/Ggrad_sample/'):
            e = rel_err(mine[k.split('/', 2)[2]].reshape(-1)[::997], data[k]); worst = max(worst, e)
            assert e < gtol, (k, e)
    print('%s: ok (worst gradient rel err %.2e)' % (tag, worst))


@pytest.mark.parametrize('name', ['tiny32', 'tiny16c1', 'thin1024'])
def test_golden_fixture(name):
    meta, data = load_fixture(name)
    G, D = build_nets(meta, DEV)
    load_fixture_params(G, data, 'G')
    load_fixture_params(D, data, 'D')
    for case in meta['cases']:
        _check_case(G, D, data, case['tag'], case, meta['cfg'], loose=(name == 'thin1024'))


def test_full_width_res32_golden():
    """Default 512-channel widths, weights re-derived from the seed (bit-exact init rule)."""
    meta, data = load_fixture('full32')
    torch.manual_seed(meta['init_seed'])
    G, D = build_nets(meta, DEV)
    for case in meta['cases']:
        _check_case(G, D, data, case['tag'], case, meta['cfg'], loose=True)


def test_trainer_trace_golden():
    meta, data = load_fixture('trace16')
    G, D = build_nets(meta, DEV)
    load_fixture_params(G, data, 'G0')
    load_fixture_params(D, data, 'D0')
    opt_g = pg.FusedAdam(G.parameters(), 0.001, betas=(0.0, 0.99))
    opt_d = pg.FusedAdam(D.parameters(), 0.001, betas=(0.0, 0.99))
    ramp = lambda nimg: pg.utils.rampup(nimg, meta['lr_rampup_kimg'])
    lrs_d, lrs_g = pg.RampupLR(opt_d, ramp), pg.RampupLR(opt_g, ramp)
    cnt = dict(real=0, z=0, mix=0)

    class Data(object):
        model_depth, alpha = 0, 1.0
    dataset = Data()

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

    tr = pg.Trainer(D, G, d_loss, pg.wgan_gp_G_loss, opt_d, opt_g, dataset, make_loader(4), make_rlg(4))
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
            losses['G'].append(float(g_cost)); losses['D'].append(float(d_cost))
    tr.register_plugin(Rec())
    for q in tr.plugin_queues.values():
        heapq.heapify(q)
    for it in range(meta['n_iter']):
        assert (tr.cur_nimg, G.depth, repr(float(G.alpha))) == (meta['nimg'][it], meta['depth'][it], meta['alpha'][it])
        tr.train()
        assert abs(losses['D'][it] - meta['D_cost'][it]) < 5e-4 * max(1.0, abs(meta['D_cost'][it])), it
        assert abs(losses['G'][it] - meta['G_cost'][it]) < 5e-4 * max(1.0, abs(meta['G_cost'][it])), it
    for pre, net in (('G1', G), ('D1', D)):
        sd = net.reference_state_dict()
        for k, v in sd.items():
            if torch.is_tensor(v):
                assert rel_err(v.cpu(), data['%s/%s' % (pre, k)]) < 5e-3, k


@pytest.mark.parametrize('res,depth,alpha,n,fmap_base,C', [(128, 5, 1.0, 2, 4096, 3), (128, 4, 0.5, 3, 4096, 3),
                                                           (128, 5, 1.0, 1, 4096, 3),     # one sample: minibatch-stddev of a single image (the 1024^2 form of this case cost 19 s of oracle time)
                                                           (256, 6, 0.25, 2, 8192, 3),
                                                           (256, 6, 1.0, 2, 4096, 1), (1024, 7, 0.5, 1, 4096, 3),
                                                           (1024, 8, 0.5, 1, 4096, 3),    # fade-in of the 1024^2 stage: the lazy pool adjoint across the fade boundary

                                                           (1024, 8, 1.0, 3, 4096, 3)])   # config 5's real minibatch: stddev couples the 3 samples
def test_against_oracle_at_baseline_widths(oracle, res, depth, alpha, n, fmap_base, C):
    """HIP vs CPU oracle on seeded inputs at BASELINE.json widths (default 4096 and the paper's 8192), incl. the
    one-channel 256^2 spectrogram shape of config 4 and a fade-in stage of the 1024^2 net."""
    torch.manual_seed(1337)
    shape = (1, C, res, res)
    G = pg.Generator(shape, fmap_base=fmap_base)
    D = pg.Discriminator(shape, fmap_base=fmap_base)
    gp, dp = G.reference_state_dict(), D.reference_state_dict()
    G.to(DEV); D.to(DEV)
    cfg = oracle.NetCfg(res, C, fmap_base=fmap_base)
    G.depth = D.depth = depth
    G.alpha = D.alpha = alpha
    real, z_d, z_g, mix = oracle.synthetic_batch(42 + depth, n, C, 4 * 2 ** depth, 512)
    pg.wgan_gp_loss.set_mixing_factors(mix)
    d_cost, d_real_loss, d_fake_loss = pg.wgan_gp_D_loss(D, G, real.to(DEV), z_d.to(DEV))
    d_cost.backward()
    ref = oracle.d_loss_and_grads(dp, gp, cfg, real, z_d, mix, depth, alpha)
    assert rel_err(d_cost, ref['D_cost']) < OUT_TOL
    assert rel_err(d_real_loss, ref['D_real_loss']) < OUT_TOL and rel_err(d_fake_loss, ref['D_fake_loss']) < OUT_TOL
    mine = reference_grads(D)
    assert sorted(mine) == sorted(ref['grads'])
    _check_grads_loose(mine, ref['grads'], 'D grads res %d depth %d' % (res, depth))
    g_cost = pg.wgan_gp_G_loss(G, D, z_g.to(DEV))
    g_cost.backward()
    gref = oracle.g_loss_and_grads(gp, dp, cfg, z_g, depth, alpha)
    assert rel_err(g_cost, gref['G_cost']) < OUT_TOL
    assert rel_err(G(z_g.to(DEV)).cpu(), gref['fake']) < OUT_TOL
    gm = reference_grads(G)
    assert sorted(gm) == sorted(gref['grads'])
    _check_grads_loose(gm, gref['grads'], 'G grads res %d depth %d' % (res, depth))


def test_gp_is_quadratic_in_lambda_and_grad_scale_linear():
    """Size-independent properties at full width: gp scales linearly with lambda; backward(gradient=s)
    scales every gradient by s."""
    torch.manual_seed(3)
    shape = (1, 3, 64, 64)
    G = pg.Generator(shape).to(DEV)
    D = pg.Discriminator(shape).to(DEV)
    G.depth = D.depth = 4
    real = (torch.rand(4, 3, 64, 64, device=DEV) * 2 - 1)
    z = torch.randn(4, 512, device=DEV)
    mix = torch.rand(4, 1)
    costs = []
    for lam in (0.0, 10.0, 20.0):
        pg.wgan_gp_loss.set_mixing_factors(mix)
        c, _, _ = pg.wgan_gp_D_loss(D, G, real, z, iwass_lambda=lam)
        costs.append(float(c))
    assert abs((costs[2] - costs[0]) - 2 * (costs[1] - costs[0])) < 1e-4 * max(1.0, abs(costs[2]))
    pg.wgan_gp_loss.set_mixing_factors(mix)
    c, _, _ = pg.wgan_gp_D_loss(D, G, real, z)
    c.backward()
    g1 = D.blocks[-1].c1.conv.weight.grad.clone()
    pg.wgan_gp_loss.set_mixing_factors(mix)
    c, _, _ = pg.wgan_gp_D_loss(D, G, real, z)
    c.backward(torch.tensor(0.5))
    g2 = D.blocks[-1].c1.conv.weight.grad
    assert rel_err(g2.cpu() * 2, g1.cpu()) < 1e-3     # atomics: run-to-run summation order differs


def test_hipgraph_replay_matches_eager():
    """Trainer iterations with the D-/G-step schedules replayed from captured hipGraphs (graphs.py) must
    reproduce the eager launches: same losses, same weights after 5 iterations (atomics => tolerance)."""
    def run(use_graphs):
        pg.wgan_gp_loss.enable_graphs(use_graphs)
        try:
            torch.manual_seed(21)
            shape = (1, 3, 32, 32)
            kw = dict(fmap_base=256, fmap_max=64)
            G = pg.Generator(shape, latent_size=64, **kw).to(DEV)
            D = pg.Discriminator(shape, **kw).to(DEV)
            G.depth = D.depth = 2
            opt_g = pg.FusedAdam(G.parameters(), 0.001, betas=(0.0, 0.99))
            opt_d = pg.FusedAdam(D.parameters(), 0.001, betas=(0.0, 0.99))
            ds = pg.utils.SyntheticDataset(32, 3, seed=5)
            ds.model_depth = 2
            pg.wgan_gp_loss.manual_seed(9)
            tr = pg.Trainer(D, G, pg.wgan_gp_D_loss, pg.wgan_gp_G_loss, opt_d, opt_g, ds, ds.loader(8),
                            pg.utils.device_latents(8, 64, seed=3))
            losses = []

            class Rec(pg.Plugin):
                def __init__(self):
                    super(Rec, self).__init__([(1, 'iteration')])

                def register(self, trainer):
                    pass

                def iteration(self, i, g_cost, d_cost, d_real, d_fake):
                    losses.append((float(g_cost), float(d_cost)))
            tr.register_plugin(Rec())
            for q in tr.plugin_queues.values():
                heapq.heapify(q)
            for _ in range(5):
                tr.train()
            shared = pg.ops._capture_workspace.get(torch.cuda.current_device())
            owners.extend(k for k, v in pg.ops._workspaces.items() if v is not None and v is shared)
            return losses, G._flat_param.clone(), D._flat_param.clone()
        finally:
            pg.wgan_gp_loss.enable_graphs(False)       # (drops the graphs and unregisters the scratch of their capture streams)
    owners = []
    l0, g0, d0 = run(False)
    assert not owners
    l1, g1, d1 = run(True)
    for it, ((a, b), (c, d)) in enumerate(zip(l0, l1)):
        # iteration 0 sees identical weights; later ones inherit sign-like Adam steps taken on fp32-atomic-ordered
        # gradients (run-to-run differences of the same size exist between two eager runs)
        tol = 2e-4 if it == 0 else 5e-3
        assert abs(a - c) < tol * max(1.0, abs(a)) and abs(b - d) < tol * max(1.0, abs(b)), (it, l0, l1)
    assert float((g0 - g1).abs().max()) < 2 * 0.001 * 5 + 1e-4
    assert float((d0 - d1).abs().max()) < 2 * 0.001 * 5 + 1e-4
    assert rel_err(g1, g0) < 2e-2 and rel_err(d1, d0) < 2e-2
    # every capture runs on a stream of its own: they all share ONE scratch per device (ops._stream_with_workspace), whatever
    # number of graphs has been captured so far, next to one per eager stream
    shared = pg.ops._capture_workspace[torch.cuda.current_device()]
    assert owners, 'no captured stream launched a conv that takes the scratch'
    assert not [k for k, v in pg.ops._workspaces.items() if v is shared]       # released with the graphs (graphs.clear)
    assert len({id(v) for v in pg.ops._workspaces.values() if v is not None}) <= 4


@pytest.mark.parametrize('foreign_optimizer', [False, True])
def test_launch_plan_replay_matches_eager(foreign_optimizer, deterministic_forward):
    """The D-/G-step schedules replayed from recorded launch plans (plans.py: the same C-ABI calls on the same streams with the same
    events, minus the Python between them) against eager launches.  Two trainers with the same seeds are stepped side by side, one
    with plans, one eager; after every iteration the PRE-ADAM gradients are compared tensor by tensor (equal up to the order of the
    atomic weight-gradient commits) and the eager trainer is re-synchronised to the other (weights, Adam moments), so replayed
    iterations are checked at identical weights.  Also: the plans are really replayed, a growth-stage change records new plans, and
    with a foreign optimizer (torch.optim.Adam: no mark_params_changed) the version check in front of every replay keeps the derived
    Winograd weights fresh (stale ones would put the losses off by O(lr))."""
    wl = pg.wgan_gp_loss

    def build():
        torch.manual_seed(11)
        shape = (1, 3, 32, 32)
        kw = dict(fmap_base=512, fmap_max=64)
        G = pg.Generator(shape, latent_size=64, **kw).cuda()
        D = pg.Discriminator(shape, **kw).cuda()
        G.depth = D.depth = 2
        mk = (lambda ps: torch.optim.Adam(ps, 0.001, betas=(0.0, 0.99))) if foreign_optimizer else (lambda ps: pg.FusedAdam(ps, 0.001, betas=(0.0, 0.99)))
        opt_g, opt_d = mk(G.parameters()), mk(D.parameters())
        ds = pg.utils.SyntheticDataset(32, 3, seed=5)
        ds.model_depth = 2
        tr = pg.Trainer(D, G, pg.wgan_gp_D_loss, pg.wgan_gp_G_loss, opt_d, opt_g, ds, ds.loader(8), pg.utils.device_latents(8, 64, seed=3))
        tr._losses = []
        return tr, ds
    wl.enable_graphs('auto')
    pg.plans.STATS.update(recorded=0, replayed=0)
    try:
        (tra, dsa), (trb, dsb) = build(), build()
        for it in range(10):
            if it == 6:                                        # growth-stage change: new plans (two eager steps, one recorded, then replay)
                for tr, ds in ((tra, dsa), (trb, dsb)):
                    tr.G.depth = tr.D.depth = ds.model_depth = 3
                    tr.dataiter = ds.loader(8)
            out = []
            for tr, plans_on in ((tra, True), (trb, False)):
                wl._use_plans = plans_on                       # (the switch itself, without dropping the recorded plans)
                wl.manual_seed(100 + it)                       # same mixing factors for both
                tr.train()
                torch.cuda.synchronize()
                out.append((grads_by_name(tr.D), grads_by_name(tr.G)))
            assert_same_contributions(out[0][0], out[1][0], tol=1e-2, total=2e-4)    # (deterministic_forward: atomic commit order only -- no LeakyReLU flip to absorb)
            assert_same_contributions(out[0][1], out[1][1], tol=0.3, total=5e-2)     # (through D after its update: see test_deferred_d_update_matches_inline)
            for a, b in ((tra.G, trb.G), (tra.D, trb.D)):
                assert float((a._flat_param - b._flat_param).abs().max()) <= 2 * _adam_step_bound(0.001, it + 1) + 1e-6
                assert _l2(a._flat_param, b._flat_param.cpu()) < 3e-3
                with torch.no_grad():
                    b._flat_param.copy_(a._flat_param)
                b.mark_params_changed()
            for oa, ob in ((tra.optimizer_g, trb.optimizer_g), (tra.optimizer_d, trb.optimizer_d)):
                if foreign_optimizer:
                    ob.load_state_dict(copy.deepcopy(oa.state_dict()))   # (load_state_dict keeps same-dtype state tensors by reference)
                else:
                    for (_, ma, va), (_, mb_, vb) in zip(oa._flat.values(), ob._flat.values()):
                        mb_.copy_(ma)
                        vb.copy_(va)
        # (D, G) x two stages: two eager steps, one recorded, then replay (iterations 3..5 and 9) for each of the two networks
        assert pg.plans.STATS['recorded'] == 4 and pg.plans.STATS['replayed'] == 2 * (3 + 1), pg.plans.STATS
    finally:
        wl._use_plans = True
        wl.enable_graphs(False)


@pytest.mark.parametrize('d_pixelnorm', [False, True])
def test_plan_replay_public_api_loop(d_pixelnorm, deterministic_forward):
    """The loop of the public API without ``Trainer``: ``c = wgan_gp_D_loss(...); c.backward(); opt.step()``.  With launch plans on, the
    recorded / replayed step leaves the weight-gradient stream un-joined and ``backward()`` must join it before the optimizer reads the
    gradient buffer (Adam used to run under the largest weight-gradient launches).  Also the Discriminator(pixelnorm=True)
    sweep under a plan: its adjoint copies / zero fills are C-ABI launches now, so a replay recomputes them (they used to be ATen ops the
    plan did not record: stale adjoints from the recording step).  Compared per step with an eager twin at identical weights."""
    wl = pg.wgan_gp_loss

    def build():
        torch.manual_seed(21)
        shape = (1, 3, 64, 64)
        kw = dict(fmap_base=1024, fmap_max=64)
        G = pg.Generator(shape, latent_size=64, **kw).cuda()
        D = pg.Discriminator(shape, pixelnorm=d_pixelnorm, **kw).cuda()
        G.depth = D.depth = 4
        return G, D, pg.FusedAdam(D.parameters(), 0.001, betas=(0.0, 0.99))
    pg.plans.STATS.update(recorded=0, replayed=0)
    gen = torch.Generator(device='cuda').manual_seed(9)
    try:
        (Ga, Da, oa), (Gb, Db, ob) = build(), build()
        for it in range(7):
            real = torch.rand((6, 3, 64, 64), device=DEV, generator=gen) * 2 - 1
            z = torch.randn((6, 64), device=DEV, generator=gen)
            mix = torch.rand((6, 1), device=DEV, generator=gen)
            costs = []
            for (G, D, opt), mode in (((Ga, Da, oa), 'auto'), ((Gb, Db, ob), False)):
                wl._use_graphs = mode                           # (the switch itself: enable_graphs(False) would drop the recorded plans)
                wl.set_mixing_factors(mix)
                c = pg.wgan_gp_D_loss(D, G, real, z)[0]
                c.backward()
                costs.append((float(c), D._flat_grad.clone()))
                opt.step()
            torch.cuda.synchronize()
            # (deterministic_forward: the two forward passes are bit-identical, so the losses are, and the pre-Adam gradients differ by the
            #  atomic commit order of the weight gradients only -- measured <= 1e-5 over 1680 twin steps; the round-5 bound of 2e-3 was sized
            #  for LeakyReLU flips of the non-deterministic split-K forward and still met a 2.3e-3 event in 1 of 27 runs)
            assert abs(costs[0][0] - costs[1][0]) <= 1e-6 * max(1.0, abs(costs[1][0])), (it, costs[0][0], costs[1][0])
            assert _l2(costs[0][1], costs[1][1].cpu()) < 2e-4, it
            with torch.no_grad():                               # keep the twins at identical weights / moments
                Db._flat_param.copy_(Da._flat_param)
            Db.mark_params_changed()
            for (_, ma, va), (_, mb_, vb) in zip(oa._flat.values(), ob._flat.values()):
                mb_.copy_(ma)
                vb.copy_(va)
        assert pg.plans.STATS['recorded'] == 1 and pg.plans.STATS['replayed'] == 4, pg.plans.STATS
    finally:
        wl.enable_graphs(False)


@pytest.mark.parametrize('ordered', [True, False])
def test_derived_refresh_ordering(ordered, monkeypatch):
    """Root cause of the round-5 lock-step failure (test_plan_replay_public_api_loop, iteration 1; docs/experiments_r6.md §1), with the
    interleaving FORCED.  In the eager public loop ``c = wgan_gp_D_loss(...); c.backward(); opt.step()`` the optimizer leaves D's derived
    (Winograd-domain / flipped) weights stale, and the next three-pass D forward refreshes them from inside its real-third pass ON THE
    SECOND STREAM -- while the mixed third + first backward of the gradient penalty on the main stream were ordered behind the image copy
    only: they could read the weights of the previous step.  Here every refresh launch is preceded by ~5 ms of filler launches on whatever
    stream issues it, so an unordered reader ALWAYS wins the race.  ``ordered=True`` (the product: engine._await_derived) must match a twin
    that refreshes on the main stream and synchronises before every step; ``ordered=False`` (PGGAN_DERIVED_EVENT=0, the round-5 code path)
    must NOT -- that half shows the regression has teeth."""
    wl, eng = pg.wgan_gp_loss, pg.engine
    filler = torch.zeros(32 << 20, device=DEV)                  # 128 MB: ~0.1 ms per pass
    real_transform = pg.ops.wino_transform_weights_batched
    delay = [False]

    def slow_transform(*a, **k):
        if delay[0]:
            for _ in range(50):
                pg.ops.axpby_mask(filler, a=1.0, out=filler)
        return real_transform(*a, **k)
    monkeypatch.setattr(pg.ops, 'wino_transform_weights_batched', slow_transform)
    monkeypatch.setattr(eng, 'DERIVED_EVENT', ordered)
    wl.enable_graphs(False)

    def build():
        torch.manual_seed(21)
        shape = (1, 3, 64, 64)
        kw = dict(fmap_base=1024, fmap_max=64)
        G = pg.Generator(shape, latent_size=64, **kw).cuda()
        D = pg.Discriminator(shape, **kw).cuda()
        G.depth = D.depth = 4
        return G, D, pg.FusedAdam(D.parameters(), 0.01, betas=(0.0, 0.99))
    gen = torch.Generator(device='cuda').manual_seed(9)
    try:
        (Ga, Da, oa), (Gb, Db, ob) = build(), build()
        worst = 0.0
        for it in range(4):
            real = torch.rand((6, 3, 64, 64), device=DEV, generator=gen) * 2 - 1
            z = torch.randn((6, 64), device=DEV, generator=gen)
            mix = torch.rand((6, 1), device=DEV, generator=gen)
            grads = []
            for (G, D, opt), racy in (((Ga, Da, oa), True), ((Gb, Db, ob), False)):
                delay[0] = racy
                if not racy:                                   # the twin: refresh on the main stream, everything drained before the step
                    D._sync_version()
                    eng._derived(D)
                    torch.cuda.synchronize()
                wl.set_mixing_factors(mix)
                c = pg.wgan_gp_D_loss(D, G, real, z)[0]
                c.backward()
                grads.append(D._flat_grad.clone())
                opt.step()
                torch.cuda.synchronize()
            if it > 0:                                         # (iteration 0: nothing stale yet)
                worst = max(worst, _l2(grads[0], grads[1].cpu()))
            with torch.no_grad():                              # keep the twins at identical weights / moments
                Db._flat_param.copy_(Da._flat_param)
            Db.mark_params_changed()
            for (_, ma, va), (_, mb_, vb) in zip(oa._flat.values(), ob._flat.values()):
                mb_.copy_(ma)
                vb.copy_(va)
            torch.cuda.synchronize()
        print('derived-refresh ordering (ordered=%s): worst pre-Adam gradient rel-L2 vs the synchronised twin %.3e' % (ordered, worst))
        if ordered:
            assert worst < 1e-4, worst                         # atomic commit order only (~1e-6)
        else:
            assert worst > 1e-3, 'the forced interleaving no longer reproduces the stale-weights read: %g' % worst
    finally:
        wl.enable_graphs(False)


def test_two_d_losses_before_backward_do_not_alias():
    """the eager D loss keeps its saved activations in a per-network arena.  A second D-loss forward on the same network
    before the first loss's backward() (two losses then backward; a validation loss between forward and backward) must not overwrite
    them: the second forward allocates its own tensors while the first loss is alive and un-back-propagated."""
    wl = pg.wgan_gp_loss
    wl.enable_graphs(False)
    torch.manual_seed(4)
    shape = (1, 3, 32, 32)
    kw = dict(fmap_base=256, fmap_max=64)
    G = pg.Generator(shape, latent_size=64, **kw).cuda()
    D = pg.Discriminator(shape, **kw).cuda()
    G.depth = D.depth = 3
    gen = torch.Generator(device='cuda').manual_seed(1)
    batches = [(torch.rand((4, 3, 32, 32), device=DEV, generator=gen) * 2 - 1, torch.randn((4, 64), device=DEV, generator=gen),
                torch.rand((4, 1), device=DEV, generator=gen)) for _ in range(2)]

    def alone(b):
        wl.set_mixing_factors(b[2])
        c = pg.wgan_gp_D_loss(D, G, b[0], b[1])[0]
        c.backward()
        return float(c), D._flat_grad.clone()
    ref = [alone(b) for b in batches]
    wl.set_mixing_factors(batches[0][2])
    c0 = pg.wgan_gp_D_loss(D, G, batches[0][0], batches[0][1])[0]
    wl.set_mixing_factors(batches[1][2])
    c1 = pg.wgan_gp_D_loss(D, G, batches[1][0], batches[1][1])[0]       # (first loss still pending: must not write into its activations)
    c0.backward()
    g0 = D._flat_grad.clone()
    c1.backward()
    g1 = D._flat_grad.clone()
    for (c, g), (cr, gr) in zip(((float(c0), g0), (float(c1), g1)), ref):
        assert abs(c - cr) <= 2e-4 * max(1.0, abs(cr))
        assert _l2(g, gr.cpu()) < 1e-4
    # a retained loss whose buffers were handed on afterwards refuses a second backward instead of computing with foreign activations
    wl.set_mixing_factors(batches[0][2])
    c2 = pg.wgan_gp_D_loss(D, G, batches[0][0], batches[0][1])[0]
    c2.backward(retain_graph=True)
    wl.set_mixing_factors(batches[1][2])
    pg.wgan_gp_D_loss(D, G, batches[1][0], batches[1][1])[0].backward()
    with pytest.raises(RuntimeError, match='overwritten'):
        c2.backward()


def test_hipgraph_replay_alternating_batch_shapes():
    """under hipGraph replay (``enable_graphs(True)``; the 4x4 stage's default until round 6) the three-pass forward's activations live in the network's single-slot arena,
    allocated outside the graph's private pool.  A D step with another batch shape replaces that arena; the graph captured for the first
    shape must keep its own alive (graphs._Graphed.keep) -- alternating two shapes, every step against an eager twin."""
    wl = pg.wgan_gp_loss
    wl.enable_graphs(False)

    def build():
        torch.manual_seed(8)
        shape = (1, 3, 16, 16)
        kw = dict(fmap_base=128, fmap_max=32)
        G = pg.Generator(shape, latent_size=32, **kw).cuda()
        D = pg.Discriminator(shape, **kw).cuda()
        G.depth = D.depth = 0
        return G, D
    gen = torch.Generator(device='cuda').manual_seed(2)
    try:
        (Ga, Da), (Gb, Db) = build(), build()
        for it in range(12):
            n = 4 if it % 2 == 0 else 8
            real = torch.rand((n, 3, 4, 4), device=DEV, generator=gen) * 2 - 1
            z = torch.randn((n, 32), device=DEV, generator=gen)
            mix = torch.rand((n, 1), device=DEV, generator=gen)
            out = []
            for (G, D), mode in (((Ga, Da), True), ((Gb, Db), False)):        # (True: hipGraph replay -- 'auto' takes launch plans since round 6)
                wl._use_graphs = mode
                wl.set_mixing_factors(mix)
                c = pg.wgan_gp_D_loss(D, G, real, z)[0]
                c.backward()
                out.append((float(c), D._flat_grad.clone()))
                if it % 4 == 3:                                # churn the allocator: freed arena blocks would be handed out here
                    junk = [torch.full((1 << 16,), float('nan'), device=DEV) for _ in range(64)]
                    del junk
            torch.cuda.synchronize()
            assert abs(out[0][0] - out[1][0]) <= 2e-4 * max(1.0, abs(out[1][0])), (it, out[0][0], out[1][0])
            assert _l2(out[0][1], out[1][1].cpu()) < 1e-4, it
        assert len(pg.graphs._CACHE) == 2 and all(g.graph is not None for g in pg.graphs._CACHE.values())
    finally:
        wl.enable_graphs(False)


def test_time_monitor_d_step_probe():
    """TimeMonitor on the device: every k-th iteration's D update is bracketed with two HIP events (start on the main stream, end on
    the stream the deferred update runs on); the tick reports their mean as stats['d_gp_ms'] next to img/s."""
    torch.manual_seed(2)
    shape = (1, 3, 16, 16)
    kw = dict(fmap_base=128, fmap_max=32)
    G = pg.Generator(shape, latent_size=32, **kw).cuda()
    D = pg.Discriminator(shape, **kw).cuda()
    G.depth = D.depth = 2
    ds = pg.utils.SyntheticDataset(16, 3, seed=5)
    ds.model_depth = 2
    tr = pg.Trainer(D, G, pg.wgan_gp_D_loss, pg.wgan_gp_G_loss, pg.FusedAdam(D.parameters(), 0.001, betas=(0.0, 0.99)),
                    pg.FusedAdam(G.parameters(), 0.001, betas=(0.0, 0.99)), ds, ds.loader(4), pg.utils.device_latents(4, 32, seed=3),
                    tick_nimg_default=4 * 12)
    tr.register_plugin(pg.TimeMonitor(sample_every=3))
    tr.run(4 * 24 / 1000.0)
    assert tr.cur_tick == 2
    assert 0.0 < tr.stats['d_gp_ms']['val'] < 1e3 and tr.stats['img/s']['val'] > 0
    assert tr.stats['sec']['tick'] > 0 and not tr.d_step_probe['pairs']          # consumed at the tick boundary


@pytest.mark.parametrize('plans_on,fake_side', [(False, True), (True, True), (True, False)])
def test_early_real_third_matches_whole_batch_forward(plans_on, fake_side, tmp_path):
    """The real third of the next D step's batched D forward runs on the second stream under the G step (engine.EarlyReal: both passes
    write ONE set of batched activations through ops.Arena, Trainer draws the next real batch one iteration ahead, DepthManager's schedule
    says whether the next iteration still is this stage).  Two trainers with the same seeds, one with the early pass and one without, stepped
    side by side through a stabilisation span, a fade (no early pass there) and a stage change: per-iteration pre-Adam gradients agree,
    weights are re-synchronised after every iteration, the same real batches are consumed in the same order, the early pass is really
    used (and dropped / not started where it must be), and a whole-module pickle taken while a pass is pending works.
    ``fake_side``: the fake third of the D step's forward on the second stream as well (three passes into one set of tensors)."""
    wl = pg.wgan_gp_loss
    eng = pg.engine
    fake_side_before, eng.FAKE_THIRD_ON_SIDE = eng.FAKE_THIRD_ON_SIDE, fake_side

    def build(early):
        torch.manual_seed(17)
        shape = (1, 3, 32, 32)
        kw = dict(fmap_base=512, fmap_max=64)
        G = pg.Generator(shape, latent_size=64, **kw).cuda()
        D = pg.Discriminator(shape, **kw).cuda()
        opt_g = pg.FusedAdam(G.parameters(), 0.001, betas=(0.0, 0.99))
        opt_d = pg.FusedAdam(D.parameters(), 0.001, betas=(0.0, 0.99))
        ds = pg.utils.SyntheticDataset(32, 3, seed=5)
        seen = []

        def loader(n):
            it = ds.loader(n)

            def gen():
                while True:
                    b = next(it)
                    seen.append(float(b.flatten()[0]))
                    yield b
            return gen()
        tr = pg.Trainer(D, G, pg.wgan_gp_D_loss, pg.wgan_gp_G_loss, opt_d, opt_g, ds, None, None, early_real_forward=early)
        span = 6 * 8
        tr.register_plugin(pg.DepthManager(loader, lambda n: pg.utils.device_latents(n, 64, seed=3), 3, minibatch_default=8,
                                           lod_training_nimg=span, lod_transition_nimg=span))
        import heapq
        for q in tr.plugin_queues.values():
            heapq.heapify(q)
        return tr, seen
    wl.enable_graphs(False)
    wl._use_graphs = 'auto' if plans_on else False
    eng.EARLY_STATS.update(passes=0, used=0, dropped=0)
    try:
        (tra, seen_a), (trb, seen_b) = build(True), build(False)
        for it in range(20):                                   # depth 0: 6 iterations, fade into depth 1: 6, depth 1: 6, fade into 2 ...
            out = []
            for tr in (tra, trb):
                wl.manual_seed(200 + it)
                tr.train()
                torch.cuda.synchronize()
                out.append((grads_by_name(tr.D), grads_by_name(tr.G)))
            assert_same_contributions(out[0][0], out[1][0])
            assert_same_contributions(out[0][1], out[1][1], tol=0.3, total=5e-2)
            assert (int(tra.D.depth), float(tra.D.alpha), tra.cur_nimg) == (int(trb.D.depth), float(trb.D.alpha), trb.cur_nimg)
            for a, b in ((tra.G, trb.G), (tra.D, trb.D)):
                with torch.no_grad():
                    b._flat_param.copy_(a._flat_param)
                b.mark_params_changed()
            for oa, ob in ((tra.optimizer_g, trb.optimizer_g), (tra.optimizer_d, trb.optimizer_d)):
                for (_, ma, va), (_, mb_, vb) in zip(oa._flat.values(), ob._flat.values()):
                    mb_.copy_(ma)
                    vb.copy_(va)
            if it == 9:                                        # a pass for iteration 10 is pending on the second stream right now
                torch.save(tra.D, str(tmp_path / 'd.dat'))
                assert torch.load(str(tmp_path / 'd.dat'), weights_only=False).__dict__.get('_early_real') is None
        # the trainer that looks ahead has drawn at most one batch more, and the batches it CONSUMED are the other one's, in order
        assert seen_a[:len(seen_b)] == seen_b and len(seen_a) - len(seen_b) in (0, 1)
        st = eng.EARLY_STATS
        assert st['passes'] >= 5 and st['used'] >= st['passes'] - 2 and st['dropped'] <= 2, st       # 5 per stabilisation span (depth 0 replays a hipGraph in 'auto': none there)
    finally:
        wl.enable_graphs(False)
        eng.FAKE_THIRD_ON_SIDE = fake_side_before


def test_call_hook_sees_the_same_launches_eager_and_replayed():
    """``_lib.CALL_HOOK`` (bench.py's per-launch timing) runs on every C-ABI call of an eager step AND of a step replayed from a launch plan:
    the same entry points the same number of times, with the stream handle as the last argument, and the results with the hook in
    place equal those without it."""
    wl = pg.wgan_gp_loss
    torch.manual_seed(31)
    shape = (1, 3, 32, 32)
    kw = dict(fmap_base=512, fmap_max=64)
    G = pg.Generator(shape, latent_size=64, **kw).cuda()
    D = pg.Discriminator(shape, **kw).cuda()
    G.depth = D.depth = 3
    gen = torch.Generator(device='cuda').manual_seed(2)
    real = torch.rand((6, 3, 32, 32), device=DEV, generator=gen) * 2 - 1
    z = torch.randn((6, 64), device=DEV, generator=gen)
    mix = torch.rand((6, 1), device=DEV, generator=gen)
    seen = {}

    def hook(fn, args, name):
        seen.setdefault(mode_tag[0], []).append((name, len(args)))
        assert args[-1] is None or isinstance(args[-1], int)
        return fn(*args)

    def step():
        wl.set_mixing_factors(mix)
        c = pg.wgan_gp_D_loss(D, G, real, z)[0]
        c.backward()
        torch.cuda.synchronize()
        return float(c), D._flat_grad.clone()
    mode_tag = ['none']
    pg.plans.clear()
    try:
        wl._use_graphs = False
        ref = step()
        pg._lib.CALL_HOOK = hook
        mode_tag[0] = 'eager'
        eager = step()
        wl._use_graphs = 'auto'
        mode_tag[0] = 'warm'
        for _ in range(3):                                     # two eager warm-up calls, one recorded
            step()
        mode_tag[0] = 'replay'
        replay = step()
        assert pg.plans.STATS['replayed'] >= 1
    finally:
        pg._lib.CALL_HOOK = None
        wl.enable_graphs(False)
        pg.plans.clear()
    assert abs(eager[0] - ref[0]) <= 2e-4 * max(1.0, abs(ref[0])) and abs(replay[0] - ref[0]) <= 2e-4 * max(1.0, abs(ref[0]))
    assert _l2(eager[1], ref[1].cpu()) < 2e-3 and _l2(replay[1], ref[1].cpu()) < 2e-3
    conv = lambda calls: sorted(n for n, _ in calls if n.startswith('pg_conv2d'))
    assert conv(seen['replay']) == conv(seen['eager']) and len(conv(seen['replay'])) > 20


@pytest.mark.parametrize('plans_on', [False, True])
def test_three_pass_d_forward_matches_whole_batch_forward(plans_on):
    """engine.REAL_THIRD_IN_STEP (default): real | fake | mixed thirds of the D forward as three passes into one set of batched tensors,
    two of them on the second stream, against the one-pass forward of the whole batch (PGGAN_REAL_SIDE=0): same losses, every
    parameter of D receives the same contributions, over several steps of the public loss -> backward loop (eager and replayed)."""
    wl, eng = pg.wgan_gp_loss, pg.engine
    before = eng.REAL_THIRD_IN_STEP
    wl.enable_graphs(False)
    wl._use_graphs = 'auto' if plans_on else False
    try:
        res = []
        for three in (True, False):
            eng.REAL_THIRD_IN_STEP = three
            pg.plans.clear()
            torch.manual_seed(23)
            shape = (1, 3, 32, 32)
            kw = dict(fmap_base=512, fmap_max=64)
            G = pg.Generator(shape, latent_size=64, **kw).cuda()
            D = pg.Discriminator(shape, **kw).cuda()
            G.depth = D.depth = 3
            ds = pg.utils.SyntheticDataset(32, 3, seed=9)
            ds.model_depth = 3
            it = ds.loader(6)
            lat = pg.utils.device_latents(6, 64, seed=4)
            out = []
            for step in range(5):
                wl.manual_seed(300 + step)
                c, rl, fl = pg.wgan_gp_D_loss(D, G, next(it), lat())
                c.backward()
                out.append((float(c), rl.clone(), fl.clone(), grads_by_name(D)))
            res.append(out)
        for (ca, ra, fa, ga), (cb, rb, fb, gb) in zip(*res):
            assert abs(ca - cb) <= 2e-4 * max(1.0, abs(cb))                # (different launch groupings: K-slice counts, summation orders)
            assert _l2(ra, rb.cpu()) < 2e-4 and _l2(fa, fb.cpu()) < 2e-4
            assert_same_contributions(ga, gb)
    finally:
        eng.REAL_THIRD_IN_STEP = before
        wl.enable_graphs(False)
        pg.plans.clear()


@pytest.mark.parametrize('mode', ['side', 'batched'])
def test_early_g_forward_is_the_same_pass(mode, deterministic_forward, monkeypatch):
    """engine.request_early_g (round 6): the generator pass that opens the G step, enqueued on the second stream inside the D step, must be
    the pass the G step would have run itself -- same kernels, same inputs.  Engine level, no optimizer in between: the G cost is
    bit-identical and G's gradients agree to the atomic commit order of the weight gradients; the pass is taken exactly when the latents
    tensor and the generator's weights / stage are the ones it was computed with."""
    eng, wl = pg.engine, pg.wgan_gp_loss
    wl.enable_graphs(False)
    monkeypatch.setattr(eng, 'EARLY_G_MIN_RES', 4)                # (default: from 256x256 up, where it pays)
    monkeypatch.setattr(eng, 'EARLY_G_MODE', mode)
    # 'side': the very same 3-image launches on another stream -> bit-identical.  'batched': one 6-image pass [z | z'] -- other K-slice
    # counts in the small-map Winograd launches, i.e. another (equally valid) fp32 summation order: last-bit differences in the fake images
    # and the occasional LeakyReLU flip downstream
    same = mode == 'side'
    torch.manual_seed(12)
    shape = (1, 3, 64, 64)
    kw = dict(fmap_base=1024, fmap_max=64)
    G = pg.Generator(shape, latent_size=64, **kw).cuda()
    D = pg.Discriminator(shape, **kw).cuda()
    G.depth = D.depth = 4
    gen = torch.Generator(device='cuda').manual_seed(3)
    real = torch.rand((6, 3, 64, 64), device=DEV, generator=gen) * 2 - 1
    z_d = torch.randn((6, 64), device=DEV, generator=gen)
    z_g = torch.randn((6, 64), device=DEV, generator=gen)
    mix = torch.rand((6, 1), device=DEV, generator=gen)

    def run(early, other_latents=False):
        stats = dict(eng.EARLY_G_STATS)
        if early:
            eng.request_early_g(D, G, z_g)
        c, _, _, st = eng.d_loss_forward(D, G, real, z_d, mix, 10.0, 0.001, 1.0)
        eng.d_loss_backward(st)
        gd = D._flat_grad.clone()
        zz = z_g.clone() if other_latents else z_g
        gc, gst = eng.g_loss_forward(G, D, zz)
        eng.g_loss_backward(gst)
        torch.cuda.synchronize()
        return float(c), gd, float(gc), G._flat_grad.clone(), {k: eng.EARLY_G_STATS[k] - stats[k] for k in stats}
    base = run(False)
    assert base[4] == dict(passes=0, used=0, dropped=0)
    got = run(True)
    assert got[4] == dict(passes=1, used=1, dropped=0), got[4]
    if same:
        assert got[0] == base[0] and got[2] == base[2], (got[0], base[0], got[2], base[2])   # bit-identical losses
        assert _l2(got[1], base[1].cpu()) < 2e-5 and _l2(got[3], base[3].cpu()) < 2e-5
    else:
        assert abs(got[0] - base[0]) <= 2e-4 * max(1.0, abs(base[0])) and abs(got[2] - base[2]) <= 2e-4 * max(1.0, abs(base[2]))
        assert _l2(got[1], base[1].cpu()) < 5e-3 and _l2(got[3], base[3].cpu()) < 5e-3
    miss = run(True, other_latents=True)                         # another latents tensor (equal values): the pass is not taken
    assert miss[4] == dict(passes=1, used=0, dropped=1), miss[4]
    if same:
        assert miss[2] == base[2] and _l2(miss[3], base[3].cpu()) < 2e-5
    else:
        assert abs(miss[2] - base[2]) <= 2e-4 * max(1.0, abs(base[2])) and _l2(miss[3], base[3].cpu()) < 5e-3
    assert G.__dict__.get('_early_fwd') is None and D.__dict__.get('_early_g_request') is None


@pytest.mark.parametrize('plans_on,mode', [(False, 'side'), (True, 'side'), (False, 'batched'), (True, 'batched')])
def test_trainer_early_g_forward_matches_in_step_forward(plans_on, mode, deterministic_forward, monkeypatch):
    """The same through ``Trainer`` (which draws the G step's latents ahead of the D loss: same position in the latents sequence) with eager
    and plan-replayed steps, through a growth-stage change: a trainer with the early generator pass against a twin without it, stepped side
    by side at identical weights -- per-iteration pre-Adam gradients; the pass is really taken in every iteration (none dropped)."""
    wl, eng = pg.wgan_gp_loss, pg.engine
    before = eng.EARLY_G_FORWARD
    monkeypatch.setattr(eng, 'EARLY_G_MIN_RES', 4)                # (default: from 256x256 up, where it pays)
    monkeypatch.setattr(eng, 'EARLY_G_MODE', mode)

    def build():
        torch.manual_seed(21)
        shape = (1, 3, 32, 32)
        kw = dict(fmap_base=512, fmap_max=64)
        G = pg.Generator(shape, latent_size=64, **kw).cuda()
        D = pg.Discriminator(shape, **kw).cuda()
        G.depth = D.depth = 2
        opt_g, opt_d = pg.FusedAdam(G.parameters(), 0.001, betas=(0.0, 0.99)), pg.FusedAdam(D.parameters(), 0.001, betas=(0.0, 0.99))
        ds = pg.utils.SyntheticDataset(32, 3, seed=5)
        ds.model_depth = 2
        return pg.Trainer(D, G, pg.wgan_gp_D_loss, pg.wgan_gp_G_loss, opt_d, opt_g, ds, ds.loader(8), pg.utils.device_latents(8, 64, seed=3)), ds
    wl.enable_graphs('auto')
    wl._use_plans = plans_on
    pg.plans.STATS.update(recorded=0, replayed=0)
    try:
        (tra, dsa), (trb, dsb) = build(), build()
        s0 = dict(eng.EARLY_G_STATS)
        for it in range(10):
            if it == 6:
                for tr, ds in ((tra, dsa), (trb, dsb)):
                    tr.G.depth = tr.D.depth = ds.model_depth = 3
                    tr.dataiter = ds.loader(8)
            out = []
            for tr, early in ((tra, True), (trb, False)):
                eng.EARLY_G_FORWARD = early
                wl.manual_seed(100 + it)
                tr.train()
                torch.cuda.synchronize()
                out.append((grads_by_name(tr.D), grads_by_name(tr.G)))
            if mode == 'side':
                assert_same_contributions(out[0][0], out[1][0], tol=1e-2, total=2e-4)
            else:                                              # (the batched pass: the fake images differ in the last bits -> LeakyReLU flips in D)
                assert_same_contributions(out[0][0], out[1][0])
            assert_same_contributions(out[0][1], out[1][1], tol=0.3, total=5e-2)     # (through D after its update: sign-like Adam on round-off noise)
            for a, b in ((tra.G, trb.G), (tra.D, trb.D)):
                assert float((a._flat_param - b._flat_param).abs().max()) <= 2 * _adam_step_bound(0.001, it + 1) + 1e-6
                with torch.no_grad():
                    b._flat_param.copy_(a._flat_param)
                b.mark_params_changed()
            for oa, ob in ((tra.optimizer_g, trb.optimizer_g), (tra.optimizer_d, trb.optimizer_d)):
                for (_, ma, va), (_, mb_, vb) in zip(oa._flat.values(), ob._flat.values()):
                    mb_.copy_(ma)
                    vb.copy_(va)
        d = {k: eng.EARLY_G_STATS[k] - s0[k] for k in s0}
        assert d == dict(passes=10, used=10, dropped=0), d
        if plans_on:
            assert pg.plans.STATS['recorded'] >= 4 and pg.plans.STATS['replayed'] >= 8, pg.plans.STATS
    finally:
        eng.EARLY_G_FORWARD = before
        wl._use_plans = True
        wl.enable_graphs(False)


def test_whole_module_pickle_roundtrip(tmp_path):
    """SaverPlugin semantics (plugins.py:155-166): ``torch.save(model)`` / ``torch.load`` of whole modules must
    preserve weights AND the equalized-lr constants c (not in the reference's state_dict), and the reloaded
    nets must keep training (flat buffers re-linked)."""
    torch.manual_seed(4)
    shape = (1, 3, 16, 16)
    kw = dict(fmap_base=128, fmap_max=32)
    G = pg.Generator(shape, latent_size=32, **kw).to(DEV)
    D = pg.Discriminator(shape, **kw).to(DEV)
    G.depth = D.depth = 2
    G.alpha = D.alpha = 0.7
    z = torch.randn(4, 32, device=DEV)
    out0 = G(z)
    s0 = D(out0)
    torch.save(G, str(tmp_path / 'g.dat'))
    torch.save(D, str(tmp_path / 'd.dat'))
    G2 = torch.load(str(tmp_path / 'g.dat'), weights_only=False)
    D2 = torch.load(str(tmp_path / 'd.dat'), weights_only=False)
    assert (G2.depth, G2.alpha, G2.block0.c1.c) == (2, 0.7, G.block0.c1.c)
    assert torch.equal(G2(z), out0) and torch.equal(D2(out0), s0)
    real = torch.rand(4, 3, 16, 16, device=DEV) * 2 - 1
    opt = pg.FusedAdam(D2.parameters(), 0.001, betas=(0.0, 0.99))
    before = D2._flat_param.clone()
    c, _, _ = pg.wgan_gp_D_loss(D2, G2, real, z)
    c.backward()
    opt.step()
    assert not torch.equal(before, D2._flat_param)
    assert torch.isfinite(D2._flat_param).all()
    assert D2.blocks[-1].c2.conv.weight.data_ptr() >= D2._flat_param.data_ptr()      # still views of the flat buffer


def test_non_default_flags_fixture_gpu():
    """ReLU / no wscale / no PixelNorm variants against the reference's own run."""
    meta, data = load_fixture('flags16')
    for case in meta['cases']:
        tag = case['tag']
        G, D = build_flag_nets(meta, case, DEV)
        load_fixture_params(G, data, tag + '/G')
        load_fixture_params(D, data, tag + '/D')
        _check_case(G, D, data, tag, case, meta['cfg'])


def test_batch_size_edge_cases(oracle):
    """Minibatch 1 and odd sizes (the reference schedule uses 3, 6, 14): D/G outputs and D_cost vs the oracle."""
    torch.manual_seed(5)
    shape = (1, 1, 16, 16)
    kw = dict(fmap_base=128, fmap_max=32)
    G = pg.Generator(shape, latent_size=32, **kw)
    D = pg.Discriminator(shape, **kw)
    gp, dp = G.reference_state_dict(), D.reference_state_dict()
    G.to(DEV); D.to(DEV)
    cfg = oracle.NetCfg(16, 1, latent_size=32, **kw)
    G.depth = D.depth = 2
    for n in (1, 3, 7, 14):
        real, z_d, z_g, mix = oracle.synthetic_batch(60 + n, n, 1, 16, 32)
        pg.wgan_gp_loss.set_mixing_factors(mix)
        d_cost, rl, fl = pg.wgan_gp_D_loss(D, G, real.to(DEV), z_d.to(DEV))
        d_cost.backward()
        ref = oracle.d_loss_and_grads(dp, gp, cfg, real, z_d, mix, 2, 1.0)
        assert rel_err(d_cost, ref['D_cost']) < OUT_TOL and rel_err(rl, ref['D_real_loss']) < OUT_TOL
        _check_grads_loose(reference_grads(D), ref['grads'], 'D grads n=%d' % n)
    with pytest.raises((RuntimeError, ValueError)):
        D(torch.zeros(2, 1, 8, 8, device=DEV))            # wrong resolution for depth 2
    with pytest.raises(RuntimeError):
        D(torch.zeros(2, 1, 16, 16))                      # CPU tensor: no fallback


def test_fused_adam_state_roundtrip():
    torch.manual_seed(1)
    D = pg.Discriminator((1, 3, 8, 8), fmap_base=64, fmap_max=16).to(DEV)
    G = pg.Generator((1, 3, 8, 8), fmap_base=64, fmap_max=16, latent_size=16).to(DEV)
    D.depth = G.depth = 1
    opt = pg.FusedAdam(D.parameters(), 0.001, betas=(0.0, 0.99))
    real, z = torch.rand(4, 3, 8, 8, device=DEV), torch.randn(4, 16, device=DEV)
    for _ in range(2):
        c, _, _ = pg.wgan_gp_D_loss(D, G, real, z); c.backward(); opt.step()
    sd = opt.state_dict()
    opt2 = pg.FusedAdam(D.parameters(), 0.001, betas=(0.0, 0.99))
    opt2.load_state_dict(sd)
    pg.wgan_gp_loss.set_mixing_factors(torch.full((4, 1), 0.3))
    c, _, _ = pg.wgan_gp_D_loss(D, G, real, z); c.backward()
    snap = D._flat_param.clone()
    opt.step(); after1 = D._flat_param.clone()
    D._flat_param.copy_(snap); D.mark_params_changed()
    opt2.step()
    assert torch.allclose(after1, D._flat_param, rtol=0, atol=1e-7)


def test_fmap_decay_and_latent_size_none(oracle):
    """fmap_decay != 1 and latent_size=None (-> nf(0), network.py:97-98) vs the oracle."""
    torch.manual_seed(9)
    shape = (1, 3, 16, 16)
    kw = dict(fmap_base=1024, fmap_decay=2.0, fmap_max=64)           # nf(0..3) = 64, 64, 64, 16
    G = pg.Generator(shape, latent_size=None, **kw)
    D = pg.Discriminator(shape, **kw)
    assert G.latent_size == 64
    gp, dp = G.reference_state_dict(), D.reference_state_dict()
    G.to(DEV); D.to(DEV)
    cfg = oracle.NetCfg(16, 3, latent_size=None, **kw)
    assert cfg.latent_size == 64 and cfg.nf(3) == 16
    G.depth = D.depth = 2
    G.alpha = D.alpha = 0.4
    real, z_d, z_g, mix = oracle.synthetic_batch(77, 4, 3, 16, 64)
    pg.wgan_gp_loss.set_mixing_factors(mix)
    d_cost, rl, fl = pg.wgan_gp_D_loss(D, G, real.to(DEV), z_d.to(DEV))
    d_cost.backward()
    ref = oracle.d_loss_and_grads(dp, gp, cfg, real, z_d, mix, 2, 0.4)
    assert rel_err(d_cost, ref['D_cost']) < OUT_TOL
    _check_grads_loose(reference_grads(D), ref['grads'], 'D grads fmap_decay')
    g_cost = pg.wgan_gp_G_loss(G, D, z_g.to(DEV))
    g_cost.backward()
    refg = oracle.g_loss_and_grads(gp, dp, cfg, z_g, 2, 0.4)
    assert rel_err(g_cost, refg['G_cost']) < OUT_TOL
    _check_grads_loose(reference_grads(G), refg['grads'], 'G grads fmap_decay')


def test_d_training_repeats(oracle):
    """Trainer(D_training_repeats=2) (trainer.py:17,90-103): two D steps on fresh real batches / latents per G
    step, cur_nimg advanced per D step; two iterations against the oracle's op sequence."""
    torch.manual_seed(11)
    shape = (1, 3, 8, 8)
    kw = dict(fmap_base=64, fmap_max=16)
    G = pg.Generator(shape, latent_size=16, **kw)
    D = pg.Discriminator(shape, **kw)
    gp, dp = G.reference_state_dict(), D.reference_state_dict()
    G.to(DEV); D.to(DEV)
    G.depth = D.depth = 1
    cfg = oracle.NetCfg(8, 3, latent_size=16, **kw)
    rs = np.random.RandomState(3)
    reals = [torch.from_numpy(rs.rand(4, 3, 8, 8).astype(np.float32) * 2 - 1) for _ in range(4)]
    zs = [torch.from_numpy(rs.randn(4, 16).astype(np.float32)) for _ in range(6)]
    mixes = [torch.from_numpy(rs.rand(4, 1).astype(np.float32)) for _ in range(4)]
    it_r, it_z, it_m = iter(reals), iter(zs), iter(mixes)

    def d_loss(Dm, Gm, real, z):
        pg.wgan_gp_loss.set_mixing_factors(next(it_m))
        return pg.wgan_gp_D_loss(Dm, Gm, real, z)
    opt_g = pg.FusedAdam(G.parameters(), 0.001, betas=(0.0, 0.99))
    opt_d = pg.FusedAdam(D.parameters(), 0.001, betas=(0.0, 0.99))
    tr = pg.Trainer(D, G, d_loss, pg.wgan_gp_G_loss, opt_d, opt_g, None, it_r, lambda: next(it_z), D_training_repeats=2)
    tr.train(); tr.train()
    assert tr.cur_nimg == 16 and tr.iterations == 2
    # oracle: same sequence
    og, od = oracle.AdamState(), oracle.AdamState()
    r, z, m = iter(reals), iter(zs), iter(mixes)
    for _ in range(2):
        lat = next(z)
        for _ in range(2):
            d = oracle.d_loss_and_grads(dp, gp, cfg, next(r), lat, next(m), 1, 1.0)
            od.step(dp, d['grads'], 0.001)
            lat = next(z)
        g = oracle.g_loss_and_grads(gp, dp, cfg, lat, 1, 1.0)
        og.step(gp, g['grads'], 0.001)
    for net, ref in ((G, gp), (D, dp)):
        for k, v in net.reference_state_dict().items():
            if torch.is_tensor(v):
                assert rel_err(v.cpu(), ref[k]) < 5e-3, k


def test_full_schedule_soak_reference_widths():
    """The 1024^2 net at the reference widths driven by Trainer + DepthManager + LRScheduler through EVERY growth
    stage 0..8 including all fade-in phases (shortened schedule, reference per-depth minibatches 16/14/6/3):
    every kernel dispatch path at every resolution runs with real shapes; losses must stay finite, the schedule
    must match the closed form, and gradients must have reached every parameter that was ever active."""
    torch.manual_seed(1337)
    np.random.seed(7)
    shape = (1, 3, 1024, 1024)
    G = pg.Generator(shape).to(DEV)
    D = pg.Discriminator(shape).to(DEV)
    opt_g = pg.FusedAdam(G.parameters(), 0.001, betas=(0.0, 0.99))
    opt_d = pg.FusedAdam(D.parameters(), 0.001, betas=(0.0, 0.99))
    ds = pg.utils.SyntheticDataset(1024, 3, seed=5)
    lat = lambda mb: (lambda: pg.utils.random_latents(mb, 512))
    tr = pg.Trainer(D, G, pg.wgan_gp_D_loss, pg.wgan_gp_G_loss, opt_d, opt_g, ds, ds.loader(16), lat(16))
    dm = pg.DepthManager(ds.loader, lat, 8, lod_training_nimg=64, lod_transition_nimg=64)
    tr.register_plugin(dm)
    ramp = lambda nimg: pg.utils.rampup(nimg, 1)
    tr.register_plugin(pg.LRScheduler(pg.RampupLR(opt_d, ramp), pg.RampupLR(opt_g, ramp)))
    seen, losses = set(), []

    class Rec(pg.Plugin):
        def __init__(self):
            super(Rec, self).__init__([(1, 'iteration')])

        def register(self, trainer):
            pass

        def iteration(self, i, g_cost, d_cost, d_real, d_fake):
            losses.append((float(g_cost), float(d_cost)))
            seen.add((G.depth, G.alpha < 1.0))
    tr.register_plugin(Rec())
    for q in tr.plugin_queues.values():
        heapq.heapify(q)
    p0 = D._flat_param.clone()
    while tr.cur_nimg < 8 * 128 + 64 + 16:
        tr.train()
        full, rem = divmod(tr.cur_nimg, 128)             # the DepthManager plugin has already set up the NEXT iteration
        tp, rem2 = divmod(rem, 64)
        depth = min(8, full + tp)
        assert G.depth == D.depth == depth
        assert G.alpha == (rem2 / 64 if (tp > 0 and full + tp == depth) else 1.0)
    torch.cuda.synchronize()
    assert all(np.isfinite(a) and np.isfinite(b) for a, b in losses), losses[-5:]
    assert {d for d, _ in seen} == set(range(9)) and any(f for _, f in seen)
    assert torch.isfinite(D._flat_param).all() and torch.isfinite(G._flat_param).all()
    moved = (D._flat_param != p0)
    assert float(moved.float().mean()) > 0.99            # every D parameter (incl. all fromRGB layers) was updated


@pytest.mark.gpu
def test_deferred_d_update_matches_inline(monkeypatch, deterministic_forward):
    """Trainer runs the tail of the D update (all-reduce, Adam, derived weights) on the second stream under the G
    forward of the G step (engine.defer_to_side).  Two trainers, overlap on / off, are stepped side by side on the same batches;
    after every iteration the PRE-ADAM gradients are compared tensor by tensor and the second trainer is re-synchronised to the
    first (weights and Adam moments), so that every iteration is compared at identical weights -- no end-state bound has to absorb
    sign-like Adam steps on round-off-sized gradient elements.  The deferred path must really be taken."""
    rs = np.random.RandomState(5)
    ITERS = 3
    reals = [torch.from_numpy(rs.rand(4, 3, 64, 64).astype(np.float32) * 2 - 1) for _ in range(ITERS)]
    zs = [torch.from_numpy(rs.randn(4, 64).astype(np.float32)) for _ in range(2 * ITERS)]
    mixes = [torch.from_numpy(rs.rand(4, 1).astype(np.float32)) for _ in range(ITERS)]
    state = dict(it=0)

    def build():
        torch.manual_seed(23)
        shape = (1, 3, 64, 64)
        kw = dict(fmap_base=512, fmap_max=64)
        G = pg.Generator(shape, latent_size=64, **kw).to(DEV)
        D = pg.Discriminator(shape, **kw).to(DEV)
        G.depth = D.depth = 4
        zi = [0]

        def d_loss(Dm, Gm, real, z):
            pg.wgan_gp_loss.set_mixing_factors(mixes[state['it']])
            return pg.wgan_gp_D_loss(Dm, Gm, real, z)

        def loader():
            while True:
                yield reals[state['it']]

        def rlg():
            z = zs[2 * state['it'] + zi[0] % 2]
            zi[0] += 1
            return z
        opt_g = pg.FusedAdam(G.parameters(), 0.001, betas=(0.0, 0.99))
        opt_d = pg.FusedAdam(D.parameters(), 0.001, betas=(0.0, 0.99))
        return pg.Trainer(D, G, d_loss, pg.wgan_gp_G_loss, opt_d, opt_g, None, loader(), rlg), opt_g, opt_d
    (tra, oga, oda), (trb, ogb, odb) = build(), build()
    taken = []
    orig = pg.engine.defer_to_side
    monkeypatch.setattr(pg.engine, 'defer_to_side', lambda net, fn: (taken.append(1), orig(net, fn))[1])
    for it in range(ITERS):
        state['it'] = it
        monkeypatch.setenv('PGGAN_OVERLAP_D_UPDATE', '1')
        tra.train()
        n_def = len(taken)
        monkeypatch.setenv('PGGAN_OVERLAP_D_UPDATE', '0')
        trb.train()
        assert n_def == it + 1 and len(taken) == n_def          # deferred in A, inline in B
        torch.cuda.synchronize()
        assert getattr(tra.D, '_pending', None) is None
        # D's gradients: computed at identical weights -> equal up to the order of the atomic weight-gradient commits, tensor by
        # tensor (a dropped or re-ordered contribution of ONE small layer shows here).  G's are computed through D AFTER its
        # update, where a sign-like Adam has already turned that noise into +-lr differences of near-zero elements: looser.
        assert_same_contributions(grads_by_name(tra.D), grads_by_name(trb.D), tol=1e-2, total=2e-4)     # (deterministic_forward: no LeakyReLU flip to absorb)
        assert_same_contributions(grads_by_name(tra.G), grads_by_name(trb.G), tol=0.3, total=5e-2)
        for a, b in ((tra.G, trb.G), (tra.D, trb.D)):           # this iteration's updates: a flipped near-zero element moves by at most twice the step bound
            assert float((a._flat_param - b._flat_param).abs().max()) <= 2 * _adam_step_bound(0.001, it + 1) + 1e-6
            assert _l2(a._flat_param, b._flat_param.cpu()) < 3e-3
            with torch.no_grad():
                b._flat_param.copy_(a._flat_param)               # re-synchronise B: the next iteration starts from identical weights ...
            b.mark_params_changed()
        for oa, ob in ((oga, ogb), (oda, odb)):                  # ... and identical Adam moments
            for (_, ma, va), (_, mb_, vb) in zip(oa._flat.values(), ob._flat.values()):
                mb_.copy_(ma)
                vb.copy_(va)
    monkeypatch.setattr(pg.engine, 'defer_to_side', orig)


@pytest.mark.gpu
def test_config2_grow_run_against_oracle(oracle):
    """BASELINE.json config 2 as written: the 32x32 network (default 512-channel widths) grown depth 0 -> 3 with alpha
    fade-ins at minibatch 64, through Trainer + DepthManager + LRScheduler + FusedAdam, against the oracle's
    ``train_iteration`` driven by the oracle's own schedule (lod spans shortened to 2 iterations so that every stage and
    every fade occurs: 14 iterations, 192 stacked images per D pass; the oracle follows the first 11, see tests/_config2_oracle.py)."""
    import _config2_oracle as c2
    N, LOD, ITERS, RAMP = c2.N, c2.LOD, c2.ITERS, c2.RAMP
    shape = c2.SHAPE
    G, D = c2.initial_nets()
    G.to(DEV)
    D.to(DEV)
    cfg = oracle.NetCfg(32, 3)
    sched = [oracle.depth_schedule(it * N, 3, LOD, LOD, minibatch_default=N) for it in range(ITERS)]
    assert sorted(set(d for d, _, _, _ in sched)) == [0, 1, 2, 3] and sum(1 for _, a, _, _ in sched if a < 1.0) == 6
    batches = [c2.batch(oracle, sched, it) for it in range(ITERS)]
    state = dict(it=0, z=0)

    class Data(object):
        model_depth, alpha = 0, 1.0

    def make_loader(mb):
        assert mb == N

        def gen():
            while True:
                yield batches[state['it']][0]
        return gen()

    def make_rlg(mb):
        def f():
            b = batches[state['it']]
            state['z'] += 1
            return b[1] if state['z'] % 2 == 1 else b[2]
        return f

    def d_loss(Dm, Gm, real, z):
        pg.wgan_gp_loss.set_mixing_factors(batches[state['it']][3])
        return pg.wgan_gp_D_loss(Dm, Gm, real, z)
    opt_g = pg.FusedAdam(G.parameters(), 0.001, betas=(0.0, 0.99))
    opt_d = pg.FusedAdam(D.parameters(), 0.001, betas=(0.0, 0.99))
    ramp = lambda nimg: pg.utils.rampup(nimg, RAMP)
    tr = pg.Trainer(D, G, d_loss, pg.wgan_gp_G_loss, opt_d, opt_g, Data(), None, None)
    tr.register_plugin(pg.DepthManager(make_loader, make_rlg, 3, minibatch_default=N, lod_training_nimg=LOD, lod_transition_nimg=LOD))
    tr.register_plugin(pg.LRScheduler(pg.RampupLR(opt_d, ramp), pg.RampupLR(opt_g, ramp)))
    losses = []

    class Rec(pg.Plugin):
        def __init__(self):
            super(Rec, self).__init__([(1, 'iteration')])

        def register(self, trainer):
            pass

        def iteration(self, i, g_cost, d_cost, d_real, d_fake):
            losses.append((float(g_cost), float(d_cost)))
    # Rec is registered last: it fires after DepthManager / LRScheduler have prepared the NEXT iteration
    tr.register_plugin(Rec())
    for q in tr.plugin_queues.values():
        heapq.heapify(q)
    # PRE-ADAM gradients, iteration by iteration, against the oracle evaluated on the HIP run's OWN weights of that iteration (what the
    # round-3 review asked for instead of end-state bounds that have to absorb 14 sign-like Adam steps): one iteration of every stage
    # and fade up to 16x16.  The D weights the G step sees are snapshotted when the G loss is entered (after D's deferred update).
    check_at = set()
    seen = set()
    for it in range(ITERS):
        key = (sched[it][0], sched[it][1] < 1.0)
        if key not in seen:
            seen.add(key)
            if key in ((0, False), (1, True), (2, True), (2, False)):     # (the 32x32 iterations at minibatch 64 cost ~40 s of oracle time each:
                check_at.add(it)                                        #  that stage has its own gradient checks in test_full_width_res32_golden)
    snap = {}

    def g_loss(Gm, Dm, z):
        if state['it'] in check_at:
            pg.engine.wait_pending(Dm)
            snap['dp_after'] = _cpu_sd(Dm.reference_state_dict())
        return pg.wgan_gp_G_loss(Gm, Dm, z)
    tr.G_loss = g_loss
    for it in range(ITERS):
        depth, alpha, mb, _ = sched[it]
        assert (tr.cur_nimg, int(G.depth), repr(float(G.alpha)), tr.stats['minibatch_size']) == (it * N, depth, repr(alpha), mb)
        state['it'], state['z'] = it, 0
        if it in check_at:
            gp_h, dp_h = _cpu_sd(G.reference_state_dict()), _cpu_sd(D.reference_state_dict())
        tr.train()
        if it == c2.ORACLE_ITERS - 1:
            torch.cuda.synchronize()
            mine_at = {'G': _cpu_sd(G.reference_state_dict()), 'D': _cpu_sd(D.reference_state_dict())}
        if it in check_at:
            real, z_d, z_g, mix = batches[it]
            # Max-norm guard 5e-2 here (3e-2 elsewhere).  Eight checks per run on weights that differ from run to run (14 Adam(beta1 = 0)
            # steps on gradients with atomic-order noise): every run draws fresh LeakyReLU-branch coincidences against the oracle.  Round 6,
            # final code, 16 runs of this test: worst rel-max per check 1e-4 .. 1.3e-2 in 14 of them, and twice ONE element of ONE tensor beyond
            # 3e-2 -- ('it 6 (depth 2 alpha 0.00) G step', 'block0.c1.conv.weight', rel-L2 1.29e-3, rel-max 3.22e-2): a single sample's branch in
            # the 4x4 block moves one entry of a 64-sample sum by a few per cent of the tensor's largest entry while the tensor's L2 distance stays
            # at 1.3e-3.  The L2 bounds (1e-2 per tensor, 4e-3 over all tensors) are unchanged: they are what a wrong mask or scale (O(1e-1)) meets.
            rd = oracle.d_loss_and_grads(dp_h, gp_h, cfg, real, z_d, mix, depth, alpha)
            _check_grads_loose(reference_grads(D), rd['grads'], 'config 2 it %d (depth %d alpha %.2f) D step' % (it, depth, alpha), max_tol=5e-2)
            rg = oracle.g_loss_and_grads(gp_h, snap['dp_after'], cfg, z_g, depth, alpha)
            _check_grads_loose(reference_grads(G), rg['grads'], 'config 2 it %d (depth %d alpha %.2f) G step' % (it, depth, alpha), max_tol=5e-2)
    # the oracle's own trajectory over the first ORACLE_ITERS iterations (tests/_config2_oracle.py)
    ref = c2.trajectory()
    gp, dp = ref['gp'], ref['dp']
    for it in range(c2.ORACLE_ITERS):
        gc, dc = losses[it]
        d_cost, g_cost = ref['losses'][it]
        # later iterations inherit sign-like Adam(beta1=0) steps on round-off-sized gradients and single LeakyReLU branch flips
        # (tests/test_fp64_adjudicator.py): the two training trajectories drift apart like any two fp32 runs of a GAN
        tol = 5e-4 if it == 0 else (3e-3 if it <= 3 else 1.5e-2)
        assert abs(dc - d_cost) < tol * max(1.0, abs(d_cost)), (it, dc, d_cost)
        assert abs(gc - g_cost) < tol * max(1.0, abs(g_cost)), (it, gc, g_cost)
    # state after ORACLE_ITERS iterations: a DRIFT GUARD only (two fp32 GAN trajectories; the per-iteration gradient checks above are the parity claim)
    for name, ref_sd in (('G', gp), ('D', dp)):
        mine = mine_at[name]
        for k, v in ref_sd.items():
            if torch.is_tensor(v):
                assert float((mine[k].cpu() - v).abs().max()) < 2 * 0.001 * c2.ORACLE_ITERS + 1e-4, (name, k)
                # (biases start at zero: after a dozen sign-like Adam steps their relative L2 distance between two fp32 trajectories is the
                # noisiest number of the run -- 1.4e-2 .. 2.02e-2 over repeated runs of the same build; the max-norm bound above is the claim)
                assert _l2(mine[k].cpu(), v) < (4e-2 if k.endswith('bias') else 2e-2), (name, k, _l2(mine[k].cpu(), v))
