"""Data-parallel path on CPU: world_size 2, gloo.  Each rank runs the product's Trainer on its own
shard (kernels emulated by tests/emu_ops.py — host logic only), gradients are SUM-all-reduced on the
flat buffers and averaged inside FusedAdam.  Checked against the oracle: two independent shard
gradients (local minibatch-stddev) averaged, then the oracle's Adam."""
import importlib
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_path, res, bucket_bytes):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import emu_ops
    import pggan_amd as pg
    from helpers import synthetic
    torch.set_num_threads(2)
    for modname in ('engine', 'optim'):
        importlib.import_module('pggan-pytorch_amd.' + modname).ops = emu_ops
    pg.engine._check_dev = lambda t, what: t.contiguous()
    pg.trainer._to_device = lambda t: t
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dp = pg.DataParallel()
    assert dp.world_size == world and dp.rank == rank
    torch.manual_seed(100 + rank)                      # deliberately different init per rank ...
    shape = (1, 3, res, res)
    kw = dict(fmap_base=64, fmap_max=16)
    G = pg.Generator(shape, latent_size=16, **kw)
    D = pg.Discriminator(shape, **kw)
    pg.parallel.MERGE_GAP = 0                          # tiny layers: keep the live spans apart so the span logic is exercised
    pg.parallel.BUCKET_BYTES = bucket_bytes            # small: several buckets leave while the backward sweep is still running
    dp.broadcast_params(G, D)                          # ... made identical by the broadcast from rank 0
    G.depth = D.depth = 2
    G.alpha = D.alpha = 0.5
    opt_g = pg.FusedAdam(G.parameters(), 0.001, betas=(0.0, 0.99), grad_scale=dp.grad_scale)
    opt_d = pg.FusedAdam(D.parameters(), 0.001, betas=(0.0, 0.99), grad_scale=dp.grad_scale)
    n = 3
    batches = [synthetic(pg.parallel.shard_seed(900 + 10 * it, rank), n, 3, 16, 16) for it in range(2)]
    state = dict(it=0)

    def loader():
        while True:
            yield batches[state['it']][0]

    zs = []

    def rlg():
        b = batches[state['it']]
        z = b[1] if len(zs) % 2 == 0 else b[2]
        zs.append(0)
        return z

    def d_loss(Dm, Gm, real, z):
        pg.wgan_gp_loss.set_mixing_factors(batches[state['it']][3])
        return pg.wgan_gp_D_loss(Dm, Gm, real, z)

    class DS(object):
        model_depth, alpha = 2, 0.5
    from helpers import reference_grads
    first = {}
    orig_all_reduce = dp.all_reduce_grads

    def recording_all_reduce(net, average=False):
        ex = getattr(net, '_grad_exchange', None)
        assert ex is not None and ex.started               # the Trainer opened a bucketed exchange for this sweep
        before = dp.stats['bytes']
        r = orig_all_reduce(net, average=average)
        key = 'D' if net is D else 'G'
        if key not in first:                               # summed (not yet averaged) gradients, iteration 0
            first[key] = {k: v.clone() for k, v in reference_grads(net).items()}
            first[key + '_reduced'] = (sum(e - s for s, e in pg.parallel.active_grad_spans(net)), net._flat_grad.numel())
            first[key + '_buckets'] = (ex.buckets, ex.sent_bytes, dp.stats['bytes'] - before)
        return r
    dp.all_reduce_grads = recording_all_reduce
    tr = pg.Trainer(D, G, d_loss, pg.wgan_gp_G_loss, opt_d, opt_g, DS(), loader(), rlg, parallel=dp)
    for it in range(2):
        state['it'] = it
        tr.train()
    assert tr.cur_nimg == 2 * n * world               # global images: the schedule stays a function of cur_nimg
    flat = torch.cat([G._flat_param, D._flat_param]).clone()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    for g in gathered:
        assert torch.equal(g, gathered[0]), 'ranks diverged'
    if rank == 0:
        torch.save(dict(G=G.reference_state_dict(), D=D.reference_state_dict(), first=first), out_path)
    dist.barrier()
    dist.destroy_process_group()


def _err(a, b):
    """max|a-b| / max(max|b|, 1e-3): linear.bias' gradient is a cancelling sum (~5e-5) of O(1/N) terms."""
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-3))


# res 32: the last growth stage is not live at depth 2 -> partial all-reduce.  bucket 1 KB: a bucket leaves after almost every
# block of the backward sweep; 16 MB: everything travels in the final flush.
@pytest.mark.parametrize('res,bucket_bytes', [(16, 1024), (32, 1024), (16, 16 << 20)])
def test_two_rank_data_parallel_matches_oracle(tmp_path, oracle, res, bucket_bytes):
    world = 2
    out_path = str(tmp_path / 'dp.pt')
    mp.spawn(_worker, args=(world, _free_port(), out_path, res, bucket_bytes), nprocs=world, join=True)
    got = torch.load(out_path, weights_only=False)
    for key in ('D', 'G'):
        buckets, sent, at_finish = got['first'][key + '_buckets']
        assert sent >= 4 * got['first'][key + '_reduced'][0] - 64          # every live gradient travelled (span padding aside)
        if bucket_bytes <= 1024:
            assert buckets >= 3 and at_finish < sent, (key, buckets, sent, at_finish)   # most of it before the sweep ended
        else:
            assert at_finish == sent, (key, buckets, sent, at_finish)
    # oracle: same start (rank 0's init), per-shard gradients averaged
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import pggan_amd as pg
    from helpers import synthetic
    from conftest import rel_err
    torch.manual_seed(100)
    shape = (1, 3, res, res)
    kw = dict(fmap_base=64, fmap_max=16)
    G = pg.Generator(shape, latent_size=16, **kw)
    D = pg.Discriminator(shape, **kw)
    gp, dp_ = G.reference_state_dict(), D.reference_state_dict()
    cfg = oracle.NetCfg(res, 3, latent_size=16, **kw)
    for key in ('D', 'G'):
        reduced, total = got['first'][key + '_reduced']
        assert reduced <= total and (total - reduced > 300) == (res > 16), (key, reduced, total)   # only the live layers travel
    og, od = oracle.AdamState(), oracle.AdamState()
    for it in range(2):
        shards = [synthetic(pg.parallel.shard_seed(900 + 10 * it, r), 3, 3, 16, 16) for r in range(world)]
        ds = [oracle.d_loss_and_grads(dp_, gp, cfg, real, z_d, mix, 2, 0.5) for (real, z_d, z_g, mix) in shards]
        avg = {k: sum(d['grads'][k] for d in ds) / world for k in ds[0]['grads']}
        if it == 0:      # the all-reduced buffer holds the SUM over ranks of the per-shard gradients
            assert sorted(avg) == sorted(got['first']['D'])
            for k in avg:
                assert _err(got['first']['D'][k], avg[k] * world) < 1e-3, ('D grad', k)
        od.step(dp_, avg, 0.001)
        gs = [oracle.g_loss_and_grads(gp, dp_, cfg, z_g, 2, 0.5) for (real, z_d, z_g, mix) in shards]
        avg = {k: sum(g['grads'][k] for g in gs) / world for k in gs[0]['grads']}
        if it == 0:
            assert sorted(avg) == sorted(got['first']['G'])
            for k in avg:
                assert _err(got['first']['G'][k], avg[k] * world) < 1e-3, ('G grad', k)
        og.step(gp, avg, 0.001)
    for name, ref, mine in (('G', gp, got['G']), ('D', dp_, got['D'])):
        for k, v in ref.items():
            if torch.is_tensor(v):
                # Adam with beta1=0 is sign-like in its first steps: a gradient within round-off of zero moves
                # a parameter by up to lr either way per step -> absolute bound 2*lr*steps on the end state
                assert float((mine[k] - v).abs().max()) < 2 * 0.001 * 2 + 1e-4, (name, k)


def _grow_worker(rank, world, port, out_path):
    """World-size-4 run THROUGH DepthManager stage boundaries: per-rank minibatch 4 -> 3 -> 2 (the reference's 16 -> 14 -> 6 -> 3
    pattern in small), loader / latent-generator switch, cur_nimg += world * N, fades in between."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import emu_ops
    import pggan_amd as pg
    torch.set_num_threads(1)
    for modname in ('engine', 'optim'):
        importlib.import_module('pggan-pytorch_amd.' + modname).ops = emu_ops
    pg.engine._check_dev = lambda t, what: t.contiguous()
    pg.trainer._to_device = lambda t: t
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dp = pg.DataParallel()
    torch.manual_seed(7 + rank)
    shape = (1, 3, 16, 16)
    kw = dict(fmap_base=32, fmap_max=8)
    G = pg.Generator(shape, latent_size=8, **kw)
    D = pg.Discriminator(shape, **kw)
    dp.broadcast_params(G, D)
    opt_g = pg.FusedAdam(G.parameters(), 0.001, betas=(0.0, 0.99))
    opt_d = pg.FusedAdam(D.parameters(), 0.001, betas=(0.0, 0.99))
    seed = pg.parallel.shard_seed(50, rank)
    made = []                                          # (minibatch) of every loader the DepthManager asked for

    class DS(object):
        model_depth, alpha = 0, 1.0

    ds = DS()

    def loader(n):
        made.append(n)
        gen = torch.Generator().manual_seed(seed + 1000 * len(made))

        def it():
            while True:
                r = 4 * 2 ** ds.model_depth
                yield torch.rand(n, 3, r, r, generator=gen) * 2 - 1
        return it()

    def rlg(n):
        gen = torch.Generator().manual_seed(seed + 77 + 1000 * len(made))
        return lambda: torch.randn(n, 8, generator=gen)

    pg.wgan_gp_loss.manual_seed(seed)
    tr = pg.Trainer(D, G, pg.wgan_gp_D_loss, pg.wgan_gp_G_loss, opt_d, opt_g, ds, None, None, parallel=dp)
    span = 3 * 4 * world                               # three iterations of the first stage per stabilise / fade span
    dm = pg.DepthManager(loader, rlg, 2, minibatch_default=4, minibatch_overrides={1: 3, 2: 2},
                         lod_training_nimg=span, lod_transition_nimg=span)
    tr.register_plugin(dm)
    log = []                                           # state every iteration RAN with: (cur_nimg before, depth, repr(alpha), minibatch)

    class Rec(pg.Plugin):
        def __init__(self):
            super(Rec, self).__init__([(1, 'iteration')])

        def register(self, trainer):
            pass

        def iteration(self, *a):
            log.append((tr.cur_nimg, int(tr.D.depth), repr(float(tr.D.alpha)), int(tr.stats['minibatch_size'])))
    tr.register_plugin(Rec())
    tr.run(4.6 * span / 1000.0)
    flat = torch.cat([G._flat_param, D._flat_param]).clone()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    for g in gathered:
        assert torch.equal(g, gathered[0]), 'ranks diverged'
    logs = [None] * world
    dist.all_gather_object(logs, (log, made, tr.cur_nimg, tr.iterations))
    if rank == 0:
        torch.save(dict(logs=logs, span=span), out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_four_rank_run_crosses_growth_stages_in_lockstep(tmp_path):
    """DP readiness without hardware (unmeasured on a multi-GPU node): four gloo ranks drive Trainer + DepthManager + FusedAdam
    through two stage changes with fades.  Every rank must see the same (cur_nimg, depth, alpha, minibatch) on every iteration,
    cur_nimg must advance by world * per-rank minibatch, the schedule must equal growth_stage() of that global counter, and
    each stage change must build exactly one new loader of the stage's per-rank minibatch on every rank."""
    world = 4
    out_path = str(tmp_path / 'grow.pt')
    mp.spawn(_grow_worker, args=(world, _free_port(), out_path), nprocs=world, join=True)
    got = torch.load(out_path, weights_only=False)
    sys.path.insert(0, ROOT)
    import pggan_amd as pg
    span = got['span']
    log0, made0, nimg0, its0 = got['logs'][0]
    for lg, made, nimg, its in got['logs'][1:]:
        assert lg == log0 and made == made0 and nimg == nimg0 and its == its0      # lock step, iteration by iteration
    mb = {0: 4, 1: 3, 2: 2}
    assert made0 == [4, 3, 2]                          # one loader per stage, with the stage's PER-RANK minibatch
    depths = [d for _, d, _, _ in log0]
    assert depths == sorted(depths) and set(depths) == {0, 1, 2}
    prev = 0
    for k, (nimg, depth, alpha, m) in enumerate(log0):
        # the plugin runs after the step: nimg is the counter the NEXT iteration starts from, (depth, alpha) what it will use
        want_depth, want_alpha = pg.plugins.growth_stage(nimg, span, span, 2)
        assert (depth, alpha) == (want_depth, repr(float(want_alpha))) and m == mb[depth], (k, nimg, depth, alpha, m)
        used = mb[log0[k - 1][1]] if k else mb[0]      # minibatch the iteration that just ended ran with
        assert nimg - prev == world * used, (k, nimg, prev, used)
        prev = nimg
    assert any(a not in ('1.0',) for _, _, a, _ in log0)        # fades happened
    assert nimg0 >= 4.6 * span and its0 == len(log0)


def _global_stddev_worker(rank, world, port, out_path, depth, alpha):
    """One D step + one G step per rank on ITS shard of a global batch, minibatch stddev in the exact-global mode."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import emu_ops
    import pggan_amd as pg
    from helpers import synthetic, reference_grads
    torch.set_num_threads(2)
    for modname in ('engine', 'optim'):
        importlib.import_module('pggan-pytorch_amd.' + modname).ops = emu_ops
    pg.engine._check_dev = lambda t, what: t.contiguous()
    pg.trainer._to_device = lambda t: t
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dp = pg.DataParallel()
    # the mode's contract -- equal per-rank minibatches -- is checked through the control plane (engine._mbstd_fwd, once per shape)
    dp.assert_same_on_all_ranks(7, 'a value every rank agrees on')
    try:
        dp.assert_same_on_all_ranks(3 + rank, 'a value the ranks disagree on')
        raise AssertionError('unequal values went unnoticed')
    except RuntimeError as exc:
        assert 'differs between the ranks' in str(exc)
    torch.manual_seed(31)
    shape = (1, 3, 16, 16)
    kw = dict(fmap_base=64, fmap_max=16)
    G = pg.Generator(shape, latent_size=16, **kw)
    D = pg.Discriminator(shape, **kw)
    G.depth = D.depth = depth
    G.alpha = D.alpha = alpha
    n = 2
    real, z_d, z_g, mix = synthetic(77, world * n, 3, 4 * 2 ** depth, 16)              # the GLOBAL batch; this rank's shard = its slice
    sl = slice(rank * n, (rank + 1) * n)
    opt_g = pg.FusedAdam(G.parameters(), 0.001, betas=(0.0, 0.99))
    opt_d = pg.FusedAdam(D.parameters(), 0.001, betas=(0.0, 0.99))

    class DS(object):
        pass
    DS.model_depth, DS.alpha = depth, alpha
    state = dict(z=[z_d[sl], z_g[sl]])

    def loader():
        while True:
            yield real[sl]

    def d_loss(Dm, Gm, r_, z_):
        pg.wgan_gp_loss.set_mixing_factors(mix[sl])
        return pg.wgan_gp_D_loss(Dm, Gm, r_, z_)
    tr = pg.Trainer(D, G, d_loss, pg.wgan_gp_G_loss, opt_d, opt_g, DS(), loader(), lambda: state['z'].pop(0) if state['z'] else z_g[sl],
                    parallel=dp, global_stddev=True)
    assert D._global_stddev is dp
    got = {}
    orig = dp.all_reduce_grads

    def rec(net, average=False):
        r = orig(net, average=average)
        got['D' if net is D else 'G'] = {k: v.clone() for k, v in reference_grads(net).items()}      # SUM over ranks, before Adam
        return r
    dp.all_reduce_grads = rec
    losses = []

    class Rec(pg.Plugin):
        def __init__(self):
            super(Rec, self).__init__([(1, 'iteration')])

        def register(self, trainer):
            pass

        def iteration(self, i, g_cost, d_cost, *rest):
            losses.append((float(g_cost), float(d_cost)))
    tr.register_plugin(Rec())
    import heapq
    for q in tr.plugin_queues.values():
        heapq.heapify(q)
    tr.train()
    t = torch.tensor(losses[0], dtype=torch.float64)
    dist.all_reduce(t)                                  # global loss = mean of the shard losses (equal shards)
    if rank == 0:
        torch.save(dict(grads=got, g_cost=float(t[0]) / world, d_cost=float(t[1]) / world), out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('depth,alpha', [(2, 1.0), (1, 0.5)])
def test_exact_global_minibatch_stddev_matches_one_process_at_global_batch(tmp_path, oracle, depth, alpha):
    """SURVEY.md §8e optional mode (Trainer(parallel=dp, global_stddev=True)): with the group statistic and the scalars of its adjoint /
    Hessian-vector term (G_sigma, <v, x - mu>, mean v) reduced over the ranks, 2 ranks x minibatch 2 reproduce ONE process at batch 4 --
    losses and the averaged gradients of the D step (first-order terms, gradient penalty incl. the stddev Hessian-vector term) and of
    the G step, against the oracle evaluated on the whole batch.  (Local-shard mode differs from this by O(1) in the stddev terms: the
    default-mode test above checks that one against per-shard oracle gradients.)"""
    world = 2
    out_path = str(tmp_path / 'gs.pt')
    mp.spawn(_global_stddev_worker, args=(world, _free_port(), out_path, depth, alpha), nprocs=world, join=True)
    got = torch.load(out_path, weights_only=False)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import pggan_amd as pg
    from helpers import synthetic
    torch.manual_seed(31)
    shape = (1, 3, 16, 16)
    kw = dict(fmap_base=64, fmap_max=16)
    G = pg.Generator(shape, latent_size=16, **kw)
    D = pg.Discriminator(shape, **kw)
    gp, dp_ = G.reference_state_dict(), D.reference_state_dict()
    cfg = oracle.NetCfg(16, 3, latent_size=16, **kw)
    real, z_d, z_g, mix = synthetic(77, 4, 3, 4 * 2 ** depth, 16)
    ref = oracle.d_loss_and_grads(dp_, gp, cfg, real, z_d, mix, depth, alpha)
    assert abs(got['d_cost'] - float(ref['D_cost'])) < 2e-4 * max(1.0, abs(float(ref['D_cost'])))
    assert sorted(ref['grads']) == sorted(got['grads']['D'])
    for k, v in ref['grads'].items():
        assert _err(got['grads']['D'][k] / world, v) < 2e-3, ('D grad', k, _err(got['grads']['D'][k] / world, v))
    # the shard-local statistic would NOT pass: make sure the case is sensitive to the mode
    loc = [oracle.d_loss_and_grads(dp_, gp, cfg, real[s], z_d[s], mix[s], depth, alpha) for s in (slice(0, 2), slice(2, 4))]
    key = [k for k in ref['grads'] if k.endswith('c1.conv.weight')][-1]
    assert _err(sum(d['grads'][key] for d in loc) / world, ref['grads'][key]) > 5e-3
    od = oracle.AdamState()
    od.step(dp_, ref['grads'], 0.001)
    refg = oracle.g_loss_and_grads(gp, dp_, cfg, z_g, depth, alpha)
    assert abs(got['g_cost'] - float(refg['G_cost'])) < 2e-4 * max(1.0, abs(float(refg['G_cost'])))
    for k, v in refg['grads'].items():
        assert _err(got['grads']['G'][k] / world, v) < 5e-3, ('G grad', k, _err(got['grads']['G'][k] / world, v))
