"""The gradient-exchange face of the C-ABI on the MI355X (include/pggan_hip.h: pg_rccl_version, pg_comm_unique_id,
pg_comm_init_rank, pg_comm_info, pg_allreduce_sum_f32, pg_comm_destroy; csrc/collective.hip) and the product path on top of
it (``parallel.DataParallel`` / ``GradExchange`` / ``Trainer(parallel=...)``).

The reference is single-GPU: its exchange points are after /root/reference/trainer.py:98 (D) and :111 (G).  The GPU box has
ONE device, so the communicator here has one rank: every collective is then the identity, which makes the checks exact —
an all-reduce must leave a random buffer bit-identical, and a data-parallel Trainer must land on the weights of the plain one.
(The two-rank arithmetic — shard sums, 1/world in Adam, per-rank minibatch accounting — is covered against the oracle by
the world-size-2 gloo tests of tests/test_parallel_cpu.py.)"""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch

import pggan_amd as pg
from pggan_amd import _lib
from helpers import assert_same_contributions, grads_by_name

pytestmark = pytest.mark.gpu
DEV = 'cuda'
PG_E_ARG = -1


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.fixture(scope='module')
def dp():
    """One-rank data-parallel group: torch.distributed as the control plane, the library's own RCCL communicator as the data plane."""
    import torch.distributed as dist
    saved = {k: os.environ.get(k) for k in ('MASTER_ADDR', 'MASTER_PORT', 'RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0', LOCAL_RANK='0', WORLD_SIZE='1')
    d = pg.DataParallel.from_env(force=True)
    yield d
    d.close()
    if dist.is_initialized():
        dist.destroy_process_group()
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def test_rccl_is_bound_and_argument_errors():
    lib = _lib.load()
    v = ctypes.c_int(-1)
    assert lib.pg_rccl_version(ctypes.byref(v)) == 0 and v.value > 0            # e.g. 22205: an RCCL really was resolved
    assert lib.pg_rccl_version(None) == PG_E_ARG
    assert lib.pg_comm_unique_id(None) == PG_E_ARG
    ident = ctypes.create_string_buffer(128)
    assert lib.pg_comm_unique_id(ident) == 0 and any(ident.raw)
    comm = ctypes.c_void_p()
    assert lib.pg_comm_init_rank(None, 1, ident, 0) == PG_E_ARG
    assert lib.pg_comm_init_rank(ctypes.byref(comm), 0, ident, 0) == PG_E_ARG      # no ranks
    assert lib.pg_comm_init_rank(ctypes.byref(comm), 1, ident, 1) == PG_E_ARG      # rank out of range
    assert lib.pg_comm_init_rank(ctypes.byref(comm), 1, ident, -1) == PG_E_ARG
    assert lib.pg_comm_init_rank(ctypes.byref(comm), 1, None, 0) == PG_E_ARG
    n, r = ctypes.c_int(), ctypes.c_int()
    assert lib.pg_comm_info(None, ctypes.byref(n), ctypes.byref(r)) == PG_E_ARG
    assert lib.pg_comm_destroy(None) == PG_E_ARG
    buf = torch.ones(16, device=DEV)
    assert lib.pg_allreduce_sum_f32(None, ctypes.c_void_p(buf.data_ptr()), 16, None) == PG_E_ARG


def test_one_rank_allreduce_is_the_identity(dp):
    lib = _lib.load()
    assert dp.comm is not None and dp.comm_ranks == 1 and dp.world_size == 1 and dp.rank == 0
    n, r = ctypes.c_int(-1), ctypes.c_int(-1)
    assert lib.pg_comm_info(dp.comm, ctypes.byref(n), ctypes.byref(r)) == 0 and (n.value, r.value) == (1, 0)
    assert lib.pg_allreduce_sum_f32(dp.comm, None, 4, None) == PG_E_ARG
    g = torch.Generator(device=DEV).manual_seed(3)
    for count in (1, 7, 4096, (16 << 20) // 4 + 3):                            # up to one 16 MB bucket, odd sizes included
        buf = torch.randn(count, device=DEV, generator=g)
        want = buf.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        assert lib.pg_allreduce_sum_f32(dp.comm, ctypes.c_void_p(buf.data_ptr()), count, ctypes.c_void_p(s.cuda_stream)) == 0
        s.synchronize()
        assert torch.equal(buf, want), count
    assert lib.pg_allreduce_sum_f32(dp.comm, ctypes.c_void_p(buf.data_ptr()), 0, None) == 0      # empty span: no-op
    # the product wrapper: a span of a larger buffer, on the current stream
    flat = torch.randn(10000, device=DEV, generator=g)
    want = flat.clone()
    before = dict(dp.stats)
    dp.all_reduce_flat(flat[100:9000])
    torch.cuda.synchronize()
    assert torch.equal(flat, want)
    assert dp.stats['collectives'] == before['collectives'] + 1 and dp.stats['bytes'] == before['bytes'] + 8900 * 4


def _train(parallel, buckets, monkeypatch, iters=3, global_stddev=False):
    monkeypatch.setenv('PGGAN_DP_BUCKETS', '1' if buckets else '0')
    monkeypatch.setattr(pg.parallel, 'BUCKET_BYTES', 1 << 16)                    # small buckets: several collectives per sweep
    torch.manual_seed(31)
    shape = (1, 3, 64, 64)
    kw = dict(fmap_base=512, fmap_max=64)
    G = pg.Generator(shape, latent_size=64, **kw).to(DEV)
    D = pg.Discriminator(shape, **kw).to(DEV)
    G.depth = D.depth = 4
    G.alpha = D.alpha = 0.7                                                      # fade-in: both fromRGB / toRGB pairs are live
    rs = np.random.RandomState(9)
    reals = iter([torch.from_numpy(rs.rand(4, 3, 64, 64).astype(np.float32) * 2 - 1) for _ in range(iters)])
    zs = iter([torch.from_numpy(rs.randn(4, 64).astype(np.float32)) for _ in range(2 * iters)])
    mixes = iter([torch.from_numpy(rs.rand(4, 1).astype(np.float32)) for _ in range(iters)])

    def d_loss(Dm, Gm, real, z):
        pg.wgan_gp_loss.set_mixing_factors(next(mixes))
        return pg.wgan_gp_D_loss(Dm, Gm, real, z)
    opt_g = pg.FusedAdam(G.parameters(), 0.001, betas=(0.0, 0.99))
    opt_d = pg.FusedAdam(D.parameters(), 0.001, betas=(0.0, 0.99))
    tr = pg.Trainer(D, G, d_loss, pg.wgan_gp_G_loss, opt_d, opt_g, None, reals, lambda: next(zs), parallel=parallel, global_stddev=global_stddev)
    grads = []
    for _ in range(iters):
        tr.train()
        grads.append((grads_by_name(D), grads_by_name(G)))
    torch.cuda.synchronize()
    assert tr.cur_nimg == 4 * iters
    return G._flat_param.clone(), D._flat_param.clone(), grads


def test_trainer_with_one_rank_communicator_matches_plain_trainer(dp, monkeypatch):
    """``Trainer(parallel=dp)`` — bucketed exchange on the third stream, and one exchange per network — against the plain
    Trainer.  With one rank every collective is the identity and 1/world = 1, so the runs differ only by stream placement:
    the pre-Adam gradients of the first iteration agree to the atomic-add order of the weight-gradient commits, the weights
    after 3 iterations to twice the sum of the per-step bounds lr*sqrt((1 - beta2^t)/(1 - beta2)) of a sign-like Adam (beta1 = 0:
    an element whose gradient was round-off noise until step t moves by that much, in either direction)."""
    g0, d0, gr0 = _train(None, True, monkeypatch)
    c0 = dp.stats['collectives']
    g1, d1, gr1 = _train(dp, True, monkeypatch)
    c1 = dp.stats['collectives']
    g2, d2, gr2 = _train(dp, False, monkeypatch)
    c2 = dp.stats['collectives']
    assert c1 - c0 > c2 - c1 >= 2 * 3            # buckets: more, smaller collectives; without: >= one span per network and step
    for name, ref, got in (('G bucketed', g0, g1), ('D bucketed', d0, d1), ('G flush', g0, g2), ('D flush', d0, d2)):
        assert float((got - ref).abs().max()) <= 2 * 0.001 * sum(((1 - 0.99 ** t) / 0.01) ** 0.5 for t in (1, 2, 3)) + 1e-6, name
        assert float((got - ref).norm() / ref.norm()) < 1e-3, name
    for other in (gr1, gr2):                     # first iteration: every layer received the same contributions
        assert_same_contributions(other[0][0], gr0[0][0])
        # (G's gradients go through D AFTER its first update, where a sign-like Adam has turned round-off noise into +-lr)
        assert_same_contributions(other[0][1], gr0[0][1], tol=0.3, total=5e-2)


def test_exact_global_stddev_with_one_rank_equals_local_mode(dp, monkeypatch):
    """``Trainer(parallel=dp, global_stddev=True)`` on the device: the split minibatch-stddev entry points + the statistic / Gs exchange
    through the library's RCCL communicator (one rank: the exchange is the identity, the global batch IS the local one), every step
    eager.  Must land where the default local-shard mode lands; the exchange really ran (a few floats per collective)."""
    g0, d0, gr0 = _train(dp, True, monkeypatch)
    c0, b0 = dp.stats['collectives'], dp.stats['bytes']
    g1, d1, gr1 = _train(dp, True, monkeypatch, global_stddev=True)
    c1, b1 = dp.stats['collectives'], dp.stats['bytes']
    g2, d2, gr2 = _train(dp, True, monkeypatch)
    extra = (c1 - c0) - (dp.stats['collectives'] - c1)
    assert extra >= 3 * 6                        # per iteration: 3 statistic gathers (D fwd, tangent, G-step D fwd) + >= 3 Gs sums
    for name, ref, got in (('G', g0, g1), ('D', d0, d1)):
        assert float((got - ref).abs().max()) <= 2 * 0.001 * sum(((1 - 0.99 ** t) / 0.01) ** 0.5 for t in (1, 2, 3)) + 1e-6, name
        assert float((got - ref).norm() / ref.norm()) < 1e-3, name
    assert_same_contributions(gr1[0][0], gr0[0][0])
    assert_same_contributions(gr1[0][1], gr0[0][1], tol=0.3, total=5e-2)


def test_launch_plan_under_data_parallelism_matches_eager_twin(dp, monkeypatch, deterministic_forward):
    """Round 6: ``Trainer(parallel=dp)`` no longer turns launch plans off when the data plane is the library's RCCL communicator -- the
    bucket collectives the backward sweep feeds (``GradExchange._flush`` -> ``pg_allreduce_sum_f32`` on the exchange stream, behind an edge
    from the weight-gradient stream) are recorded and replayed with the rest of the step; ``finish()`` (what is left + the join) stays eager
    in the update.  A plan-issued data-parallel trainer against an eagerly issued twin at identical weights: per-iteration pre-Adam
    gradients of D and G, the number of collectives and bytes per step (the host-side counters are put back from the recording), and
    lock-step weights.  One rank: every collective is the identity, so any difference is an ordering defect."""
    wl = pg.wgan_gp_loss
    monkeypatch.setattr(pg.parallel, 'BUCKET_BYTES', 1 << 16)                    # small buckets: several collectives inside each sweep

    def build():
        torch.manual_seed(21)
        shape = (1, 3, 32, 32)
        kw = dict(fmap_base=512, fmap_max=64)
        G = pg.Generator(shape, latent_size=64, **kw).cuda()
        D = pg.Discriminator(shape, **kw).cuda()
        G.depth = D.depth = 3
        opt_g = pg.FusedAdam(G.parameters(), 0.001, betas=(0.0, 0.99))
        opt_d = pg.FusedAdam(D.parameters(), 0.001, betas=(0.0, 0.99))
        ds = pg.utils.SyntheticDataset(32, 3, seed=5)
        ds.model_depth = 3
        return pg.Trainer(D, G, pg.wgan_gp_D_loss, pg.wgan_gp_G_loss, opt_d, opt_g, ds, ds.loader(8), pg.utils.device_latents(8, 64, seed=3), parallel=dp)
    wl.enable_graphs('auto')
    wl.enable_plans(True)
    pg.plans.STATS.update(recorded=0, replayed=0)
    try:
        tra, trb = build(), build()
        assert wl._use_plans, 'Trainer(parallel=<RCCL data plane>) must leave launch plans on'
        per_step = []
        for it in range(7):
            out = []
            for tr, plans_on in ((tra, True), (trb, False)):
                wl._use_plans = plans_on
                wl.manual_seed(100 + it)
                c0, b0 = dp.stats['collectives'], dp.stats['bytes']
                tr.train()
                torch.cuda.synchronize()
                out.append((grads_by_name(tr.D), grads_by_name(tr.G), dp.stats['collectives'] - c0, dp.stats['bytes'] - b0))
            assert_same_contributions(out[0][0], out[1][0], tol=1e-2, total=2e-4)    # (deterministic_forward: the atomic commit order of the weight gradients only)
            assert_same_contributions(out[0][1], out[1][1], tol=0.3, total=5e-2)     # (through D after its update: sign-like Adam on round-off noise)
            assert out[0][2:] == out[1][2:] and out[0][2] > 4, (it, out[0][2:], out[1][2:])   # same collectives / bytes, several buckets per step
            per_step.append(out[0][2:])
            for a, b in ((tra.G, trb.G), (tra.D, trb.D)):
                assert float((a._flat_param - b._flat_param).abs().max()) <= 2 * 0.001 * ((1 - 0.99 ** (it + 1)) / 0.01) ** 0.5 + 1e-6
                with torch.no_grad():
                    b._flat_param.copy_(a._flat_param)
                b.mark_params_changed()
            for oa, ob in ((tra.optimizer_g, trb.optimizer_g), (tra.optimizer_d, trb.optimizer_d)):
                for (_, ma, va), (_, mb_, vb) in zip(oa._flat.values(), ob._flat.values()):
                    mb_.copy_(ma)
                    vb.copy_(va)
        assert len(set(per_step)) == 1, per_step                                  # eager warm-up, recording and replayed steps exchange the same
        assert pg.plans.STATS['recorded'] == 2 and pg.plans.STATS['replayed'] == 2 * 4, pg.plans.STATS
    finally:
        wl._use_plans = True
        wl.enable_graphs(False)
