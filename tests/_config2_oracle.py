"""The oracle's side of tests/test_e2e_gpu.py::test_config2_grow_run_against_oracle: BASELINE.json config 2 (32x32 network, default
512-channel widths, grown depth 0 -> 3 with fade-ins at minibatch 64) trained by ``oracle.train_iteration`` under the oracle's own
schedule.  The trajectory depends on nothing the HIP run produces (same seeded initial weights, same synthetic batches).  It stops
after ORACLE_ITERS = 11 of the HIP run's 14 iterations -- through the growth-stage switch to 32x32 and its first iteration; every further
32x32 iteration at minibatch 64 costs 25-60 s of CPU time depending on the box (the gradients of that stage are pinned by
test_full_width_res32_golden, the schedule of all 14 iterations by the HIP run's own assertions).
(Running it as a background process beside the
other GPU tests was tried: their oracle calls then fight it for the host cores, 866 s instead of 714 s for the suite.)
Test infrastructure only."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N, LOD, ITERS, RAMP, SEED, SHAPE = 64, 128, 14, 0.256, 77, (1, 3, 32, 32)
ORACLE_ITERS = 11


def initial_nets():
    import pggan_amd as pg
    torch.manual_seed(SEED)
    return pg.Generator(SHAPE), pg.Discriminator(SHAPE)


def schedule(oracle):
    return [oracle.depth_schedule(it * N, 3, LOD, LOD, minibatch_default=N) for it in range(ITERS)]


def batch(oracle, sched, it):
    return oracle.synthetic_batch(300 + it, N, 3, 4 * 2 ** sched[it][0], 512)


def trajectory():
    """Returns dict(losses=[(D_cost, G_cost)] per iteration, gp=, dp= the oracle's parameter dicts after ORACLE_ITERS iterations)."""
    from oracle import pggan_cpu as oracle
    G, D = initial_nets()
    gp, dp = G.reference_state_dict(), D.reference_state_dict()
    cfg = oracle.NetCfg(32, 3)
    sched = schedule(oracle)
    og, od = oracle.AdamState(), oracle.AdamState()
    losses = []
    for it in range(ORACLE_ITERS):
        depth, alpha, _, _ = sched[it]
        real, z_d, z_g, mix = batch(oracle, sched, it)
        lr = 0.001 * oracle.rampup(it * N, RAMP)
        d, g = oracle.train_iteration(gp, dp, cfg, og, od, real, z_d, z_g, mix, depth, alpha, lr, lr)
        losses.append((float(d['D_cost']), float(g['G_cost'])))
    return dict(losses=losses, gp=gp, dp=dp)
