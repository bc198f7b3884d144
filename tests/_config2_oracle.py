"""The oracle's side of tests/test_e2e_gpu.py::test_config2_grow_run_against_oracle: BASELINE.json config 2 (32x32 network, default
512-channel widths, grown depth 0 -> 3 with fade-ins at minibatch 64) trained by ``oracle.train_iteration`` under the oracle's own
schedule.  The trajectory depends on nothing the HIP run produces (same seeded initial weights, same synthetic batches), and it is
~150 s of CPU work -- so tests/conftest.py starts it as a background process when the GPU session begins (this file run as a script)
and the test picks the result up; run on its own, the test computes it in-process.  Test infrastructure only."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N, LOD, ITERS, RAMP, SEED, SHAPE = 64, 128, 14, 0.256, 77, (1, 3, 32, 32)


def initial_nets():
    import pggan_amd as pg
    torch.manual_seed(SEED)
    return pg.Generator(SHAPE), pg.Discriminator(SHAPE)


def schedule(oracle):
    return [oracle.depth_schedule(it * N, 3, LOD, LOD, minibatch_default=N) for it in range(ITERS)]


def batch(oracle, sched, it):
    return oracle.synthetic_batch(300 + it, N, 3, 4 * 2 ** sched[it][0], 512)


def trajectory():
    """Returns dict(losses=[(D_cost, G_cost)] per iteration, gp=, dp= the oracle's end-state parameter dicts)."""
    from oracle import pggan_cpu as oracle
    G, D = initial_nets()
    gp, dp = G.reference_state_dict(), D.reference_state_dict()
    cfg = oracle.NetCfg(32, 3)
    sched = schedule(oracle)
    og, od = oracle.AdamState(), oracle.AdamState()
    losses = []
    for it in range(ITERS):
        depth, alpha, _, _ = sched[it]
        real, z_d, z_g, mix = batch(oracle, sched, it)
        lr = 0.001 * oracle.rampup(it * N, RAMP)
        d, g = oracle.train_iteration(gp, dp, cfg, og, od, real, z_d, z_g, mix, depth, alpha, lr, lr)
        losses.append((float(d['D_cost']), float(g['G_cost'])))
    return dict(losses=losses, gp=gp, dp=dp)


_BACKGROUND = {}


def start_background(tmpdir):
    import subprocess
    out = os.path.join(tmpdir, 'config2_oracle.pt')
    env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')      # (the oracle never touches the GPU)
    _BACKGROUND.update(proc=subprocess.Popen([sys.executable, os.path.abspath(__file__), out], env=env), out=out)


def result():
    if _BACKGROUND:
        rc = _BACKGROUND['proc'].wait(timeout=900)
        assert rc == 0, 'background oracle trajectory failed (rc %d)' % rc
        return torch.load(_BACKGROUND['out'], weights_only=False)
    return trajectory()


if __name__ == '__main__':
    torch.set_num_threads(max(4, (os.cpu_count() or 8) // 2))                   # (the other tests' oracle calls run beside it)
    res = trajectory()
    torch.save(res, sys.argv[1] + '.tmp')
    os.replace(sys.argv[1] + '.tmp', sys.argv[1])
