"""Un-traced phase timeline of the 1024^2 (or any) train step: engine.PROBES timing events at the phase boundaries of the D and G
schedules, re-recorded by the launch plan on every replay; prints, for the last of a few steps, when each stream reached each
boundary (us from the start of the D step) and what the main stream spent between consecutive boundaries.

    python tools/phase_timeline.py [--depth 8] [--steps 30]

Reading: 'D.loss_end - D.gp_bwd_end' much larger than the three tiny launches between them = the main stream waited for the fake third
on the second stream; 'D.mix_end - D.g_fwd_end' larger than gp_mix = it waited for the real copy / host; G.start - D.sweep_end = the
iteration's hand-over from the D step to the G step (Trainer's python between them, the deferred update is on the second stream)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import pggan_amd as pg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--depth', type=int, default=8)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--minibatch', type=int, default=0)
    a = ap.parse_args()
    pg.wgan_gp_loss.enable_graphs('auto')
    mb = a.minibatch or bench.REF_MINIBATCH.get(a.depth, 16)
    tr = bench.make_trainer(pg, 1024, a.depth, 1.0, mb, 1337, None, fmap_base=4096, channels=3)
    for _ in range(20):
        tr.train()
    torch.cuda.synchronize()
    pg.plans.clear()                       # re-record with the probes in the plan
    pg.engine.PROBES = {}
    for _ in range(a.steps):
        tr.train()
    torch.cuda.synchronize()
    P = pg.engine.PROBES
    # Events recorded inside a plan are re-recorded at every replay: the last entry of every tag that sits in a plan holds the LAST
    # step; the eager ones (update_end) get a fresh event per step -- take the last of each list.
    last = {k: v[-1] for k, v in P.items()}
    base = last['D.start']
    rows = []
    for k, ev in last.items():
        try:
            rows.append((base.elapsed_time(ev) * 1e3, k))
        except Exception as exc:           # noqa: BLE001
            print('skip', k, exc)
    # the D-step probes of the last step precede its G-step probes; update_end events of the last step are after both
    rows.sort()
    print('phase boundaries of the last step (us after D.start; minibatch %d, depth %d):' % (mb, a.depth))
    for t, k in rows:
        print('  %9.1f  %s' % (t, k))
    t = dict((k, v) for v, k in rows)
    def d(a_, b_):
        return t[b_] - t[a_] if a_ in t and b_ in t else float('nan')
    print('main stream, D step: G fwd %.0f | -> gp_mix done %.0f | mixed fwd %.0f | GP first bwd %.0f | wait fake + d_loss %.0f | tangent %.0f | sweep %.0f'
          % (d('D.start', 'D.g_fwd_end'), d('D.g_fwd_end', 'D.mix_end'), d('D.mix_end', 'D.mixed_fwd_end'), d('D.mixed_fwd_end', 'D.gp_bwd_end'),
             d('D.gp_bwd_end', 'D.loss_end'), d('D.loss_end', 'D.tangent_end'), d('D.tangent_end', 'D.sweep_end')))
    print('second stream, D step: starts %.0f | real third until %.0f | fake third %.0f -> %.0f | weight gradients done %.0f | update done %.0f'
          % (t.get('D.side_start', float('nan')), t.get('D.real_end', float('nan')), t.get('D.fake_start', float('nan')), t.get('D.fake_end', float('nan')),
             t.get('D.wgrad_end', float('nan')), t.get('D.update_end', float('nan'))))
    if 'D.early_g_start' in t:
        print('second stream, G step\'s generator pass inside the D step: %.0f -> %.0f' % (t['D.early_g_start'], t['D.early_g_end']))
    print('main stream, G step: hand-over %.0f | G fwd %.0f | D fwd %.0f | D bwd %.0f | G bwd %.0f | (wgrad done at %.0f, update done %.0f)'
          % (d('D.sweep_end', 'G.start'), d('G.start', 'G.g_fwd_end'), d('G.g_fwd_end', 'G.d_fwd_end'), d('G.d_fwd_end', 'G.d_bwd_end'),
             d('G.d_bwd_end', 'G.g_bwd_end'), t.get('G.wgrad_end', float('nan')), t.get('G.update_end', float('nan'))))


if __name__ == '__main__':
    main()
