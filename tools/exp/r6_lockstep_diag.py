"""Round 6, item 1: what is the 2.28e-3 event of tests/test_e2e_gpu.py::test_plan_replay_public_api_loop?

Two EAGER twins (no plan, no hipGraph) of the test's network step side by side on identical weights / inputs / mixing factors through
engine.d_loss_forward + d_loss_backward; per iteration the LeakyReLU sign patterns of every saved activation of the two forward passes are
compared next to the pre-Adam gradients.  If the forward pass were deterministic no sign could differ and the gradients would agree to the
atomic-commit order of the weight gradients (~1e-6); a gradient event that coincides with >= 1 flipped branch in a low-resolution layer is
forward non-determinism (the direct conv's split-K launches of < 192-workgroup shapes commit with fp32 atomics: csrc/conv_igemm.hip
launch_conv), not a missing stream edge.  ``--no-splitk`` forces those launches to one K slice (pg_debug_set_tuning(2, 1)).

    python tools/exp/r6_lockstep_diag.py --reps 150 [--no-splitk] [--pixelnorm]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import pggan_amd as pg  # noqa: E402

DEV = 'cuda'


def l2(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def signs(t):
    if t.dtype == torch.uint8:
        return t
    return t > 0


def collect(ctx):
    out = {}
    for k, rec in enumerate(ctx['recs']):
        for name in ('inp', 'a1', 'a2'):
            if name in rec and torch.is_tensor(rec[name]):
                out['blk%d.%s@%d' % (k, name, rec['H'])] = signs(rec[name]).clone()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=100)
    ap.add_argument('--no-splitk', action='store_true')
    ap.add_argument('--pixelnorm', action='store_true')
    a = ap.parse_args()
    eng, wl = pg.engine, pg.wgan_gp_loss
    wl.enable_graphs(False)
    if a.no_splitk:
        pg._lib.load().pg_debug_set_tuning(2, 1)

    def build():
        torch.manual_seed(21)
        shape = (1, 3, 64, 64)
        kw = dict(fmap_base=1024, fmap_max=64)
        G = pg.Generator(shape, latent_size=64, **kw).cuda()
        D = pg.Discriminator(shape, pixelnorm=a.pixelnorm, **kw).cuda()
        G.depth = D.depth = 4
        return G, D, pg.FusedAdam(D.parameters(), 0.001, betas=(0.0, 0.99))
    events = []
    steps = 0
    for rep in range(a.reps):
        gen = torch.Generator(device='cuda').manual_seed(9)
        (Ga, Da, oa), (Gb, Db, ob) = build(), build()
        for it in range(7):
            real = torch.rand((6, 3, 64, 64), device=DEV, generator=gen) * 2 - 1
            z = torch.randn((6, 64), device=DEV, generator=gen)
            mix = torch.rand((6, 1), device=DEV, generator=gen)
            res = []
            for G, D, opt in ((Ga, Da, oa), (Gb, Db, ob)):
                D.zero_grad()
                c, _, _, state = eng.d_loss_forward(D, G, real, z, mix, 10.0, 0.001, 1.0)
                sg = collect(state['ctx'])
                eng.d_loss_backward(state)
                res.append((float(c), D._flat_grad.clone(), sg))
                opt.step()
            torch.cuda.synchronize()
            steps += 1
            e = l2(res[0][1], res[1][1])
            flips = {k: int((res[0][2][k] != res[1][2][k]).sum()) for k in res[0][2]}
            nfl = sum(flips.values())
            if e > 1e-4 or nfl:
                events.append((rep, it, e, nfl, {k: v for k, v in flips.items() if v}))
            with torch.no_grad():
                Db._flat_param.copy_(Da._flat_param)
            Db.mark_params_changed()
            for (_, ma, va), (_, mb_, vb) in zip(oa._flat.values(), ob._flat.values()):
                mb_.copy_(ma)
                vb.copy_(va)
    print('mode: split-K %s, D pixelnorm %s: %d twin steps, %d with a gradient difference > 1e-4 or a flipped branch'
          % ('OFF' if a.no_splitk else 'on (default)', a.pixelnorm, steps, len(events)))
    big = [e for e in events if e[2] > 1e-3]
    print('  gradient rel-L2 > 1e-3: %d   of those with >= 1 flipped LeakyReLU branch in the forward passes: %d' % (len(big), sum(1 for e in big if e[3] > 0)))
    print('  flipped-branch steps with gradient rel-L2 <= 1e-3: %d' % sum(1 for e in events if e[3] > 0 and e[2] <= 1e-3))
    for rep, it, e, nfl, fl in events[:25]:
        print('    rep %3d it %d  grad rel-L2 %.3e  flipped branches %d %s' % (rep, it, e, nfl, fl))


if __name__ == '__main__':
    main()
