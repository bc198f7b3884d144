"""Fade-in (alpha 0.5) vs fully grown (alpha 1) step time per growth stage, and the fallbacks a fade-in step takes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
import pggan_amd as pg
for d in (2, 4, 6, 7, 8):
    mb = bench.REF_MINIBATCH.get(d, 16)
    res = {}
    for alpha in (1.0, 0.5):
        tr = bench.make_trainer(pg, 1024, d, alpha, mb, 1337, None)
        for _ in range(6): tr.train()
        torch.cuda.synchronize(); pg.engine.FALLBACKS.clear(); t0 = time.perf_counter()
        n = 30
        for _ in range(n): tr.train()
        torch.cuda.synchronize(); res[alpha] = (time.perf_counter() - t0) / n * 1e3
        fb = dict(pg.engine.FALLBACKS)
        del tr; torch.cuda.empty_cache(); pg.plans.clear()
    print('depth %d: alpha 1 %.3f ms, alpha 0.5 %.3f ms (+%.1f %%)  fade fallbacks: %s' % (d, res[1.0], res[0.5], 100 * (res[0.5] / res[1.0] - 1), fb or 'none'), flush=True)
