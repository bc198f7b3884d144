"""Which sign-byte requests fell back to fp32 masks (PG_E_UNSUP) during one train step, per growth stage."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
import pggan_amd as pg
for d in (4, 5, 6, 7, 8):
    mb = bench.REF_MINIBATCH.get(d, 16)
    tr = bench.make_trainer(pg, 1024, d, 1.0, mb, 1337, None)
    tr.train(); torch.cuda.synchronize()
    pg.engine.FALLBACKS.clear()
    tr.train(); torch.cuda.synchronize()
    print('depth %d:' % d, dict(pg.engine.FALLBACKS) or 'no fallbacks')
    del tr; torch.cuda.empty_cache()
