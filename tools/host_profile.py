"""Host-side (Python) cost of one train step at depth 8: cProfile over 20 iterations, GPU work asynchronous."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import pggan_amd as pg
pg.wgan_gp_loss.enable_graphs('auto')
tr = bench.make_trainer(pg, 1024, 8, 1.0, 3, 1337, None)
for _ in range(20):
    tr.train()
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(20):
    tr.train()
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host enqueue time %.2f ms per step (profiled, ~2x slower than unprofiled); GPU drained %.2f ms after the last enqueue' % ((t1 - t0) * 1e3 / 20, (t2 - t1) * 1e3))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
print('\n'.join(l[:150] for l in s.getvalue().splitlines()[:40]))
s = io.StringIO()
pstats.Stats(pr, stream=s).print_callers('_named_members|named_modules|current_stream')
print('\n'.join(l[:170] for l in s.getvalue().splitlines()[:60]))
t0 = time.perf_counter()
for _ in range(20):
    tr.train()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('unprofiled: host returns after %.2f ms per step; total %.2f ms per step' % ((t1 - t0) * 1e3 / 20, (t2 - t0) * 1e3 / 20))
