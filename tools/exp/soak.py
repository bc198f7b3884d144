"""Soak: N train steps at the headline workload with the default step issue (launch plans); memory must not grow, losses stay finite,
and a growth-stage change in the middle (depth 7 -> 8 with a fade-in) must work with plans on."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import pggan_amd as pg
tr = bench.make_trainer(pg, 1024, 7, 1.0, 6, 1337, None)
losses = []
class Rec(pg.Plugin):
    def __init__(self): super(Rec, self).__init__([(50, 'iteration')])
    def register(self, trainer): pass
    def iteration(self, i, g_cost, d_cost, d_real, d_fake): losses.append((i, float(g_cost), float(d_cost)))
tr.register_plugin(Rec())
import heapq
for q in tr.plugin_queues.values(): heapq.heapify(q)
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tr.train()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('depth 7: %.3f ms/step' % run(300), 'mem %.2f GB' % (torch.cuda.memory_allocated() / 2**30), flush=True)
# fade-in to depth 8
tr.G.depth = tr.D.depth = tr.dataset.model_depth = 8
ds = tr.dataset
tr.dataiter = ds.loader(3)
tr.random_latents_generator = pg.utils.device_latents(3, 512, seed=7)
for a in (0.0, 0.25, 0.5, 0.75):
    tr.G.alpha = tr.D.alpha = ds.alpha = a
    print('depth 8 alpha %.2f: %.3f ms/step' % (a, run(20)), flush=True)
tr.G.alpha = tr.D.alpha = ds.alpha = 1.0
m0 = None
for k in range(5):
    ms = run(400)
    mem = torch.cuda.memory_allocated() / 2**30
    print('depth 8 block %d: %.3f ms/step, mem %.2f GB, reserved %.2f GB, plans %s' % (k, ms, mem, torch.cuda.memory_reserved() / 2**30, pg.plans.STATS), flush=True)
    if m0 is None: m0 = mem
    assert mem <= m0 + 0.05, 'memory grows'
assert all(abs(g) < 1e6 and abs(d) < 1e6 and g == g and d == d for _, g, d in losses), 'non-finite loss'
print('losses (every 50 iterations):', ' '.join('%d:%.3g/%.3g' % l for l in losses[::6]))
print('OK')
