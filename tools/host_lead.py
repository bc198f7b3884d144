"""How far does the host run ahead of the device?  Per train step: host time at the call vs the device time at which an event recorded at that
point completes (both relative to step 0).  lead = device - host: ~0 means the device waits for the host there.
   python tools/host_lead.py [depth] [minibatch]"""
import os
import sys
import time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pggan_amd as pg  # noqa: E402

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mb = int(sys.argv[2]) if len(sys.argv) > 2 else bench.REF_MINIBATCH.get(depth, 16)
torch.cuda.set_device(0)
pg.wgan_gp_loss.enable_graphs('auto')
tr = bench.make_trainer(pg, 1024, depth, 1.0, mb, 1337, None)
for _ in range(30):
    tr.train()
torch.cuda.synchronize()
n = 40
evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
mids = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
host, hmid = [], []
orig = tr.optimizer_g.step


def step_g(*a, **k):                      # host / device time right before G's optimizer step (the iteration boundary)
    i = len(hmid)
    hmid.append(time.perf_counter())
    mids[i].record()
    return orig(*a, **k)
tr.optimizer_g.step = step_g
t0 = time.perf_counter()
for i in range(n):
    host.append(time.perf_counter())
    evs[i].record()
    tr.train()
evs[n].record()
host.append(time.perf_counter())
torch.cuda.synchronize()
print('step  host_ms  device_ms  lead_ms | before Adam(G): host  device  lead')
for i in range(0, n, 2):
    h, d = 1e3 * (host[i] - host[0]), evs[0].elapsed_time(evs[i])
    hm, dm = 1e3 * (hmid[i] - host[0]), evs[0].elapsed_time(mids[i])
    print('%4d %8.2f %9.2f %8.2f | %8.2f %8.2f %8.2f' % (i, h, d, d - h, hm, dm, dm - hm))
