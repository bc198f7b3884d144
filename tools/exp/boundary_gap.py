"""The ~100 us hole in front of wino_weights_batched_kernel at the iteration boundary (profiles/r05_step_timeline.txt).  Variants (VAR=...):
   base     the train loop as it is
   marker   a 4-byte pg_zero (fillBuffer kernel) right after G's optimizer step: where does the hole go?
   nowait   FusedAdam.step without engine._await_backward_copies
Run:  rocprofv3 --kernel-trace --output-format csv -d /tmp/bg -o bg -- python tools/exp/boundary_gap.py ; python tools/step_timeline.py "/tmp/bg/**/bg_kernel_trace.csv" 3 | grep -B6 -A4 wino_weights"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import pggan_amd as pg
torch.cuda.set_device(0)
var = os.environ.get('VAR', 'base')
tr = bench.make_trainer(pg, 1024, 8, 1.0, 3, 1337, None)
tiny = torch.zeros(4, device='cuda')
if var == 'marker':
    orig = tr.optimizer_g.step

    def step(*a, **k):
        r = orig(*a, **k)
        pg.ops.zero_(tiny)
        return r
    tr.optimizer_g.step = step
if var == 'nowait':
    pg.engine._await_backward_copies = lambda net: None
    pg.optim.FusedAdam.step.__globals__  # (step imports engine lazily: the patched attribute is what it sees)
for _ in range(16):
    tr.train()
torch.cuda.synchronize()
print('done', var)
