"""Which host call sites launch the small device copies / fills / torch elementwise kernels inside one 1024^2 train step
(torch.profiler with stacks): python tools/exp/find_copies.py"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import pggan_amd as pg
pg.wgan_gp_loss.enable_graphs('auto')
torch.cuda.set_device(0)
tr = bench.make_trainer(pg, 1024, 8, 1.0, 3, 1, None)
for _ in range(5):
    tr.train()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(2):
        tr.train()
    torch.cuda.synchronize()
agg = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith('aten::'):
        continue
    if ev.name in ('aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::mul', 'aten::add', 'aten::sub', 'aten::div', 'aten::mean', 'aten::sum',
                   'aten::uniform_', 'aten::_local_scalar_dense', 'aten::cat', 'aten::mul_', 'aten::add_', 'aten::neg', 'aten::pow', 'aten::sqrt'):
        st = [s for s in ev.stack if 'pggan' in s or 'bench' in s][:3]
        agg[(ev.name, ' <- '.join(s.split('/')[-1] for s in st))] += 1
for (name, where), n in sorted(agg.items(), key=lambda kv: -kv[1]):
    print('%4d  %-28s %s' % (n, name, where))
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=25, max_name_column_width=60))
