import sys, os, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
import pggan_amd as pg
tr = bench.make_trainer(pg, 1024, 8, 1.0, 3, 1337, None)
for _ in range(3): tr.train()
cnt = collections.Counter()
orig = pg.ops.conv2d_wino
def wrap(x, u, bias, N, H, W, scale, slope=1.0, mask=None, mask_slope=0.2, ups=False, out=None, pool=False, other=None, a=1.0, b=0.0,
         pool_only=False, unpool=False, upmask=None, up_mul=1.0, y_bytes=False, signs_out=False):
    key = ('mask:' + ('none' if mask is None else str(mask.dtype).split('.')[-1]), 'pool' if pool else '', 'pool_only' if pool_only else '', 'unpool' if unpool else '',
           'y_bytes' if y_bytes else '', 'signs_out' if signs_out else '', 'H%d' % H)
    cnt[key] += 1
    return orig(x, u, bias, N, H, W, scale, slope, mask, mask_slope, ups, out, pool, other, a, b, pool_only, unpool, upmask, up_mul, y_bytes, signs_out)
pg.ops.conv2d_wino = wrap
pg.engine.ops.conv2d_wino = wrap
tr.train()
torch.cuda.synchronize()
agg = collections.Counter()
for k, v in cnt.items():
    agg[k[:6]] += v
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(v, [x for x in k if x])
print(sum(cnt.values()))
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
    if k[0] == 'mask:float32': print(v, k)
