"""Cycle-level phase breakdown of conv_wino_strip_kernel (s_memtime stamps of lane 0 of every wave of the first 1024 workgroups).
Build the traced library in the build container (tools/exp/build_trace.sh), then on the GPU box:
  PGGAN_HIP_LIB=ab/libpggan_trace.so python tools/exp/wino_strip_trace.py N H Cin Cout kind      (kind: plain | poolb | maskb | mpool | unpool)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import pggan_amd as pg
ops, lib = pg.ops, pg._lib.load()
N, H, ci, co = [int(v) for v in sys.argv[1:5]]
kind = sys.argv[5] if len(sys.argv) > 5 else 'poolb'
lib.pg_debug_set_wino(21)
g = torch.Generator(device='cuda').manual_seed(1)
x = torch.randn(N, H, H, ci, device='cuda', generator=g)
u = ops.wino_transform_weights(torch.randn(3, 3, co, ci, device='cuda', generator=g) * 0.05)
b = torch.randn(co, device='cuda', generator=g)
mb = (torch.randn(N, H, H, co // 4, device='cuda', generator=g) > 0).to(torch.uint8) * 5
umb = (torch.randn(N, 2 * H, 2 * H, co // 4, device='cuda', generator=g) > 0).to(torch.uint8) * 5
fn = {'plain': lambda: ops.conv2d_wino(x, u, b, N, H, H, 0.5, 0.2),
      'poolb': lambda: ops.conv2d_wino(x, u, b, N, H, H, 0.5, 0.2, pool=True, y_bytes=True),
      'maskb': lambda: ops.conv2d_wino(x, u, None, N, H, H, 0.5, mask=mb, mask_slope=0.2),
      'mpool': lambda: ops.conv2d_wino(x, u, None, N, H, H, 0.5, mask=mb, mask_slope=0.2, pool=True, pool_only=True),
      'unpool': lambda: ops.conv2d_wino(x, u, None, N, H, H, 0.5, mask_slope=0.2, unpool=True, upmask=umb, up_mul=0.7)}[kind]
for _ in range(3):
    fn()
tr = torch.zeros(1024 * 4 * 16 * 8, dtype=torch.int64, device='cuda')
lib.pg_debug_wino_trace.argtypes = [ctypes.c_void_p]
lib.pg_debug_wino_trace(tr.data_ptr())
fn()
torch.cuda.synchronize()
print(lib.pg_debug_last_wino_kernel().decode())
lib.pg_debug_wino_trace(None)
t = tr.cpu().numpy().reshape(1024, 4, 16, 8).astype(np.float64)
ok = t[:, :, 15, 0] > 0
nst = int((t[0, 0, :15, 0] > 0).sum())
print('workgroups traced', int(ok[:, 0].sum()), 'steps traced per workgroup', nst)
t0, tend = t[:, :, 15, 0], t[:, :, 15, 1]
tot = (tend - t0)[ok]
print('wave lifetime: mean %.0f  min %.0f  max %.0f cycles' % (tot.mean(), tot.min(), tot.max()))
print('prologue (entry -> first step top): mean %.0f' % ((t[:, :, 0, 0] - t0)[ok].mean()))
names = ['wait dma', 'barrier', 'dma issue + prefetch', 'patch reads', 'transform', 'frag reads + mfma', 'epilogue']
for st in range(nst):
    seg = [(t[:, :, st, i + 1] - t[:, :, st, i])[ok].mean() for i in range(7)]
    print('step %2d: ' % st + '  '.join('%s %.0f' % (n, v) for n, v in zip(names, seg)) + '   total %.0f' % sum(seg))
base = t0[ok].min()
for wg in (0, 1, 255, 256, 511, 512, 700, 1023):
    if ok[wg, 0]:
        print('wg %4d: entry %.0f  end %.0f' % (wg, t0[wg, 0] - base, tend[wg, 0] - base))
