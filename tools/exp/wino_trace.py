"""Cycle-level phase breakdown of conv_wino_kernel (s_memtime stamps, lane 0 of every wave of the first 1024 workgroups).
Build the traced library in the build container:   hipcc ... -DPG_WINO_TRACE -o ab/libpggan_trace.so   (tools/exp/build_trace.sh)
Run on the GPU box:   PGGAN_HIP_LIB=ab/libpggan_trace.so python tools/exp/wino_trace.py N H Cin Cout"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import pggan_amd as pg
ops, lib = pg.ops, pg._lib.load()
N, H, ci, co = [int(v) for v in sys.argv[1:5]]
gen = int(sys.argv[5]) if len(sys.argv) > 5 else 4          # pg_debug_set_wino: 4 first generation, 11 / 12 second
lib.pg_debug_set_wino(gen)
KC = 16 if gen == 4 else 8
x = torch.randn(N, H, H, ci, device='cuda'); w = torch.randn(3, 3, co, ci, device='cuda') * 0.05; b = torch.randn(co, device='cuda')
u = ops.wino_transform_weights(w)
y = torch.empty(N, H, H, co, device='cuda')
for _ in range(3):
    ops.conv2d_wino(x, u, b, N, H, H, 0.5, 0.2, out=y)
tr = torch.zeros(1024 * 4 * 8 * 8, dtype=torch.int64, device='cuda')
lib.pg_debug_wino_trace.argtypes = [ctypes.c_void_p]
lib.pg_debug_wino_trace(tr.data_ptr())
ops.conv2d_wino(x, u, b, N, H, H, 0.5, 0.2, out=y)
torch.cuda.synchronize()
lib.pg_debug_wino_trace(None)
t = tr.cpu().numpy().reshape(1024, 4, 8, 8).astype(np.float64)
nch = min(8, ci // KC)
ok = t[:, :, 0, 7] > 0
print('workgroups traced', int(ok[:, 0].sum()), 'chunks', nch)
t0 = t[:, :, 0, 7]                       # kernel entry of the wave
tend = t[:, :, 1, 7]
names = ['lds store', 'barrier1', 'fetch issue', 'patch reads + transform', 'mfma issue', 'barrier2'] if gen == 4 else \
    ['wait dma', 'barrier', 'patch reads + dma issue', 'transform', 'frag reads + mfma issue', '-']
tot = (tend - t0)[ok]
print('wave lifetime: mean %.0f  min %.0f  max %.0f cycles' % (tot.mean(), tot.min(), tot.max()))
print('prologue (entry -> first chunk top): mean %.0f' % ((t[:, :, 0, 0] - t0)[ok].mean()))
for c in range(nch):
    seg = [(t[:, :, c, i + 1] - t[:, :, c, i])[ok].mean() for i in range(6)]
    nxt = (t[:, :, c + 1, 0] - t[:, :, c, 6])[ok].mean() if c + 1 < nch else float('nan')
    print('chunk %d: ' % c + '  '.join('%s %.0f' % (n, v) for n, v in zip(names, seg)) + '   total %.0f  (to next top %.0f)' % (sum(seg), nxt))
print('epilogue (last barrier2 -> end): mean %.0f' % ((tend - t[:, :, nch - 1, 6])[ok].mean()))
# concurrency picture: start/end of the first workgroups relative to the earliest entry
base = t0[ok].min()
for wg in (0, 1, 255, 256, 511, 512, 700, 1023):
    if ok[wg, 0]:
        print('wg %4d: entry %.0f  end %.0f' % (wg, t0[wg, 0] - base, tend[wg, 0] - base))
