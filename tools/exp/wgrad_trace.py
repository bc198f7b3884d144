"""Cycle-level phase breakdown of conv_wgrad_thin_kernel (s_memtime stamps per wave and tile, first 8 tiles of the first 1024
workgroups).  Build: tools/exp/build_trace.sh; run on the GPU box:
    PGGAN_HIP_LIB=ab/libpggan_trace.so python tools/exp/wgrad_trace.py N H Cin Cout"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import pggan_amd as pg
ops, lib = pg.ops, pg._lib.load()
N, H, ci, co = [int(v) for v in sys.argv[1:5]]
wino = len(sys.argv) > 5 and sys.argv[5] == 'wino'           # trace the Winograd weight gradient instead of the block-MFMA kernel
x = torch.randn(N, H, H, ci, device='cuda'); gz = torch.randn(N, H, H, co, device='cuda')
dw = torch.zeros(3, 3, co, ci, device='cuda'); db = torch.zeros(co, device='cuda')
run = (lambda: ops.conv2d_wgrad_wino(x, gz, dw, db, N, H, H, 0.5)) if wino else (lambda: ops.conv2d_wgrad(x, gz, dw, db, N, H, H, 3, 1, 0.5))
for _ in range(3):
    run()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(10):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
sym = (lib.pg_debug_last_wino_wgrad_kernel() if wino else lib.pg_debug_last_conv_kernel()).decode()
print('%s  %.1f us  %.2f TB/s algorithmic  %.1f algorithmic TF' % (sym, dt * 1e6, 4.0 * N * H * H * (ci + co) / dt / 1e12, 2.0 * N * H * H * ci * co * 9 / dt / 1e12))
tr = torch.zeros(1024 * 4 * 8 * 8, dtype=torch.int64, device='cuda')
setter = lib.pg_debug_wino_wgrad_trace if wino else lib.pg_debug_wgrad_trace
setter.argtypes = [ctypes.c_void_p]
setter(tr.data_ptr())
run()
torch.cuda.synchronize()
setter(None)
t = tr.cpu().numpy().reshape(1024, 4, 8, 8).astype(np.float64)
ok = t[:, :, 0, 0] > 0
names = ['lds store (waits for the prefetch) | dma kernel: vmcnt wait', 'barrier', 'fetch issue', 'fragment reads + transforms + mfma' if wino else 'fragment reads + mfma', 'barrier']
for c in range(8):
    seg = [(t[:, :, c, i + 1] - t[:, :, c, i])[ok].mean() for i in range(5)]
    nxt = (t[:, :, c + 1, 0] - t[:, :, c, 5])[ok].mean() if c < 7 else float('nan')
    print('tile %d: ' % c + '  '.join('%s %.0f' % (n, v) for n, v in zip(names, seg)) + '   total %.0f (to next %.0f)' % (sum(seg), nxt))
