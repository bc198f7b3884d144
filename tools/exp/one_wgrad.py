"""One thin weight-gradient layer (for rocprofv3 --pmc): python tools/exp/one_wgrad.py N H Cin Cout [unpooled]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pggan_amd as pg
ops = pg.ops
N, H, ci, co = [int(v) for v in sys.argv[1:5]]
unp = len(sys.argv) > 5
x = torch.randn(N, H, H, ci, device='cuda')
dw = torch.zeros(3, 3, co, ci, device='cuda'); db = torch.zeros(co, device='cuda')
if unp:
    g = torch.randn(N, H // 2, H // 2, co, device='cuda'); gb = torch.randint(0, 16, (N, H, H, co // 4), device='cuda', dtype=torch.uint8)
    f = lambda: ops.conv2d_wgrad_unpooled(x, g, gb, 0.25, 0.2, dw, db, N, H, H, 0.5)
else:
    gz = torch.randn(N, H, H, co, device='cuda')
    f = lambda: ops.conv2d_wgrad(x, gz, dw, db, N, H, H, 3, 1, 0.5)
for _ in range(3): f()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): f()
torch.cuda.synchronize(); print('%.1f us' % ((time.perf_counter() - t0) / 5 * 1e6), pg._lib.load().pg_debug_last_conv_kernel().decode())
