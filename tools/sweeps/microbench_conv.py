"""GPU microbenchmark of pg_conv2d_nhwc / wgrad on given shapes (tuning aid).
usage: python tools/sweeps/microbench_conv.py N H Cin Cout [mask]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pggan_amd as pg
ops = pg.ops
N, H, ci, co = [int(v) for v in sys.argv[1:5]]
mask = len(sys.argv) > 5
x = torch.randn(N, H, H, ci, device='cuda'); w = torch.randn(3, 3, co, ci, device='cuda') * 0.1; b = torch.randn(co, device='cuda')
m = torch.randn(N, H, H, co, device='cuda') if mask else None
y = torch.empty(N, H, H, co, device='cuda')
def run(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
fl = 2.0 * N * H * H * ci * co * 9
hb = 4.0 * N * H * H * (ci + co * (2 if mask else 1))
for cfg in os.environ.get('CFGS', '').split(',') if os.environ.get('CFGS') else [None]:
    if cfg is not None: os.environ['PGGAN_CONV_CFG'] = cfg
    t = run(lambda: ops.conv2d(x, w, None if mask else b, N, H, H, 3, 1, 0.5, 0.2, mask=m, out=y))
    print('conv cfg %-4s %s  %.1f us  %.1f TFLOP/s  %.2f TB/s' % (cfg, pg._lib.load().pg_debug_last_conv_kernel().decode(), t * 1e6, fl / t / 1e12, hb / t / 1e12))
gz = torch.randn(N, H, H, co, device='cuda'); dw = torch.zeros(3, 3, co, ci, device='cuda'); db = torch.zeros(co, device='cuda')
t = run(lambda: ops.conv2d_wgrad(x, gz, dw, db, N, H, H, 3, 1, 0.5))
print('wgrad %s  %.1f us  %.1f TFLOP/s  %.2f TB/s' % (pg._lib.load().pg_debug_last_conv_kernel().decode(), t * 1e6, fl / t / 1e12, 4.0 * N * H * H * (ci + co) / t / 1e12))
