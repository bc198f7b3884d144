"""GPU timing of pg_conv2d_nhwc on the depth-8 layer shapes (tuning aid; PGGAN_HIP_LIB selects an experimental build).
usage: python tools/sweeps/sweep_conv.py [tile candidates...]   (pg_debug_set_tuning key 0; -1 = built-in cost model)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pggan_amd as pg
ops, lib = pg.ops, pg._lib.load()
cfgs = [int(v) for v in sys.argv[1:]] or [-1]
SHAPES = [(9, 16, 512, 512), (3, 16, 512, 512), (9, 32, 256, 512), (3, 32, 256, 256), (9, 64, 128, 256), (3, 64, 128, 128),
          (9, 128, 64, 128), (3, 128, 64, 64), (9, 256, 32, 64), (3, 256, 32, 32), (9, 512, 16, 32), (3, 512, 16, 16),
          (3, 1024, 8, 8), (3, 1024, 16, 8), (9, 8, 512, 512), (3, 8, 512, 512), (9, 4, 512, 512)]
def run(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for (N, H, ci, co) in SHAPES:
    x = torch.randn(N, H, H, ci, device='cuda'); w = torch.randn(3, 3, co, ci, device='cuda') * 0.05; b = torch.randn(co, device='cuda')
    y = torch.empty(N, H, H, co, device='cuda')
    fl = 2.0 * N * H * H * ci * co * 9
    line = 'conv n%d @%d %d->%d:' % (N, H, ci, co)
    for c in cfgs:
        lib.pg_debug_set_tuning(0, c)
        try:
            ops.conv2d(x, w, b, N, H, H, 3, 1, 0.5, 0.2, out=y)
        except RuntimeError:
            line += '  [%d] unsup' % c
            continue
        t = run(lambda: ops.conv2d(x, w, b, N, H, H, 3, 1, 0.5, 0.2, out=y))
        line += '  [%d] %s %.1fus %.0fTF' % (c, lib.pg_debug_last_conv_kernel().decode().replace('conv_igemm_kernel', ''), t * 1e6, fl / t / 1e12)
    lib.pg_debug_set_tuning(0, -1)
    print(line, flush=True)
