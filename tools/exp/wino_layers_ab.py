"""Per-layer timing of the Winograd tile conv on the depth-8 layer shapes, for A/B runs of two builds of the library:
     PGGAN_HIP_LIB=ab/libpggan_x.so python tools/exp/wino_layers_ab.py   (prints one line per layer; HIP events, 3 rotating inputs)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pggan_amd as pg
ops, lib = pg.ops, pg._lib.load()
SHAPES = [(9, 256, 32, 64), (3, 256, 32, 64), (9, 256, 64, 32), (3, 256, 64, 32), (9, 128, 64, 128), (3, 128, 64, 128), (9, 128, 128, 64), (3, 128, 128, 64),
          (9, 64, 128, 256), (3, 64, 128, 256), (9, 64, 256, 128), (3, 64, 256, 128), (9, 32, 256, 512), (9, 32, 512, 256), (3, 32, 512, 256), (3, 32, 256, 256),
          (9, 16, 512, 512), (3, 16, 512, 512), (9, 8, 512, 512), (3, 8, 512, 512), (9, 512, 32, 16), (3, 512, 32, 16)]
def timed(fn, reps=40):
    for i in range(4): fn(i)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for i in range(reps): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
tot = 0.0
for (N, H, ci, co) in SHAPES:
    xs = [torch.randn(N, H, H, ci, device='cuda') for _ in range(3)]
    w = torch.randn(3, 3, co, ci, device='cuda') * 0.05; b = torch.randn(co, device='cuda')
    u = ops.wino_transform_weights(w)
    y = ops.conv2d_wino(xs[0], u, b, N, H, H, 0.5, 0.2)
    t = timed(lambda i: ops.conv2d_wino(xs[i % 3], u, b, N, H, H, 0.5, 0.2, out=y))
    name = lib.pg_debug_last_wino_kernel().decode()
    fl = 2.0 * N * H * H * ci * co * 9
    tot += t
    print('n%d @%d %d->%d: %7.1f us  %5.1f TF alg  %s' % (N, H, ci, co, t, fl / t / 1e6, name), flush=True)
print('sum %.1f us' % tot)
