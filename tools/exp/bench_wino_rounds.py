import importlib, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
pg = importlib.import_module('pggan-pytorch_amd')
ops, lib = pg.ops, pg._lib.load()
def timed(fn, reps=30):
    for i in range(3): fn(i)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for i in range(reps): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for H, ci, co in ((1024, 8, 16), (512, 16, 16), (512, 8, 16), (1024, 16, 16)):
    for N in (1, 2, 3, 4, 6, 9):
        if N * H * H * max(ci, co) * 4 > 3e9: continue
        xs = [torch.randn(N, H, H, ci, device='cuda') for _ in range(3)]
        w = torch.randn(3, 3, co, ci, device='cuda') * 0.2; b = torch.randn(co, device='cuda')
        u = ops.wino_transform_weights(w)
        t = timed(lambda i: ops.conv2d_wino(xs[i % 3], u, b, N, H, H, 0.5, 0.2))
        wgs = N * (H // 16) ** 2
        print('@%d %d->%d n%d: %6.1f us  %5d workgroups  %.1f per us   %.2f TB/s' % (H, ci, co, N, t, wgs, wgs / t, N * H * H * (ci + co) * 4 / t * 1e-6), flush=True)
