import importlib, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
pg = importlib.import_module('pggan-pytorch_amd')
ops, lib = pg.ops, pg._lib.load()
def timed(fn, reps=20):
    for i in range(3): fn(i)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for i in range(reps): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for N, H, ci, co in ((9, 1024, 8, 16), (3, 1024, 8, 16), (9, 512, 16, 16), (9, 512, 16, 32)):
    xs = [torch.randn(N, H, H, ci, device='cuda') for _ in range(3)]
    w = torch.randn(3, 3, co, ci, device='cuda') * 0.2; b = torch.randn(co, device='cuda')
    u = ops.wino_transform_weights(w)
    t = timed(lambda i: ops.conv2d_wino(xs[i % 3], u, b, N, H, H, 0.5, 0.2, pool=True, y_bytes=True))
    t2 = timed(lambda i: ops.conv2d_wino(xs[i % 3], u, b, N, H, H, 0.5, 0.2))
    print('n%d @%d %d->%d: pool+bytes %.1f us   plain %.1f us  (%s)' % (N, H, ci, co, t, t2, lib.pg_debug_last_wino_kernel().decode()), flush=True)
