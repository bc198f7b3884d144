import importlib, os, sys, torch
sys.path.insert(0, '.')
pg = importlib.import_module('pggan-pytorch_amd'); ops = pg.ops
def timed(fn, reps=50):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for N in (3, 6):
    H = 1024; C = 3
    x = torch.randn(N, H, H, 8, device='cuda'); w = torch.randn(3, 3, 8, 8, device='cuda') * 0.2; b = torch.randn(8, device='cuda')
    tw, tb = torch.randn(C, 8, device='cuda'), torch.randn(C, device='cuda')
    out = torch.empty(N, C, H, H, device='cuda')
    t1 = timed(lambda: ops.conv2d_pixelnorm(x, w, b, N, H, H, 3, 1, 0.4, 0.2, 1e-8))
    y, r = ops.conv2d_pixelnorm(x, w, b, N, H, H, 3, 1, 0.4, 0.2, 1e-8)
    t2 = timed(lambda: ops.torgb_fwd(y, tw, tb, N, C, H, H, 0.7, out=out))
    t12 = timed(lambda: ops.torgb_fwd(ops.conv2d_pixelnorm(x, w, b, N, H, H, 3, 1, 0.4, 0.2, 1e-8)[0], tw, tb, N, C, H, H, 0.7, out=out))
    t3 = timed(lambda: ops.conv2d_pixelnorm_torgb(x, w, b, tw, tb, N, C, H, H, 0.4, 0.2, 0.7, 1e-8, out=out))
    print('n%d: conv+pn %.1f us, torgb %.1f us, back to back %.1f | fused %.1f us' % (N, t1, t2, t12, t3))
