"""Does a launch pay for a cold instruction cache?  Kernel A (a 3-image Winograd tile-kernel launch) timed with HIP-event pairs when it
follows itself (AAAA...) and when every A follows launches of other kernels with large code (weight gradient, row-streaming conv,
another Winograd variant) that evict it (ABCD A BCD A ...).  Inputs rotate, so the data caches see the same thing in both patterns."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pggan_amd as pg
ops = pg.ops
def mk(N, H, ci, co):
    xs = [torch.randn(N, H, H, ci, device='cuda') for _ in range(3)]
    w = torch.randn(3, 3, co, ci, device='cuda') * 0.05
    return xs, ops.wino_transform_weights(w), torch.randn(co, device='cuda'), w
for shape in ((3, 256, 32, 64), (3, 64, 128, 256), (3, 16, 512, 512)):
    N, H, ci, co = shape
    xs, u, b, w = mk(*shape)
    y = ops.conv2d_wino(xs[0], u, b, N, H, H, 0.5, 0.2)
    # evictors: different symbols
    x2, u2, b2, w2 = mk(3, 128, 64, 64)
    m2 = (torch.rand(3, 128, 128, 16, device='cuda') * 255).to(torch.uint8)
    g3 = torch.randn(3, 64, 64, 128, device='cuda'); x3 = torch.randn(3, 64, 64, 128, device='cuda')
    dw3, db3 = torch.zeros(3, 3, 128, 128, device='cuda'), torch.zeros(128, device='cuda')
    x4 = torch.randn(3, 1024, 1024, 8, device='cuda'); w4 = torch.randn(3, 3, 8, 8, device='cuda') * 0.1; b4 = torch.randn(8, device='cuda')
    def evict(i):
        ops.conv2d_wino(x2[i % 3], u2, None, 3, 128, 128, 0.5, mask=m2)
        ops.conv2d_wgrad_wino(x3, g3, dw3, db3, 3, 64, 64, 0.5)
        ops.conv2d(x4, w4, b4, 3, 1024, 1024, 3, 1, 0.5, 0.2)
    def timed(pattern, reps=30):
        pairs = []
        for i in range(reps + 3):
            if pattern == 'cold':
                evict(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.conv2d_wino(xs[i % 3], u, b, N, H, H, 0.5, 0.2, out=y); e1.record()
            if i >= 3:
                pairs.append((e0, e1))
        torch.cuda.synchronize()
        v = sorted(a.elapsed_time(c) * 1e3 for a, c in pairs)
        return v[len(v) // 2]
    warm, cold = timed('warm'), timed('cold')
    warm2, cold2 = timed('warm'), timed('cold')
    print('A = n%d @%d %d->%d: follows itself %.1f / %.1f us, follows three other kernels %.1f / %.1f us (HIP-event pairs incl. ~7 us of pair overhead)' % (N, H, ci, co, warm, warm2, cold, cold2), flush=True)
