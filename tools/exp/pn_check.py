"""PixelNorm epilogue of the Winograd conv vs the plain epilogue vs conv + separate PixelNorm pass vs the direct fused kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pggan_amd as pg
ops = pg.ops
def run(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
for (N, H, ci, co, ups) in [(3, 512, 16, 16, 0), (3, 512, 32, 16, 1), (3, 256, 32, 32, 0), (3, 256, 64, 32, 1), (3, 128, 64, 64, 0)]:
    hin = H // 2 if ups else H
    x = torch.randn(N, hin, hin, ci, device='cuda'); w = torch.randn(3, 3, co, ci, device='cuda') * 0.05; b = torch.randn(co, device='cuda')
    u = ops.wino_transform_weights(w)
    y = torch.empty(N, H, H, co, device='cuda')
    t_plain = run(lambda: ops.conv2d_wino(x, u, b, N, H, H, 0.5, 0.2, ups=bool(ups), out=y))
    t_sep = run(lambda: ops.pixelnorm_fwd(ops.conv2d_wino(x, u, b, N, H, H, 0.5, 0.2, ups=bool(ups), out=y), 1e-8, inplace=True))
    try:
        t_pn = run(lambda: ops.conv2d_wino_pixelnorm(x, u, b, N, H, H, 0.5, 0.2, 1e-8, ups=bool(ups)))
    except RuntimeError:
        t_pn = float('nan')
    try:
        t_dir = run(lambda: ops.conv2d_pixelnorm(x, w, b, N, H, H, 3, 1, 0.5, 0.2, 1e-8, ups=bool(ups)))
    except RuntimeError:
        t_dir = float('nan')
    t_so = run(lambda: ops.conv2d_wino(x, u, b, N, H, H, 0.5, 0.2, ups=bool(ups), out=y, signs_out=True)) if not ups else float('nan')
    print('signs_out %.1f us' % t_so, end=' | ')
    print('n%d @%d %d->%d ups%d: wino plain %.1f us | wino + pn pass %.1f | wino pn epilogue %.1f | direct fused %.1f' % (N, H, ci, co, ups, t_plain, t_sep, t_pn, t_dir), flush=True)
