"""HBM rate of the 1x1 RGB kernels at the 1024^2 / 512^2 stages (bytes = the tensors each kernel must touch once)."""
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
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for (N, H, C) in [(9, 1024, 8), (3, 1024, 8), (9, 512, 16), (3, 512, 16)]:
    img = torch.randn(N, 3, H, H, device='cuda'); w = torch.randn(C, 3, device='cuda'); b = torch.randn(C, device='cuda')
    x = torch.randn(N, H, H, C, device='cuda'); wt = torch.randn(3, C, device='cuda'); bt = torch.randn(3, device='cuda')
    gimg = torch.empty_like(img); out = torch.empty_like(img)
    px = N * H * H
    t = run(lambda: ops.fromrgb_fwd(img, w, b, N, 3, H, H, 0.5, 0.2))
    print('n%d @%d C%d fromrgb_fwd      %.1fus %.2f TB/s' % (N, H, C, t * 1e6, px * 4 * (3 + C) / t / 1e12))
    t = run(lambda: ops.fromrgb_fwd(img, w, None, N, 3, H, H, 0.5, 0.2, mask=x))
    print('n%d @%d C%d fromrgb_fwd mask %.1fus %.2f TB/s' % (N, H, C, t * 1e6, px * 4 * (3 + 2 * C) / t / 1e12))
    t = run(lambda: ops.fromrgb_bwd_data(x, w, gimg, N, 3, H, H, 0.5))
    print('n%d @%d C%d fromrgb_bwd_data %.1fus %.2f TB/s' % (N, H, C, t * 1e6, px * 4 * (3 + C) / t / 1e12))
    t = run(lambda: ops.torgb_fwd(x, wt, bt, N, 3, H, H, 0.5, out=out))
    print('n%d @%d C%d torgb_fwd        %.1fus %.2f TB/s' % (N, H, C, t * 1e6, px * 4 * (3 + C) / t / 1e12))
    t = run(lambda: ops.torgb_bwd_data(img, wt, N, 3, H, H, 0.5))
    print('n%d @%d C%d torgb_bwd_data   %.1fus %.2f TB/s' % (N, H, C, t * 1e6, px * 4 * (3 + C) / t / 1e12), flush=True)
