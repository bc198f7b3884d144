"""A/B of the 4x4x1 block-MFMA kernel for 8-cout layers vs the generic halo kernel (tuning key 3: 2 = generic, 4 = TH 8)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pggan_amd as pg
ops, lib = pg.ops, pg._lib.load()
SHAPES = [(3, 1024, 8, 8), (3, 1024, 16, 8), (3, 1024, 8, 16), (9, 1024, 8, 16), (3, 512, 16, 16), (9, 512, 16, 16), (3, 512, 32, 16), (9, 512, 32, 16), (6, 512, 32, 8), (14, 256, 16, 16)]
def run(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for (N, H, ci, co) in SHAPES:
    x = torch.randn(N, H, H, ci, device='cuda'); w = torch.randn(3, 3, co, ci, device='cuda') * 0.05; b = torch.randn(co, device='cuda')
    m = torch.randn(N, H, H, co, device='cuda'); y = torch.empty(N, H, H, co, device='cuda')
    fl = 2.0 * N * H * H * ci * co * 9
    line = 'conv n%d @%d %d->%d:' % (N, H, ci, co)
    for masked in (False, True):
        ref = None
        for mode in (2, -1):
            lib.pg_debug_set_tuning(3, mode)
            f = (lambda: ops.conv2d(x, w, None, N, H, H, 3, 1, 0.5, mask=m, mask_slope=0.2, out=y)) if masked else (lambda: ops.conv2d(x, w, b, N, H, H, 3, 1, 0.5, 0.2, out=y))
            f(); torch.cuda.synchronize()
            if ref is None: ref = y.clone()
            err = float((y - ref).abs().max() / ref.abs().max())
            t = run(f)
            hb = 4.0 * N * H * H * (ci + co * (2 if masked else 1))
            line += '  %s%s %.1fus %.0fTF %.2fTB/s%s' % ('mask ' if masked else '', {2: 'halo', -1: 'b16', 4: 'b8'}[mode], t * 1e6, fl / t / 1e12, hb / t / 1e12, '' if err < 1e-5 else ' ERR %.1e' % err)
    lib.pg_debug_set_tuning(3, -1)
    print(line, flush=True)
