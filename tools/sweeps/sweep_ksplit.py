"""In-workgroup K-split conv kernel for small-M layers vs the generic split-K path (tuning key 3: 8 = generic, 9 = extend to M <= 2304)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pggan_amd as pg
ops, lib = pg.ops, pg._lib.load()
SHAPES = [(3, 4, 512, 512), (9, 4, 512, 512), (3, 4, 528, 512), (3, 8, 512, 512), (9, 8, 512, 512), (3, 16, 512, 512), (9, 16, 512, 512), (16, 4, 512, 512), (48, 4, 512, 512), (16, 8, 512, 512),
          (3, 16, 256, 512), (3, 32, 256, 256)]
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
        for mode in (8, 9):
            lib.pg_debug_set_tuning(3, mode)
            f = (lambda: ops.conv2d(x, w, None, N, H, H, 3, 1, 0.5, mask=m, mask_slope=0.2, out=y)) if masked else (lambda: ops.conv2d(x, w, b, N, H, H, 3, 1, 0.5, 0.2, out=y))
            f(); torch.cuda.synchronize()
            if ref is None: ref = y.clone()
            err = float((y - ref).abs().max() / ref.abs().max())
            t = run(f)
            line += '  %s%s %.1fus %.0fTF%s' % ('mask ' if masked else '', lib.pg_debug_last_conv_kernel().decode().replace('conv_', '').replace('_kernel', ''), t * 1e6, fl / t / 1e12, '' if err < 2e-5 else ' ERR %.1e' % err)
    lib.pg_debug_set_tuning(3, -1)
    print(line, flush=True)
