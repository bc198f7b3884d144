import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pggan_amd as pg
ops, lib = pg.ops, pg._lib.load()
cfgs = [int(v) for v in sys.argv[1:]] or [-1, 9]
SHAPES = [(9, 1024, 8, 8), (9, 1024, 8, 16), (9, 512, 16, 16), (3, 512, 16, 16), (6, 512, 16, 16), (9, 512, 16, 32), (3, 512, 16, 32), (3, 512, 32, 16), (6, 512, 32, 16), (14, 256, 16, 16), (2, 64, 16, 32)]
def run(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for (N, H, ci, co) in SHAPES:
    x = torch.randn(N, H, H, ci, device='cuda'); gz = torch.randn(N, H, H, co, device='cuda')
    dw = torch.zeros(3, 3, co, ci, device='cuda'); db = torch.zeros(co, device='cuda')
    fl = 2.0 * N * H * H * ci * co * 9
    ref = None
    line = 'wgrad n%d @%d %d->%d:' % (N, H, ci, co)
    for c in cfgs:
        lib.pg_debug_set_tuning(1, c)
        dw.zero_(); db.zero_()
        try:
            ops.conv2d_wgrad(x, gz, dw, db, N, H, H, 3, 1, 0.5)
        except RuntimeError as e:
            line += '  [%d] unsup' % c
            continue
        torch.cuda.synchronize()
        if ref is None: ref = dw.clone()
        err = float((dw - ref).abs().max() / ref.abs().max())
        t = run(lambda: ops.conv2d_wgrad(x, gz, dw, db, N, H, H, 3, 1, 0.5))
        line += '  [%d] %.1fus %.0fTF %.2fTB/s%s' % (c, t * 1e6, fl / t / 1e12, 4.0 * N * H * H * (ci + co) / t / 1e12, '' if err < 1e-4 else ' ERR%.1e' % err)
    lib.pg_debug_set_tuning(1, -1)
    print(line, flush=True)
