"""Winograd F(2x2,3x3) conv vs the direct implicit-GEMM kernel on the wide layers."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pggan_amd as pg
ops, lib = pg.ops, pg._lib.load()
SHAPES = [(9, 16, 512, 512), (3, 16, 512, 512), (9, 32, 256, 512), (3, 32, 256, 256), (9, 64, 128, 256), (3, 64, 128, 128), (9, 128, 64, 128), (3, 128, 64, 64),
          (9, 256, 32, 64), (3, 256, 32, 32), (9, 512, 32, 16), (9, 8, 512, 512), (3, 8, 512, 512), (16, 16, 512, 512), (48, 16, 512, 512), (16, 32, 512, 512),
          (9, 512, 16, 16), (3, 512, 16, 16), (9, 512, 16, 32), (3, 512, 16, 32), (9, 1024, 16, 16),
          (9, 1024, 8, 16), (3, 1024, 8, 16), (9, 1024, 16, 8), (3, 1024, 16, 8), (9, 1024, 8, 8), (3, 1024, 8, 8)]
if len(sys.argv) > 1 and sys.argv[1] == 'thin':
    SHAPES = SHAPES[-11:]
def run(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for (N, H, ci, co) in SHAPES:
    x = torch.randn(N, H, H, ci, device='cuda'); w = torch.randn(3, 3, co, ci, device='cuda') * 0.05; b = torch.randn(co, device='cuda')
    m = torch.randn(N, H, H, co, device='cuda')
    u = ops.wino_transform_weights(w)
    fl = 2.0 * N * H * H * ci * co * 9
    y0 = ops.conv2d(x, w, b, N, H, H, 3, 1, 0.5, 0.2)
    ym0 = ops.conv2d(x, w, None, N, H, H, 3, 1, 0.5, mask=m)
    t0 = run(lambda: ops.conv2d(x, w, b, N, H, H, 3, 1, 0.5, 0.2, out=y0))
    res = []
    for v in (4, 11, 12, 0):                 # first generation (16-channel chunks); second generation with 16 / 32 couts per workgroup; built-in choice
        lib.pg_debug_set_wino(v)
        try:
            y1 = ops.conv2d_wino(x, u, b, N, H, H, 0.5, 0.2)
        except RuntimeError:
            res.append('%s unsupported' % {4: 'gen1', 11: 'gen2/16', 12: 'gen2/32', 0: 'auto'}[v]); continue
        ym1 = ops.conv2d_wino(x, u, None, N, H, H, 0.5, mask=m)
        e = max(float((y1 - y0).abs().max() / y0.abs().max()), float((ym1 - ym0).abs().max() / ym0.abs().max()))
        t = run(lambda: ops.conv2d_wino(x, u, b, N, H, H, 0.5, 0.2, out=y1))
        res.append('%s %.1fus %.0fTF (%.2fx) err %.0e' % ({4: 'gen1', 11: 'gen2/16', 12: 'gen2/32', 0: 'auto'}[v], t * 1e6, fl / t / 1e12, t0 / t, e))
    lib.pg_debug_set_wino(4)
    print('conv n%d @%d %d->%d: direct %.1fus %.0fTF | %s' % (N, H, ci, co, t0 * 1e6, fl / t0 / 1e12, ' | '.join(res)), flush=True)
