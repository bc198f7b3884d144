"""All conv tile candidates (built-in split-K rule) on every 3x3 layer shape of the depth-0..8 schedule."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pggan_amd as pg
ops, lib = pg.ops, pg._lib.load()
nf = [512, 512, 512, 512, 256, 128, 64, 32, 16, 8]
shapes = set()
for n in (3, 9):                      # depth 8 minibatch 3: single / batched [real|fake|mixed]
    for s in range(1, 9):
        r = 4 * 2 ** s
        a, b = nf[s], nf[s + 1] if s + 1 < 10 else None
        # G block s: c1 nf(s)->nf(s+1)?  use D naming: block at res r has channels nf(s) (in) -> nf(s-1) (out)
        shapes.add((n, r, nf[s + 1] if s + 1 < 10 else nf[s], nf[s + 1] if s + 1 < 10 else nf[s]))
for n in (3, 9):
    for s in range(0, 9):
        r = 4 * 2 ** s
        cin_hi, cin_lo = nf[s + 1], nf[s]          # D block at res r: c1 hi->hi, c2 hi->lo ; G block: lo->hi (ups), hi->hi
        shapes.add((n, r, cin_hi, cin_hi)); shapes.add((n, r, cin_hi, cin_lo)); shapes.add((n, r, cin_lo, cin_hi))
for n in (16, 48):                     # low depths use minibatch 16
    for s in range(0, 4):
        r = 4 * 2 ** s
        shapes.add((n, r, 512, 512))
shapes = sorted(shapes, key=lambda t: (t[1], t[0], t[2], t[3]))
def run(f, reps=8):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for (N, H, ci, co) in shapes:
    x = torch.randn(N, H, H, ci, device='cuda'); w = torch.randn(3, 3, co, ci, device='cuda') * 0.05; b = torch.randn(co, device='cuda')
    y = torch.empty(N, H, H, co, device='cuda')
    fl = 2.0 * N * H * H * ci * co * 9
    res = {}
    for c in [-1, 0, 1, 2, 3, 4, 5, 6, 7]:
        lib.pg_debug_set_tuning(0, c)
        try:
            ops.conv2d(x, w, b, N, H, H, 3, 1, 0.5, 0.2, out=y)
        except RuntimeError:
            continue
        res[c] = (run(lambda: ops.conv2d(x, w, b, N, H, H, 3, 1, 0.5, 0.2, out=y)), lib.pg_debug_last_conv_kernel().decode())
    lib.pg_debug_set_tuning(0, -1)
    best = min((k for k in res if k >= 0), key=lambda k: res[k][0])
    print('n%-2d @%-4d %3d->%-3d  auto %6.1fus %5.1fTF %-16s best c%d %6.1fus %5.1fTF (%+.0f%%) | ' % (
        N, H, ci, co, res[-1][0] * 1e6, fl / res[-1][0] / 1e12, res[-1][1].replace('conv_igemm_kernel', ''), best, res[best][0] * 1e6,
        fl / res[best][0] / 1e12, 100 * (res[-1][0] / res[best][0] - 1)) + ' '.join('c%d:%.0f' % (k, res[k][0] * 1e6) for k in sorted(res) if k >= 0), flush=True)
