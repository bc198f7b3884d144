"""8->16 conv + pool with sign bytes out at 1024^2: generic tile kernel vs the block-MFMA kernel (default; tuning key 3 = 17 selects the generic kernel)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pggan_amd as pg
ops, lib = pg.ops, pg._lib.load()
def run(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
for N in (9, 3):
    H = 1024
    x = torch.randn(N, H, H, 8, device='cuda'); w = torch.randn(3, 3, 16, 8, device='cuda') * 0.1; b = torch.randn(16, device='cuda')
    res = {}
    for mode in (17, -1):
        lib.pg_debug_set_tuning(3, mode)
        yb, yp = ops.conv2d_pool(x, w, b, N, H, H, 3, 1, 0.4, 0.2, y_bytes=True)
        res[mode] = (yb.clone(), yp.clone(), run(lambda: ops.conv2d_pool(x, w, b, N, H, H, 3, 1, 0.4, 0.2, y_bytes=True)), lib.pg_debug_last_conv_kernel().decode())
    lib.pg_debug_set_tuning(3, -1)
    same = torch.equal(res[-1][0], res[17][0]), float((res[-1][1] - res[17][1]).abs().max())
    print('n%d: %s %.0f us | %s %.0f us | bytes equal %s, pooled max diff %.1e' % (N, res[17][3], res[17][2], res[-1][3], res[-1][2], same[0], same[1]), flush=True)
