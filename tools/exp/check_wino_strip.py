"""Strip vs tile kernel, every specialised epilogue, several shapes: prints max abs difference (0 expected: same sums, same order)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pggan_amd as pg
ops, lib = pg.ops, pg._lib.load()
g = torch.Generator(device='cuda').manual_seed(3)
for (N, H, ci, co) in [(1, 128, 8, 16), (1, 128, 16, 16), (1, 128, 16, 32), (1, 128, 8, 32), (1, 128, 32, 16), (1, 128, 32, 64), (2, 256, 16, 32)]:
    x = torch.randn(N, H, H, ci, device='cuda', generator=g)
    u = ops.wino_transform_weights(torch.randn(3, 3, co, ci, device='cuda', generator=g) * 0.2)
    b = torch.randn(co, device='cuda', generator=g)
    mb = (torch.randn(N, H, H, co // 4, device='cuda', generator=g) > 0).to(torch.uint8) * 5
    umb = (torch.randn(N, 2 * H, 2 * H, co // 4, device='cuda', generator=g) > 0).to(torch.uint8) * 9
    oth = torch.randn(N, H // 2, H // 2, co, device='cuda', generator=g)
    forms = {
        'plain': lambda: [ops.conv2d_wino(x, u, b, N, H, H, 0.37, 0.2)],
        'maskb': lambda: [ops.conv2d_wino(x, u, None, N, H, H, 0.37, mask=mb, mask_slope=0.2)],
        'signs': lambda: list(ops.conv2d_wino(x, u, b, N, H, H, 0.37, 0.2, signs_out=True)),
        'poolb': lambda: list(ops.conv2d_wino(x, u, b, N, H, H, 0.37, 0.2, pool=True, other=oth, a=0.6, b=0.4, y_bytes=True)),
        'mpool': lambda: [ops.conv2d_wino(x, u, None, N, H, H, 0.37, mask=mb, mask_slope=0.2, pool=True, pool_only=True)[1]],
        'unpool': lambda: [ops.conv2d_wino(x, u, None, N, H, H, 0.37, mask_slope=0.2, unpool=True, upmask=umb, up_mul=0.7)],
    }
    for name, fn in forms.items():
        lib.pg_debug_set_wino(0)
        a = fn(); ka = lib.pg_debug_last_wino_kernel().decode()
        lib.pg_debug_set_wino(20)
        t = fn()
        lib.pg_debug_set_wino(0)
        d = max(float((p.float() - q.float()).abs().max()) for p, q in zip(a, t))
        print('%d %d %d->%d %-7s %-36s maxdiff %.3g' % (N, H, ci, co, name, ka, d), flush=True)
