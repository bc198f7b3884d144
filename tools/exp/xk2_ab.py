"""Stage mode of the Winograd tile kernel (PG_WINO_XK=2: 16 channels per DMA wait + barrier, 80 KB of LDS) vs the 8-channel form, on the
wide layers of the 1024^2 step.  Run once per setting (the switch is read once per process); prints us per launch and a checksum."""
import hashlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pggan_amd as pg
ops, lib = pg.ops, pg._lib.load()
ROT = 4
def timeit(fn, reps=20):
    for i in range(ROT): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = torch.Generator(device='cuda').manual_seed(1)
CASES = [(9, 256, 32, 32, 'signs'), (9, 256, 32, 64, 'pool'), (9, 256, 64, 32, 'maskb'), (9, 128, 64, 64, 'signs'), (9, 128, 64, 128, 'pool'), (9, 128, 128, 64, 'maskb'),
         (9, 64, 128, 128, 'signs'), (9, 64, 128, 256, 'pool'), (9, 64, 256, 128, 'maskb'), (9, 32, 256, 256, 'plain'), (9, 32, 256, 512, 'pool'), (9, 32, 512, 256, 'mask32'),
         (3, 256, 64, 32, 'maskb'), (3, 128, 128, 64, 'maskb'), (3, 128, 64, 64, 'plain'), (3, 256, 32, 64, 'pool')]
for n, H, ci, co, kind in CASES:
    xs = [torch.randn(n, H, H, ci, device='cuda', generator=g) for _ in range(ROT)]
    u = ops.wino_transform_weights(torch.randn(3, 3, co, ci, device='cuda', generator=g) * 0.2)
    b = torch.randn(co, device='cuda', generator=g)
    mb = (torch.randn(n, H, H, co // 4, device='cuda', generator=g) > 0).to(torch.uint8) * 5
    m32 = torch.randn(n, H, H, co, device='cuda', generator=g)
    if kind == 'plain': fn = lambda i: ops.conv2d_wino(xs[i % ROT], u, b, n, H, H, 0.37, 0.2)
    elif kind == 'signs': fn = lambda i: ops.conv2d_wino(xs[i % ROT], u, b, n, H, H, 0.37, 0.2, signs_out=True)
    elif kind == 'pool': fn = lambda i: ops.conv2d_wino(xs[i % ROT], u, b, n, H, H, 0.37, 0.2, pool=True, y_bytes=True)
    elif kind == 'maskb': fn = lambda i: ops.conv2d_wino(xs[i % ROT], u, None, n, H, H, 0.37, mask=mb, mask_slope=0.2)
    else: fn = lambda i: ops.conv2d_wino(xs[i % ROT], u, None, n, H, H, 0.37, mask=m32, mask_slope=0.2)
    t = min(timeit(fn), timeit(fn))
    y = fn(0)
    h = hashlib.sha1()
    for tns in (y if isinstance(y, (tuple, list)) else [y]):
        if torch.is_tensor(tns): h.update(tns.cpu().numpy().tobytes())
    print('%-6s n%d @%-3d %3d->%-3d %7.1f us  %s  %s' % (kind, n, H, ci, co, t, h.hexdigest()[:10], lib.pg_debug_last_wino_kernel().decode()), flush=True)
