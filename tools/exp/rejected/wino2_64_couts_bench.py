"""conv_wino2_kernel with 16 / 32 / 64 couts per workgroup (pg_debug_set_wino 11 / 12 / 14) on the wide layers, cold rotating inputs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pggan_amd as pg
ops, lib = pg.ops, pg._lib.load()
ROT = 4
def timeit(fn, reps=16):
    for i in range(ROT): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = torch.Generator(device='cuda').manual_seed(1)
for (N, H, ci, co) in [(9, 64, 128, 256), (9, 64, 128, 128), (9, 128, 64, 128), (9, 128, 64, 64), (9, 32, 256, 512), (9, 32, 256, 256), (9, 16, 512, 512), (9, 256, 32, 64),
                       (3, 64, 128, 256), (3, 128, 64, 128), (3, 32, 256, 512), (3, 16, 512, 512)]:
    xs = [torch.randn(N, H, H, ci, device='cuda', generator=g) for _ in range(ROT)]
    u = ops.wino_transform_weights(torch.randn(3, 3, co, ci, device='cuda', generator=g) * 0.05)
    b = torch.randn(co, device='cuda', generator=g)
    fl = 2.0 * N * H * H * ci * co * 9
    res = []
    ref = None
    for v in (0, 11, 12, 14):
        lib.pg_debug_set_wino(v)
        try:
            y = ops.conv2d_wino(xs[0], u, b, N, H, H, 0.5, 0.2)
        except RuntimeError as e:
            res.append('%d: unsupported' % v); continue
        if ref is None: ref = y
        err = float((y - ref).abs().max())
        ts = min(timeit(lambda i: ops.conv2d_wino(xs[i % ROT], u, b, N, H, H, 0.5, 0.2)) for _ in range(3))
        res.append('%d: %6.1f us %5.1f TF err %.0e [%s]' % (v, ts, fl / ts / 1e6, err, lib.pg_debug_last_wino_kernel().decode()[17:]))
    lib.pg_debug_set_wino(0)
    print('n%d @%-3d %3d->%3d | %s' % (N, H, ci, co, ' | '.join(res)), flush=True)
