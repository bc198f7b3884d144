"""Third-generation Winograd conv (csrc/conv_wino3.hip: pipelined K loop, 16 / 32 / 64 couts per workgroup = pg_debug_set_wino 31 / 32 / 34)
against the built-in choice on the wide layers of the 1024x1024 schedule; cold rotating inputs, results compared bit for bit."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
SHAPES = [(9, 64, 128, 256), (9, 64, 128, 128), (9, 64, 256, 128), (9, 128, 64, 128), (9, 128, 64, 64), (9, 128, 128, 64), (9, 32, 256, 512), (9, 32, 256, 256), (9, 32, 512, 256),
          (9, 16, 512, 512), (9, 256, 32, 64), (9, 256, 64, 32), (3, 64, 128, 256), (3, 64, 128, 128), (3, 128, 64, 128), (3, 128, 64, 64), (3, 32, 256, 512), (3, 32, 256, 256), (3, 16, 512, 512)]
for (N, H, ci, co) in SHAPES:
    xs = [torch.randn(N, H, H, ci, device='cuda', generator=g) for _ in range(ROT)]
    u = ops.wino_transform_weights(torch.randn(3, 3, co, ci, device='cuda', generator=g) * 0.05)
    b = torch.randn(co, device='cuda', generator=g)
    fl = 2.0 * N * H * H * ci * co * 9
    res, ref = [], None
    for v in (0, 31, 32, 34):
        lib.pg_debug_set_wino(v)
        try:
            y = ops.conv2d_wino(xs[0], u, b, N, H, H, 0.5, 0.2)
        except RuntimeError:
            res.append('%d: unsupported' % v); continue
        if ref is None: ref = y
        same = bool(torch.equal(y, ref)) or float((y - ref).abs().max()) < 1e-5 * float(ref.abs().max())
        ts = min(timeit(lambda i: ops.conv2d_wino(xs[i % ROT], u, b, N, H, H, 0.5, 0.2)) for _ in range(3))
        res.append('%d: %6.1f us %5.1f TF %s' % (v, ts, fl / ts / 1e6, 'ok' if same else 'MISMATCH'))
    lib.pg_debug_set_wino(0)
    print('n%d @%-3d %3d->%3d | %s' % (N, H, ci, co, ' | '.join(res)), flush=True)
