"""Pool-adjoint (unpool) epilogue of the Winograd tile kernel: 16 vs 32 couts per workgroup (pg_debug_set_wino(11) / (12)) on the
layers of the 1024^2 step that take it -- 32 couts per workgroup write whole 128-byte lines of the 4x-sized output."""
import os, sys
import torch
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
for n, H, ci, co, bytes_ in [(9, 256, 32, 32, True), (9, 128, 64, 64, True), (3, 128, 64, 64, True), (9, 64, 128, 128, True), (3, 64, 128, 128, True), (9, 32, 256, 256, True), (3, 32, 256, 256, True),
                             (9, 16, 512, 512, False), (3, 16, 512, 512, False), (3, 8, 512, 512, False)]:
    xs = [torch.randn(n, H, H, ci, device='cuda', generator=g) for _ in range(ROT)]
    u = ops.wino_transform_weights(torch.randn(3, 3, co, ci, device='cuda', generator=g) * 0.2)
    um = (torch.randn(n, 2 * H, 2 * H, co // 4, device='cuda', generator=g) > 0).to(torch.uint8) * 5 if bytes_ else torch.randn(n, 2 * H, 2 * H, co, device='cuda', generator=g)
    fn = lambda i: ops.conv2d_wino(xs[i % ROT], u, None, n, H, H, 0.37, mask_slope=0.2, unpool=True, upmask=um, up_mul=0.7)
    out = []
    for mode in (0, 11, 12, 0, 11, 12):
        lib.pg_debug_set_wino(mode)
        out.append((mode, timeit(fn), lib.pg_debug_last_wino_kernel().decode()))
    lib.pg_debug_set_wino(0)
    print('unpool n%d @%-3d %3d->%-3d ' % (n, H, ci, co) + '  '.join('%d: %6.1f us' % (m, t) for m, t, _ in out) + '   [' + out[0][2] + ' | ' + out[2][2] + ']', flush=True)
