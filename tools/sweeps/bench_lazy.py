"""Pool adjoint in the gather vs materialised gz2: backward-data conv and weight gradient of the 8->16 c2 layer at 1024^2."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import pggan_amd as pg
ops = pg.ops
def run(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
for N in (9, 3):
    H, cg, co = 1024, 16, 8
    g = torch.randn(N, H // 2, H // 2, cg, device='cuda'); a2 = torch.randn(N, H, H, cg, device='cuda')
    a1 = torch.randn(N, H, H, co, device='cuda')
    _, _ = None, None
    gb = ((a2 > 0).to(torch.uint8).view(N, H, H, cg // 4, 4) * torch.tensor([1, 2, 4, 8], device='cuda', dtype=torch.uint8)).sum(-1).to(torch.uint8).contiguous()
    a1b = ((a1 > 0).to(torch.uint8).view(N, H, H, co // 4, 4) * torch.tensor([1, 2, 4, 8], device='cuda', dtype=torch.uint8)).sum(-1).to(torch.uint8).contiguous()
    wt = torch.randn(3, 3, co, cg, device='cuda') * 0.1
    dw = torch.zeros(3, 3, cg, co, device='cuda'); db = torch.zeros(cg, device='cuda')
    gz2 = ops.avgpool2_bwd(g, ops.signbytes_to_mask(gb), 1.0, 0.2)
    t_unpool = run(lambda: ops.avgpool2_bwd(g, a2, 1.0, 0.2))
    t_d0 = run(lambda: ops.conv2d(gz2, wt, None, N, H, H, 3, 1, 0.3, mask=a1b, mask_slope=0.2))
    t_d1 = run(lambda: ops.conv2d_unpooled(g, wt, gb, 0.25, 0.2, N, H, H, 0.3, mask=a1b, mask_slope=0.2))
    t_w0 = run(lambda: ops.conv2d_wgrad(a1, gz2, dw, db, N, H, H, 3, 1, 0.4))
    t_w1 = run(lambda: ops.conv2d_wgrad_unpooled(a1, g, gb, 0.25, 0.2, dw, db, N, H, H, 0.4))
    print('n%d: unpool pass %.0f us | dgrad materialised %.0f us, in-gather %.0f us | wgrad materialised %.0f us, in-gather %.0f us' % (N, t_unpool, t_d0, t_d1, t_w0, t_w1), flush=True)
