"""G forward at minibatch n vs 2n (is one batched generator pass for the D step's fakes and the G step's fakes cheaper than two?)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pggan_amd as pg
for depth, n in ((8, 3), (7, 6), (6, 14), (5, 16), (3, 16)):
    torch.manual_seed(1)
    G = pg.Generator((1, 3, 1024, 1024)).cuda()
    G.depth = depth
    res = {}
    for m in (n, 2 * n):
        z = torch.randn(m, 512, device='cuda')
        for _ in range(5): pg.engine.generator_forward(G, z, save=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): pg.engine.generator_forward(G, z, save=True)
        torch.cuda.synchronize(); res[m] = (time.perf_counter() - t0) / 30 * 1e3
    print('depth %d: G forward n%d %.3f ms, n%d %.3f ms -> two separate %.3f vs one batched %.3f: saves %.3f ms' % (depth, n, res[n], 2 * n, res[2 * n], 2 * res[n], res[2 * n], 2 * res[n] - res[2 * n]), flush=True)
    del G
