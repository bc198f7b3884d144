"""GB/s of the input / output step kernels at BASELINE sizes (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pggan_amd as pg
def run(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for n, r in ((16, 128), (14, 256), (3, 1024), (16, 1024)):
    x = torch.randint(0, 256, (n, 3, r, r), dtype=torch.uint8, device='cuda')
    for a in (0.5, 1.0):
        t = run(lambda: pg.ops.real_prepare_u8(x, a))
        print('real_prepare_u8 n=%d %dx%d alpha=%.1f  %.1f us  %.2f TB/s (1 B in + 4 B out per pixel-channel)' % (n, r, r, a, t * 1e6, x.numel() * 5 / t / 1e12))
imgs = torch.randn(6, 3, 1024, 1024, device='cuda')
t = run(lambda: pg.ops.image_grid_u8(imgs, (-1, 1), 1))
print('image_grid_u8 6x3x1024x1024  %.1f us  %.2f TB/s' % (t * 1e6, imgs.numel() * 5 / t / 1e12))
