"""Per-kernel parity on the MI355X: every C-ABI entry point vs the torch-CPU statement of its contract
(tests/emu_ops.py, itself pinned to the reference through tests/test_engine_host.py).
Calls go through pggan-pytorch_amd/ops.py -> ctypes -> libpggan_hip.so (the C-ABI)."""
import numpy as np
import pytest
import torch

import emu_ops as E
from conftest import rel_err

import pggan_amd as pg

pytestmark = pytest.mark.gpu
ops = pg.ops
TOL = 2e-5


def dev(t):
    return None if t is None else t.cuda()


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g)


def check(name, a, b, tol=TOL):
    e = rel_err(a.cpu(), b)
    print('%-60s rel_err %.2e' % (name, e))
    assert e < tol, (name, e)


CONV_CASES = [
    # N, Hin(after ups), Cin, Cout, ks, pad, ups
    (2, 32, 64, 64, 3, 1, 0), (3, 16, 128, 96, 3, 1, 0), (2, 32, 32, 32, 3, 1, 0), (2, 64, 16, 16, 3, 1, 0),
    (1, 64, 8, 8, 3, 1, 0), (2, 32, 8, 16, 3, 1, 0), (2, 16, 4, 4, 3, 1, 0), (2, 16, 12, 20, 3, 1, 0),
    (5, 4, 32, 16, 3, 1, 0), (3, 4, 48, 64, 3, 1, 0), (7, 4, 16, 16, 4, 0, 0), (48, 4, 64, 64, 4, 0, 0),
    (6, 1, 16, 32, 4, 3, 0), (20, 1, 64, 64, 4, 3, 0), (2, 16, 32, 32, 1, 0, 0), (2, 16, 16, 16, 3, 1, 1),
    (3, 8, 64, 32, 3, 1, 1), (3, 8, 528, 512, 3, 1, 0), (1, 256, 8, 8, 3, 1, 0), (2, 128, 16, 32, 3, 1, 1),
    (70, 1, 32, 32, 4, 3, 0), (130, 4, 16, 16, 4, 0, 0), (16, 1, 512, 512, 4, 3, 0), (48, 4, 512, 512, 4, 0, 0),
    (33, 1, 128, 48, 4, 3, 0), (9, 4, 80, 32, 4, 0, 0), (2, 32, 16, 8, 3, 1, 0), (3, 128, 8, 8, 3, 1, 0), (5, 8, 8, 16, 3, 1, 0),
    (2, 4, 8, 8, 3, 1, 0), (2, 64, 16, 8, 3, 1, 1), (1, 32, 8, 8, 3, 1, 1), (2, 32, 32, 16, 3, 1, 0), (1, 64, 32, 8, 3, 1, 0),
    (2, 64, 16, 16, 3, 1, 1), (2, 64, 32, 16, 3, 1, 1), (3, 32, 16, 32, 3, 1, 0), (3, 4, 256, 64, 3, 1, 0), (9, 8, 128, 48, 3, 1, 0),
    (5, 4, 528, 512, 3, 1, 0), (1, 16, 144, 32, 3, 1, 0),
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv2d_fwd_and_masked(case):
    N, H, ci, co, ks, pad, ups = case
    hin = H // 2 if ups else H
    x, w, b = rnd(N, hin, hin, ci), rnd(ks, ks, co, ci, seed=1) * 0.2, rnd(co, seed=2)
    ho = H + 2 * pad - ks + 1
    y = ops.conv2d(dev(x), dev(w), dev(b), N, H, H, ks, pad, 0.37, slope=0.2, ups=bool(ups))
    check('conv fwd %s' % (case,), y, E.conv2d(x, w, b, N, H, H, ks, pad, 0.37, slope=0.2, ups=bool(ups)))
    m = rnd(N, ho, ho, co, seed=3)
    y = ops.conv2d(dev(x), dev(w), None, N, H, H, ks, pad, 0.37, mask=dev(m), mask_slope=0.2, ups=bool(ups))
    check('conv masked %s' % (case,), y, E.conv2d(x, w, None, N, H, H, ks, pad, 0.37, mask=m, mask_slope=0.2, ups=bool(ups)))
    y = ops.conv2d(dev(x), dev(w), dev(b), N, H, H, ks, pad, 1.0, slope=1.0, ups=bool(ups))
    check('conv linear %s' % (case,), y, E.conv2d(x, w, b, N, H, H, ks, pad, 1.0, slope=1.0, ups=bool(ups)))


@pytest.mark.parametrize('case', [(3, 512, 512), (9, 80, 32), (48, 64, 64), (130, 16, 16)])
def test_conv_4x4_to_1x1_split_over_pixels(case):
    """D's last conv (4x4 -> 1x1, network.py:213-221) with one workgroup per (cout block, input pixel) and the last-arriver
    fix-up through the stream's scratch (conv_k4_reduce_split_kernel) against the one-workgroup-per-cout-block kernel
    (pg_debug_set_tuning(3, 21)); repeats are bit-identical (the 16 partial sums are added in pixel order)."""
    N, ci, co = case
    lib = pg._lib.load()
    x, w, b = dev(rnd(N, 4, 4, ci)), dev(rnd(4, 4, co, ci, seed=1) * 0.2), dev(rnd(co, seed=2))
    ys = [ops.conv2d(x, w, b, N, 4, 4, 4, 0, 0.37, slope=0.2) for _ in range(3)]
    assert lib.pg_debug_last_conv_kernel().decode().startswith('conv_k4_reduce_split_kernel')
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2])
    assert lib.pg_debug_set_tuning(3, 21) == 0
    try:
        ref = ops.conv2d(x, w, b, N, 4, 4, 4, 0, 0.37, slope=0.2)
        assert lib.pg_debug_last_conv_kernel().decode().startswith('conv_k4_reduce_kernel')
    finally:
        lib.pg_debug_set_tuning(3, -1)
    assert rel_err(ys[0], ref) < 1e-6


POOL_CASES = [(2, 32, 64, 64), (3, 16, 128, 96), (2, 64, 16, 16), (1, 64, 8, 8), (2, 32, 8, 16), (2, 16, 12, 20), (5, 4, 32, 16),
              (3, 8, 528, 512), (1, 256, 8, 16), (3, 128, 16, 32), (2, 64, 32, 64), (9, 16, 256, 256), (3, 32, 256, 512), (1, 8, 64, 32),
              (2, 2, 16, 16), (2, 32, 16, 16), (1, 64, 32, 16), (3, 32, 8, 16)]


@pytest.mark.parametrize('case', POOL_CASES)
@pytest.mark.parametrize('cand', [-1, 0, 1, 2, 3, 4, 5, 6, 7])
def test_conv2d_pool_fused(case, cand):
    """Fused conv + 2x2 pool (+ fade-in blend) == conv followed by avgpool, bit for bit, for every tile shape."""
    N, H, ci, co = case
    lib = pg._lib.load()
    x, w, b = rnd(N, H, H, ci), rnd(3, 3, co, ci, seed=1) * 0.2, rnd(co, seed=2)
    other, m = rnd(N, H // 2, H // 2, co, seed=4), rnd(N, H, H, co, seed=3)
    lib.pg_debug_set_tuning(0, cand)
    try:
        try:
            y, yp = ops.conv2d_pool(dev(x), dev(w), dev(b), N, H, H, 3, 1, 0.37, slope=0.2, other=dev(other), a=0.6, b=0.4)
        except RuntimeError:
            pytest.skip('tile candidate not available for this shape')
        lib.pg_debug_set_tuning(0, -1)
        ry = ops.conv2d(dev(x), dev(w), dev(b), N, H, H, 3, 1, 0.37, slope=0.2)
        check('conv+pool y %s' % (case,), y, E.conv2d(x, w, b, N, H, H, 3, 1, 0.37, slope=0.2))
        assert torch.equal(yp, ops.avgpool2_fwd(y, dev(other), 0.6, 0.4)), 'pooled output differs from the unfused pair'
        lib.pg_debug_set_tuning(0, cand)
        _, yp2 = ops.conv2d_pool(dev(x), dev(w), None, N, H, H, 3, 1, 0.37, mask=dev(m), mask_slope=0.2, pool_only=True)
        lib.pg_debug_set_tuning(0, -1)
        ym = ops.conv2d(dev(x), dev(w), None, N, H, H, 3, 1, 0.37, mask=dev(m), mask_slope=0.2)
        check('conv+pool masked, pooled only %s' % (case,), yp2, ops.avgpool2_fwd(ym).cpu())   # other tile / split-K: fp32 order
    finally:
        lib.pg_debug_set_tuning(0, -1)


@pytest.mark.parametrize('case', [(1, 64, 8, 8, 0), (2, 64, 16, 8, 1), (2, 32, 16, 8, 0), (2, 64, 16, 16, 0), (3, 32, 32, 16, 1), (2, 32, 32, 32, 0),
                                  (1, 128, 64, 32, 1), (2, 16, 64, 64, 0), (3, 8, 512, 512, 0), (5, 4, 32, 16, 0), (2, 16, 12, 20, 0)])
def test_conv2d_pixelnorm_fused(case):
    """Generator layer (conv -> bias -> LeakyReLU -> PixelNorm) in one launch vs the separate kernels and the emulation."""
    N, H, ci, co, ups = case
    hin = H // 2 if ups else H
    x, w, b = rnd(N, hin, hin, ci), rnd(3, 3, co, ci, seed=1) * 0.2, rnd(co, seed=2)
    y, r = ops.conv2d_pixelnorm(dev(x), dev(w), dev(b), N, H, H, 3, 1, 0.37, 0.2, 1e-8, ups=bool(ups))
    print(pg._lib.load().pg_debug_last_conv_kernel().decode())
    ry, rr = E.conv2d_pixelnorm(x, w, b, N, H, H, 3, 1, 0.37, 0.2, 1e-8, ups=bool(ups))
    check('conv+pixelnorm y %s' % (case,), y, ry)
    check('conv+pixelnorm r %s' % (case,), r, rr)
    pg._lib.load().pg_debug_set_tuning(3, 11)
    try:
        y2, r2 = ops.conv2d_pixelnorm(dev(x), dev(w), dev(b), N, H, H, 3, 1, 0.37, 0.2, 1e-8, ups=bool(ups))
    finally:
        pg._lib.load().pg_debug_set_tuning(3, -1)
    check('fused vs separate kernels', y, y2.cpu(), 2e-6)


@pytest.mark.parametrize('case', [(1, 64, 8, 8), (2, 64, 16, 8), (2, 64, 16, 16), (3, 32, 32, 16), (2, 32, 32, 32), (1, 128, 64, 32),
                                  (2, 16, 64, 64), (3, 8, 512, 512), (5, 4, 32, 16), (2, 16, 12, 20)])
@pytest.mark.parametrize('with_r', [True, False])
def test_conv2d_pnbwd_fused(case, with_r):
    """Backward-data conv + adjoint of the previous layer's (LeakyReLU -> PixelNorm) in one launch vs the emulation."""
    N, H, ci, co = case
    x, w = rnd(N, H, H, ci), rnd(3, 3, co, ci, seed=1) * 0.2
    h = rnd(N, H, H, co, seed=4)
    ys, r = E.pixelnorm_fwd(h)
    if not with_r:
        ys, r = h, None
    out = ops.conv2d_pnbwd(dev(x), dev(w), dev(ys), None if r is None else dev(r), N, H, H, 3, 1, 0.37, 0.2)
    print(pg._lib.load().pg_debug_last_conv_kernel().decode())
    check('conv+pnbwd %s r=%s' % (case, with_r), out, E.conv2d_pnbwd(x, w, ys, r, N, H, H, 3, 1, 0.37, 0.2), 5e-5)


@pytest.mark.parametrize('case', POOL_CASES)
@pytest.mark.parametrize('cand', [-1, 0, 1, 2, 3, 4, 5, 6, 7])
def test_conv2d_unpool_fused(case, cand):
    """Backward-data conv + pool adjoint + LeakyReLU' mask in one kernel == conv followed by avgpool2_bwd."""
    N, H, ci, co = case
    lib = pg._lib.load()
    x, w = rnd(N, H, H, ci), rnd(3, 3, co, ci, seed=1) * 0.2
    m = rnd(N, 2 * H, 2 * H, co, seed=3)
    lib.pg_debug_set_tuning(0, cand)
    try:
        try:
            up = ops.conv2d_unpool(dev(x), dev(w), N, H, H, 3, 1, 0.37, upmask=dev(m), mul=0.7, mask_slope=0.2)
            y = ops.conv2d(dev(x), dev(w), None, N, H, H, 3, 1, 0.37)
        except RuntimeError:
            pytest.skip('tile candidate not available for this shape')
        # (not torch.equal: when the shape needs split-K both calls fall back to atomics, whose order differs run to run)
        check('conv+unpool vs unfused pair %s' % (case,), up, ops.avgpool2_bwd(y, dev(m), 0.7, 0.2).cpu(), 2e-6)
        check('conv+unpool vs emulation %s' % (case,), up, E.conv2d_unpool(x, w, N, H, H, 3, 1, 0.37, upmask=m, mul=0.7, mask_slope=0.2))
        up2 = ops.conv2d_unpool(dev(x), dev(w), N, H, H, 3, 1, 0.37)
        check('conv+unpool (no mask) %s' % (case,), up2, ops.avgpool2_bwd(y).cpu(), 2e-6)
    finally:
        lib.pg_debug_set_tuning(0, -1)


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv2d_wgrad(case):
    N, H, ci, co, ks, pad, ups = case
    hin = H // 2 if ups else H
    ho = H + 2 * pad - ks + 1
    x, gz = rnd(N, hin, hin, ci), rnd(N, ho, ho, co, seed=5)
    dw0, db0 = rnd(ks, ks, co, ci, seed=6), rnd(co, seed=7)
    dw, db = dev(dw0.clone()), dev(db0.clone())
    ops.conv2d_wgrad(dev(x), dev(gz), dw, db, N, H, H, ks, pad, 0.41, ups=bool(ups))
    rw, rb = dw0.clone(), db0.clone()
    E.conv2d_wgrad(x, gz, rw, rb, N, H, H, ks, pad, 0.41, ups=bool(ups))
    check('wgrad dw %s' % (case,), dw, rw, 5e-5)
    check('wgrad db %s' % (case,), db, rb, 5e-5)
    dw = dev(dw0.clone())
    ops.conv2d_wgrad(dev(x), dev(gz), dw, None, N, H, H, ks, pad, 0.41, ups=bool(ups))
    check('wgrad dw (no bias) %s' % (case,), dw, rw, 5e-5)


@pytest.mark.parametrize('ks,co,ci', [(3, 64, 32), (1, 16, 8), (4, 20, 12), (3, 512, 528)])
def test_pack_dgrad(ks, co, ci):
    w = rnd(ks, ks, co, ci)
    wt = torch.empty(ks, ks, ci, co).cuda()
    ops.pack_dgrad_weights(dev(w), wt)
    ref = torch.empty(ks, ks, ci, co)
    E.pack_dgrad_weights(w, ref)
    assert torch.equal(wt.cpu(), ref)


def test_dgrad_is_adjoint_of_conv():
    """<conv(x), g> == <x, dgrad(g)> with the packed backward-data weights (transpose-detecting)."""
    for (N, H, ci, co, ks, pad) in [(2, 16, 16, 32, 3, 1), (3, 4, 32, 16, 4, 0), (2, 8, 8, 8, 1, 0)]:
        x, w = rnd(N, H, H, ci), rnd(ks, ks, co, ci, seed=1)
        ho = H + 2 * pad - ks + 1
        g = rnd(N, ho, ho, co, seed=2)
        y = ops.conv2d(dev(x), dev(w), None, N, H, H, ks, pad, 1.0)
        wt = torch.empty(ks, ks, ci, co).cuda()
        ops.pack_dgrad_weights(dev(w), wt)
        gx = ops.conv2d(dev(g), wt, None, N, ho, ho, ks, ks - 1 - pad, 1.0)
        a = float((y.cpu().double() * g.double()).sum())
        b = float((x.double() * gx.cpu().double()).sum())
        print('adjoint', a, b)
        assert abs(a - b) < 1e-4 * max(1.0, abs(a))


RGB_CASES = [(2, 3, 16, 16), (3, 1, 8, 32), (2, 3, 4, 512), (1, 3, 64, 8), (2, 4, 8, 12), (1, 3, 256, 8), (2, 3, 256, 16),
             (1, 1, 256, 32), (16, 3, 4, 512), (5, 1, 8, 256), (3, 4, 16, 512)]


@pytest.mark.parametrize('N,C,H,co', RGB_CASES)
@pytest.mark.parametrize('pool', [False, True])
def test_fromrgb(N, C, H, co, pool):
    hi = 2 * H if pool else H
    img, w, b = rnd(N, C, hi, hi), rnd(co, C, seed=1), rnd(co, seed=2)
    y = ops.fromrgb_fwd(dev(img), dev(w), dev(b), N, C, H, H, 0.8, 0.2, pool=pool)
    check('fromrgb fwd', y, E.fromrgb_fwd(img, w, b, N, C, H, H, 0.8, 0.2, pool=pool))
    m = rnd(N, H, H, co, seed=3)
    y = ops.fromrgb_fwd(dev(img), dev(w), None, N, C, H, H, 0.8, 1.0, pool=pool, mask=dev(m), mask_slope=0.2)
    check('fromrgb masked', y, E.fromrgb_fwd(img, w, None, N, C, H, H, 0.8, 1.0, pool=pool, mask=m, mask_slope=0.2))
    gz = rnd(N, H, H, co, seed=4)
    for acc in (False, True):
        g0 = rnd(N, C, hi, hi, seed=5)
        gi = dev(g0.clone())
        ops.fromrgb_bwd_data(dev(gz), dev(w), gi, N, C, H, H, 0.8, pool=pool, accumulate=acc)
        r = g0.clone()
        E.fromrgb_bwd_data(gz, w, r, N, C, H, H, 0.8, pool=pool, accumulate=acc)
        check('fromrgb bwd_data acc=%s' % acc, gi, r)
    dw0, db0 = rnd(co, C, seed=6), rnd(co, seed=7)
    dw, db = dev(dw0.clone()), dev(db0.clone())
    ops.fromrgb_wgrad(dev(gz), dev(img), dw, db, N, C, H, H, 0.8, pool=pool)
    rw, rb = dw0.clone(), db0.clone()
    E.fromrgb_wgrad(gz, img, rw, rb, N, C, H, H, 0.8, pool=pool)
    check('fromrgb wgrad', dw, rw, 5e-5)
    check('fromrgb bgrad', db, rb, 5e-5)


@pytest.mark.parametrize('N,C,H,ci', RGB_CASES)
def test_torgb(N, C, H, ci):
    x, w, b = rnd(N, H, H, ci), rnd(C, ci, seed=1), rnd(C, seed=2)
    prev = rnd(N, C, H // 2, H // 2, seed=3)
    y = ops.torgb_fwd(dev(x), dev(w), dev(b), N, C, H, H, 0.7)
    check('torgb fwd', y, E.torgb_fwd(x, w, b, N, C, H, H, 0.7))
    y = ops.torgb_fwd(dev(x), dev(w), dev(b), N, C, H, H, 0.7, out_mul=0.3, prev=dev(prev), prev_mul=0.7)
    check('torgb blend', y, E.torgb_fwd(x, w, b, N, C, H, H, 0.7, out_mul=0.3, prev=prev, prev_mul=0.7))
    g = rnd(N, C, H, H, seed=4)
    check('torgb bwd_data', ops.torgb_bwd_data(dev(g), dev(w), N, C, H, H, 0.21), E.torgb_bwd_data(g, w, N, C, H, H, 0.21))
    ys, rs = rnd(N, H, H, ci, seed=8), rnd(N * H * H, seed=9).abs() + 0.5        # fused (8 features, large maps) or two launches
    check('torgb bwd_data + pn adjoint', ops.torgb_bwd_data_pnbwd(dev(g), dev(w), dev(ys), dev(rs), N, C, H, H, 0.21, 0.2),
          E.torgb_bwd_data_pnbwd(g, w, ys, rs, N, C, H, H, 0.21, 0.2))
    g2 = rnd(N, C, 2 * H, 2 * H, seed=5)
    check('torgb bwd_data down', ops.torgb_bwd_data(dev(g2), dev(w), N, C, H, H, 0.21, down=True),
          E.torgb_bwd_data(g2, w, N, C, H, H, 0.21, down=True))
    for down, gg in ((False, g), (True, g2)):
        dw0, db0 = rnd(C, ci, seed=6), rnd(C, seed=7)
        dw, db = dev(dw0.clone()), dev(db0.clone())
        ops.torgb_wgrad(dev(gg), dev(x), dw, db, N, C, H, H, 0.21, 0.3, down=down)
        rw, rb = dw0.clone(), db0.clone()
        E.torgb_wgrad(gg, x, rw, rb, N, C, H, H, 0.21, 0.3, down=down)
        check('torgb wgrad down=%s' % down, dw, rw, 5e-5)
        check('torgb bgrad down=%s' % down, db, rb, 5e-5)


@pytest.mark.parametrize('N,H,C', [(2, 8, 16), (3, 4, 512), (1, 64, 8), (2, 2, 4)])
def test_pool_upsample_axpby(N, H, C):
    x, o = rnd(N, 2 * H, 2 * H, C), rnd(N, H, H, C, seed=1)
    check('avgpool', ops.avgpool2_fwd(dev(x)), E.avgpool2_fwd(x))
    check('avgpool blend', ops.avgpool2_fwd(dev(x), dev(o), 0.3, 0.7), E.avgpool2_fwd(x, o, 0.3, 0.7))
    gy, m = rnd(N, H, H, C, seed=2), rnd(N, 2 * H, 2 * H, C, seed=3)
    check('avgpool bwd', ops.avgpool2_bwd(dev(gy), dev(m), 0.3, 0.2), E.avgpool2_bwd(gy, m, 0.3, 0.2))
    check('avgpool bwd nomask', ops.avgpool2_bwd(dev(gy)), E.avgpool2_bwd(gy))
    check('upsample bwd', ops.upsample2_bwd(dev(x)), E.upsample2_bwd(x))
    check('axpby', ops.axpby_mask(dev(x), dev(m), dev(m), 0.3, 0.7, 0.2), E.axpby_mask(x, m, m, 0.3, 0.7, 0.2))
    check('scale', ops.axpby_mask(dev(x), a=0.3), E.axpby_mask(x, a=0.3))


@pytest.mark.parametrize('P,C', [(64, 512), (1000, 16), (4096, 8), (37, 4), (5, 32), (300, 256), (16, 64)])
def test_pixelnorm(P, C):
    x = rnd(P, C)
    y, r = ops.pixelnorm_fwd(dev(x))
    ry, rr = E.pixelnorm_fwd(x)
    check('pn fwd', y, ry)
    check('pn r', r, rr)
    gy = rnd(P, C, seed=1)
    check('pn bwd', ops.pixelnorm_lrelu_bwd(dev(gy), y, r, 0.2), E.pixelnorm_lrelu_bwd(gy, ry, rr, 0.2))
    check('lrelu-only bwd', ops.pixelnorm_lrelu_bwd(dev(gy), y, None, 0.2), E.pixelnorm_lrelu_bwd(gy, ry, None, 0.2))


@pytest.mark.parametrize('G,n,C,world', [(1, 4, 16, 2), (3, 3, 512, 3), (3, 16, 512, 2), (2, 6, 32, 3)])
def test_mbstd_exact_global_mode(G, n, C, world):
    """The exact-global minibatch stddev entry points (pg_mbstd_stats / _write / _tangent_stats / _tangent_write / _gsum / _bwd_global,
    SURVEY.md §8e optional mode) with the ranks of a data-parallel group played by slices of one batch on one device: every "rank" holds
    n / world images of each group, the partial rows are stacked rank-major as the all-gather would, the Gs sums are added as the
    all-reduce would -- and forward, tangent, adjoint and Hessian-vector term of every shard must equal the single-process kernels (and
    the torch statement) on the whole batch."""
    cp = C + 16
    per = n // world
    x = rnd(G * n, 4, 4, C) + 0.3
    tx = rnd(G * n, 4, 4, C, seed=1)
    gy, gf = rnd(G * n, 4, 4, cp, seed=2), rnd(G * n, 4, 4, cp, seed=3)
    idx = [torch.cat([torch.arange(g * n + r * per, g * n + (r + 1) * per) for g in range(G)]) for r in range(world)]   # shard r: its slice of EVERY group
    xs, txs, gys, gfs = ([dev(t[i]) for i in idx] for t in (x, tx, gy, gf))
    ry, rst = E.mbstd_fwd(x, G, cp)
    rty, rts = E.mbstd_tangent(x, tx, rst, cp)
    parts = [ops.mbstd_stats(xr, G) for xr in xs]
    gathered = torch.stack(parts).contiguous()
    outs = [ops.mbstd_write(xr, parts[r].clone(), gathered, cp) for r, xr in enumerate(xs)]
    for r in range(world):
        check('global mbstd fwd, shard %d' % r, outs[r][0], ry[idx[r]])
        check('global mbstd stats, shard %d' % r, outs[r][1][:, :2], rst)
        assert torch.equal(outs[r][1][:, :2], outs[0][1][:, :2])                 # bit-identical mu / sigma on every rank
    stats = [o[1] for o in outs]
    tparts = [ops.mbstd_tangent_stats(xs[r], txs[r], stats[r]) for r in range(world)]
    tgath = torch.stack(tparts).contiguous()
    touts = [ops.mbstd_tangent_write(txs[r], tparts[r].clone(), tgath, stats[r], cp) for r in range(world)]
    for r in range(world):
        check('global mbstd tangent, shard %d' % r, touts[r][0], rty[idx[r]], 1e-4)
        check('global mbstd tstats, shard %d' % r, touts[r][1][:, :2], rts, 1e-4)
    for am in (False, True):
        for use_gy, use_tx in ((True, False), (True, True), (False, True)):
            gsum = sum(ops.mbstd_gsum(gys[r] if use_gy else None, gfs[r] if use_tx else None, G, tuple(xs[r].shape), cp) for r in range(world))
            ref = E.mbstd_bwd(gy if use_gy else None, x, rst, cp, am, 0.2, tx=tx if use_tx else None, tstats=rts if use_tx else None,
                              gy_first=gf if use_tx else None)
            for r in range(world):
                got = ops.mbstd_bwd_global(gys[r] if use_gy else None, xs[r], stats[r], cp, am, gsum, world, 0.2,
                                           tx=txs[r] if use_tx else None, tstats=touts[r][1] if use_tx else None,
                                           gy_first=gfs[r] if use_tx else None)
                check('global mbstd bwd mask=%s gy=%s hvp=%s shard %d' % (am, use_gy, use_tx, r), got, ref[idx[r]], 1e-4)


@pytest.mark.parametrize('G,n,C', [(1, 4, 16), (3, 3, 512), (3, 16, 512), (2, 5, 32)])
def test_mbstd(G, n, C):
    cp = C + 16
    x = rnd(G * n, 4, 4, C) + 0.3
    y, st = ops.mbstd_fwd(dev(x), G, cp)
    ry, rst = E.mbstd_fwd(x, G, cp)
    check('mbstd fwd', y, ry)
    check('mbstd stats', st[:, :2], rst)
    tx = rnd(G * n, 4, 4, C, seed=1)
    ty, ts = ops.mbstd_tangent(dev(x), dev(tx), st, cp)
    rty, rts = E.mbstd_tangent(x, tx, rst, cp)
    check('mbstd tangent', ty, rty)
    check('mbstd tstats', ts[:, :2], rts, 1e-4)
    gy, gf = rnd(G * n, 4, 4, cp, seed=2), rnd(G * n, 4, 4, cp, seed=3)
    for am in (False, True):
        check('mbstd bwd mask=%s' % am, ops.mbstd_bwd(dev(gy), dev(x), st, cp, am, 0.2), E.mbstd_bwd(gy, x, rst, cp, am, 0.2))
        check('mbstd bwd+hvp', ops.mbstd_bwd(dev(gy), dev(x), st, cp, am, 0.2, tx=dev(tx), tstats=ts, gy_first=dev(gf)),
              E.mbstd_bwd(gy, x, rst, cp, am, 0.2, tx=tx, tstats=rts, gy_first=gf), 1e-4)
        check('mbstd hvp only', ops.mbstd_bwd(None, dev(x), st, cp, am, 0.2, tx=dev(tx), tstats=ts, gy_first=dev(gf)),
              E.mbstd_bwd(None, x, rst, cp, am, 0.2, tx=tx, tstats=rts, gy_first=gf), 1e-4)


def test_linear_gp_loss_adam():
    N, C = 9, 512
    h, w, b = rnd(N, 1, 1, C), rnd(1, C, seed=1), rnd(1, seed=2)
    s = ops.linear1_fwd(dev(h), dev(w), dev(b))
    check('linear fwd', s, E.linear1_fwd(h, w, b))
    gs = rnd(N, seed=3)
    check('linear bwd', ops.linear1_bwd_data(dev(gs), dev(w), dev(h), h.shape, 0.2), E.linear1_bwd_data(gs, w, h, h.shape, 0.2))
    dw0, db0 = rnd(1, C, seed=4), rnd(1, seed=5)
    dw, db = dev(dw0.clone()), dev(db0.clone())
    ops.linear1_wgrad(dev(gs), dev(h), dw, db)
    rw, rb = dw0.clone(), db0.clone()
    E.linear1_wgrad(gs, h, rw, rb)
    check('linear wgrad', dw, rw)
    check('linear bgrad', db, rb)
    real, fake, m = rnd(4, 3, 16, 16), rnd(4, 3, 16, 16, seed=1), torch.rand(4)
    check('gp mix', ops.gp_mix(dev(real), dev(fake), dev(m)), E.gp_mix(real, fake, m))
    ss = ops.row_sumsq(dev(real))
    check('row sumsq', ss, E.row_sumsq(real))
    gp, u = ops.gp_seed(dev(real), ss, 10.0, 1.0, 0.25)
    rgp, ru = E.gp_seed(real, E.row_sumsq(real), 10.0, 1.0, 0.25)
    check('gp', gp, rgp)
    check('gp seed', u, ru)
    sc, gpv = rnd(12, seed=7), torch.rand(4)
    out = ops.d_loss(dev(sc), dev(gpv), 4, 0.001)
    ref = E.d_loss(sc, gpv, 4, 0.001)
    for nm, a, r in zip(('d_cost', 'd_real_loss', 'd_fake_loss', 'gscore'), out, ref):
        check('d_loss ' + nm, a, r)
    gc, gsc = ops.g_loss(dev(sc))
    rgc, rgsc = E.g_loss(sc)
    check('g_cost', gc, rgc)
    check('g gscore', gsc, rgsc)
    n = 1003
    p, g, mm, vv = rnd(n), rnd(n, seed=1), rnd(n, seed=2) * 0.1, rnd(n, seed=3).abs() * 0.1
    dp, dg, dm, dv = dev(p.clone()), dev(g), dev(mm.clone()), dev(vv.clone())
    ops.adam(dp, dg, dm, dv, 1e-3, 0.0, 0.99, 1e-8, 1.0, 0.3, 0.5)
    E.adam(p, g, mm, vv, 1e-3, 0.0, 0.99, 1e-8, 1.0, 0.3, 0.5)
    check('adam p', dp, p)
    check('adam m', dm, mm)
    check('adam v', dv, vv)
    # beta1 != 0 (the other instantiation: with beta1 == 0 the kernel does not read the old first moment), bias corrections != 1
    ops.adam(dp, dg, dm, dv, 2e-3, 0.9, 0.999, 1e-8, 0.19, 0.0447, 1.0)
    E.adam(p, g, mm, vv, 2e-3, 0.9, 0.999, 1e-8, 0.19, 0.0447, 1.0)
    check('adam p (beta1 0.9)', dp, p)
    check('adam m (beta1 0.9)', dm, mm)
    check('adam v (beta1 0.9)', dv, vv)


@pytest.mark.gpu
@pytest.mark.parametrize('slope', [0.2, 0.0])            # 0.0: ReLU (Discriminator nonlinearity flag) -- the sign byte is (y > 0) either way
@pytest.mark.parametrize('case', [(2, 64, 16, 32), (3, 32, 8, 16), (1, 128, 32, 64), (2, 64, 64, 32), (5, 16, 16, 16)])
def test_sign_byte_activations(case, slope):
    """PG_FLAG_Y_BYTES / PG_FLAG_MASK_BYTES: the conv+pool epilogue writes the sign bytes of its output instead of the
    fp32 activation, the unpool and masked(+pool) epilogues read them as LeakyReLU' masks -- direct and Winograd kernels.
    Results must equal the fp32-mask launches exactly (same kernels, same arithmetic, only the mask source differs)."""
    N, H, ci, co = case
    ops = pg.ops
    dev = lambda t: t.cuda()
    x, w, b = rnd(N, H, H, ci), rnd(3, 3, co, ci, seed=1) * 0.2, rnd(co, seed=2)
    other = rnd(N, H // 2, H // 2, co, seed=3)
    # producer
    y32, yp32 = ops.conv2d_pool(dev(x), dev(w), dev(b), N, H, H, 3, 1, 0.4, slope, other=dev(other), a=0.6, b=0.4)
    yb, ypb = ops.conv2d_pool(dev(x), dev(w), dev(b), N, H, H, 3, 1, 0.4, slope, other=dev(other), a=0.6, b=0.4, y_bytes=True)
    assert yb.dtype == torch.uint8 and tuple(yb.shape) == (N, H, H, co // 4)
    assert torch.equal(yb.cpu(), E.signbytes_of(y32.cpu())) and rel_err(ypb, yp32) < 2e-6
    assert torch.equal(ops.signbytes_to_mask(yb).cpu(), E.signbytes_to_mask(yb.cpu()))
    # unpool consumer: gradient at the pooled resolution -> fine resolution, masked by the signs of y
    g = rnd(N, H // 2, H // 2, co, seed=4)
    wt = rnd(3, 3, co, co, seed=5) * 0.2
    up32 = ops.conv2d_unpool(dev(g), dev(wt), N, H // 2, H // 2, 3, 1, 0.3, upmask=y32, mul=0.7, mask_slope=slope)
    upb = ops.conv2d_unpool(dev(g), dev(wt), N, H // 2, H // 2, 3, 1, 0.3, upmask=yb, mul=0.7, mask_slope=slope)
    assert rel_err(upb, up32) < 2e-6
    # tangent consumer: masked conv + pool_only
    _, t32 = ops.conv2d_pool(dev(x), dev(w), None, N, H, H, 3, 1, 0.4, 1.0, mask=y32, mask_slope=slope, other=dev(other), a=0.6, b=0.4, pool_only=True)
    _, tb = ops.conv2d_pool(dev(x), dev(w), None, N, H, H, 3, 1, 0.4, 1.0, mask=yb, mask_slope=slope, other=dev(other), a=0.6, b=0.4, pool_only=True)
    assert rel_err(tb, t32) < 2e-6
    if ci % 16 == 0 and H >= 8:                                  # the Winograd kernel's epilogues
        u = ops.wino_transform_weights(dev(w))
        ybw, ypw = ops.conv2d_wino(dev(x), u, dev(b), N, H, H, 0.4, slope, pool=True, other=dev(other), a=0.6, b=0.4, y_bytes=True)
        y32w, _ = ops.conv2d_wino(dev(x), u, dev(b), N, H, H, 0.4, slope, pool=True, other=dev(other), a=0.6, b=0.4)
        assert torch.equal(ybw.cpu(), E.signbytes_of(y32w.cpu())) and rel_err(ypw, yp32) < 2e-5
        _, tw = ops.conv2d_wino(dev(x), u, None, N, H, H, 0.4, 1.0, mask=ybw, mask_slope=slope, pool=True, other=dev(other), a=0.6, b=0.4, pool_only=True)
        _, tw32 = ops.conv2d_wino(dev(x), u, None, N, H, H, 0.4, 1.0, mask=y32w, mask_slope=slope, pool=True, other=dev(other), a=0.6, b=0.4, pool_only=True)
        assert rel_err(tw, tw32) < 2e-6
    if co % 16 == 0 and H >= 16:
        ut = ops.wino_transform_weights(dev(wt))
        uw = ops.conv2d_wino(dev(g), ut, None, N, H // 2, H // 2, 0.3, mask_slope=slope, unpool=True, upmask=yb, up_mul=0.7)
        uw32 = ops.conv2d_wino(dev(g), ut, None, N, H // 2, H // 2, 0.3, mask_slope=slope, unpool=True, upmask=y32, up_mul=0.7)
        assert rel_err(uw, uw32) < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(2, 64, 8, 8), (1, 64, 16, 8), (2, 32, 8, 16), (2, 32, 16, 16), (8, 32, 32, 64), (8, 32, 64, 32)])
def test_sign_bytes_of_fp32_activations(case):
    """PG_FLAG_SIGNS_OUT: forward launches that write y AND its sign bytes (DBlock c1 / fromRGB outputs), and masked launches
    that read the bytes instead of the fp32 activation -- block-MFMA (8 couts), generic and Winograd kernels, fromRGB.
    (Launches too small to fill the chip without split-K answer PG_E_UNSUP by contract: see the last test below.)"""
    N, H, ci, co = case
    ops = pg.ops
    dev = lambda t: t.cuda()
    x, w, b = rnd(N, H, H, ci), rnd(3, 3, co, ci, seed=1) * 0.2, rnd(co, seed=2)
    y32 = ops.conv2d(dev(x), dev(w), dev(b), N, H, H, 3, 1, 0.4, 0.2)
    y, yb = ops.conv2d(dev(x), dev(w), dev(b), N, H, H, 3, 1, 0.4, 0.2, signs_out=True)
    assert torch.equal(y, y32) and torch.equal(yb.cpu(), E.signbytes_of(y32.cpu()))
    g, wt = rnd(N, H, H, co, seed=3), rnd(3, 3, ci, co, seed=4) * 0.2          # backward-data direction: co -> ci, masked by x's signs
    xb = E.signbytes_of(x).cuda()
    d32 = ops.conv2d(dev(g), dev(wt), None, N, H, H, 3, 1, 0.3, mask=dev(x), mask_slope=0.2)
    db = ops.conv2d(dev(g), dev(wt), None, N, H, H, 3, 1, 0.3, mask=xb, mask_slope=0.2)
    assert rel_err(db, d32) < 2e-6
    if ci % 16 == 0:
        u = ops.wino_transform_weights(dev(w))
        yw32 = ops.conv2d_wino(dev(x), u, dev(b), N, H, H, 0.4, 0.2)
        yw, ywb = ops.conv2d_wino(dev(x), u, dev(b), N, H, H, 0.4, 0.2, signs_out=True)
        assert torch.equal(yw, yw32) and torch.equal(ywb.cpu(), E.signbytes_of(yw32.cpu()))
    if co % 16 == 0:
        ut = ops.wino_transform_weights(dev(wt))
        assert rel_err(ops.conv2d_wino(dev(g), ut, None, N, H, H, 0.3, mask=xb, mask_slope=0.2),
                       ops.conv2d_wino(dev(g), ut, None, N, H, H, 0.3, mask=dev(x), mask_slope=0.2)) < 2e-6
    # fromRGB: narrow per-pixel kernels (8/16 couts on >= 65536 pixels) and the generic one
    for (n, hh, cc) in ((1, 256, 8), (1, 256, 16), (2, 16, 32)):
        img, fw, fb = rnd(n, 3, hh, hh, seed=5), rnd(cc, 3, seed=6), rnd(cc, seed=7)
        a32 = ops.fromrgb_fwd(dev(img), dev(fw), dev(fb), n, 3, hh, hh, 0.7, 0.2)
        a, ab = ops.fromrgb_fwd(dev(img), dev(fw), dev(fb), n, 3, hh, hh, 0.7, 0.2, signs_out=True)
        assert torch.equal(a, a32) and torch.equal(ab.cpu(), E.signbytes_of(a32.cpu()))
        t32 = ops.fromrgb_fwd(dev(img), dev(fw), None, n, 3, hh, hh, 0.7, 1.0, mask=a32, mask_slope=0.2)
        tb = ops.fromrgb_fwd(dev(img), dev(fw), None, n, 3, hh, hh, 0.7, 1.0, mask=ab, mask_slope=0.2)
        assert torch.equal(tb, t32)


@pytest.mark.gpu
def test_sign_bytes_unsupported_is_reported():
    """Configurations without a byte-aware epilogue (here a 1x1 conv) answer PG_E_UNSUP -> ops.Unsupported, never a wrong result."""
    ops = pg.ops
    x, w, b = rnd(2, 16, 16, 32).cuda(), (rnd(1, 1, 32, 32, seed=1) * 0.1).cuda(), rnd(32, seed=2).cuda()
    with pytest.raises(ops.Unsupported):
        ops.conv2d(x, w, b, 2, 16, 16, 1, 0, 0.4, 0.2, signs_out=True)


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(2, 64, 16, 8), (3, 32, 8, 8), (1, 128, 16, 8)])
def test_pool_adjoint_in_the_gather(case):
    """pg_conv2d_unpooled_nhwc / pg_conv2d_wgrad_unpooled_nhwc: the pool adjoint (x 1/4 x mul x LeakyReLU' sign byte) of the
    coarse gradient evaluated in the consumers' gathers == materialising it with avgpool2_bwd first."""
    N, H, cg, co = case                                  # g has cg channels at H/2; backward-data conv cg -> co
    ops = pg.ops
    g, a2 = rnd(N, H // 2, H // 2, cg), rnd(N, H, H, cg, seed=1)
    gb = E.signbytes_of(a2)
    wt, a1 = rnd(3, 3, co, cg, seed=2) * 0.2, rnd(N, H, H, co, seed=3)
    gz2 = E.avgpool2_bwd(g, a2, 0.7, 0.2)
    ref = E.conv2d(gz2, wt, None, N, H, H, 3, 1, 0.3, mask=a1, mask_slope=0.2)
    y = ops.conv2d_unpooled(g.cuda(), wt.cuda(), gb.cuda(), 0.25 * 0.7, 0.2, N, H, H, 0.3, mask=E.signbytes_of(a1).cuda(), mask_slope=0.2)
    assert rel_err(y, ref) < 2e-5
    y = ops.conv2d_unpooled(g.cuda(), wt.cuda(), gb.cuda(), 0.25 * 0.7, 0.2, N, H, H, 0.3, mask=a1.cuda(), mask_slope=0.2)
    assert rel_err(y, ref) < 2e-5
    # weight gradient of the forward conv co -> cg:  dw [3,3,cg,co] += sum gz2 (x) a1
    dw0, db0 = rnd(3, 3, cg, co, seed=4), rnd(cg, seed=5)
    rdw, rdb = dw0.clone(), db0.clone()
    E.conv2d_wgrad(a1, gz2, rdw, rdb, N, H, H, 3, 1, 0.41)
    dw, db = dw0.cuda(), db0.cuda()
    ops.conv2d_wgrad_unpooled(a1.cuda(), g.cuda(), gb.cuda(), 0.25 * 0.7, 0.2, dw, db, N, H, H, 0.41)
    assert rel_err(dw, rdw) < 2e-5 and rel_err(db, rdb) < 2e-5


from philox_ref import _philox4x32_10  # noqa: E402


@pytest.mark.parametrize('n,seed,offset', [(4, 0, 0), (3, 1337, 0), (16, 1337, 7), (4099, (5 << 40) + 3, (9 << 33) + 1)])
def test_uniform_f32_is_philox_bit_for_bit(n, seed, offset):
    """pg_uniform_f32 (the gradient-penalty mixing factors, wgan_gp_loss.py:15-17): element i = word (i % 4) of
    Philox4x32-10(counter = (i / 4, offset), key = seed), top 24 bits / 2^24 -- bit-exact integer work -- and a U[0,1) sample."""
    out = torch.empty(n, device='cuda')
    ops.uniform_(out, seed, offset)
    got = out.cpu().numpy()
    for i in sorted(set(list(range(min(n, 12))) + [n - 1, n // 2])):
        w = _philox4x32_10([(i // 4) & 0xffffffff, (i // 4) >> 32, offset & 0xffffffff, offset >> 32], [seed & 0xffffffff, seed >> 32])[i % 4]
        assert got[i] == np.float32((w >> 8) / 16777216.0), (i, got[i])
    assert got.min() >= 0.0 and got.max() < 1.0
    big = torch.empty(1 << 20, device='cuda')
    ops.uniform_(big, seed, offset)
    b = big.double()
    assert abs(float(b.mean()) - 0.5) < 2e-3 and abs(float(b.var()) - 1.0 / 12) < 2e-3
    again = torch.empty(1 << 20, device='cuda')
    ops.uniform_(again, seed, offset)
    assert torch.equal(big, again)
    ops.uniform_(again, seed, offset + 1)
    assert not torch.equal(big, again) and abs(float((big * again).mean()) - 0.25) < 2e-3       # next draw: independent
