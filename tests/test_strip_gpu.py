"""Row-streaming kernels of the 8/16-channel 1024^2 layers (csrc/conv_strip.hip) on the MI355X.

conv_strip_kernel serves the same C-ABI entry points as the tile kernel it replaces (pg_conv2d_nhwc, pg_conv2d_pool_nhwc,
pg_conv2d_pixelnorm_nhwc, pg_conv2d_pnbwd_nhwc for 3x3 / pad 1 / 8 couts / 8 or 16 input channels; reference network.py:33-36
and its adjoint forms) with the SAME MFMA accumulation order, so for every fused epilogue its result must be BIT-IDENTICAL to
the tile kernel's (selected with pg_debug_set_tuning(3, 20)) and within fp32 round-off of the torch-CPU statement of the
contract (tests/emu_ops.py).  Shapes: every strip / segment geometry the launcher can pick (64 .. 512 wide, 1 .. 9 images,
segments of 16 / 32 / 64 rows), image borders on all four sides of a strip, the x2-upsample gather."""
import numpy as np
import pytest
import torch

import emu_ops as E
from conftest import rel_err

import pggan_amd as pg

pytestmark = pytest.mark.gpu
ops = pg.ops
lib = pg._lib.load()


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g)


def last_kernel():
    return lib.pg_debug_last_conv_kernel().decode()


class tile_kernel(object):
    """``with tile_kernel():`` launches go to conv_thin_kernel (the A/B switch of the dispatcher)."""
    def __enter__(self):
        assert lib.pg_debug_set_tuning(3, 20) == 0

    def __exit__(self, *exc):
        lib.pg_debug_set_tuning(3, -1)
        return False


def both(fn):
    """fn() through the strip kernel and through the tile kernel; asserts that each really ran."""
    a = fn()
    assert last_kernel().startswith('conv_strip_kernel'), last_kernel()
    with tile_kernel():
        b = fn()
        assert last_kernel().startswith('conv_thin_kernel'), last_kernel()
    torch.cuda.synchronize()
    return a, b


def same(a, b):
    """Strip vs tile kernel: same MFMA order, so the results agree to the last bits — up to the one place where hipcc may or may
    not contract ``acc * scale + bias`` into an FMA in the two kernels (1 ulp); integer outputs (sign bytes) must be equal."""
    if isinstance(a, (tuple, list)):
        return all(same(x, y) for x, y in zip(a, b))
    if a is None or b is None:
        return a is None and b is None
    if a.dtype == torch.uint8:
        return float((a != b).float().mean()) < 1e-5          # (a value within 1 ulp of zero may take the other sign)
    return rel_err(a, b) < 1e-6


CASES = [(1, 64, 64, 8, 8), (2, 128, 128, 8, 8), (3, 256, 256, 8, 8), (1, 64, 64, 16, 8), (2, 128, 128, 16, 8),
         (9, 64, 64, 8, 8), (1, 512, 512, 8, 8), (1, 256, 256, 16, 8)]      # N, H, W, Cin, Cout


@pytest.mark.parametrize('case', CASES)
def test_strip_equals_tile_kernel_and_contract(case):
    N, H, W, ci, co = case
    assert H == W
    x, w, b = rnd(N, H, H, ci), rnd(3, 3, co, ci, seed=1) * 0.2, rnd(co, seed=2)
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    m = rnd(N, H, H, co, seed=3)
    md, mb = m.cuda(), E.signbytes_of(m).cuda()
    # forward (bias + LeakyReLU), masked linear map with fp32 / byte masks, linear
    s, t = both(lambda: ops.conv2d(xd, wd, bd, N, H, H, 3, 1, 0.37, slope=0.2))
    assert same(s, t) and rel_err(s, E.conv2d(x, w, b, N, H, H, 3, 1, 0.37, slope=0.2)) < 2e-5
    s, t = both(lambda: ops.conv2d(xd, wd, None, N, H, H, 3, 1, 0.37, mask=md, mask_slope=0.2))
    assert same(s, t) and rel_err(s, E.conv2d(x, w, None, N, H, H, 3, 1, 0.37, mask=m, mask_slope=0.2)) < 2e-5
    s2, t2 = both(lambda: ops.conv2d(xd, wd, None, N, H, H, 3, 1, 0.37, mask=mb, mask_slope=0.2))
    assert same(s2, t2) and same(s2, s)
    s, t = both(lambda: ops.conv2d(xd, wd, bd, N, H, H, 3, 1, 0.4, 0.2, signs_out=True))
    assert same(s, t) and torch.equal(s[1].cpu(), E.signbytes_of(s[0].cpu()))        # (bytes consistent with the SAME launch's y: exact)
    # PixelNorm epilogue and the adjoint of (LeakyReLU -> PixelNorm)
    s, t = both(lambda: ops.conv2d_pixelnorm(xd, wd, bd, N, H, H, 3, 1, 0.37, 0.2, 1e-8))
    assert same(s, t)
    yref, rref = E.conv2d_pixelnorm(x, w, b, N, H, H, 3, 1, 0.37, 0.2, 1e-8) if hasattr(E, 'conv2d_pixelnorm') else (None, None)
    if yref is not None:
        assert rel_err(s[0], yref) < 2e-5 and rel_err(s[1], rref) < 2e-5
    ys, rs = s
    g = rnd(N, H, H, ci, seed=5).cuda()
    wt = (rnd(3, 3, co, ci, seed=6) * 0.2).cuda()
    s, t = both(lambda: ops.conv2d_pnbwd(g, wt, ys, rs, N, H, H, 3, 1, 0.3, 0.2))
    assert same(s, t)


@pytest.mark.parametrize('case', [(2, 64, 8, 8), (1, 128, 16, 8), (3, 128, 8, 8), (1, 256, 8, 8)])
def test_strip_pool_epilogues(case):
    N, H, ci, co = case
    x, w, b = rnd(N, H, H, ci), rnd(3, 3, co, ci, seed=1) * 0.2, rnd(co, seed=2)
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    other = rnd(N, H // 2, H // 2, co, seed=4).cuda()
    m = rnd(N, H, H, co, seed=3)
    md = m.cuda()
    s, t = both(lambda: ops.conv2d_pool(xd, wd, bd, N, H, H, 3, 1, 0.37, 0.2))
    assert same(s, t)
    ref = E.conv2d(x, w, b, N, H, H, 3, 1, 0.37, slope=0.2)
    assert rel_err(s[0], ref) < 2e-5 and rel_err(s[1], E.avgpool2_fwd(ref, None, 1.0, 0.0)) < 2e-5
    s, t = both(lambda: ops.conv2d_pool(xd, wd, bd, N, H, H, 3, 1, 0.37, 0.2, other=other, a=0.3, b=0.7))
    assert same(s, t) and rel_err(s[1], E.avgpool2_fwd(ref, other.cpu(), 0.3, 0.7)) < 2e-5
    s, t = both(lambda: ops.conv2d_pool(xd, wd, None, N, H, H, 3, 1, 0.37, 1.0, mask=md, mask_slope=0.2, other=other, a=0.3, b=0.7,
                                        pool_only=True))
    assert same(s[1], t[1])
    tref = E.conv2d(x, w, None, N, H, H, 3, 1, 0.37, mask=m, mask_slope=0.2)
    assert rel_err(s[1], E.avgpool2_fwd(tref, other.cpu(), 0.3, 0.7)) < 2e-5


@pytest.mark.parametrize('case', [(2, 64, 16, 8), (1, 128, 8, 8), (2, 256, 16, 8)])
def test_strip_upsampled_gather(case):
    """nearest x2 upsample fused into the row gather (the generator's c1 layers, network.py:62-66)."""
    N, H, ci, co = case
    x, w, b = rnd(N, H // 2, H // 2, ci), rnd(3, 3, co, ci, seed=1) * 0.2, rnd(co, seed=2)
    s, t = both(lambda: ops.conv2d(x.cuda(), w.cuda(), b.cuda(), N, H, H, 3, 1, 0.37, slope=0.2, ups=True))
    assert same(s, t) and rel_err(s, E.conv2d(x, w, b, N, H, H, 3, 1, 0.37, slope=0.2, ups=True)) < 2e-5
    s, t = both(lambda: ops.conv2d_pixelnorm(x.cuda(), w.cuda(), b.cuda(), N, H, H, 3, 1, 0.37, 0.2, 1e-8, ups=True))
    assert same(s, t)


def test_strip_segment_geometries(monkeypatch):
    """The launcher picks 16 / 32 / 64-row segments from the launch size; every choice must give the same result (each image
    row is the last row of one segment and the halo of the next in one of them)."""
    N, H, ci, co = 2, 256, 8, 8
    x, w, b = rnd(N, H, H, ci).cuda(), (rnd(3, 3, co, ci, seed=1) * 0.2).cuda(), rnd(co, seed=2).cuda()
    with tile_kernel():
        ref = ops.conv2d(x, w, b, N, H, H, 3, 1, 0.37, slope=0.2)
    y = ops.conv2d(x, w, b, N, H, H, 3, 1, 0.37, slope=0.2)
    assert last_kernel().startswith('conv_strip_kernel') and same(y, ref)
    # small launches (few strips) fall to 16-row segments, large ones keep 64: N = 1 @64 (1 strip) .. N = 9 @512 (72 strips)
    for n, h in ((1, 64), (9, 512)):
        xx = rnd(n, h, h, ci, seed=7).cuda()
        with tile_kernel():
            r = ops.conv2d(xx, w, b, n, h, h, 3, 1, 0.37, slope=0.2)
        assert same(ops.conv2d(xx, w, b, n, h, h, 3, 1, 0.37, slope=0.2), r)


def test_shapes_outside_the_strip_kernel_keep_the_tile_kernel():
    x, w, b = rnd(2, 32, 32, 8).cuda(), (rnd(3, 3, 8, 8, seed=1) * 0.2).cuda(), rnd(8, seed=2).cuda()
    ops.conv2d(x, w, b, 2, 32, 32, 3, 1, 0.37, slope=0.2)              # 32 columns: narrower than a strip
    assert last_kernel().startswith('conv_thin_kernel'), last_kernel()


class tile_wgrad(object):
    """``with tile_wgrad():`` weight-gradient launches go to conv_wgrad_thin_kernel (A/B switch of the dispatcher)."""
    def __enter__(self):
        assert lib.pg_debug_set_tuning(1, 20) == 0

    def __exit__(self, *exc):
        lib.pg_debug_set_tuning(1, -1)
        return False


@pytest.mark.parametrize('case', [(2, 64, 8, 8, 0), (1, 128, 8, 16, 0), (1, 128, 16, 8, 0), (3, 256, 8, 8, 0), (9, 64, 8, 8, 0),
                                  (2, 128, 16, 8, 1), (1, 256, 8, 8, 1), (1, 512, 8, 8, 0), (1, 1024, 8, 8, 0)])
def test_wgrad_strip_against_contract_and_tile_kernel(case):
    """wgrad_strip_kernel (pg_conv2d_wgrad_nhwc for 3x3 / pad 1 layers with 8 / 16 channels on >= 64-wide maps): accumulates into dW
    and db like the tile kernel, same result up to the order of the fp32 sums; the x2-upsample gather of the generator's c1 layers."""
    N, H, ci, co, ups = case
    hin = H // 2 if ups else H
    x, gz = rnd(N, hin, hin, ci), rnd(N, H, H, co, seed=1)
    dw0, db0 = rnd(3, 3, co, ci, seed=2), rnd(co, seed=3)
    rdw, rdb = dw0.clone(), db0.clone()
    E.conv2d_wgrad(x, gz, rdw, rdb, N, H, H, 3, 1, 0.41, ups=bool(ups))
    dw, db = dw0.cuda(), db0.cuda()
    ops.conv2d_wgrad(x.cuda(), gz.cuda(), dw, db, N, H, H, 3, 1, 0.41, ups=bool(ups))
    assert last_kernel().startswith('wgrad_strip_kernel'), last_kernel()
    assert rel_err(dw, rdw) < 2e-5 and rel_err(db, rdb) < 2e-5, (rel_err(dw, rdw), rel_err(db, rdb))
    tdw, tdb = dw0.cuda(), db0.cuda()
    with tile_wgrad():
        ops.conv2d_wgrad(x.cuda(), gz.cuda(), tdw, tdb, N, H, H, 3, 1, 0.41, ups=bool(ups))
        assert last_kernel().startswith('conv_wgrad_thin_kernel'), last_kernel()
    assert rel_err(dw, tdw) < 1e-5 and rel_err(db, tdb) < 1e-5
    # without a bias gradient (the tangent-term launches), accumulating twice
    dw2 = dw0.cuda()
    ops.conv2d_wgrad(x.cuda(), gz.cuda(), dw2, None, N, H, H, 3, 1, 0.41, ups=bool(ups))
    ops.conv2d_wgrad(x.cuda(), gz.cuda(), dw2, None, N, H, H, 3, 1, 0.41, ups=bool(ups))
    assert rel_err(dw2 - dw0.cuda(), 2 * (rdw - dw0)) < 2e-5


@pytest.mark.parametrize('case', [(2, 64, 16, 8), (1, 128, 8, 8), (2, 128, 16, 8), (1, 256, 16, 8), (3, 256, 8, 8)])
def test_strip_pool_adjoint_in_the_gather(case):
    """pg_conv2d_unpooled_nhwc / pg_conv2d_wgrad_unpooled_nhwc on the row-streaming kernels: the pool adjoint (x 1/4 x mul x
    LeakyReLU' sign byte) of the coarse gradient is evaluated on the way into the LDS ring (registers -> ds_write) instead of
    being materialised at the fine resolution; against the materialised form (emu) and the tile kernels."""
    N, H, cg, co = case                                  # g has cg channels at H/2; backward-data conv cg -> co
    g, a2 = rnd(N, H // 2, H // 2, cg), rnd(N, H, H, cg, seed=1)
    gb = E.signbytes_of(a2).cuda()
    wt, a1 = rnd(3, 3, co, cg, seed=2) * 0.2, rnd(N, H, H, co, seed=3)
    gz2 = E.avgpool2_bwd(g, a2, 0.7, 0.2)
    ref = E.conv2d(gz2, wt, None, N, H, H, 3, 1, 0.3, mask=a1, mask_slope=0.2)
    gd, wtd, a1b = g.cuda(), wt.cuda(), E.signbytes_of(a1).cuda()
    s, t = both(lambda: ops.conv2d_unpooled(gd, wtd, gb, 0.25 * 0.7, 0.2, N, H, H, 0.3, mask=a1b, mask_slope=0.2))
    assert same(s, t) and rel_err(s, ref) < 2e-5
    s, t = both(lambda: ops.conv2d_unpooled(gd, wtd, gb, 0.25 * 0.7, 0.2, N, H, H, 0.3, mask=a1.cuda(), mask_slope=0.2))
    assert same(s, t) and rel_err(s, ref) < 2e-5
    # weight gradient of the forward conv co -> cg:  dw [3,3,cg,co] += sum gz2 (x) a1
    dw0, db0 = rnd(3, 3, cg, co, seed=4), rnd(cg, seed=5)
    rdw, rdb = dw0.clone(), db0.clone()
    E.conv2d_wgrad(a1, gz2, rdw, rdb, N, H, H, 3, 1, 0.41)
    dw, db = dw0.cuda(), db0.cuda()
    ops.conv2d_wgrad_unpooled(a1.cuda(), gd, gb, 0.25 * 0.7, 0.2, dw, db, N, H, H, 0.41)
    assert last_kernel().startswith('wgrad_strip_kernel'), last_kernel()
    assert rel_err(dw, rdw) < 2e-5 and rel_err(db, rdb) < 2e-5
    tdw, tdb = dw0.cuda(), db0.cuda()
    with tile_wgrad():
        ops.conv2d_wgrad_unpooled(a1.cuda(), gd, gb, 0.25 * 0.7, 0.2, tdw, tdb, N, H, H, 0.41)
        assert last_kernel().startswith('conv_wgrad_thin_kernel'), last_kernel()
    assert rel_err(dw, tdw) < 1e-5 and rel_err(db, tdb) < 1e-5


@pytest.mark.parametrize('N,H,W,C', [(3, 64, 64, 3), (2, 128, 256, 3), (1, 1024, 1024, 3), (9, 32, 128, 1), (1, 16, 64, 2), (5, 32, 256, 3), (2, 16, 512, 2)])
def test_conv_with_fromrgb_in_the_gather(N, H, W, C):
    """pg_conv2d_fromrgb_nhwc (conv_strip_rgb_kernel): c1(fromRGB(img)) of a DBlock in one launch -- the 1x1 conv + LeakyReLU of
    reference network.py:145 evaluated in the row gather of the 3x3 conv (network.py:33-36), its output never written.  Against the
    torch statement of the two layers, and against pg_fromrgb_fwd followed by pg_conv2d_nhwc (same order of operations per pixel: the
    sign bytes of fromRGB's output are identical; same MFMA order: the conv output agrees to the last bits).  Image borders on all four
    sides of a strip / segment, one to three image channels, segments of 16 / 32 / 64 rows."""
    img = rnd(N, C, H, W, seed=1)
    rw, rb = rnd(8, C, seed=2) * 0.7, rnd(8, seed=3) * 0.3
    w, b = rnd(3, 3, 8, 8, seed=4) * 0.2, rnd(8, seed=5) * 0.1
    d = lambda t: t.cuda()
    x0, x0b = ops.fromrgb_fwd(d(img), d(rw), d(rb), N, C, H, W, 0.61, 0.2, signs_out=True)
    want, wantb = ops.conv2d(x0, d(w), d(b), N, H, W, 3, 1, 0.37, 0.2, signs_out=True)
    assert last_kernel().startswith('conv_strip_kernel'), last_kernel()
    y, yb, xb = ops.conv2d_fromrgb(d(img), d(rw), d(rb), 0.61, 0.2, d(w), d(b), N, C, H, W, 0.37, 0.2)
    assert last_kernel().startswith('conv_strip_rgb_kernel'), last_kernel()
    torch.cuda.synchronize()
    assert torch.equal(xb, x0b)                                # fromRGB's output: bit-identical (explicit roundings in both kernels)
    assert same(y, want) and same(yb, wantb)                   # (the conv: same MFMA order; see ``same`` for the one free contraction)
    ex0 = E.fromrgb_fwd(img, rw, rb, N, C, H, W, 0.61, 0.2)
    assert rel_err(y, E.conv2d(ex0, w, b, N, H, W, 3, 1, 0.37, slope=0.2)) < 2e-5
    assert bool((xb.cpu() == E.signbytes_of(x0.cpu())).all())


def test_conv_with_fromrgb_unsupported_shapes():
    """Outside the 8 -> 8 layer on strip-sized maps the entry point refuses (ops.Unsupported: the caller keeps the two launches)."""
    d = lambda t: t.cuda()
    for (N, H, W, C, cm) in [(1, 32, 32, 3, 8), (1, 64, 64, 4, 8), (1, 64, 64, 3, 16)]:
        with pytest.raises(ops.Unsupported):
            ops.conv2d_fromrgb(d(rnd(N, C, H, W)), d(rnd(cm, C)), d(rnd(cm)), 1.0, 0.2, d(rnd(3, 3, 8, cm)), d(rnd(8)), N, C, H, W, 1.0, 0.2)


@pytest.mark.parametrize('N,H,W,C', [(3, 64, 64, 3), (2, 128, 256, 3), (1, 1024, 1024, 3), (9, 32, 128, 1), (1, 16, 64, 2)])
def test_pixelnorm_conv_with_torgb_in_the_epilogue(N, H, W, C):
    """pg_conv2d_pixelnorm_torgb_nhwc: the generator's last conv (+ bias + LeakyReLU + PixelNorm, reference network.py:33-41) with the
    block's toRGB layer (network.py:49, :138) in the same epilogue.  y and r exactly as pg_conv2d_pixelnorm_nhwc writes them (the same
    kernel body), the image against pg_torgb_fwd on that y (the two lanes of a pixel add their partial sums: another order of the
    eight products, 1e-6) and against the torch statement of the three layers; the image written into a caller's buffer."""
    x, w, b = rnd(N, H, W, 8, seed=1), rnd(3, 3, 8, 8, seed=2) * 0.2, rnd(8, seed=3) * 0.1
    tw, tb = rnd(C, 8, seed=4) * 0.5, rnd(C, seed=5) * 0.2
    d = lambda t: t.cuda()
    y0, r0 = ops.conv2d_pixelnorm(d(x), d(w), d(b), N, H, W, 3, 1, 0.37, 0.2, 1e-8)
    assert last_kernel().startswith('conv_strip_kernel<8, 8, 3'), last_kernel()
    img0 = ops.torgb_fwd(y0, d(tw), d(tb), N, C, H, W, 0.71)
    buf = torch.full((N + 1, C, H, W), float('nan'), device='cuda')
    y, r, img = ops.conv2d_pixelnorm_torgb(d(x), d(w), d(b), d(tw), d(tb), N, C, H, W, 0.37, 0.2, 0.71, 1e-8, out=buf[1:])
    assert last_kernel().startswith('conv_strip_x_kernel<3, '), last_kernel()
    torch.cuda.synchronize()
    assert torch.equal(y, y0) and torch.equal(r, r0)
    assert img.data_ptr() == buf[1:].data_ptr() and bool(torch.isnan(buf[0]).all())
    assert rel_err(img, img0) < 1e-6
    ey, er = E.conv2d_pixelnorm(x, w, b, N, H, W, 3, 1, 0.37, 0.2, 1e-8)
    assert rel_err(img, E.torgb_fwd(ey, tw, tb, N, C, H, W, 0.71)) < 2e-5


@pytest.mark.parametrize('N,H,W,C', [(3, 64, 64, 3), (2, 128, 256, 3), (1, 1024, 1024, 3), (9, 32, 128, 1), (1, 16, 64, 2)])
@pytest.mark.parametrize('keep', [True, False])
def test_masked_backward_conv_with_fromrgb_adjoint_in_the_epilogue(N, H, W, C, keep):
    """pg_conv2d_masked_fromrgb_bwd_nhwc: the entry block's backward-data conv (x LeakyReLU' from sign bytes) with fromRGB's backward-data
    (the adjoint of reference network.py:145) in the same epilogue.  The 8-channel gradient exactly as pg_conv2d_nhwc writes it (same
    kernel body) -- or not written at all (``keep=False``) --, the image gradient against pg_fromrgb_bwd_data on it and against
    the torch statement of the two steps."""
    gz, wt = rnd(N, H, W, 8, seed=1), rnd(3, 3, 8, 8, seed=2) * 0.2
    m = rnd(N, H, W, 8, seed=3)
    rw = rnd(8, C, seed=4) * 0.5
    d = lambda t: t.cuda()
    mb = E.signbytes_of(m)
    gf0 = ops.conv2d(d(gz), d(wt), None, N, H, W, 3, 1, 0.37, 1.0, mask=d(mb), mask_slope=0.2)
    assert last_kernel().startswith('conv_strip_kernel<8, 8, 2'), last_kernel()
    gi0 = torch.empty(N, C, H, W, device='cuda')
    ops.fromrgb_bwd_data(gf0, d(rw), gi0, N, C, H, W, 0.61)
    gf, gi = ops.conv2d_masked_fromrgb_bwd(d(gz), d(wt), d(mb), 0.2, d(rw), 0.61, N, C, H, W, 0.37, keep_gf=keep)
    assert last_kernel().startswith('conv_strip_x_kernel<2, '), last_kernel()
    torch.cuda.synchronize()
    assert (gf is None) == (not keep)
    if keep:
        assert torch.equal(gf, gf0)
    assert rel_err(gi, gi0) < 1e-6
    egf = E.conv2d(gz, wt, None, N, H, W, 3, 1, 0.37, mask=m, mask_slope=0.2)
    egi = torch.zeros(N, C, H, W)
    E.fromrgb_bwd_data(egf, rw, egi, N, C, H, W, 0.61)
    assert rel_err(gi, egi) < 2e-5


@pytest.mark.parametrize('N,H,W,C', [(3, 64, 64, 3), (2, 128, 256, 3), (1, 1024, 1024, 3), (9, 32, 128, 1), (1, 16, 64, 2)])
@pytest.mark.parametrize('keep,want_gimg', [(False, False), (True, True), (False, True)])
def test_masked_backward_conv_with_fromrgb_weight_gradient_in_the_epilogue(N, H, W, C, keep, want_gimg):
    """pg_conv2d_masked_fromrgb_bwd_nhwc with (img, rgb_dw, rgb_db): fromRGB's weight and bias gradient (reference: autograd of network.py:145
    under trainer.py:98) accumulated in the epilogue of the entry block's backward-data conv, one commit per workgroup -- against
    pg_fromrgb_wgrad on the 8-channel gradient pg_conv2d_nhwc writes and against the torch statement; accumulates INTO dw / db; alone (the
    batched sweep: nothing else is written), with the image gradient, with the 8-channel gradient."""
    gz, wt = rnd(N, H, W, 8, seed=1), rnd(3, 3, 8, 8, seed=2) * 0.2
    m, img = rnd(N, H, W, 8, seed=3), rnd(N, C, H, W, seed=6)
    rw = rnd(8, C, seed=4) * 0.5
    d = lambda t: t.cuda()
    mb = E.signbytes_of(m)
    gf0 = ops.conv2d(d(gz), d(wt), None, N, H, W, 3, 1, 0.37, 1.0, mask=d(mb), mask_slope=0.2)
    dw0, db0 = torch.full((8, C, 1, 1), 0.25, device='cuda'), torch.full((8,), -0.5, device='cuda')
    dw, db = dw0.clone(), db0.clone()
    ops.fromrgb_wgrad(gf0, d(img), dw0, db0, N, C, H, W, 0.61)
    ops.fromrgb_wgrad(gf0, d(img), dw0, db0, N, C, H, W, 0.61)
    for _ in range(2):                                      # (accumulates: twice the sums)
        gf, gi = ops.conv2d_masked_fromrgb_bwd(d(gz), d(wt), d(mb), 0.2, d(rw), 0.61, N, C, H, W, 0.37, keep_gf=keep, want_gimg=want_gimg,
                                               img=d(img), rgb_dw=dw, rgb_db=db)
    assert last_kernel() == 'conv_strip_x_kernel<2, false, 3>', last_kernel()
    torch.cuda.synchronize()
    assert (gf is None) == (not keep) and (gi is None) == (not want_gimg)
    if keep:
        assert same(gf, gf0)
    assert rel_err(dw, dw0) < 2e-5 and rel_err(db, db0) < 2e-5
    egf = E.conv2d(gz, wt, None, N, H, W, 3, 1, 0.37, mask=m, mask_slope=0.2)
    edw, edb = torch.full((8, C), 0.25), torch.full((8,), -0.5)
    E.fromrgb_wgrad(egf, img, edw, edb, N, C, H, W, 0.61)
    E.fromrgb_wgrad(egf, img, edw, edb, N, C, H, W, 0.61)
    assert rel_err(dw.view(8, C), edw) < 5e-5 and rel_err(db, edb) < 5e-5
    if want_gimg:
        egi = torch.zeros(N, C, H, W)
        E.fromrgb_bwd_data(egf, rw, egi, N, C, H, W, 0.61)
        assert rel_err(gi, egi) < 2e-5
