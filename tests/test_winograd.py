"""Winograd F(2x2,3x3) path (csrc/conv_wino.hip): kernel vs the direct-conv emulation, and the engine with the path forced
on for every eligible layer vs the CPU oracle (host emulation on CPU, HIP on the GPU)."""
import importlib
import os
import sys

import pytest
import torch

from conftest import rel_err
from helpers import reference_grads
import emu_ops as E

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
pg = importlib.import_module('pggan-pytorch_amd')
oracle = importlib.import_module('oracle.pggan_cpu')


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g)


def test_emulated_weight_transform_roundtrip():
    w = rnd(3, 3, 32, 48)
    assert rel_err(E._unwino(E.wino_transform_weights(w)), w) < 1e-5


@pytest.fixture()
def emu(monkeypatch):
    for modname in ('engine', 'optim'):
        mod = importlib.import_module('pggan-pytorch_amd.' + modname)
        monkeypatch.setattr(mod, 'ops', E)
    monkeypatch.setattr(pg.engine, '_check_dev', lambda t, what: t.contiguous())
    yield


def _engine_vs_oracle(dev, monkeypatch, res, depth, alpha, n, kw=None):
    monkeypatch.setattr(pg.engine, 'WINO_MIN_WORKGROUPS', 0)          # every eligible layer takes the Winograd path
    torch.manual_seed(21)
    shape = (1, 3, res, res)
    kw = kw or dict(fmap_base=256, fmap_max=64)
    G = pg.Generator(shape, latent_size=64, **kw)
    D = pg.Discriminator(shape, **kw)
    gp, dp = G.reference_state_dict(), D.reference_state_dict()
    G.to(dev); D.to(dev)
    cfg = oracle.NetCfg(res, 3, latent_size=64, **kw)
    G.depth = D.depth = depth
    G.alpha = D.alpha = alpha
    real, z_d, z_g, mix = oracle.synthetic_batch(300 + depth, n, 3, 4 * 2 ** depth, 64)
    pg.wgan_gp_loss.set_mixing_factors(mix)
    d_cost, rl, fl = pg.wgan_gp_D_loss(D, G, real.to(dev), z_d.to(dev))
    d_cost.backward()
    ref = oracle.d_loss_and_grads(dp, gp, cfg, real, z_d, mix, depth, alpha)
    assert rel_err(d_cost, ref['D_cost']) < 2e-4
    mine = reference_grads(D)
    for k, v in ref['grads'].items():
        assert rel_err(mine[k], v) < 2e-2, (k, rel_err(mine[k], v))
    g_cost = pg.wgan_gp_G_loss(G, D, z_g.to(dev))
    g_cost.backward()
    refg = oracle.g_loss_and_grads(gp, dp, cfg, z_g, depth, alpha)
    assert rel_err(g_cost, refg['G_cost']) < 2e-4
    assert rel_err(G(z_g.to(dev)).cpu(), refg['fake']) < 2e-4
    gm = reference_grads(G)
    for k, v in refg['grads'].items():
        assert rel_err(gm[k], v) < 2e-2, (k, rel_err(gm[k], v))
    used = [m for m in D._layers() if getattr(m, '_wu', None) is not None]
    assert used, 'no layer was eligible for the Winograd path: the test does not exercise it'


@pytest.mark.parametrize('depth,alpha,n', [(2, 1.0, 2), (3, 0.6, 2)])
def test_engine_with_winograd_host(emu, monkeypatch, depth, alpha, n):
    _engine_vs_oracle('cpu', monkeypatch, 32, depth, alpha, n)


@pytest.mark.gpu
@pytest.mark.parametrize('depth,alpha,n', [(2, 1.0, 3), (3, 0.6, 2), (4, 1.0, 2)])
def test_engine_with_winograd_gpu(monkeypatch, depth, alpha, n):
    _engine_vs_oracle('cuda', monkeypatch, 64, depth, alpha, n)


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(2, 32, 64, 64, 0), (3, 16, 128, 96, 0), (1, 8, 32, 48, 0), (9, 8, 512, 512, 0), (2, 64, 32, 64, 1),
                                  (3, 16, 528, 512, 0), (5, 8, 64, 36, 0), (1, 128, 32, 32, 1), (2, 8, 16, 32, 0),
                                  (2, 32, 8, 16, 0), (3, 16, 16, 16, 0), (1, 64, 24, 32, 1), (2, 16, 8, 40, 1), (1, 128, 16, 32, 0)])
@pytest.mark.parametrize('gen', [0, 11, 12, 4])
def test_conv2d_wino_kernel(case, gen):
    """Every Winograd conv kernel against the torch restatement of the conv contract: gen 0 = the built-in choice (second
    generation), 11 / 12 = second generation with 16 / 32 couts per workgroup, 4 = the round-1 kernel (pg_debug_set_wino)."""
    lib = pg._lib.load()
    if gen == 4 and case[2] % 16:
        pytest.skip('the first-generation kernel stages 16-channel chunks (8-channel layers: second generation only)')
    assert lib.pg_debug_set_wino(gen) == 0
    try:
        _wino_kernel_case(case)
        want = 'conv_wino_kernel<4>' if gen == 4 else 'conv_wino2_kernel<'
        name = lib.pg_debug_last_wino_kernel().decode()
        assert name.startswith(want) or (gen == 0 and name.startswith('conv_wino_strip_kernel<')), name
    finally:
        lib.pg_debug_set_wino(0)


def _wino_kernel_case(case):
    N, H, ci, co, ups = case
    ops = pg.ops
    hin = H // 2 if ups else H
    x, w, b = rnd(N, hin, hin, ci), rnd(3, 3, co, ci, seed=1) * 0.2, rnd(co, seed=2)
    m, other, um = rnd(N, H, H, co, seed=3), rnd(N, H // 2, H // 2, co, seed=4), rnd(N, 2 * H, 2 * H, co, seed=5)
    dev = lambda t: t.cuda()
    u = ops.wino_transform_weights(dev(w))
    assert rel_err(ops.wino_unpack(u), E.wino_transform_weights(w)) < 1e-6      # (the device stores 8-channel packs)
    tol = 2e-5
    y = ops.conv2d_wino(dev(x), u, dev(b), N, H, H, 0.37, 0.2, ups=bool(ups))
    assert rel_err(y, E.conv2d(x, w, b, N, H, H, 3, 1, 0.37, slope=0.2, ups=bool(ups))) < tol
    y = ops.conv2d_wino(dev(x), u, None, N, H, H, 0.37, mask=dev(m), mask_slope=0.2, ups=bool(ups))
    assert rel_err(y, E.conv2d(x, w, None, N, H, H, 3, 1, 0.37, mask=m, mask_slope=0.2, ups=bool(ups))) < tol
    if not ups:
        y, yp = ops.conv2d_wino(dev(x), u, dev(b), N, H, H, 0.37, 0.2, pool=True, other=dev(other), a=0.6, b=0.4)
        ry, ryp = E.conv2d_pool(x, w, b, N, H, H, 3, 1, 0.37, slope=0.2, other=other, a=0.6, b=0.4)
        assert rel_err(y, ry) < tol and rel_err(yp, ryp) < tol
        _, yp = ops.conv2d_wino(dev(x), u, None, N, H, H, 0.37, pool=True, a=4.0, pool_only=True)
        assert rel_err(yp, E.conv2d_pool(x, w, None, N, H, H, 3, 1, 0.37, a=4.0)[1]) < tol
        yu = ops.conv2d_wino(dev(x), u, None, N, H, H, 0.37, mask_slope=0.2, unpool=True, upmask=dev(um), up_mul=0.7)
        assert rel_err(yu, E.conv2d_unpool(x, w, N, H, H, 3, 1, 0.37, upmask=um, mul=0.7, mask_slope=0.2)) < tol
        # sign-byte forms: mask given as bytes, full-resolution output kept as bytes next to the pooled one, byte copy of the output
        mb = E.signbytes_of(m)
        y = ops.conv2d_wino(dev(x), u, None, N, H, H, 0.37, mask=dev(mb), mask_slope=0.2)
        assert rel_err(y, E.conv2d(x, w, None, N, H, H, 3, 1, 0.37, mask=m, mask_slope=0.2)) < tol
        yb, yp = ops.conv2d_wino(dev(x), u, dev(b), N, H, H, 0.37, 0.2, pool=True, y_bytes=True)
        ry, ryp = E.conv2d_pool(x, w, b, N, H, H, 3, 1, 0.37, slope=0.2)
        assert yb.dtype == torch.uint8 and rel_err(yp, ryp) < tol
        agree = (yb.cpu() == E.signbytes_of(ry)).float().mean()
        assert float(agree) > 0.9999                       # (an output within round-off of zero may land on either side)
        y, ys = ops.conv2d_wino(dev(x), u, dev(b), N, H, H, 0.37, 0.2, signs_out=True)
        assert rel_err(y, E.conv2d(x, w, b, N, H, H, 3, 1, 0.37, slope=0.2)) < tol
        assert bool((ys.cpu() == E.signbytes_of(y.cpu())).all())               # the byte copy is the sign of the very value stored


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(3, 16, 512, 512, 0), (3, 32, 256, 128, 0), (9, 16, 128, 96, 0), (2, 32, 64, 64, 1), (3, 8, 64, 36, 0),
                                  (1, 16, 40, 32, 0), (5, 32, 16, 24, 1)])
@pytest.mark.parametrize('slices', [-1, 2, 3, 8])
def test_conv2d_wino_k_split(case, slices):
    """The same contract with the K loop sliced across workgroups (pg_set_workspace + the last-arriver fix-up of conv_wino2_kernel<..,
    true>): -1 = the built-in choice (splits the 16x16 / 32x32 maps at minibatch 3), n = n slices forced; every fused epilogue; and
    the result must not depend on which workgroup arrives last (bit-identical repeats)."""
    lib = pg._lib.load()
    N, H, ci, co, ups = case
    assert lib.pg_debug_set_wino_ksplit(slices) == 0
    try:
        _wino_kernel_case(case)
        name = lib.pg_debug_last_wino_kernel().decode()
        nblk = -(-(N * (H // 2) ** 2) // 64) * -(-co // 16)
        if (slices > 1 and ci >= 8 * slices) or (slices == -1 and nblk <= 256 and ci >= 64):
            assert name.count(', true, ') == 1, name
        x, u = rnd(N, H // 2 if ups else H, H // 2 if ups else H, ci).cuda(), pg.ops.wino_transform_weights((rnd(3, 3, co, ci, seed=1) * 0.2).cuda())
        ys = [pg.ops.conv2d_wino(x, u, None, N, H, H, 0.37, 0.2, ups=bool(ups)) for _ in range(4)]
        assert all(torch.equal(ys[0], y) for y in ys[1:])
    finally:
        lib.pg_debug_set_wino_ksplit(-1)


@pytest.mark.gpu
@pytest.mark.parametrize('shapes', [[(64, 32), (16, 48), (512, 512)], [(8, 16), (40, 8), (24, 136)]])
def test_backward_data_form_straight_from_the_parameter(shapes):
    """pg_wino_transform_weights_batched(transposed): the Winograd form of the flipped / transposed weights computed from the forward
    parameter in one pass == the forward-form transform of pg_pack_dgrad_weights' output, bit for bit (same arithmetic, another
    read order), for several layers of one flat buffer; couts that are not multiples of 32, cins of one pack."""
    ops = pg.ops
    ws = [rnd(3, 3, co, ci, seed=7 + i) * 0.3 for i, (co, ci) in enumerate(shapes)]
    flat = torch.cat([w.reshape(-1) for w in ws]).cuda()
    offs, o = [], 0
    for w in ws:
        offs.append(o); o += w.numel()
    uoffs, uo = [], 0
    for co, ci in shapes:
        uoffs.append(uo); uo += 16 * co * ci
    wt = torch.zeros_like(flat)
    ops.pack_dgrad_weights_batched(flat, wt, [(offs[i], 3, co, ci) for i, (co, ci) in enumerate(shapes)])
    want = torch.zeros(uo, device='cuda')
    ops.wino_transform_weights_batched(wt, want, [(offs[i], uoffs[i], ci, co) for i, (co, ci) in enumerate(shapes)])
    got = torch.full((uo,), float('nan'), device='cuda')
    ops.wino_transform_weights_batched(flat, got, [(offs[i], uoffs[i], ci, co) for i, (co, ci) in enumerate(shapes)], transposed=True)
    assert torch.equal(got, want)
    host = torch.zeros(uo)
    E.wino_transform_weights_batched(flat.cpu(), host, [(offs[i], uoffs[i], ci, co) for i, (co, ci) in enumerate(shapes)], transposed=True)
    for i, (co, ci) in enumerate(shapes):                                    # (the device stores 8-channel packs)
        u = got[uoffs[i]:uoffs[i] + 16 * co * ci].view(16, ci, co)                # nominal shape of the backward-data form: [16][cout' = ci][cin' = co]
        assert rel_err(ops.wino_unpack(u), host[uoffs[i]:uoffs[i] + 16 * co * ci].view(16, ci, co)) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(3, 64, 128, 128, 0), (2, 32, 64, 48, 1), (3, 16, 512, 512, 0), (1, 128, 16, 32, 0), (2, 8, 64, 36, 0)])
def test_specialised_epilogues_match_the_general_one(case):
    """conv_wino2_kernel<.., EPI_PLAIN / EPI_MASKB> (bias + LeakyReLU -> y; sign-byte LeakyReLU' factors -> y: 32-bit offsets, no
    option tests) against the general epilogue of the same kernel (pg_debug_set_wino_epi(0)), with and without K slices, and against
    the torch restatement; couts that are not a multiple of 16 (dead lanes), fused-upsample input."""
    N, H, ci, co, ups = case
    lib, ops = pg._lib.load(), pg.ops
    hin = H // 2 if ups else H
    x, w, b = rnd(N, hin, hin, ci), rnd(3, 3, co, ci, seed=1) * 0.2, rnd(co, seed=2)
    m = rnd(N, H, H, co, seed=3)
    mb = E.signbytes_of(m).cuda()
    u = ops.wino_transform_weights(w.cuda())

    def run():
        y = ops.conv2d_wino(x.cuda(), u, b.cuda(), N, H, H, 0.37, 0.2, ups=bool(ups))
        n1 = lib.pg_debug_last_wino_kernel().decode()
        ym = ops.conv2d_wino(x.cuda(), u, None, N, H, H, 0.37, mask=mb, mask_slope=0.2, ups=bool(ups))
        return y, ym, n1, lib.pg_debug_last_wino_kernel().decode()
    y, ym, n1, n2 = run()
    assert n1.endswith(', 1>') and n2.endswith(', 2>'), (n1, n2)
    assert lib.pg_debug_set_wino_epi(0) == 0
    try:
        gy, gym, g1, g2 = run()
        assert g1.endswith(', 0>') and g2.endswith(', 0>'), (g1, g2)
    finally:
        lib.pg_debug_set_wino_epi(-1)
    assert rel_err(y, gy) < 1e-6 and rel_err(ym, gym) < 1e-6
    assert rel_err(y, E.conv2d(x, w, b, N, H, H, 3, 1, 0.37, slope=0.2, ups=bool(ups))) < 2e-5
    assert rel_err(ym, E.conv2d(x, w, None, N, H, H, 3, 1, 0.37, mask=m, mask_slope=0.2, ups=bool(ups))) < 2e-5


@pytest.mark.gpu
def test_workspace_registration_errors_and_unsplit_without_scratch():
    lib = pg._lib.load()
    import ctypes
    s = torch.cuda.Stream()
    buf = torch.zeros(1 << 20, dtype=torch.uint8, device='cuda')
    h = ctypes.c_void_p(s.cuda_stream)
    assert lib.pg_set_workspace(h, ctypes.c_void_p(buf.data_ptr()), 0) == -1           # pointer without a size
    assert lib.pg_set_workspace(h, None, 4096) == -1
    assert lib.pg_set_workspace(h, ctypes.c_void_p(buf.data_ptr()), 1 << 14) == -1     # no room behind the tickets
    assert lib.pg_set_workspace(h, ctypes.c_void_p(buf.data_ptr() + 4), 1 << 19) == -1  # alignment
    assert lib.pg_set_workspace(h, None, 0) == 0                                        # clearing an absent entry is fine
    x, w = rnd(3, 16, 16, 64).cuda(), (rnd(3, 3, 64, 64, seed=1) * 0.2).cuda()
    u = pg.ops.wino_transform_weights(w)
    want = pg.ops.conv2d_wino(x, u, None, 3, 16, 16, 0.37, 0.2)                          # current stream: scratch registered by ops
    assert lib.pg_debug_last_wino_kernel().decode().count(', true, ') == 1
    torch.cuda.synchronize()
    y = torch.empty_like(want)
    args = [ctypes.c_void_p(t.data_ptr()) for t in (x, u)] + [None, None, ctypes.c_void_p(y.data_ptr()), None, None, 1.0, 0.0, 0, None, None, 1.0,
                                                              3, 16, 16, 64, 64, 0, 0.37, 0.2, 0.2, h]
    assert lib.pg_conv2d_wino_nhwc(*args) == 0                                          # a stream without scratch: unsplit launch
    assert not lib.pg_debug_last_wino_kernel().decode().count(', true, ') == 1
    s.synchronize()
    assert rel_err(y, want) < 1e-6
    small = torch.zeros((1 << 14) + (1 << 15), dtype=torch.uint8, device='cuda')        # room for two slices of ONE block only
    assert lib.pg_set_workspace(h, ctypes.c_void_p(small.data_ptr()), small.numel()) == 0
    assert lib.pg_conv2d_wino_nhwc(*args) == 0                                          # 12 blocks x 8 slices do not fit: unsplit
    assert not lib.pg_debug_last_wino_kernel().decode().count(', true, ') == 1
    assert lib.pg_set_workspace(h, ctypes.c_void_p(buf.data_ptr()), buf.numel()) == 0   # re-registering replaces the entry
    assert lib.pg_conv2d_wino_nhwc(*args) == 0
    assert lib.pg_debug_last_wino_kernel().decode().count(', true, ') == 1
    s.synchronize()
    assert rel_err(y, want) < 1e-6
    assert lib.pg_set_workspace(h, None, 0) == 0


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(3, 64, 16, 16, 0), (2, 32, 32, 32, 1), (1, 16, 64, 24, 0), (5, 8, 8, 16, 0), (3, 128, 32, 16, 1),
                                  (2, 16, 16, 8, 0), (1, 32, 128, 32, 0)])
def test_conv2d_wino_pixelnorm_kernel(case):
    """conv -> bias -> LeakyReLU -> PixelNorm in the Winograd epilogue (pg_conv2d_wino_pixelnorm_nhwc) against the torch
    restatement of network.py:32-52; couts 8 / 16 (one block), 24 / 32 (two blocks), upsample-fused input; > 32 couts refused."""
    N, H, ci, co, ups = case
    ops = pg.ops
    hin = H // 2 if ups else H
    x, w, b = rnd(N, hin, hin, ci), rnd(3, 3, co, ci, seed=1) * 0.2, rnd(co, seed=2)
    u = ops.wino_transform_weights(w.cuda())
    y, r = ops.conv2d_wino_pixelnorm(x.cuda(), u, b.cuda(), N, H, H, 0.37, 0.2, 1e-8, ups=bool(ups))
    ry, rr = E.conv2d_pixelnorm(x, w, b, N, H, H, 3, 1, 0.37, 0.2, 1e-8, ups=bool(ups))
    assert rel_err(y, ry) < 2e-5 and rel_err(r, rr.reshape(-1)) < 2e-5
    wide = ops.wino_transform_weights(rnd(3, 3, 48, 16, seed=3).cuda())
    with pytest.raises(ops.Unsupported):
        ops.conv2d_wino_pixelnorm(rnd(1, 16, 16, 16).cuda(), wide, rnd(48).cuda(), 1, 16, 16, 1.0, 0.2)


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(3, 64, 16, 16), (2, 32, 64, 32), (1, 16, 32, 24), (5, 8, 16, 8), (3, 128, 32, 16), (1, 32, 128, 32)])
@pytest.mark.parametrize('pool', [False, True])
def test_conv2d_wino_pnbwd_kernel(case, pool):
    """Backward-data conv (+ the pool that is the upsample's adjoint, + fade-in blend) + the adjoint of (LeakyReLU -> PixelNorm) in the
    Winograd epilogue (pg_conv2d_wino_pnbwd_nhwc) against the torch restatement; couts 8 .. 32; > 32 couts refused."""
    N, H, ci, co = case
    ops = pg.ops
    ho = H // 2 if pool else H
    x, w = rnd(N, H, H, ci), rnd(3, 3, co, ci, seed=1) * 0.2
    ys, rs, other = rnd(N, ho, ho, co, seed=2), rnd(N * ho * ho, seed=3).abs() + 0.5, rnd(N, ho, ho, co, seed=4)
    u = ops.wino_transform_weights(w.cuda())
    for oth in ((None, other) if pool else (None,)):
        y = ops.conv2d_wino_pnbwd(x.cuda(), u, ys.cuda(), rs.cuda(), N, H, H, 0.37, 0.2, pool=pool,
                                  other=None if oth is None else oth.cuda(), a=4.0, b=1.0 if oth is not None else 0.0)
        ref = E.conv2d_wino_pnbwd(x, E.wino_transform_weights(w), ys, rs, N, H, H, 0.37, 0.2, pool=pool, other=oth, a=4.0,
                                  b=1.0 if oth is not None else 0.0)
        assert rel_err(y, ref) < 3e-5
    wide = ops.wino_transform_weights(rnd(3, 3, 48, 16, seed=5).cuda())
    with pytest.raises(ops.Unsupported):
        ops.conv2d_wino_pnbwd(rnd(1, 16, 16, 16).cuda(), wide, rnd(1, 16, 16, 48).cuda(), rnd(256).cuda(), 1, 16, 16, 1.0, 0.2)


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(2, 16, 32, 32, 0), (3, 16, 64, 96, 0), (1, 32, 36, 48, 0), (9, 16, 128, 64, 0), (2, 64, 32, 64, 1),
                                  (5, 32, 8, 40, 1), (1, 128, 32, 32, 0), (7, 16, 100, 36, 0), (2, 16, 32, 32, 1),
                                  (2, 16, 16, 16, 0), (3, 32, 16, 32, 0), (2, 32, 32, 16, 1), (1, 64, 12, 20, 0), (3, 64, 8, 16, 0),
                                  (6, 32, 256, 256, 0), (5, 64, 200, 144, 0), (8, 64, 96, 160, 1)])     # (the last three: pair mapping)
def test_conv2d_wgrad_wino_kernel(case):
    """csrc/conv_wino_wgrad.hip vs the torch restatement of the weight gradient, accumulating onto non-zero dw/db;
    ragged channel counts (not multiples of the 32x32 block) and the upsample-fused input included."""
    N, H, ci, co, ups = case
    ops = pg.ops
    hin = H // 2 if ups else H
    x, gz = rnd(N, hin, hin, ci), rnd(N, H, H, co, seed=1)
    dw0, db0 = rnd(3, 3, co, ci, seed=2), rnd(co, seed=3)
    rdw, rdb = dw0.clone(), db0.clone()
    E.conv2d_wgrad(x, gz, rdw, rdb, N, H, H, 3, 1, 0.41, ups=bool(ups))
    dw, db = dw0.cuda(), db0.cuda()
    ops.conv2d_wgrad_wino(x.cuda(), gz.cuda(), dw, db, N, H, H, 0.41, ups=bool(ups))
    sym = pg._lib.load().pg_debug_last_wino_wgrad_kernel().decode()
    assert sym.startswith('conv_wino_wgrad_')
    if N >= 5 and min(ci, co) >= 96:
        assert sym == 'conv_wino_wgrad_pair_kernel'         # >= 6 regions per workgroup: the wave-pair mapping
    assert rel_err(dw, rdw) < 2e-5 and rel_err(db, rdb) < 2e-5
    dw2 = dw0.cuda()
    ops.conv2d_wgrad_wino(x.cuda(), gz.cuda(), dw2, None, N, H, H, 0.41, ups=bool(ups))
    assert rel_err(dw2, rdw) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(6, 3, 32, 32, 32, 0), (2, 1, 16, 64, 96, 0), (1, 3, 64, 16, 16, 0), (4, 2, 32, 20, 36, 1), (9, 3, 16, 128, 64, 0),
                                  (6, 3, 32, 256, 256, 0)])
@pytest.mark.parametrize('bias2', [False, True])
def test_conv2d_wgrad_wino_two_batches(case, bias2):
    """pg_conv2d_wgrad_wino2_nhwc: two batches of one layer in one launch == the two launches (the deferred gradient-penalty
    tangent term riding in the launch of D's batched adjoint sweep); db from the first batch, optionally the second."""
    N, N2, H, ci, co, ups = case
    ops = pg.ops
    hin = H // 2 if ups else H
    x, gz = rnd(N, hin, hin, ci), rnd(N, H, H, co, seed=1)
    x2, gz2 = rnd(N2, hin, hin, ci, seed=4), rnd(N2, H, H, co, seed=5)
    dw0, db0 = rnd(3, 3, co, ci, seed=2), rnd(co, seed=3)
    rdw, rdb = dw0.clone(), db0.clone()
    E.conv2d_wgrad(x, gz, rdw, rdb, N, H, H, 3, 1, 0.41, ups=bool(ups))
    E.conv2d_wgrad(x2, gz2, rdw, rdb if bias2 else None, N2, H, H, 3, 1, 0.41, ups=bool(ups))
    dw, db = dw0.cuda(), db0.cuda()
    ops.conv2d_wgrad_wino(x.cuda(), gz.cuda(), dw, db, N, H, H, 0.41, ups=bool(ups), second=(x2.cuda(), gz2.cuda(), N2, bias2))
    assert rel_err(dw, rdw) < 2e-5 and rel_err(db, rdb) < 2e-5
    dw = dw0.cuda()                                  # no bias gradient at all
    ops.conv2d_wgrad_wino(x.cuda(), gz.cuda(), dw, None, N, H, H, 0.41, ups=bool(ups), second=(x2.cuda(), gz2.cuda(), N2, bias2))
    assert rel_err(dw, rdw) < 2e-5


@pytest.mark.gpu
def test_deferred_tangent_wgrad_matches_separate_launches(monkeypatch):
    """engine.DEFER_TANGENT_WGRAD: D's gradients with the tangent term carried by the sweep's launches == launched on its own,
    and the two-batch entry point is really taken."""
    from helpers import synthetic
    calls = []
    orig = pg.ops.conv2d_wgrad_wino
    monkeypatch.setattr(pg.ops, 'conv2d_wgrad_wino', lambda *a, **k: (calls.append(k.get('second') is not None), orig(*a, **k))[1])
    grads = {}
    for flag in (True, False):
        monkeypatch.setattr(pg.engine, 'DEFER_TANGENT_WGRAD', flag)
        torch.manual_seed(5)
        shape = (1, 3, 64, 64)
        G, D = pg.Generator(shape, fmap_base=1024).cuda(), pg.Discriminator(shape, fmap_base=1024).cuda()
        G.depth = D.depth = 4
        G.alpha = D.alpha = 1.0
        real, z_d, _, mix = synthetic(3, 3, 3, 64, G.latent_size)
        pg.wgan_gp_loss.set_mixing_factors(mix)
        try:
            cost, _, _ = pg.wgan_gp_D_loss(D, G, real.cuda(), z_d.cuda())
            cost.backward()
        finally:
            pg.wgan_gp_loss.set_mixing_factors(None)
        grads[flag] = reference_grads(D)
        if flag:
            assert any(calls), 'no launch carried a deferred contribution'
            calls.clear()
        else:
            assert not any(calls)
    assert grads[True].keys() == grads[False].keys() and len(grads[True]) > 8
    # Two runs of the same step agree only to the branch-flip level of tests/test_e2e_gpu.py (split-K atomics in the low-resolution
    # layers move activations by an ulp, a LeakyReLU branch may flip: 1e-4 .. 5e-3 per tensor); a lost or doubled tangent term is O(0.1 .. 1)
    num = sum(float(((grads[True][n] - grads[False][n]).double() ** 2).sum()) for n in grads[True])
    den = sum(float((grads[False][n].double() ** 2).sum()) for n in grads[True])
    assert (num / den) ** 0.5 < 4e-3
    for name in grads[True]:
        assert rel_err(grads[True][name], grads[False][name]) < 3e-2, name


@pytest.mark.gpu
def test_conv2d_wgrad_wino_rejects_small_maps():
    ops = pg.ops
    x, gz = torch.zeros(2, 8, 8, 32, device='cuda'), torch.zeros(2, 8, 8, 32, device='cuda')
    with pytest.raises(RuntimeError):
        ops.conv2d_wgrad_wino(x, gz, torch.zeros(3, 3, 32, 32, device='cuda'), None, 2, 8, 8, 1.0)


@pytest.mark.parametrize('depth,alpha,n', [(2, 1.0, 2), (3, 0.6, 2)])
def test_engine_with_sign_bytes_host(emu, monkeypatch, depth, alpha, n):
    """DBlock c2 activations kept as sign bytes (engine.USE_SIGN_BYTES) from 8x8 up: host emulation vs the oracle."""
    monkeypatch.setattr(pg.engine, 'SIGN_BYTES_MIN_H', 8)
    monkeypatch.setattr(pg.engine, 'USE_SIGN_BYTES', True)
    _engine_vs_oracle('cpu', monkeypatch, 32, depth, alpha, n)


@pytest.mark.gpu
@pytest.mark.parametrize('depth,alpha,n,wino', [(3, 1.0, 3, True), (4, 0.6, 2, True), (4, 1.0, 2, False)])
def test_engine_with_sign_bytes_gpu(monkeypatch, depth, alpha, n, wino):
    """Same on the GPU with the byte format forced down to 8x8 maps (Winograd and direct kernels)."""
    monkeypatch.setattr(pg.engine, 'SIGN_BYTES_MIN_H', 8)
    monkeypatch.setattr(pg.engine, 'USE_SIGN_BYTES', True)
    if not wino:
        monkeypatch.setattr(pg.engine, 'USE_WINOGRAD', False)
    _engine_vs_oracle('cuda', monkeypatch, 64, depth, alpha, n)


@pytest.mark.gpu
@pytest.mark.parametrize('alpha', [1.0, 0.5])
def test_engine_with_lazy_pool_adjoint_gpu(monkeypatch, alpha):
    """128x128 network with 8/16-channel top stages: D's backward takes the pool adjoint between the 64^2 and 128^2 blocks
    inside the consumers' gathers (engine.USE_LAZY_UNPOOL) -- checked against the oracle, and that the path is really taken."""
    calls = []
    orig = pg.ops.conv2d_unpooled
    monkeypatch.setattr(pg.ops, 'conv2d_unpooled', lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    _engine_vs_oracle('cuda', monkeypatch, 128, 5, alpha, 2, kw=dict(fmap_base=512, fmap_max=64))
    assert len(calls) > 0                             # ... also across the fade-in boundary (x alpha) since round 4


@pytest.mark.gpu
def test_torch_adam_keeps_derived_weights_fresh(monkeypatch):
    """with torch's own Adam (or any update that does not go through FusedAdam.mark_params_changed) the
    Winograd-domain / backward-data weight copies must still be refreshed — every engine entry point syncs the parameter
    versions.  Three Trainer iterations with torch.optim.Adam at a large learning rate on a Winograd-eligible network against the
    oracle's Adam: stale copies would put iteration 2's losses off by O(lr)."""
    monkeypatch.setattr(pg.engine, 'WINO_MIN_WORKGROUPS', 0)
    torch.manual_seed(33)
    res, depth, n, lr = 32, 3, 4, 0.01
    kw = dict(fmap_base=256, fmap_max=64)
    G = pg.Generator((1, 3, res, res), latent_size=64, **kw)
    D = pg.Discriminator((1, 3, res, res), **kw)
    gp, dp = G.reference_state_dict(), D.reference_state_dict()
    G.cuda(); D.cuda()
    G.depth = D.depth = depth
    cfg = oracle.NetCfg(res, 3, latent_size=64, **kw)
    opt_d = torch.optim.Adam(D.parameters(), lr, betas=(0.0, 0.99))
    opt_g = torch.optim.Adam(G.parameters(), lr, betas=(0.0, 0.99))
    batches = [oracle.synthetic_batch(700 + it, n, 3, res, 64) for it in range(3)]
    state = dict(it=0, z=0)

    def loader():
        while True:
            yield batches[state['it']][0]

    def rlg():
        b = batches[state['it']]
        state['z'] += 1
        return b[1] if state['z'] % 2 == 1 else b[2]

    def d_loss(Dm, Gm, real, z):
        pg.wgan_gp_loss.set_mixing_factors(batches[state['it']][3])
        return pg.wgan_gp_D_loss(Dm, Gm, real, z)
    tr = pg.Trainer(D, G, d_loss, pg.wgan_gp_G_loss, opt_d, opt_g, None, loader(), rlg)
    losses = []

    class Rec(pg.Plugin):
        def __init__(self):
            super(Rec, self).__init__([(1, 'iteration')])

        def register(self, trainer):
            pass

        def iteration(self, i, g_cost, d_cost, d_real, d_fake):
            losses.append((float(g_cost), float(d_cost)))
    tr.register_plugin(Rec())
    og, od = oracle.AdamState(), oracle.AdamState()
    used = [m for m in list(D._layers()) + list(G._layers()) if getattr(m, '_wu', None) is not None]
    for it in range(3):
        state['it'], state['z'] = it, 0
        tr.train()
        real, z_d, z_g, mix = batches[it]
        d, g = oracle.train_iteration(gp, dp, cfg, og, od, real, z_d, z_g, mix, depth, 1.0, lr, lr)
        gc, dc = losses[it]
        tol = 5e-4 if it == 0 else 2e-2       # Adam(beta1 = 0) at lr 0.01 is sign-like: round-off-sized gradients move weights by +-lr
        assert abs(dc - float(d['D_cost'])) < tol * max(1.0, abs(float(d['D_cost']))), (it, dc, float(d['D_cost']))
        assert abs(gc - float(g['G_cost'])) < tol * max(1.0, abs(float(g['G_cost']))), (it, gc, float(g['G_cost']))
        # the sharp form: outputs computed now (derived copies as the engine finds them) == outputs after forcing a re-derivation
        zt, xt = z_g.cuda(), real.cuda()
        y1, s1 = G(zt).clone(), D(xt).clone()
        G.mark_params_changed()
        D.mark_params_changed()
        # (split-K layers accumulate with atomics: bit-equality is not guaranteed; stale copies would differ by O(lr) = 1e-2)
        assert rel_err(G(zt), y1) < 1e-5 and rel_err(D(xt), s1) < 1e-5, 'derived weight copies were stale after torch.optim.Adam.step()'
    used = [m for m in list(D._layers()) + list(G._layers()) if getattr(m, '_wu', None) is not None]
    assert used, 'no layer took the Winograd path'


STRIP_CASES = [(2, 128, 8, 16, 0), (1, 128, 16, 32, 1), (2, 128, 32, 32, 0), (1, 256, 16, 16, 0), (1, 128, 32, 64, 0),
               (1, 128, 8, 32, 0), (1, 256, 32, 16, 1)]


@pytest.mark.gpu
@pytest.mark.parametrize('case', STRIP_CASES)
def test_conv2d_wino_strip_kernel(case):
    """The row-streaming Winograd conv (csrc/conv_wino_strip.hip: 64-column strips, ring of five row pairs, weights resident in LDS)
    against the torch restatement of the conv contract with every fused epilogue (bias + LeakyReLU, fp32 / sign-byte masks, pool +
    blend, pool only, pool adjoint, sign bytes out), and bit for bit against the tile kernel it replaces (pg_debug_set_wino(20)):
    same sums in the same order.  Cin 8 / 16 / 32, 16 .. 64 couts (one and two cout blocks per workgroup, cout groups), fused
    upsample, several segments per image (256 rows)."""
    lib, ops = pg._lib.load(), pg.ops
    N, H, ci, co, ups = case
    assert lib.pg_debug_set_wino(21) == 0                  # (21: the row-streaming form wherever it exists -- the built-in choice keeps the tile kernel for 32 input channels)
    try:
        _wino_kernel_case(case)
        name = lib.pg_debug_last_wino_kernel().decode()
        assert name.startswith('conv_wino_strip_kernel<%d, ' % ci), name
        _strip_vs_tile(case)
    finally:
        lib.pg_debug_set_wino(0)


def _strip_vs_tile(case):
    lib, ops = pg._lib.load(), pg.ops
    N, H, ci, co, ups = case
    hin = H // 2 if ups else H
    x, w, b = rnd(N, hin, hin, ci).cuda(), rnd(3, 3, co, ci, seed=1) * 0.2, rnd(co, seed=2).cuda()
    u = ops.wino_transform_weights(w.cuda())
    mb = E.signbytes_of(rnd(N, H, H, co, seed=3)).cuda()

    m, other, um = rnd(N, H, H, co, seed=3), rnd(N, H // 2, H // 2, co, seed=4).cuda(), rnd(N, 2 * H, 2 * H, co, seed=5)
    umb = E.signbytes_of(um).cuda()
    names = []

    def run():
        out = [ops.conv2d_wino(x, u, b, N, H, H, 0.37, 0.2, ups=bool(ups)),
               ops.conv2d_wino(x, u, None, N, H, H, 0.37, mask=mb, mask_slope=0.2, ups=bool(ups))]
        out += list(ops.conv2d_wino(x, u, b, N, H, H, 0.37, 0.2, ups=bool(ups), signs_out=True))
        names.append(lib.pg_debug_last_wino_kernel().decode())
        if not ups:
            out += list(ops.conv2d_wino(x, u, b, N, H, H, 0.37, 0.2, pool=True, y_bytes=True))
            out += list(ops.conv2d_wino(x, u, b, N, H, H, 0.37, 0.2, pool=True, other=other, a=0.6, b=0.4, y_bytes=True))
            names.append(lib.pg_debug_last_wino_kernel().decode())
            out.append(ops.conv2d_wino(x, u, None, N, H, H, 0.37, mask=mb, mask_slope=0.2, pool=True, pool_only=True)[1])
            out.append(ops.conv2d_wino(x, u, None, N, H, H, 0.37, mask=mb, mask_slope=0.2, pool=True, other=other, a=0.6, b=0.4, pool_only=True)[1])
            names.append(lib.pg_debug_last_wino_kernel().decode())
            out.append(ops.conv2d_wino(x, u, None, N, H, H, 0.37, mask_slope=0.2, unpool=True, upmask=umb, up_mul=0.7))
            names.append(lib.pg_debug_last_wino_kernel().decode())
        return out
    got = run()
    ncb = 2 if (co % 32 == 0 and ci <= 16) else 1
    want_names = ['<%d, %d, 3>' % (ci, 1)] + ([] if ups else ['<%d, %d, 4>' % (ci, ncb), '<%d, %d, 5>' % (ci, ncb), '<%d, %d, 6>' % (ci, 1)])
    if ncb == 1:                                           # (two cout blocks per workgroup: only the forms the 16->32 / 8->32 layers launch are specialised)
        assert [nm[nm.index('<'):] for nm in names] == want_names, names
    assert lib.pg_debug_set_wino(20) == 0
    try:
        want = run()
    finally:
        lib.pg_debug_set_wino(21)
    assert names[-1].startswith('conv_wino2_kernel<'), names
    for i, (a, b_) in enumerate(zip(got, want)):
        if i in (8, 9):                                    # masked + pooled: hipcc contracts mul + add of the pooling differently in the two kernels
            assert rel_err(a, b_) < 1e-6
        else:
            assert torch.equal(a, b_), i
    if not ups:                                            # the specialised forms against the torch restatement
        ry, ryp = E.conv2d_pool(x.cpu(), w, b.cpu(), N, H, H, 3, 1, 0.37, slope=0.2, other=other.cpu(), a=0.6, b=0.4)
        assert rel_err(got[7], ryp) < 2e-5 and float((got[6].cpu() == E.signbytes_of(ry)).float().mean()) > 0.9999
        assert rel_err(got[9], E.conv2d_pool(x.cpu(), w, None, N, H, H, 3, 1, 0.37, mask=m, mask_slope=0.2, other=other.cpu(), a=0.6, b=0.4)[1]) < 2e-5
        assert rel_err(got[10], E.conv2d_unpool(x.cpu(), w, N, H, H, 3, 1, 0.37, upmask=um, mul=0.7, mask_slope=0.2)) < 2e-5


@pytest.mark.gpu
def test_k_sliced_launches_stress_two_streams():
    """The hand-rolled hand-over of the K-sliced launches (agent-scope sc1 stores of the partial sums, s_waitcnt vmcnt(0), relaxed
    agent-scope ticket, last arriver reads with sc1 loads: conv_wino.hip) under load: 2 000 launches with 8 forced slices, four
    shapes round robin on the main stream, while a second stream runs its own K-sliced launches (its own scratch) over the same
    CUs all the time.  Every result must equal the first result of its shape bit for bit (slices are added in slice order, whoever
    arrives last) -- a stale read of a slice or of a ticket would show up as a mismatch -- and the unsplit launch to rounding."""
    lib, ops = pg._lib.load(), pg.ops
    shapes = [(3, 16, 512, 512), (3, 32, 256, 256), (9, 16, 128, 128), (2, 8, 512, 64)]
    data = []
    for i, (N, H, ci, co) in enumerate(shapes):
        x = rnd(N, H, H, ci, seed=i).cuda()
        u = ops.wino_transform_weights((rnd(3, 3, co, ci, seed=10 + i) * 0.1).cuda())
        mb = E.signbytes_of(rnd(N, H, H, co, seed=20 + i)).cuda()
        data.append((x, u, mb, N, H))
    assert lib.pg_debug_set_wino_ksplit(0) == 0
    try:
        unsplit = [ops.conv2d_wino(x, u, None, N, H, H, 0.37, mask=mb, mask_slope=0.2) for x, u, mb, N, H in data]
        assert lib.pg_debug_set_wino_ksplit(8) == 0
        ref = [ops.conv2d_wino(x, u, None, N, H, H, 0.37, mask=mb, mask_slope=0.2) for x, u, mb, N, H in data]
        assert lib.pg_debug_last_wino_kernel().decode().count(', true, ') == 1
        for a, b in zip(ref, unsplit):
            assert rel_err(a, b) < 1e-5
        bad = torch.zeros((), dtype=torch.int64, device='cuda')
        bad2 = torch.zeros((), dtype=torch.int64, device='cuda')
        side = torch.cuda.Stream()
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        for it in range(2000):
            k = it % len(data)
            x, u, mb, N, H = data[k]
            y = ops.conv2d_wino(x, u, None, N, H, H, 0.37, mask=mb, mask_slope=0.2)
            bad += torch.count_nonzero(y != ref[k])
            if it % 2 == 0:                                   # the second stream: its own sliced launches, other shape phase
                with torch.cuda.stream(side):
                    k2 = (it // 2 + 1) % len(data)
                    x2, u2, mb2, N2, H2 = data[k2]
                    assert lib.pg_debug_set_wino_ksplit(8) == 0          # (thread-local, but the setting is per call site: keep it explicit)
                    y2 = ops.conv2d_wino(x2, u2, None, N2, H2, H2, 0.37, mask=mb2, mask_slope=0.2)
                    bad2 += torch.count_nonzero(y2 != ref[k2])
        main.wait_stream(side)
        torch.cuda.synchronize()
        assert int(bad) == 0 and int(bad2) == 0, (int(bad), int(bad2))
    finally:
        lib.pg_debug_set_wino_ksplit(-1)


@pytest.mark.gpu
def test_workspace_bytes_query():
    """pg_workspace_bytes (scratch sizes are queried, the caller owns the buffer): a scratch of exactly the queried size
    makes the launch slice, one byte less and it runs unsplit; every layer of the 1024x1024 schedule at minibatch 3 / 9 fits the 32 MB
    the host layer registers."""
    import ctypes
    lib, ops = pg._lib.load(), pg.ops
    N, H, ci, co = 3, 16, 256, 256
    need = ops.workspace_bytes(0, N, H, H, ci, co)
    assert need > 16384 and ops.workspace_bytes(0, 9, 256, 256, 64, 64) == 0            # (a full grid never slices)
    x, w = rnd(N, H, H, ci).cuda(), (rnd(3, 3, co, ci, seed=1) * 0.2).cuda()
    u = ops.wino_transform_weights(w)
    s = torch.cuda.Stream()
    h = ctypes.c_void_p(s.cuda_stream)
    buf = torch.zeros(need, dtype=torch.uint8, device='cuda')
    y = torch.empty(N, H, H, co, device='cuda')
    args = [ctypes.c_void_p(t.data_ptr()) for t in (x, u)] + [None, None, ctypes.c_void_p(y.data_ptr()), None, None, 1.0, 0.0, 0, None, None, 1.0,
                                                              N, H, H, ci, co, 0, 0.37, 0.2, 0.2, h]
    try:
        assert lib.pg_set_workspace(h, ctypes.c_void_p(buf.data_ptr()), need) == 0
        assert lib.pg_conv2d_wino_nhwc(*args) == 0
        assert lib.pg_debug_last_wino_kernel().decode().count(', true, ') == 1
        assert lib.pg_set_workspace(h, ctypes.c_void_p(buf.data_ptr()), need - 16) == 0
        assert lib.pg_conv2d_wino_nhwc(*args) == 0
        assert lib.pg_debug_last_wino_kernel().decode().count(', true, ') == 0
        s.synchronize()
    finally:
        lib.pg_set_workspace(h, None, 0)
    worst = 0
    for n in (3, 9):
        for hh, c_in, c_out in ((8, 512, 512), (16, 512, 512), (32, 512, 256), (32, 256, 512), (64, 256, 128), (64, 128, 256), (128, 128, 64)):
            worst = max(worst, ops.workspace_bytes(0, n, hh, hh, c_in, c_out))
        worst = max(worst, ops.workspace_bytes(1, n, 4, 4, 512, 512))
    assert 0 < worst <= ops.WORKSPACE_BYTES
    err = ctypes.c_size_t(0)
    assert lib.pg_workspace_bytes(7, 1, 8, 8, 8, 16, ctypes.addressof(err)) == -1 and lib.pg_workspace_bytes(0, 1, 8, 8, 8, 16, None) == -1
