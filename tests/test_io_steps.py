"""The steps either side of the hot path: dataset fade-in + dynamic range on the way in,
ImageSaver's grid/uint8 conversion on the way out.  Oracle (numpy) vs the reference's own outputs on CPU;
HIP kernels vs the oracle on the GPU — integer/byte outputs and the fp64-evaluated input step are BIT-EXACT."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

from oracle import io_steps as oio


def _fx():
    return np.load(os.path.join(GOLDEN, 'io_steps.npz'))


def test_oracle_matches_reference_bit_exact():
    fx = _fx()
    for tag in ('a', 'b', 'c'):
        out = oio.real_prepare(fx['real/%s/in' % tag], float(fx['real/%s/alpha' % tag]))
        assert out.dtype == np.float32 and np.array_equal(out, fx['real/%s/out' % tag])
    for tag in ('g6', 'g1', 'g5', 'g4'):
        res = int(fx['grid/%s/res' % tag])
        out = oio.image_grid_u8(fx['grid/%s/in' % tag], (-1, 1), None if res < 0 else res)
        assert out.dtype == np.uint8 and np.array_equal(out, fx['grid/%s/out' % tag])
    for tag in ('p1', 'p2', 'p3', 'p1b'):
        out = oio.create_datapoint_from_depth(fx['pyr/%s/in' % tag], int(fx['pyr/%s/diff' % tag]))
        assert out.dtype == np.uint8 and np.array_equal(out, fx['pyr/%s/out' % tag])


@pytest.mark.gpu
def test_kernels_match_oracle_bit_exact():
    import pggan_amd as pg
    fx = _fx()
    for tag in ('a', 'b', 'c'):
        x = torch.from_numpy(fx['real/%s/in' % tag]).cuda()
        out = pg.ops.real_prepare_u8(x, float(fx['real/%s/alpha' % tag]))
        assert np.array_equal(out.cpu().numpy(), fx['real/%s/out' % tag]), tag
    for tag in ('g6', 'g1', 'g5', 'g4'):
        imgs = fx['grid/%s/in' % tag]
        res = int(fx['grid/%s/res' % tag])
        up = 1 if res < 0 else res // imgs.shape[-1]
        grid = pg.ops.image_grid_u8(torch.from_numpy(imgs).cuda(), (-1, 1), up).cpu().numpy()
        ref = fx['grid/%s/out' % tag]
        assert np.array_equal(grid.reshape(ref.shape), ref), tag
    for tag in ('p1', 'p2', 'p3', 'p1b'):
        x = torch.from_numpy(fx['pyr/%s/in' % tag]).cuda()
        out = pg.ops.pyramid_level_u8(x, int(fx['pyr/%s/diff' % tag]))
        assert np.array_equal(out.cpu().numpy(), fx['pyr/%s/out' % tag]), tag
    big = np.random.RandomState(3).randint(0, 256, size=(2, 3, 1024, 1024)).astype(np.uint8)
    levels = pg.utils.build_pyramid(torch.from_numpy(big).cuda(), 3)
    for d in (1, 2, 3):
        ref = np.stack([oio.create_datapoint_from_depth(im, d) for im in big])
        assert np.array_equal(levels[d].cpu().numpy(), ref), d
    # full-size property checks against the oracle: 1024x1024 batch (sizes of BASELINE config 5)
    rs = np.random.RandomState(0)
    x = rs.randint(0, 256, size=(3, 3, 1024, 1024)).astype(np.uint8)
    for alpha in (0.25, 1.0):
        out = pg.ops.real_prepare_u8(torch.from_numpy(x).cuda(), alpha).cpu().numpy()
        assert np.array_equal(out, oio.real_prepare(x, alpha))
    imgs = (rs.randn(6, 3, 256, 256) * 0.8).astype(np.float32)
    grid = pg.ops.image_grid_u8(torch.from_numpy(imgs).cuda(), (-1, 1), 2).cpu().numpy()
    assert np.array_equal(grid, oio.image_grid_u8(imgs, (-1, 1), 512))
    # alpha == 1 is the identity on the image content (only the range changes); fade is idempotent at alpha == 0
    a0 = pg.ops.real_prepare_u8(torch.from_numpy(x).cuda(), 0.0, (0, 255), (0, 255))
    blocks = a0.reshape(3, 3, 512, 2, 512, 2)
    assert torch.equal(blocks[:, :, :, 0, :, 0], blocks[:, :, :, 1, :, 1])
