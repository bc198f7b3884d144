"""CPU oracle (numpy) for the steps either side of the hot path.  TEST INFRASTRUCTURE ONLY (see
oracle/pggan_cpu.py for the rules).  Pinned by tests/golden/io_steps.npz, which is produced by the
reference's own functions (tests/golden/make_golden.py: make_io_steps)."""
import numpy as np


def adjust_dynamic_range(data, range_in, range_out):
    """reference utils.py:24-30."""
    if range_in != range_out:
        (min_in, max_in) = range_in
        (min_out, max_out) = range_out
        scale_factor = (max_out - min_out) / (max_in - min_in)
        data = (data - min_in) * scale_factor + min_out
    return data


def alpha_fade(datapoint, alpha):
    """reference dataset.py:109-113 (identical in FolderDataset :238-242): blend with the 2x2 box-filtered copy."""
    c, h, w = datapoint.shape
    t = datapoint.reshape(c, h // 2, 2, w // 2, 2).mean((2, 4)).repeat(2, 1).repeat(2, 2)
    return datapoint + (t - datapoint) * (1 - alpha)


def real_prepare(batch_u8, alpha, range_in=(0, 255), range_out=(-1, 1)):
    """DepthDataset.__getitem__ (dataset.py:54-67) applied to every image of a uint8 batch [N,C,H,W]."""
    out = []
    for x in batch_u8:
        d = x
        if alpha < 1.0:                                                   # :62
            d = alpha_fade(d, alpha)
        d = adjust_dynamic_range(d, range_in, range_out)                  # :65
        out.append(d.astype('float32'))                                   # :67
    return np.stack(out)


def upsample_nearest(x, scale):
    """reference utils.py:33-53 for the last two dims with an integer factor."""
    return x.repeat(scale, axis=-2).repeat(scale, axis=-1) if scale > 1 else x


def image_grid_u8(images, drange=(-1, 1), resolution=None):
    """ImageSaver.__call__ up to the PIL hand-off (output_postprocess.py:35-62): returns the uint8 HWC array
    ([H,W] for one channel) that ``PIL.Image.fromarray`` receives."""
    if resolution is not None:
        images = upsample_nearest(images, resolution // images.shape[-1])            # :64
    count, channels, img_h, img_w = images.shape
    grid_w = max(int(np.ceil(np.sqrt(count))), 1)                                    # :38
    grid_h = max((count - 1) // grid_w + 1, 1)                                       # :39
    grid = np.zeros((channels,) + (grid_h * img_h, grid_w * img_w), dtype=images.dtype)
    for i in range(count):
        x = (i % grid_w) * img_w
        y = (i // grid_w) * img_h
        grid[:, y: y + img_h, x: x + img_w] = images[i]
    image = grid[0] if channels == 1 else grid.transpose(1, 2, 0)                    # :51-56
    image = adjust_dynamic_range(image, drange, (0, 255))                            # :58
    return image.round().clip(0, 255).astype(np.uint8)                               # :60


def create_datapoint_from_depth(datapoint, depthdiff, range_in=(0, 255), scale_factor=2):
    """reference dataset.py:243-250: one level of the multi-depth pyramid from a uint8 image [C,H,W].  The
    reference adds the scale_factor^2 sub-grids taken with stride scale_factor**depthdiff (a 2x2 box mean for
    depthdiff 1, a 4-sample subsampling for larger differences), divides, rounds half-to-even, clips, uint8."""
    d = datapoint.astype(np.float32)
    st = scale_factor ** depthdiff
    acc = 0
    for a in range(scale_factor):
        for b in range(scale_factor):
            acc = acc + d[:, a::st, b::st]
    acc = acc / (scale_factor ** 2)
    return np.uint8(np.clip(np.round(acc), range_in[0], range_in[1]))
