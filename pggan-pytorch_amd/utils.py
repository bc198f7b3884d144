"""Host-side helpers with the reference's names (/root/reference/utils.py) that the hot path's callers use."""
import numpy as np
import torch


def random_latents(num_latents, latent_size):
    """reference utils.py:56-57 — host-side ``np.random.randn`` -> fp32 tensor (moved to the GPU by Trainer)."""
    return torch.from_numpy(np.random.randn(num_latents, latent_size).astype(np.float32))


def generate_samples(generator, gen_input):
    """reference utils.py:8-11."""
    out = generator.forward(gen_input)
    return out.cpu().data.numpy()


def adjust_dynamic_range(data, range_in, range_out):
    """reference utils.py:24-30."""
    if range_in != range_out:
        (min_in, max_in) = range_in
        (min_out, max_out) = range_out
        scale_factor = (max_out - min_out) / (max_in - min_in)
        data = (data - min_in) * scale_factor + min_out
    return data


def rampup(cur_nimg, lr_rampup_kimg=40):
    """Learning-rate ramp-up multiplier.  reference train.py:151-156."""
    if cur_nimg < lr_rampup_kimg * 1000:
        p = max(0.0, 1 - cur_nimg / (lr_rampup_kimg * 1000))
        return np.exp(-p * p * 5.0)
    return 1.0


class SyntheticDataset(object):
    """Seeded synthetic images in [-1,1) with the DepthDataset protocol the DepthManager drives
    (``model_depth``, ``alpha``, ``shape``; reference dataset.py:31-70).  Batches are generated on the
    device, so the benchmark measures the train step and not a host data loader.
    ``ring`` > 0: a loader pre-generates that many batches and cycles through them, so that no RNG kernel runs inside
    the timed train steps (the reference's DataLoader workers produce batches off the step's critical path too)."""

    def __init__(self, resolution, num_channels=3, seed=1337, device='cuda', ring=0, host=False):
        self.shape = (1, num_channels, resolution, resolution)
        self.model_depth = 0
        self.alpha = 1.0
        self.device = device
        self.ring = int(ring)
        self.host = bool(host)       # batches live in PINNED HOST memory (what a DataLoader with pin_memory hands over): the Trainer uploads them
        self._gen = torch.Generator(device=device)
        self._gen.manual_seed(int(seed))

    def batch(self, n):
        r = 4 * 2 ** self.model_depth
        x = torch.rand((n, self.shape[1], r, r), device=self.device, dtype=torch.float32, generator=self._gen)
        x = x.mul_(2).sub_(1)
        if self.host:
            h = torch.empty(x.shape, dtype=x.dtype, pin_memory=True)
            h.copy_(x)
            return h
        return x

    def loader(self, minibatch_size):
        if self.ring > 0:
            batches = [self.batch(minibatch_size) for _ in range(self.ring)]
            i = 0
            while True:
                yield batches[i]
                i = (i + 1) % len(batches)
        while True:
            yield self.batch(minibatch_size)

    def close(self):
        pass


def device_latents(minibatch_size, latent_size, seed=1337, device='cuda', ring=0):
    """Device-side replacement of ``random_latents`` for benchmarks (no H2D copy per iteration).  ``ring`` > 0:
    cycle through that many pre-generated draws (no RNG kernel inside the timed steps)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))

    def draw():
        return torch.randn((minibatch_size, latent_size), device=device, dtype=torch.float32, generator=gen)
    if ring <= 0:
        return draw
    draws = [draw() for _ in range(ring)]
    state = [0]

    def cycle():
        z = draws[state[0]]
        state[0] = (state[0] + 1) % len(draws)
        return z
    return cycle


def prepare_real_batch(batch_u8, alpha, range_in=(0, 255), range_out=(-1, 1)):
    """Device-side DepthDataset.__getitem__ for a whole uint8 batch [N,C,H,W] (reference dataset.py:54-67):
    fade-in blend with the 2x2 box-filtered copy when alpha < 1, dynamic-range change, fp32.  Doing this on the
    device also removes the reference's fork-staleness of ``dataset.alpha`` in DataLoader workers."""
    from . import ops
    return ops.real_prepare_u8(batch_u8, alpha, range_in, range_out)


def build_pyramid(batch_u8, max_depth_diff, range_in=(0, 255)):
    """Device-side multi-depth pyramid of a uint8 batch (DefaultImageFolderDataset.load / create_datapoint_from_depth,
    dataset.py:205-216,243-250): returns [level 0 (= input), level 1, ...], level d derived from the full-resolution
    image with depth difference d exactly like the reference does."""
    from . import ops
    return [batch_u8] + [ops.pyramid_level_u8(batch_u8, d, range_in) for d in range(1, max_depth_diff + 1)]


class DeviceImageSaver(object):
    """Postprocessor with the reference hook signature ``proc(out, description)`` (output_postprocess.py:21-71,
    plugins.py:188-192).  The grid / nearest upsample / range / uint8 conversion runs on the device
    (pg_image_grid_u8); only the final uint8 grid crosses PCIe and is written as PNG."""

    output_file_format = 'fakes_{}.png'
    accepts_device_tensors = True            # plugins.OutputGenerator then skips the D2H copy of the fp32 samples

    def __init__(self, samples_path='.', drange=(-1, 1), resolution=512, create_subdirs=True):
        import os
        self.samples_path, self.drange, self.resolution = samples_path, drange, resolution
        if create_subdirs:
            os.makedirs(self.samples_path, exist_ok=True)

    def to_grid(self, output):
        from . import ops
        t = output if torch.is_tensor(output) else torch.from_numpy(np.ascontiguousarray(output, dtype=np.float32))
        t = t.cuda().contiguous()
        up = 1 if self.resolution is None else max(1, self.resolution // t.shape[-1])
        return ops.image_grid_u8(t, self.drange, up)

    def __call__(self, output, description):
        import os
        import PIL.Image
        grid = self.to_grid(output).cpu().numpy()
        im = PIL.Image.fromarray(grid[:, :, 0], 'L') if grid.shape[2] == 1 else PIL.Image.fromarray(grid, 'RGB')
        fname = self.output_file_format
        if type(description) is int:
            fname = fname.format('{:06}')
        im.save(os.path.join(self.samples_path, fname.format(description)))


def output_samples(generator_path, num_samples, postprocessors, description):
    """reference generate.py:18-30: load a whole-module generator snapshot, sample, run the postprocessors."""
    G = torch.load(generator_path, weights_only=False)
    G.cuda()
    latent_size = getattr(G, 'latent_size', 512)
    gen_input = random_latents(num_samples, latent_size).cuda()
    out_dev = G.forward(gen_input)
    out_host = None
    for proc in postprocessors:
        if getattr(proc, 'accepts_device_tensors', False):
            proc(out_dev, description)
        else:
            if out_host is None:
                out_host = out_dev.cpu().numpy()
            proc(out_host, description)
    return out_dev
