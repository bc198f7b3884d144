"""Trainer plugins of the hot path: DepthManager and LRScheduler (/root/reference/plugins.py:13-99).

``torch.utils.trainer.plugins`` — the base classes the reference imports (plugins.py:8-9) — was removed
from PyTorch after 0.3, so the minimal ``Plugin`` protocol is restated here: an object with
``trigger_interval = [(n, unit), ...]``, ``register(trainer)`` and one method per unit
(``iteration`` / ``epoch`` / ``s`` / ``end``) — exactly what ``Trainer.call_plugins`` relies on.
All schedule arithmetic is Python int (+ one IEEE double division for alpha): bit-exact by construction."""
import os
import time
from datetime import timedelta
from glob import glob


class Plugin(object):
    """The plugin protocol ``Trainer`` dispatches on: ``trigger_interval`` = list of (period, unit) pairs, ``register(trainer)``,
    and one method per unit it listens on."""

    def __init__(self, interval=None):
        self.trigger_interval = [] if interval is None else interval

    def register(self, trainer):
        raise NotImplementedError


_STAT_FMT = ['{val:.2f}']


def growth_stage(cur_nimg, lod_training_nimg, lod_transition_nimg, max_depth):
    """The progressive-growing schedule as a pure integer function of the images shown so far (behaviour of reference
    plugins.py:58-63, bit for bit): time is cut into cycles of ``stabilise`` (lod_training_nimg) + ``fade``
    (lod_transition_nimg) images; the stage index is the number of completed cycles plus the number of whole
    ``stabilise`` spans already inside the current one, clamped to ``max_depth``; while a new stage is still un-clamped
    and past its stabilise span, alpha is the position inside the remaining span over ``fade`` (one IEEE double division),
    otherwise exactly 1.0.  Returns (depth, alpha)."""
    cycle, inside = divmod(cur_nimg, lod_training_nimg + lod_transition_nimg)
    spans, position = divmod(inside, lod_training_nimg)
    stage = cycle + spans
    depth = stage if stage < max_depth else max_depth
    fading = spans > 0 and stage == depth
    return depth, (position / lod_transition_nimg if fading else 1.0)


class DepthManager(Plugin):
    """Growth-stage controller (reference plugins.py:13-81: same constructor arguments, same trainer/dataset fields
    written, same ``stats`` keys).  On a stage change it rebuilds the data iterator and the latent generator for the
    stage's minibatch size and sets the tick length; alpha is pushed to D, G and the dataset whenever it changes."""

    def __init__(self, create_dataloader_fun, create_rlg, max_depth, minibatch_default=16,
                 minibatch_overrides={6: 14, 7: 6, 8: 3}, tick_kimg_default=20,
                 tick_kimg_overrides={3: 10, 4: 10, 5: 5, 6: 2, 7: 2, 8: 1},
                 lod_training_nimg=100 * 1000, lod_transition_nimg=100 * 1000, max_lod=None, depth_offset=None):
        super(DepthManager, self).__init__([(1, 'iteration')])
        self.create_dataloader_fun, self.create_rlg = create_dataloader_fun, create_rlg
        self.max_depth = max_depth
        self.minibatch_default, self.minibatch_overrides = minibatch_default, minibatch_overrides
        self.tick_kimg_default, self.tick_kimg_overrides = tick_kimg_default, tick_kimg_overrides
        self.lod_training_nimg, self.lod_transition_nimg = lod_training_nimg, lod_transition_nimg
        self.max_lod, self.depth_offset = max_lod, depth_offset      # only for the 'lod' statistic of the original paper
        self.trainer = None
        self.depth = self.alpha = -1                                 # "nothing applied yet"

    @property
    def _reports_lod(self):
        return self.max_lod is not None and self.depth_offset is not None

    @property
    def lod(self):
        return self.max_lod - self.depth_offset - self.depth - self.alpha + 1 if self._reports_lod else -1

    def schedule(self, cur_nimg):
        return growth_stage(cur_nimg, self.lod_training_nimg, self.lod_transition_nimg, self.max_depth)

    def register(self, trainer):
        self.trainer = trainer
        stats = trainer.stats
        stats['minibatch_size'] = self.minibatch_default
        stats['alpha'] = dict(log_name='alpha', log_epoch_fields=list(_STAT_FMT), val=self.alpha)
        if self._reports_lod:
            stats['lod'] = dict(log_name='lod', log_epoch_fields=list(_STAT_FMT), val=self.lod)
        self.iteration()                                             # stage 0 is applied before the first step

    def _enter_stage(self, depth):
        tr = self.trainer
        for target in (tr.D, tr.G):
            target.depth = depth
        tr.dataset.model_depth = depth
        self.depth = depth
        batch = self.minibatch_overrides.get(depth, self.minibatch_default)
        tr.dataiter = iter(self.create_dataloader_fun(batch))
        tr.random_latents_generator = self.create_rlg(batch)
        tr.tick_duration_nimg = 1000 * self.tick_kimg_overrides.get(depth, self.tick_kimg_default)
        tr.stats['minibatch_size'] = batch

    def _set_alpha(self, alpha):
        tr = self.trainer
        for target in (tr.D, tr.G, tr.dataset):
            target.alpha = alpha
        self.alpha = alpha

    def iteration(self, *args):
        tr = self.trainer
        depth, alpha = self.schedule(tr.cur_nimg)
        if depth != self.depth:
            self._enter_stage(depth)
        if alpha != self.alpha:
            self._set_alpha(alpha)
        tr.stats['depth'] = depth
        tr.stats['alpha']['val'] = alpha
        if self._reports_lod:
            tr.stats['lod']['val'] = self.lod


class LRScheduler(Plugin):
    """Steps both learning-rate schedules with the image counter after every iteration and once at registration
    (reference plugins.py:84-99); the schedulers only need ``step(cur_nimg)``."""

    def __init__(self, lr_scheduler_d, lr_scheduler_g):
        super(LRScheduler, self).__init__([(1, 'iteration')])
        self.lrs_d, self.lrs_g = lr_scheduler_d, lr_scheduler_g

    def register(self, trainer):
        self.trainer = trainer
        self.iteration()

    def iteration(self, *args):
        shown = self.trainer.cur_nimg
        for sched in (self.lrs_d, self.lrs_g):
            sched.step(shown)


class RampupLR(object):
    """``LambdaLR(opt, rampup).step(cur_nimg)`` without the scheduler machinery: sets
    ``lr = base_lr * fn(cur_nimg)`` on every param group (what plugins.py:97-99 drives).  Like torch's
    schedulers the base rate is kept in the group as ``initial_lr``, so it survives an optimizer
    state_dict round trip (a resumed run must not ramp from the already-ramped rate)."""

    def __init__(self, optimizer, fn):
        self.optimizer, self.fn = optimizer, fn
        for g in optimizer.param_groups:
            g.setdefault('initial_lr', g['lr'])
        self.base_lrs = [g['initial_lr'] for g in optimizer.param_groups]

    def step(self, cur_nimg):
        for g, base in zip(self.optimizer.param_groups, self.base_lrs):
            g['lr'] = base * self.fn(cur_nimg)


class TimeMonitor(Plugin):
    """Per-tick timing stats under the reference's names (plugins.py:114-139: ``stats['time']``, ``stats['sec']['tick']``,
    ``stats['sec']['kimg']``) plus the two numbers this project's benchmark is quoted in: ``stats['img/s']`` (images shown per
    second over the tick, all ranks) and ``stats['d_gp_ms']`` (device time of the D step + gradient penalty + Adam(D), mean of the
    sampled iterations of the tick; ``Trainer`` brackets every ``sample_every``-th D update with two HIP events on the streams it
    runs on).  The device is synchronised at tick boundaries only -- the host clock is the device's there and nowhere else."""

    stat_name = 'time'

    def __init__(self, base_time=0, sample_every=16):
        super(TimeMonitor, self).__init__([(1, 'epoch')])
        self.base_time = base_time
        self.sample_every = int(sample_every)

    def register(self, trainer):
        self.trainer = trainer
        self.start_time = self.epoch_start = time.time()
        self.start_nimg = trainer.cur_nimg
        trainer.stats['sec'] = {'log_format': ':.1f'}
        trainer.stats['img/s'] = dict(val=0.0, log_epoch_fields=['{val:.1f}'], log_name='img/s')
        trainer.stats['d_gp_ms'] = dict(val=0.0, log_epoch_fields=['{val:.3f}'], log_name='d_gp_ms')
        trainer.d_step_probe = dict(every=max(1, self.sample_every), pairs=[])

    def epoch(self, epoch_index):
        import torch
        tr = self.trainer
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        now = time.time()
        tick_time = now - self.epoch_start
        self.epoch_start = now
        nimg = max(1, tr.cur_nimg - self.start_nimg)
        self.start_nimg = tr.cur_nimg
        tr.stats['time'] = timedelta(seconds=now - self.start_time + self.base_time)
        tr.stats['sec']['tick'] = tick_time
        tr.stats['sec']['kimg'] = tick_time / nimg * 1000
        tr.stats['img/s']['val'] = nimg / max(tick_time, 1e-9)
        probe = getattr(tr, 'd_step_probe', None)
        if probe and probe['pairs']:
            ms = [a.elapsed_time(b) for a, b in probe['pairs']]
            tr.stats['d_gp_ms']['val'] = sum(ms) / len(ms)
            del probe['pairs'][:]


AbsoluteTimeMonitor = TimeMonitor      # the reference's name (plugins.py:114; train.py: ``AbsoluteTimeMonitor(params['resume_time'])``): drop-in


class SaverPlugin(Plugin):
    """Network snapshots with the reference's file names and whole-module pickles (plugins.py:142-174):
    ``network-snapshot-{generator|discriminator}-{kimg:06}.dat`` every ``network_snapshot_ticks`` ticks and at
    the end.  The pickles carry the packed weights AND the equalized-lr constants ``c`` (network.py:19-20 keeps
    ``c`` outside the state_dict, so whole-module pickling is what makes a snapshot resumable).

    Addition over the reference: ``network-snapshot-trainer-{kimg:06}.dat`` with both Adam
    states and ``cur_nimg`` — the reference restarts Adam from zero moments on resume."""

    last_pattern = 'network-snapshot-{}-{}.dat'

    def __init__(self, checkpoints_path, keep_old_checkpoints=False, network_snapshot_ticks=40):
        super(SaverPlugin, self).__init__([(network_snapshot_ticks, 'epoch'), (1, 'end')])
        self.checkpoints_path = checkpoints_path
        self.keep_old_checkpoints = keep_old_checkpoints

    def register(self, trainer):
        self.trainer = trainer

    def epoch(self, epoch_index):
        import torch
        tr = self.trainer
        if tr.parallel is not None and tr.parallel.rank != 0:
            return                                                   # replicas are identical: rank 0 writes
        if not self.keep_old_checkpoints:
            self._clear(self.last_pattern.format('*', '*'))
        kimg = '{:06}'.format(tr.cur_nimg // 1000)
        for model, name in [(tr.G, 'generator'), (tr.D, 'discriminator')]:
            torch.save(model, os.path.join(self.checkpoints_path, self.last_pattern.format(name, kimg)))
        state = {'cur_nimg': tr.cur_nimg, 'optimizer_d': tr.optimizer_d.state_dict(),
                 'optimizer_g': tr.optimizer_g.state_dict()}
        torch.save(state, os.path.join(self.checkpoints_path, self.last_pattern.format('trainer', kimg)))

    def end(self, *args):
        self.epoch(*args)

    def _clear(self, pattern):
        for file_name in glob(os.path.join(self.checkpoints_path, pattern)):
            os.remove(file_name)


def load_models(resume_network, result_dir, logger=None):
    """reference train.py:60-64: ``resume_network`` is a pattern with one ``{}`` for generator/discriminator."""
    import torch
    if logger is not None:
        logger.log('Resuming {}'.format(resume_network))
    G = torch.load(os.path.join(result_dir, resume_network.format('generator')), weights_only=False)
    D = torch.load(os.path.join(result_dir, resume_network.format('discriminator')), weights_only=False)
    return G, D


def load_trainer_state(resume_network, result_dir, optimizer_d, optimizer_g):
    """Restore the Adam moments / step counts written by SaverPlugin; returns ``resume_nimg`` for Trainer."""
    import torch
    state = torch.load(os.path.join(result_dir, resume_network.format('trainer')), weights_only=False)
    optimizer_d.load_state_dict(state['optimizer_d'])
    optimizer_g.load_state_dict(state['optimizer_g'])
    return state['cur_nimg']


class OutputGenerator(Plugin):
    """Sample-grid hook (plugins.py:177-195): every ``output_snapshot_ticks`` ticks run G on fresh latents and
    hand the fp32 ``[n,C,H,W]`` array to each postprocessor as ``proc(out, kimg)``.  A postprocessor exposing
    ``accepts_device_tensors`` (``utils.DeviceImageSaver``) gets the device tensor instead, so only the final
    uint8 grid crosses PCIe."""

    def __init__(self, sample_fn, output_postprocessors, samples_count=6, output_snapshot_ticks=3):
        super(OutputGenerator, self).__init__([(output_snapshot_ticks, 'epoch'), (1, 'end')])
        self.sample_fn = sample_fn
        self.output_postprocessors = output_postprocessors
        self.samples_count = samples_count

    def register(self, trainer):
        self.trainer = trainer

    def epoch(self, epoch_index):
        tr = self.trainer
        if tr.parallel is not None and tr.parallel.rank != 0:
            return
        gen_input = self.sample_fn(self.samples_count).cuda()
        out_dev = tr.G.forward(gen_input)
        out_host = None
        for proc in self.output_postprocessors:
            if getattr(proc, 'accepts_device_tensors', False):
                proc(out_dev, tr.cur_nimg // 1000)
            else:
                if out_host is None:
                    out_host = out_dev.cpu().numpy()
                proc(out_host, tr.cur_nimg // 1000)

    def end(self, *args):
        self.epoch(*args)
