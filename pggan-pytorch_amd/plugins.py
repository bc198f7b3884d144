"""Trainer plugins of the hot path: DepthManager and LRScheduler (/root/reference/plugins.py:13-99).

``torch.utils.trainer.plugins`` — the base classes the reference imports (plugins.py:8-9) — was removed
from PyTorch after 0.3, so the minimal ``Plugin`` protocol is restated here: an object with
``trigger_interval = [(n, unit), ...]``, ``register(trainer)`` and one method per unit
(``iteration`` / ``epoch`` / ``s`` / ``end``) — exactly what ``Trainer.call_plugins`` relies on.
All schedule arithmetic is Python int (+ one IEEE double division for alpha): bit-exact by construction."""
import time
from datetime import timedelta


class Plugin(object):
    def __init__(self, interval=None):
        if interval is None:
            interval = []
        self.trigger_interval = interval

    def register(self, trainer):
        raise NotImplementedError


class DepthManager(Plugin):
    """reference plugins.py:13-81 (same constructor, same stats keys)."""

    def __init__(self,
                 create_dataloader_fun,
                 create_rlg,
                 max_depth,
                 minibatch_default=16,
                 minibatch_overrides={6: 14, 7: 6, 8: 3},
                 tick_kimg_default=20,
                 tick_kimg_overrides={3: 10, 4: 10, 5: 5, 6: 2, 7: 2, 8: 1},
                 lod_training_nimg=100 * 1000,
                 lod_transition_nimg=100 * 1000,
                 max_lod=None,
                 depth_offset=None):
        super(DepthManager, self).__init__([(1, 'iteration')])
        self.minibatch_default = minibatch_default
        self.minibatch_overrides = minibatch_overrides
        self.tick_kimg_default = tick_kimg_default
        self.tick_kimg_overrides = tick_kimg_overrides
        self.create_dataloader_fun = create_dataloader_fun
        self.create_rlg = create_rlg
        self.lod_training_nimg = lod_training_nimg
        self.lod_transition_nimg = lod_transition_nimg
        self.trainer = None
        self.depth = -1
        self.alpha = -1
        self.max_depth = max_depth
        self.max_lod = max_lod
        self.depth_offset = depth_offset

    def register(self, trainer):
        self.trainer = trainer
        self.trainer.stats['minibatch_size'] = self.minibatch_default
        self.trainer.stats['alpha'] = {'log_name': 'alpha', 'log_epoch_fields': ['{val:.2f}'], 'val': self.alpha}
        if self.max_lod is not None and self.depth_offset is not None:
            self.trainer.stats['lod'] = {'log_name': 'lod', 'log_epoch_fields': ['{val:.2f}'], 'val': self.lod}
        self.iteration()

    @property
    def lod(self):
        if self.max_lod is not None and self.depth_offset is not None:
            return self.max_lod - self.depth_offset - self.depth - self.alpha + 1
        return -1

    def schedule(self, cur_nimg):
        """(depth, alpha) as a pure function of cur_nimg.  plugins.py:58-63."""
        full_passes, remaining_nimg = divmod(cur_nimg, self.lod_training_nimg + self.lod_transition_nimg)
        train_passes_rem, remaining_nimg = divmod(remaining_nimg, self.lod_training_nimg)
        depth = min(self.max_depth, full_passes + train_passes_rem)
        alpha = remaining_nimg / self.lod_transition_nimg \
            if train_passes_rem > 0 and full_passes + train_passes_rem == depth else 1.0
        return depth, alpha

    def iteration(self, *args):
        depth, alpha = self.schedule(self.trainer.cur_nimg)
        dataset = self.trainer.dataset
        if depth != self.depth:                                                # plugins.py:65-74
            self.trainer.D.depth = self.trainer.G.depth = dataset.model_depth = depth
            self.depth = depth
            minibatch_size = self.minibatch_overrides.get(depth, self.minibatch_default)
            self.trainer.dataiter = iter(self.create_dataloader_fun(minibatch_size))
            self.trainer.random_latents_generator = self.create_rlg(minibatch_size)
            tick_duration_kimg = self.tick_kimg_overrides.get(depth, self.tick_kimg_default)
            self.trainer.tick_duration_nimg = tick_duration_kimg * 1000
            self.trainer.stats['minibatch_size'] = minibatch_size
        if alpha != self.alpha:                                                # plugins.py:75-77
            self.trainer.D.alpha = self.trainer.G.alpha = dataset.alpha = alpha
            self.alpha = alpha
        self.trainer.stats['depth'] = depth
        self.trainer.stats['alpha']['val'] = alpha
        if self.max_lod is not None and self.depth_offset is not None:
            self.trainer.stats['lod']['val'] = self.lod


class LRScheduler(Plugin):
    """reference plugins.py:84-99."""

    def __init__(self, lr_scheduler_d, lr_scheduler_g):
        super(LRScheduler, self).__init__([(1, 'iteration')])
        self.lrs_d = lr_scheduler_d
        self.lrs_g = lr_scheduler_g

    def register(self, trainer):
        self.trainer = trainer
        self.iteration()

    def iteration(self, *args):
        self.lrs_d.step(self.trainer.cur_nimg)
        self.lrs_g.step(self.trainer.cur_nimg)


class RampupLR(object):
    """``LambdaLR(opt, rampup).step(cur_nimg)`` without the scheduler machinery: sets
    ``lr = base_lr * fn(cur_nimg)`` on every param group (what plugins.py:97-99 drives)."""

    def __init__(self, optimizer, fn):
        self.optimizer, self.fn = optimizer, fn
        self.base_lrs = [g['lr'] for g in optimizer.param_groups]

    def step(self, cur_nimg):
        for g, base in zip(self.optimizer.param_groups, self.base_lrs):
            g['lr'] = base * self.fn(cur_nimg)


class ThroughputMonitor(Plugin):
    """Adds ``img/s`` per tick next to the reference's ``sec.tick`` / ``sec.kimg`` (plugins.py:114-139)."""

    def __init__(self, base_time=0):
        super(ThroughputMonitor, self).__init__([(1, 'epoch')])
        self.base_time = base_time

    def register(self, trainer):
        self.trainer = trainer
        self.start_time = self.epoch_start = time.time()
        self.start_nimg = trainer.cur_nimg
        self.trainer.stats['sec'] = {'log_format': ':.1f'}

    def epoch(self, epoch_index):
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        cur = time.time()
        tick_time = cur - self.epoch_start
        self.epoch_start = cur
        nimg = max(1, self.trainer.cur_nimg - self.start_nimg)
        self.start_nimg = self.trainer.cur_nimg
        self.trainer.stats['time'] = timedelta(seconds=cur - self.start_time + self.base_time)
        self.trainer.stats['sec']['tick'] = tick_time
        self.trainer.stats['sec']['kimg'] = tick_time / nimg * 1000
        self.trainer.stats['img/s'] = nimg / tick_time
