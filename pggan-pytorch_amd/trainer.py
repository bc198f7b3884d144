"""Trainer: the reference's step driver and plugin queues (/root/reference/trainer.py), unchanged in
API — ``Trainer(D, G, D_loss, G_loss, optimizer_d, optimizer_g, dataset, dataiter,
random_latents_generator, D_training_repeats=1, tick_nimg_default=2000, resume_nimg=0)``,
``register_plugin``, ``run(total_kimg)``, ``train()`` and the public mutable fields.

Additions for the MI355X data-parallel path (both optional, default off):
  * ``parallel``: a ``parallel.DataParallel`` helper; gradients of D (after trainer.py:98) and of G
    (after trainer.py:111) are sum-all-reduced over RCCL on the networks' flat gradient buffers, and
    ``cur_nimg`` advances by ``world_size * minibatch`` so the depth/alpha schedule stays a pure
    function of the number of images shown;
  * inputs that are not yet on the device are moved there (the reference's ``.cuda()`` calls)."""
import heapq
import os

from . import engine


def _to_device(t):
    return t if getattr(t, 'is_cuda', False) else t.cuda()


class Trainer(object):

    def __init__(self,
                 D,
                 G,
                 D_loss,
                 G_loss,
                 optimizer_d,
                 optimizer_g,
                 dataset,
                 dataiter,
                 random_latents_generator,
                 D_training_repeats=1,  # trainer
                 tick_nimg_default=2 * 1000,  # trainer
                 resume_nimg=0,
                 parallel=None):
        self.D = D
        self.G = G
        self.D_loss = D_loss
        self.G_loss = G_loss
        self.D_training_repeats = D_training_repeats
        self.optimizer_d = optimizer_d
        self.optimizer_g = optimizer_g
        self.dataiter = dataiter
        self.dataset = dataset
        self.cur_nimg = resume_nimg
        self.random_latents_generator = random_latents_generator
        self.tick_start_nimg = self.cur_nimg
        self.tick_duration_nimg = tick_nimg_default
        self.iterations = 0
        self.cur_tick = 0
        self.time = 0
        self.parallel = parallel
        self.stats = {
            'kimg_stat': {'val': self.cur_nimg / 1000., 'log_epoch_fields': ['{val:8.3f}'], 'log_name': 'kimg'},
            'tick_stat': {'val': self.cur_tick, 'log_epoch_fields': ['{val:5}'], 'log_name': 'tick'}
        }
        self.plugin_queues = {
            'iteration': [],
            'epoch': [],
            's': [],
            'end': []
        }

    def register_plugin(self, plugin):
        plugin.register(self)
        intervals = plugin.trigger_interval
        if not isinstance(intervals, list):
            intervals = [intervals]
        for (duration, unit) in intervals:
            queue = self.plugin_queues[unit]
            queue.append((duration, len(queue), plugin))

    def call_plugins(self, queue_name, time, *args):
        args = (time,) + args
        queue = self.plugin_queues[queue_name]
        if len(queue) == 0:
            return
        while queue[0][0] <= time:
            plugin = queue[0][2]
            getattr(plugin, queue_name)(*args)
            for trigger in plugin.trigger_interval:
                if trigger[1] == queue_name:
                    interval = trigger[0]
            new_item = (time + interval, queue[0][1], plugin)
            heapq.heappushpop(queue, new_item)

    def run(self, total_kimg=1):
        for q in self.plugin_queues.values():
            heapq.heapify(q)
        while self.cur_nimg < total_kimg * 1000:
            self.train()
            if self.cur_nimg >= self.tick_start_nimg + self.tick_duration_nimg or self.cur_nimg >= total_kimg * 1000:
                self.cur_tick += 1
                self.tick_start_nimg = self.cur_nimg
                self.stats['kimg_stat']['val'] = self.cur_nimg / 1000.
                self.stats['tick_stat']['val'] = self.cur_tick
                self.call_plugins('epoch', self.cur_tick)
        self.call_plugins('end', 1)

    def _d_update(self):
        if self.parallel is not None:
            self.parallel.all_reduce_grads(self.D)
        self.optimizer_d.step()                                                   # reference trainer.py:100
        if getattr(self.D, '_flat_param', None) is not None and self.D._flat_param.is_cuda:
            engine._derived(self.D)

    def _can_overlap_d_update(self):
        """Only with the product's own G loss (it reaches D's weights through the engine, which waits for the deferred
        update) and when the step is launched eagerly (a captured hipGraph step has no second stream to overlap with)."""
        from . import wgan_gp_loss
        if os.environ.get('PGGAN_OVERLAP_D_UPDATE', '1') == '0':
            return False
        return (self.G_loss is wgan_gp_loss.wgan_gp_G_loss and hasattr(self.D, '_flat_param')
                and not wgan_gp_loss._graphs_on(self.D) and not wgan_gp_loss._graphs_on(self.G))

    def train(self):
        """One iteration.  reference trainer.py:85-115."""
        world = 1 if self.parallel is None else self.parallel.world_size
        fake_latents_in = _to_device(self.random_latents_generator())            # :86
        d_losses = [0, 0, 0]
        for i in range(self.D_training_repeats):                                  # :90
            real_images_expr = _to_device(next(self.dataiter))                    # :92
            self.cur_nimg += real_images_expr.size(0) * world                     # :93 (global images)
            d_losses = self.D_loss(self.D, self.G, real_images_expr, fake_latents_in)   # :95
            d_losses = tuple(d_losses)
            D_loss = d_losses[0]
            defer = i == self.D_training_repeats - 1 and self._can_overlap_d_update()
            self.D._skip_join = defer                # the update runs on the second stream, behind the weight gradients
            try:
                D_loss.backward()                                                 # :98
            finally:
                self.D._skip_join = False
            if defer:
                # the tail of the last D update (all-reduce, Adam, derived weights) runs on the second stream under the
                # generator forward that opens the G step; the main stream re-joins at its first use of D (engine.wait_pending)
                engine.defer_to_side(self.D, self._d_update)
            else:
                self._d_update()
            fake_latents_in = _to_device(self.random_latents_generator())         # :103
        g_losses = self.G_loss(self.G, self.D, fake_latents_in)                   # :105
        if type(g_losses) is list:
            g_losses = tuple(g_losses)
        elif type(g_losses) is not tuple:
            g_losses = (g_losses,)
        G_loss = g_losses[0]
        G_loss.backward()                                                         # :111
        if self.parallel is not None:
            self.parallel.all_reduce_grads(self.G)
        self.optimizer_g.step()                                                   # :112
        engine.wait_pending(self.D)                  # (a G_loss that never ran D: nothing may outlive the iteration)
        self.iterations += 1
        self.call_plugins('iteration', self.iterations, *(g_losses + d_losses))   # :115
