"""Trainer: step driver + plugin dispatcher behind the reference's API (/root/reference/trainer.py:7-115).

Same public surface as the reference — ``Trainer(D, G, D_loss, G_loss, optimizer_d, optimizer_g, dataset, dataiter,
random_latents_generator, D_training_repeats=1, tick_nimg_default=2000, resume_nimg=0)``, ``register_plugin``,
``call_plugins``, ``run(total_kimg)``, ``train()``, the mutable fields plugins poke (``dataiter``,
``random_latents_generator``, ``tick_duration_nimg``, ``stats`` ...) and ``plugin_queues`` (unit -> list of
``(due, registration order, plugin)``) — with the same observable firing order, pinned by the reference trace
fixture (``tests/golden/trace16``).  The body is this project's own.

MI355X data-parallel additions (optional, default off):
  * ``parallel``: a ``parallel.DataParallel``; D's gradients (after reference trainer.py:98) and G's (after :111)
    are sum-all-reduced over RCCL on the flat gradient buffers, ``cur_nimg`` advances by ``world_size * minibatch``
    so the depth/alpha schedule stays a pure function of the images shown, and the optimizers' gradient pre-scale is
    set to 1/world_size here (a plain ``torch.optim`` optimizer gets averaged gradients from the all-reduce instead);
  * ``global_stddev=True`` (with ``parallel``): exact-global minibatch stddev -- the statistic of reference network.py:174-187 and the scalars
    of its adjoint / Hessian-vector term are reduced over all ranks, so world x minibatch equals one process at batch world * minibatch
    (default: every rank's own minibatch);
  * host inputs are moved to the device (the reference's ``.cuda()`` calls)."""
import heapq
import os

from . import engine

PLUGIN_UNITS = ('iteration', 'epoch', 's', 'end')

import torch


def _to_device(t):
    """Small host inputs (the latents of reference trainer.py:86,103: a few KB) without a host-device synchronisation: staged in
    pinned memory and copied asynchronously on the current stream (a pageable ``.cuda()`` waits for the stream to drain, which
    would take away the host's run-ahead once per iteration)."""
    if getattr(t, 'is_cuda', False):
        return t
    if not torch.cuda.is_available():
        return t.cuda()                                  # (raises the usual "no device" error)
    pin = t if t.is_pinned() else torch.empty(t.shape, dtype=t.dtype, pin_memory=True).copy_(t)
    out = torch.empty(t.shape, dtype=t.dtype, device='cuda')
    out.copy_(pin, non_blocking=True)
    _PINNED_IN_FLIGHT.append((pin, torch.cuda.Event()))
    _PINNED_IN_FLIGHT[-1][1].record()
    while len(_PINNED_IN_FLIGHT) > 8:                    # keep the staging buffers alive until their copies have run
        _PINNED_IN_FLIGHT.pop(0)[1].synchronize()
    return out


_PINNED_IN_FLIGHT = []


class _InputPrefetch(object):
    """Host -> device path of the real-image batches (reference trainer.py:92: ``next(self.dataiter).cuda()``; 37.7 MB per step
    at 1024x1024, ~0.6 ms of PCIe).  ``take`` draws the batch exactly where the reference does, stages it in pinned memory
    (nothing to do for a pinned source: DataLoader(pin_memory=True)) and uploads it on a COPY STREAM; the main stream waits for
    the upload by event.  The host enqueues a step ~2 ms ahead of the device, so the upload of iteration k + 1 runs while the
    device is still finishing iteration k and the host never blocks (a pageable ``.cuda()`` waits for the stream to drain).
    ``lookahead`` (opt-in, ``Trainer(prefetch_inputs=True)``): the batch of iteration k + 1 is already drawn at the END of
    iteration k, after the plugins have run (DepthManager has installed the loader / alpha / depth it must be drawn with) --
    for loaders that do not depend on state the caller changes between iterations; a batch of a replaced iterator is dropped.
    Device-resident batches pass through untouched."""

    def __init__(self):
        self.src = None            # iterator the pending batch was drawn from
        self.pending = None        # (device tensor, ready event) | exception raised by the iterator | None
        self.stream = None
        self.hits = self.misses = 0

    def _upload(self, t):
        if getattr(t, 'is_cuda', False):
            return (t, None)
        if not torch.cuda.is_available():
            return (_to_device(t), None)                 # (no device: the module-level hook decides -- the CPU test-suite makes it the identity)
        if self.stream is None:
            self.stream = torch.cuda.Stream()
        pin = t if t.is_pinned() else torch.empty(t.shape, dtype=t.dtype, pin_memory=True).copy_(t)
        main = torch.cuda.current_stream()
        with torch.cuda.stream(self.stream):
            dev = torch.empty(t.shape, dtype=t.dtype, device='cuda')
            dev.copy_(pin, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        dev.record_stream(main)                          # consumed on the main stream: the allocator waits for that use
        self._keep = (pin, ev)                           # (staging buffer alive until the next upload replaces it: its copy is an iteration old by then)
        return (dev, ev)

    def prefetch(self, iterator):
        """Draw and start uploading the next batch of ``iterator`` (called at the end of an iteration)."""
        self.src = iterator
        try:
            self.pending = self._upload(next(iterator))
        except Exception as e:                           # an exhausted / failing loader: raised when that batch is asked for
            self.pending = e

    def take(self, iterator):
        """The next batch of ``iterator`` on the device, ordered on the current stream."""
        if self.pending is None or self.src is not iterator:
            self.misses += 1
            self.pending = None
            dev, ev = self._upload(next(iterator))
        else:
            self.hits += 1
            got, self.pending = self.pending, None
            if isinstance(got, Exception):
                raise got
            dev, ev = got
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
        return dev


def _triggers(plugin):
    """``plugin.trigger_interval`` as a list of (period, unit) pairs (a bare pair is accepted like a one-element list)."""
    trig = plugin.trigger_interval
    return trig if isinstance(trig, list) else [trig]


def _period(plugin, unit):
    """Re-arm period of ``plugin`` on the ``unit`` queue: the LAST matching trigger wins, as in the reference's scan
    (trainer.py:63-65)."""
    found = None
    for period, u in _triggers(plugin):
        if u == unit:
            found = period
    if found is None:
        raise ValueError('plugin %r fired on %r without a trigger for it' % (plugin, unit))
    return found


def _as_tuple(losses):
    """Loss functions may return a scalar, a tuple or a list (trainer.py:105-110)."""
    if isinstance(losses, (list, tuple)):
        return tuple(losses)
    return (losses,)


class Trainer(object):

    def __init__(self, D, G, D_loss, G_loss, optimizer_d, optimizer_g, dataset, dataiter, random_latents_generator,
                 D_training_repeats=1, tick_nimg_default=2 * 1000, resume_nimg=0, parallel=None, input_transform=None,
                 prefetch_inputs=False, global_stddev=False, early_real_forward=None):
        # networks, losses, optimizers, data sources: the names are API (plugins read and replace them)
        self.D, self.G = D, G
        self.D_loss, self.G_loss = D_loss, G_loss
        self.optimizer_d, self.optimizer_g = optimizer_d, optimizer_g
        self.dataset, self.dataiter = dataset, dataiter
        self.random_latents_generator = random_latents_generator
        self.D_training_repeats = D_training_repeats
        # progress counters
        self.cur_nimg = self.tick_start_nimg = resume_nimg
        self.tick_duration_nimg = tick_nimg_default
        self.iterations = self.cur_tick = self.time = 0
        self.stats = {
            'kimg_stat': dict(val=self.cur_nimg / 1000., log_epoch_fields=['{val:8.3f}'], log_name='kimg'),
            'tick_stat': dict(val=self.cur_tick, log_epoch_fields=['{val:5}'], log_name='tick'),
        }
        self.plugin_queues = {unit: [] for unit in PLUGIN_UNITS}
        self.parallel = parallel
        # input path: host -> device upload of the real batches on a copy stream (no-op for device-resident batches);
        # ``input_transform(device_batch) -> fp32 images`` runs on the device when the batch is consumed, e.g.
        # ``lambda u8: utils.prepare_real_batch(u8, dataset.alpha)`` for uint8 sources (a quarter of the PCIe bytes)
        self._inputs = _InputPrefetch()
        self.prefetch_inputs = bool(prefetch_inputs)
        self.input_transform = input_transform
        # look-ahead form of the split D forward: the real third of the NEXT D step under this iteration's G step (engine.EarlyReal).
        # Off by default since the in-step form (engine.REAL_THIRD_IN_STEP: the real third next to the generator's forward of the same
        # step) measures the same step time without drawing a batch early; True / PGGAN_EARLY_REAL=1 turn it on
        self.early_real_forward = (os.environ.get('PGGAN_EARLY_REAL', '0') == '1') if early_real_forward is None else bool(early_real_forward)
        self._next_reals = None               # (batch, iterator it was drawn from) consumed one iteration ahead for that pass
        self._average_in_allreduce = {}
        self._exchanges = {}
        # plugins.TimeMonitor: {'every': k, 'pairs': [(start event, end event), ...]} -- every k-th iteration the last D update
        # (D loss + gradient penalty + backward + exchange + Adam(D)) is bracketed with two HIP events; read at tick boundaries
        self.d_step_probe = None
        if global_stddev and parallel is None:
            raise ValueError('global_stddev=True needs parallel= (the exact-global minibatch stddev is a data-parallel mode)')
        if parallel is not None and hasattr(D, '_flat_param'):
            # minibatch stddev under data parallelism: local-shard statistics by default, exact-global on request
            D._global_stddev = parallel if global_stddev else None
        if parallel is not None:
            from . import wgan_gp_loss
            if getattr(parallel, 'comm', None) is None:
                # the reduction goes through torch.distributed (CPU hosts / PGGAN_DP_TORCH_ALLREDUCE=1): not a C-ABI call, a launch plan
                # cannot replay it.  With the library's own RCCL communicator the bucket collectives are recorded like any other
                # launch (round 6: plans.py), and the data-parallel step is issued from a plan exactly as the single-GPU one.
                wgan_gp_loss.enable_plans(False)
            for net, opt in ((D, optimizer_d), (G, optimizer_g)):
                if hasattr(opt, 'grad_scale'):
                    opt.grad_scale = parallel.grad_scale          # FusedAdam folds 1/world into its update
                    self._average_in_allreduce[id(net)] = False
                else:                                             # e.g. torch.optim.Adam: hand it averaged gradients
                    self._average_in_allreduce[id(net)] = True

    # ---------------------------------------------------------------- plugins
    def register_plugin(self, plugin):
        plugin.register(self)
        for period, unit in _triggers(plugin):
            pending = self.plugin_queues[unit]
            pending.append((period, len(pending), plugin))       # first firing at t == period; ties by registration order

    def call_plugins(self, queue_name, time, *args):
        """Fire every plugin of the ``queue_name`` queue that is due at ``time`` (earliest due first, then registration
        order) and re-arm each at ``time + period``."""
        pending = self.plugin_queues[queue_name]
        while pending and pending[0][0] <= time:
            _, order, plugin = pending[0]
            getattr(plugin, queue_name)(time, *args)
            heapq.heapreplace(pending, (time + _period(plugin, queue_name), order, plugin))

    def _end_tick(self):
        self.cur_tick += 1
        self.tick_start_nimg = self.cur_nimg
        self.stats['kimg_stat']['val'] = self.cur_nimg / 1000.
        self.stats['tick_stat']['val'] = self.cur_tick
        self.call_plugins('epoch', self.cur_tick)

    def run(self, total_kimg=1):
        for pending in self.plugin_queues.values():
            heapq.heapify(pending)
        total_nimg = total_kimg * 1000
        while self.cur_nimg < total_nimg:
            self.train()
            # a tick closes when its image budget is used up, and always with the last iteration of the run
            if self.cur_nimg >= min(total_nimg, self.tick_start_nimg + self.tick_duration_nimg):
                self._end_tick()
        self.call_plugins('end', 1)

    # ---------------------------------------------------------------- one iteration
    def _open_exchange(self, net, layers_fn):
        """Under data parallelism, let the backward sweep that follows hand finished blocks to a bucketed all-reduce
        (``parallel.GradExchange``).  Eager launches of the product's own networks only: a captured hipGraph step has
        no hook points, and a foreign network has no flat gradient buffer."""
        if self.parallel is None or not hasattr(net, '_flat_param'):
            return
        from . import parallel as par, wgan_gp_loss
        if wgan_gp_loss._replay_mode(net) == 'graph' and float(net.alpha) >= 1.0 and net._flat_param.is_cuda:
            return
        if os.environ.get('PGGAN_DP_BUCKETS', '1') == '0':
            return
        ex = self._exchanges.get(id(net))
        if ex is None:
            ex = self._exchanges[id(net)] = par.GradExchange(self.parallel, net)
        ex.begin(layers_fn(net, int(net.depth), float(net.alpha)))
        net._grad_exchange, net._grad_hook = ex, ex.ready

    def _exchange(self, net):
        if self.parallel is not None:
            self.parallel.all_reduce_grads(net, average=self._average_in_allreduce.get(id(net), False))

    def _d_update(self):
        self._exchange(self.D)
        self.optimizer_d.step()                                                   # reference trainer.py:100
        if getattr(self.D, '_flat_param', None) is not None and self.D._flat_param.is_cuda:
            engine._derived(self.D)
            engine.probe('D.update_end')
        if self._probe_open is not None:             # end of the D+GP window, on the stream the update ran on
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            self.d_step_probe['pairs'].append((self._probe_open, end))
            del self.d_step_probe['pairs'][:-64]
            self._probe_open = None

    _probe_open = None

    def _can_overlap_d_update(self):
        """Only with the product's own G loss (it reaches D's weights through the engine, which waits for the deferred
        update) and when the step is launched eagerly (a captured hipGraph step has no second stream to overlap with)."""
        from . import wgan_gp_loss
        if os.environ.get('PGGAN_OVERLAP_D_UPDATE', '1') == '0':
            return False
        return (self.G_loss is wgan_gp_loss.wgan_gp_G_loss and hasattr(self.D, '_flat_param')
                and not wgan_gp_loss._graphs_on(self.D) and not wgan_gp_loss._graphs_on(self.G))

    def _early_g_ok(self):
        """May the generator pass that opens the G step run inside the D step (second stream, engine.request_early_g)?  The product's G loss
        (it is what picks the pass up), both networks on the device with flat buffers, two streams, no hipGraph capture of either step."""
        from . import wgan_gp_loss
        return (engine.early_g_mode(self.G.depth) is not None and engine.ASYNC_WGRAD and self.G_loss is wgan_gp_loss.wgan_gp_G_loss
                and hasattr(self.D, '_flat_param') and hasattr(self.G, '_flat_param') and self.G._flat_param.is_cuda
                and not wgan_gp_loss._graphs_on(self.D) and not wgan_gp_loss._graphs_on(self.G))

    def _draw_reals(self):
        reals = self._inputs.take(self.dataiter)                                  # :92
        if self.input_transform is not None:
            reals = self.input_transform(reals)
        return reals

    def _early_real_ok(self):
        """May the real third of the NEXT iteration's D forward run now?  Only when that iteration is known to use this stage's networks
        unchanged: alpha == 1 now and after the image counter's next value (every plugin with a ``schedule`` -- DepthManager -- is asked),
        one D update per iteration, the product's own networks and losses, eager or plan-replayed steps (not the captured 4x4 stage), no
        statistic exchange inside the pass."""
        from . import wgan_gp_loss
        D = self.D
        if not (self.early_real_forward and self.D_training_repeats == 1 and hasattr(D, '_flat_param') and D._flat_param.is_cuda
                and self.D_loss is wgan_gp_loss.wgan_gp_D_loss and engine.ASYNC_WGRAD):
            return False
        if float(D.alpha) < 1.0 or getattr(D, 'pixelnorm', False) or D.__dict__.get('_global_stddev') is not None:
            return False
        if wgan_gp_loss._replay_mode(D) == 'graph':
            return False
        for unit in self.plugin_queues.values():
            for _, _, plugin in unit:
                sched = getattr(plugin, 'schedule', None)
                if sched is not None:
                    depth, alpha = sched(self.cur_nimg)
                    if depth != int(D.depth) or alpha < 1.0:
                        return False
        return True

    def train(self):
        """One iteration: ``D_training_repeats`` discriminator updates, then one generator update
        (reference trainer.py:85-115; line numbers below refer to it)."""
        self._train_iteration()

    def _train_iteration(self):
        world = 1 if self.parallel is None else self.parallel.world_size
        latents = _to_device(self.random_latents_generator())                     # :86
        d_losses = (0, 0, 0)
        for rep in range(self.D_training_repeats):                                # :90
            ahead, self._next_reals = self._next_reals, None
            if ahead is not None and ahead[1] is self.dataiter:
                reals = ahead[0]                                                  # drawn (in order) at the end of the previous iteration
            else:
                if ahead is not None:                                             # (the iterator was replaced in between: that batch is not this stage's)
                    self.D.__dict__.pop('_early_real', None)
                reals = self._draw_reals()
            self.cur_nimg += reals.size(0) * world                                # :93 (global images)
            last = rep == self.D_training_repeats - 1
            probe = self.d_step_probe
            if probe and last and self.iterations % probe['every'] == 0 and getattr(reals, 'is_cuda', False):
                self._probe_open = torch.cuda.Event(enable_timing=True)
                self._probe_open.record()
            # (the exchange is opened BEFORE the loss call: a plan-replayed loss has its backward sweep -- and the bucket collectives the
            #  sweep feeds -- inside that call; an eager forward never reports a finished block)
            self._open_exchange(self.D, engine.d_exchange_layers)
            z_g = None
            if last and self._early_g_ok():
                # the latents of the G step, drawn now (the same position in the generator's sequence as trainer.py:103: nothing else is
                # drawn in between) so that G(z_g) can run on the second stream inside the D step (engine.request_early_g)
                z_g = _to_device(self.random_latents_generator())                 # :103
                if getattr(z_g, 'is_cuda', False):
                    engine.request_early_g(self.D, self.G, z_g)
            try:
                d_losses = _as_tuple(self.D_loss(self.D, self.G, reals, latents)) # :95
            except BaseException:
                self.D._grad_hook = None
                raise
            finally:
                self.D.__dict__.pop('_early_g_request', None)                     # (a D loss that never reached the engine's second-stream pass)
            defer = last and self._can_overlap_d_update()
            self.D._skip_join = defer                # the update runs on the second stream, behind the weight gradients
            try:
                d_losses[0].backward()                                            # :98
            finally:
                self.D._skip_join = False
                self.D._grad_hook = None
            if defer:
                # the tail of the last D update (all-reduce, Adam, derived weights) runs on the second stream under the
                # generator forward that opens the G step; the main stream re-joins at its first use of D (engine.wait_pending)
                engine.defer_to_side(self.D, self._d_update)
                if self._early_real_ok():
                    # ... and behind it the real third of the NEXT D step's forward (D's weights are final now; the G step that follows
                    # is 3-image, latency-bound work on the main stream)
                    nxt = self._draw_reals()
                    if getattr(nxt, 'is_cuda', False):
                        self._next_reals = (nxt, self.dataiter)
                        engine.early_real_on_side(self.D, nxt)
                    else:
                        self._next_reals = (nxt, self.dataiter)
            else:
                engine._join_side()                  # (a replayed plan leaves the weight gradients un-joined; a second join is free)
                self._d_update()
            latents = z_g if z_g is not None else _to_device(self.random_latents_generator())     # :103
        self._open_exchange(self.G, engine.g_exchange_layers)
        try:
            g_losses = _as_tuple(self.G_loss(self.G, self.D, latents))            # :105-110
            g_losses[0].backward()                                                # :111
        finally:
            self.G._grad_hook = None
        self._exchange(self.G)
        self.optimizer_g.step()                                                   # :112
        if getattr(self.G, '_flat_param', None) is not None and self.G._flat_param.is_cuda:
            engine._derived(self.G)                  # (the next generator pass needs them first thing; never part of a replayed plan)
            engine.probe('G.update_end')
        engine.wait_pending(self.D)                  # (a G_loss that never ran D: nothing may outlive the iteration)
        if getattr(self.G, '__dict__', {}).pop('_early_fwd', None) is not None:   # (... nor a generator pass nobody took)
            engine.EARLY_G_STATS['dropped'] += 1
        self.iterations += 1
        self.call_plugins('iteration', self.iterations, *(g_losses + d_losses))   # :115
        if self.prefetch_inputs and self._inputs.stream is not None and self.dataiter is not None:
            # host-resident batches, look-ahead on: the next one starts travelling now, with the loader / alpha / depth the plugins just installed
            self._inputs.prefetch(self.dataiter)
