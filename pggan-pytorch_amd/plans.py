"""Launch-plan replay of the D-step and G-step schedules: the host side of a step as a table built once per growth stage.

The schedules in ``engine`` are pure launch sequences with no host synchronisation and no data-dependent control flow, so for a
fixed (network pair, depth, minibatch, loss hyper-parameters) and alpha == 1 every step issues the SAME C-ABI calls with the same
arguments on the same streams, separated by the same event record / wait pairs -- only ~5 us of each ~25 us launch is the launch
itself, the rest is Python (shape logic, ``torch.empty``, pointer checks, stream contexts; ``tools/host_profile.py``).  A plan
records that sequence once -- while executing it for real -- as a flat list

    (CALL, ctypes function, argument tuple)          the stream handle is one of the arguments
    (RECORD, event, stream) / (WAIT, event, stream)  the fork / join points between the main and the weight-gradient stream
    (PENDING, net) / (BWD_COPIES, net)               waits whose event changes from step to step (deferred D update, side-stream
                                                     refresh of the backward-only derived weights): looked up at replay time

and replays it with one Python loop: same kernels, same streams, same events, same overlap as the eager path (unlike a hipGraph,
whose replay serialises the two streams on ROCm 7.2: 12.2 vs 11.0 ms at 1024^2, 4.7 vs 2.2 ms at 8x8, ``bench.py --graphs``).
Every tensor whose pointer enters a recorded call is kept alive by the plan, so the caching allocator can never hand its block to
somebody else (the price: the plan pins one step's worth of activations, ~10 GB at 1024^2 of 288 GB).  Inputs are copied into
static buffers, losses come back in static tensors (copies are handed out), gradients land in the networks' flat gradient buffers
exactly as in the eager path.

Not in a plan: Adam and the derived-weight refresh (host scalars / version checks: ``Trainer`` runs them eagerly after each update,
``_prologue`` re-checks the versions before every replay so that a foreign optimizer or ``load_state_dict`` is noticed), the tail of
the RCCL exchange (``GradExchange.finish``: what the sweep has not sent, and the join -- part of the update Trainer runs eagerly; the bucket
collectives the sweep itself issues ARE recorded, round 6: ``pg_allreduce_sum_f32`` is a C-ABI call like any other and the edges to the
exchange stream go through ``engine._wait_stream``), fade-in phases (a new alpha per
iteration), the mixing-factor draw (a counter-based RNG call with a new counter per step, written into the static buffer)."""
import collections

import torch

from . import _lib, engine, ops

CALL, RECORD, WAIT, PENDING, BWD_COPIES = 0, 1, 2, 3, 4


class _Plan(object):
    def __init__(self):
        self.entries = None          # the recorded list (None until recorded)
        self.keep = {}               # id -> tensor: everything whose pointer is baked into the entries
        self.static_in = None
        self.static_out = None
        self.dp_state = None         # (GradExchange snapshot, collectives, bytes) of the recorded step under data parallelism
        self.early_g = None          # engine.EarlyG the recorded D step leaves for the G step (its tensors are rewritten by every replay)
        self.warm = 0


_CACHE = collections.OrderedDict()
MAX_PLANS = 4                        # (D, G) of the current network pair + one more pair: every plan pins GBs of activations
STATS = {'recorded': 0, 'replayed': 0}


def clear():
    _CACHE.clear()


class _Recorder(object):
    """Context manager: while active every ``_lib.call`` is executed AND logged, the stream / event operations the engine issues
    are executed AND logged with events the plan owns, and every tensor whose pointer passes ``ops._p`` is kept."""

    def __init__(self, plan):
        self.plan = plan
        self.entries = []

    def __enter__(self):
        rec = self
        lib = _lib.load()
        self._saved = dict(call=_lib.call, p=ops._p, wait_stream=engine._wait_stream, wait_event=engine._wait_event,
                           record_event=engine._record_event, wait_pending=engine.wait_pending, await_bwd=engine._await_backward_copies)
        sv = self._saved

        def call(name, *args):
            fn = getattr(lib, name)
            _lib.check(fn(*args) if _lib.CALL_HOOK is None else _lib.CALL_HOOK(fn, args, name), name)
            rec.entries.append((CALL, fn, args, name))

        def p(t):
            if t is not None:
                rec.plan.keep[id(t)] = t
            return sv['p'](t)

        def wait_stream(waiter, other):            # = record an event on ``other``, make ``waiter`` wait for it (events the plan owns)
            ev = torch.cuda.Event()
            ev.record(other)
            waiter.wait_event(ev)
            rec.entries.append((RECORD, ev, other))
            rec.entries.append((WAIT, ev, waiter))

        def wait_event(stream, ev):
            stream.wait_event(ev)
            rec.entries.append((WAIT, ev, stream))

        def record_event(ev, stream):
            ev.record(stream)
            rec.entries.append((RECORD, ev, stream))

        def wait_pending(net):
            sv['wait_pending'](net)
            rec.entries.append((PENDING, net, torch.cuda.current_stream()))

        def await_bwd(net):
            sv['await_bwd'](net)
            rec.entries.append((BWD_COPIES, net, torch.cuda.current_stream()))

        # (module attributes of this package only; ``ops`` and ``engine`` reach the library through ``_lib.call`` / their own globals)
        _lib.call = call
        ops._p = p
        engine._wait_stream = wait_stream
        engine._wait_event = wait_event
        engine._record_event = record_event
        engine.wait_pending = wait_pending
        engine._await_backward_copies = await_bwd
        return self

    def __exit__(self, *exc):
        sv = self._saved
        _lib.call = sv['call']
        ops._p = sv['p']
        engine._wait_stream = sv['wait_stream']
        engine._wait_event = sv['wait_event']
        engine._record_event = sv['record_event']
        engine.wait_pending = sv['wait_pending']
        engine._await_backward_copies = sv['await_bwd']
        if exc[0] is None:
            self.plan.entries = self.entries
            STATS['recorded'] += 1
        return False


def _replay(plan):
    check = _lib.check
    hook = _lib.CALL_HOOK            # (bench.py's per-launch timing: None in production)
    for e in plan.entries:
        kind = e[0]
        if kind == CALL:
            rc = e[1](*e[2]) if hook is None else hook(e[1], e[2], e[3])
            if rc:
                check(rc, e[3])
        elif kind == RECORD:
            e[1].record(e[2])
        elif kind == WAIT:
            e[2].wait_event(e[1])
        elif kind == PENDING:
            ev = getattr(e[1], '_pending', None)
            if ev is not None:
                e[2].wait_event(ev)
                e[1]._pending = None
        else:
            with torch.cuda.stream(e[2]):             # (no-op when e[2] is the current stream, which is the usual case)
                engine._await_backward_copies(e[1])
    STATS['replayed'] += 1


def _new_plan(key):
    """A plan pins one step's worth of activations: when a network pair moves to another growth stage / minibatch, the plans of
    the stage it left are dropped (one D plan and one G plan per network pair at any time)."""
    for k in [k for k in _CACHE if k[:3] == key[:3] and k[:9] != key[:9]]:      # (the with / without early-real-third variants of a stage live side by side)
        del _CACHE[k]
    while len(_CACHE) >= MAX_PLANS:
        _CACHE.popitem(last=False)          # least recently used
    g = _CACHE[key] = _Plan()
    return g


def _prologue(*nets):
    """What the bodies would notice at their entry points: a parameter update that did not go through FusedAdam, a growth-stage
    change -- the derived weights are refreshed eagerly (never inside a plan)."""
    for net in nets:
        net._sync_version()
        net._ensure_buffers()
        if net._derived_ver != (net._param_version, int(net.depth)):
            engine._derived(net)
        engine.order_side_behind_derived(net)


def _dp_tag(net):
    """What a plan recorded under data parallelism depends on: the bucketed exchange the sweep feeds (``Trainer._open_exchange`` installs it
    BEFORE the loss call in plan mode) and whether its collectives are being left out (bench.py's exposed-exchange estimate)."""
    ex = net.__dict__.get('_grad_exchange') if net.__dict__.get('_grad_hook') is not None else None
    return (0, None) if ex is None else ((id(ex), bool(ex.dp.skip_exchange)), ex)


def _dp_begin(ex):
    return None if ex is None else (ex.dp.stats['collectives'], ex.dp.stats['bytes'])


def _dp_recorded(g, ex, before):
    if ex is not None:
        g.dp_state = (ex.snapshot(), ex.dp.stats['collectives'] - before[0], ex.dp.stats['bytes'] - before[1])


def _dp_replayed(g, ex):
    if ex is not None:
        snap, ncoll, nbytes = g.dp_state
        ex.restore(snap)
        ex.dp.stats['collectives'] += ncoll
        ex.dp.stats['bytes'] += nbytes


def d_step(D, G, real, latents, mix, lam, eps, target):
    """Plan-replayed ``d_loss_forward`` + ``d_loss_backward``.  Returns (d_cost, d_real_loss, d_fake_loss)."""
    _prologue(D, G)
    # the real third may already be through D (engine.EarlyReal, left by Trainer for exactly this batch): a plan of its own -- its body
    # starts from the batched tensors of that pass, whose addresses are stable, and never touches ``real``
    early = D.__dict__.get('_early_real')
    if early is not None and not (early.real is real and early.stamp == (D._param_version, int(D.depth), float(D.alpha))):
        early = D.__dict__.pop('_early_real') and None
        engine.EARLY_STATS['dropped'] += 1
    dp_tag, ex = _dp_tag(D)
    # the G step's generator pass rides in this step on the second stream (engine.request_early_g): its latents are a fourth static input
    req = D.__dict__.get('_early_g_request')
    if req is not None and engine.early_g_mode(req[0].depth) is None:
        req = None
    if req is None:
        D.__dict__.pop('_early_g_request', None)
    zg = engine._check_dev(req[1], 'latents') if req is not None else None
    key = ('D', D._flat_param.data_ptr(), G._flat_param.data_ptr(), int(D.depth), tuple(real.shape), tuple(latents.shape), float(lam), float(eps), float(target),
           id(early.arena) if early is not None else 0, dp_tag, (req[0]._flat_param.data_ptr(), tuple(zg.shape)) if req is not None else 0)
    g = _CACHE.get(key)
    if g is None:
        g = _new_plan(key)
        g.static_in = (torch.empty_like(real), torch.empty_like(latents), torch.empty_like(mix)) + ((torch.empty_like(zg),) if req is not None else ())
    else:
        _CACHE.move_to_end(key)
    if req is not None:
        g.static_in[3].copy_(zg)
        D._early_g_request = (req[0], g.static_in[3])
    for dst, src in zip(g.static_in, (real, latents, mix)) if early is None else zip(g.static_in[1:], (latents, mix)):
        dst.copy_(src)
    if early is not None:
        early.real = g.static_in[0]          # (what the body hands to d_loss_forward; identity is all take_early_real compares)
        if g.entries is not None:            # replay: the wait d_loss_forward would issue
            D.__dict__.pop('_early_real', None)
            engine.EARLY_STATS['used'] += 1
            if early.event is not None:
                torch.cuda.current_stream().wait_event(early.event)

    def body():
        # The plan itself never joins the weight-gradient stream: whether the caller's next launch needs the join is not known
        # here.  ``backward()`` of the returned loss does it (wgan_gp_loss._replayed_backward) unless the caller has set
        # ``D._skip_join`` by then (Trainer: the whole update follows on the weight-gradient stream, in order behind them).
        caller = getattr(D, '_skip_join', False)
        D._skip_join = True
        try:
            c, rl, fl, state = engine.d_loss_forward(D, G, g.static_in[0], g.static_in[1], g.static_in[2], lam, eps, target)
            engine.d_loss_backward(state)
        finally:
            D._skip_join = caller
        return c, rl, fl
    if D.__dict__.get('_plan_unjoined', False):     # a step whose loss never had backward() called: its weight gradients may still be in flight
        engine._join_side()
    D._plan_unjoined = True

    def early_g_handover(eg):
        # what the body (or the replay) left for the G step is keyed to the CALLER's latents tensor, not to the static copy
        D.__dict__.pop('_early_g_request', None)         # (a body without a second-stream pass does not take the request)
        if req is not None and eg is not None:
            eg.latents = zg
            eg.stamp = (req[0]._param_version, int(req[0].depth), float(req[0].alpha))
            req[0]._early_fwd = eg
    if g.entries is None:
        if g.warm < 2:                       # eager warm-up (kernel attributes, first-request derived copies, allocator pools)
            g.warm += 1
            out = body()
            early_g_handover(req[0].__dict__.get('_early_fwd') if req is not None else None)
            return out
        before = _dp_begin(ex)
        with _Recorder(g):
            g.static_out = body()
        _dp_recorded(g, ex, before)
        g.early_g = req[0].__dict__.get('_early_fwd') if req is not None else None
    else:
        _replay(g)
        _dp_replayed(g, ex)
        if g.early_g is not None:
            engine.EARLY_G_STATS['passes'] += 1
    early_g_handover(g.early_g)
    engine._assign_grads(D, engine.d_active_params(D, int(D.depth), 1.0), linear=True)
    # the static outputs are overwritten by the next replay: hand out copies (a plugin may keep loss tensors) -- one device copy when
    # they are views of one buffer (ops.d_loss)
    base = g.static_out[0]._base
    if base is not None and all(t._base is base for t in g.static_out):
        c = base.clone()
        return tuple(c.as_strided(t.shape, t.stride(), t.storage_offset()) for t in g.static_out)
    return tuple(t.clone() for t in g.static_out)


def g_step(G, D, latents):
    """Plan-replayed ``g_loss_forward`` + ``g_loss_backward``.  Returns g_cost."""
    dp_tag, ex = _dp_tag(G)
    # the generator pass may already be through (engine.EarlyG, left by the D step for exactly these latents): a plan of its own, whose
    # body starts from that pass's tensors -- stable addresses: the D step's plan rewrites them at every replay
    eg = G.__dict__.get('_early_fwd')
    if eg is not None and not (eg.latents is latents and eg.stamp == (G._param_version, int(G.depth), float(G.alpha))):
        eg = G.__dict__.pop('_early_fwd') and None
        engine.EARLY_G_STATS['dropped'] += 1
    key = ('G', D._flat_param.data_ptr(), G._flat_param.data_ptr(), int(G.depth), tuple(latents.shape), dp_tag, eg is not None)
    g = _CACHE.get(key)
    if g is None:
        g = _new_plan(key)
        g.static_in = (torch.empty_like(latents),)
    else:
        _CACHE.move_to_end(key)
        if g.entries is not None and eg is not None and g.early_g is not eg.ctx:
            # recorded against the tensors of another generator pass (the D step's plan was re-recorded): record again, now
            g.entries, g.keep, g.warm = None, {}, 2
    g.static_in[0].copy_(latents)
    if eg is not None:
        eg.latents = g.static_in[0]          # (what the body hands to g_loss_forward; identity is all take_early_g compares)
        if g.entries is not None:            # replay: the wait on the pass's event is one of the recorded entries
            G.__dict__.pop('_early_fwd', None)
            engine.EARLY_G_STATS['used'] += 1
    if getattr(D, '_pending', None) is None:  # (a deferred D update refreshes D's derived weights itself, on the second stream)
        _prologue(D)
    _prologue(G)

    def body():
        c, state = engine.g_loss_forward(G, D, g.static_in[0])
        engine.g_loss_backward(state)
        return c, state['active_g']
    if g.entries is None:
        if g.warm < 2:
            g.warm += 1
            return body()[0]
        before = _dp_begin(ex)
        g.early_g = eg.ctx if eg is not None else None     # (the D plan's replays rewrite these very tensors)
        with _Recorder(g):
            g.static_out = body()
        _dp_recorded(g, ex, before)
    else:
        _replay(g)
        _dp_replayed(g, ex)
    engine._assign_grads(G, g.static_out[1])
    return g.static_out[0].clone()
