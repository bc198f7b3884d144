"""WGAN-GP losses with the reference's call signatures (/root/reference/wgan_gp_loss.py).

``wgan_gp_D_loss(D, G, real, latents, ...) -> (D_cost, D_real_loss, D_fake_loss)`` and
``wgan_gp_G_loss(G, D, latents) -> G_cost`` return device tensors; ``D_cost.backward()`` /
``G_cost.backward()`` (called by ``Trainer.train``, trainer.py:98,111) run the explicit HIP
backward schedules of ``engine`` and leave the gradients in ``param.grad`` (views of each
network's flat gradient buffer), exactly where ``optimizer.step()`` expects them."""
import torch

from . import engine

# wgan_gp_loss.py:4-5 keeps module-global scratch; the only state kept here is the injectable RNG.
mixing_factors = None
_seed = None                # [seed, draws so far, torch seed it came from] of the mixing-factor stream; None: seeded from torch's RNG at first use
_use_graphs = 'auto'        # 'auto': launch plans (hipGraphs at the 4x4 stage only with GRAPH_STAGE0); True: hipGraphs everywhere; False: eager


def enable_graphs(flag=True):
    """How the D-step / G-step schedules are issued whenever alpha == 1:
    ``'auto'`` (default): every stage is replayed from a recorded LAUNCH PLAN (plans.py: the same kernels on the same streams with the
    same events as the eager path, minus the Python between them; ``enable_plans(False)`` turns that off) -- the 4x4 stage included since
    round 6 (``GRAPH_STAGE0``: it used to replay captured hipGraphs, graphs.py);
    ``True``: hipGraph replay everywhere (measured slower from 8x8 on: the replay serialises the weight-gradient stream);
    ``False``: eager launches (per-launch instrumentation, debugging)."""
    global _use_graphs
    _use_graphs = 'auto' if flag == 'auto' else bool(flag)
    if not flag:
        from . import graphs, plans
        graphs.clear()
        plans.clear()


def enable_plans(flag=True):
    """Launch-plan replay of the stages above 4x4 (see ``enable_graphs``).  Under data parallelism the bucket collectives of the
    sweep are part of the plan (the library's RCCL communicator); Trainer turns plans off only when the reduction goes through
    torch.distributed (CPU hosts)."""
    global _use_plans
    _use_plans = bool(flag)
    if not flag:
        from . import plans
        plans.clear()


_use_plans = __import__('os').environ.get('PGGAN_PLANS', '1') != '0'
# The 4x4 stage used to replay captured hipGraphs in 'auto' mode (round 3: 1.44x the eager launches).  Since the launch plans exist the plan
# is the faster form there as well (round 6, 4x4 stage, minibatch 16, ms per step graph | plan: 1.121 | 0.924; with a one-rank RCCL
# communicator 4.09 | 1.09 -- a graph replay serialises the streams and cannot carry the bucket collectives).  PGGAN_GRAPH_STAGE0=1 restores
# the graph at that stage; ``enable_graphs(True)`` still captures every stage.
GRAPH_STAGE0 = __import__('os').environ.get('PGGAN_GRAPH_STAGE0', '0') == '1'


def _replay_mode(net):
    """'graph' | 'plan' | None (eager) for a step of ``net`` at its current growth stage (alpha == 1 is checked by the callers)."""
    if _use_graphs is False:
        return None
    if _use_graphs is True or (int(net.depth) == 0 and GRAPH_STAGE0):
        return 'graph'
    return 'plan' if _use_plans else None


def _exchange_instrumented(net):
    """bench.py times every collective with a fresh HIP event pair (``DataParallel.record_events``): those steps are issued eagerly."""
    ex = net.__dict__.get('_grad_exchange') if net.__dict__.get('_grad_hook') is not None else None
    return ex is not None and ex.dp.record_events


def _graphs_on(net):
    return _replay_mode(net) == 'graph'


def set_mixing_factors(m):
    """Inject the U[0,1) draw of wgan_gp_loss.py:15-17 for the NEXT wgan_gp_D_loss call (parity tests)."""
    global mixing_factors
    mixing_factors = m


def manual_seed(seed, device='cuda'):
    """Seed the stream the mixing factors are drawn from (counter-based: ``pg_uniform_f32(seed, draw number, element)``).
    Without this call the stream follows ``torch.manual_seed``: it is (re)seeded from ``torch.initial_seed()`` at first use and
    again whenever that value has changed since (two same-seed runs in one process draw the same factors)."""
    global _seed
    _seed = [int(seed), 0, None]


def _draw_mixing_factors(n, device):
    """U[0,1) [n, 1] on the device (wgan_gp_loss.py:15-17) by the library's own generator: no ATen RNG launch inside the step."""
    global _seed
    ts = int(torch.initial_seed())
    if _seed is None or (_seed[2] is not None and _seed[2] != ts):   # [seed, draws so far, torch seed it was derived from (None: manual_seed)]
        _seed = [ts, 0, ts]
    mix = torch.empty((n, 1), device=device, dtype=torch.float32)
    if mix.is_cuda:
        from . import ops
        ops.uniform_(mix, _seed[0], _seed[1])
    else:                                                                # (host emulation in the CPU test-suite only)
        mix.copy_(torch.rand((n, 1), generator=torch.Generator().manual_seed((_seed[0] * 1000003 + _seed[1]) % (1 << 63))))
    _seed[1] += 1
    return mix


class LossTensor(torch.Tensor):
    """A scalar device tensor whose ``backward()`` launches an explicit HIP backward schedule."""

    @staticmethod
    def wrap(t, backward_fn):
        out = torch.Tensor._make_subclass(LossTensor, t)
        out._pg_backward = backward_fn
        return out

    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        if create_graph:
            raise NotImplementedError('third-order differentiation is not part of the path')
        fn = getattr(self, '_pg_backward', None)
        if fn is None:
            raise RuntimeError('backward() was already consumed (or this tensor is a derived copy)')
        scale = 1.0 if gradient is None else float(gradient)
        fn(scale)
        if not retain_graph:
            self._pg_backward = None

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        # derived tensors (``.mean()``, ``.item()`` ...) are plain tensors
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **(kwargs or {}))


def _replayed_backward(net):
    """``backward()`` of a loss whose step was replayed (hipGraph or launch plan): the gradient launches are already enqueued.  A
    replayed plan leaves the weight-gradient stream un-joined; the join happens HERE, so that ``loss.backward(); optimizer.step()``
    keeps the contract of the eager path (engine.d_loss_backward: gradients complete for whatever the caller does next on this
    stream) -- unless the caller has set ``net._skip_join`` (Trainer: the update follows on the weight-gradient stream itself).
    A ``gradient`` other than 1 rescales the flat gradient buffer afterwards, as the eager path does."""
    def fn(scale):
        if not (getattr(net, '_skip_join', False) and scale == 1.0):
            engine._join_side()
        net._plan_unjoined = False           # joined, or the caller took over (its update is ordered behind the weight-gradient stream)
        if scale != 1.0:
            from . import ops
            ops.axpby_mask(net._flat_grad, a=scale, out=net._flat_grad)
    return fn


def wgan_gp_D_loss(D, G, real_images_in, fake_latents_in,
                   iwass_lambda=10.0,
                   iwass_epsilon=0.001,
                   iwass_target=1.0,
                   return_all=True):
    """reference wgan_gp_loss.py:36-65."""
    global mixing_factors
    D.zero_grad()                                                        # :42
    G.zero_grad()                                                        # :43
    n = real_images_in.size(0)
    if mixing_factors is not None:
        mix = mixing_factors.to(device=real_images_in.device, dtype=torch.float32).reshape(n, 1)
        mixing_factors = None
    else:                                                                # :15-17 (device RNG)
        mix = _draw_mixing_factors(n, real_images_in.device)
    mode = _replay_mode(D) if (float(D.alpha) >= 1.0 and real_images_in.is_cuda and hasattr(D, '_flat_param')
                               and D.__dict__.get('_global_stddev') is None) else None      # (exact-global stddev: collectives inside the step -> eager)
    if mode == 'plan' and _exchange_instrumented(D):
        mode = None
    if mode is not None:
        from . import graphs, plans
        real_c = engine._check_dev(real_images_in, 'real images')
        z_c = engine._check_dev(fake_latents_in, 'latents')
        d_cost, d_real_loss, d_fake_loss = (graphs if mode == 'graph' else plans).d_step(D, G, real_c, z_c, mix.contiguous(), iwass_lambda,
                                                                                         iwass_epsilon, iwass_target)
        d_cost = LossTensor.wrap(d_cost, _replayed_backward(D))  # gradients are already in param.grad
        if return_all:
            return d_cost, d_real_loss, d_fake_loss
        return d_cost
    d_cost, d_real_loss, d_fake_loss, state = engine.d_loss_forward(
        D, G, real_images_in, fake_latents_in, mix, iwass_lambda, iwass_epsilon, iwass_target)
    d_cost = LossTensor.wrap(d_cost, lambda scale: engine.d_loss_backward(state, scale))
    if return_all:
        return d_cost, d_real_loss, d_fake_loss
    return d_cost


def wgan_gp_G_loss(G, D, fake_latents_in):
    """reference wgan_gp_loss.py:68-74."""
    G.zero_grad()                                                        # :69
    mode = _replay_mode(G) if (float(G.alpha) >= 1.0 and fake_latents_in.is_cuda and hasattr(G, '_flat_param')
                               and D.__dict__.get('_global_stddev') is None) else None
    if mode == 'plan' and _exchange_instrumented(G):
        mode = None
    if mode is not None:
        from . import graphs, plans
        g_cost = (graphs if mode == 'graph' else plans).g_step(G, D, engine._check_dev(fake_latents_in, 'latents'))
        return LossTensor.wrap(g_cost, _replayed_backward(G))
    g_cost, state = engine.g_loss_forward(G, D, fake_latents_in)
    return LossTensor.wrap(g_cost, lambda scale: engine.g_loss_backward(state, scale))
