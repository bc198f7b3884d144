"""hipGraph replay of the D-step and G-step launch schedules.

The schedules in ``engine`` are pure launch sequences on one stream with no host synchronisation, so
for a fixed (network pair, depth, minibatch, loss hyper-parameters) and alpha == 1 they are captured
once into a hipGraph (through ``torch.cuda.CUDAGraph``, i.e. hipStreamBeginCapture / hipGraphLaunch)
and replayed: ~200 kernel launches collapse into one graph launch, which is what makes the 4x4 ...
32x32 growth stages GPU-bound instead of host-bound.  Inputs are copied into static buffers
(device-to-device), the losses come back in static tensors, gradients land in the networks' flat
gradient buffers exactly as in the eager path.

Not captured: Adam (its bias corrections / learning rate are host scalars that change every step)
and the RCCL all-reduce.  Fade-in phases (alpha < 1, a new alpha every iteration) run eagerly.
Enable with ``wgan_gp_loss.enable_graphs(True)``."""
import torch

from . import engine, ops


class _Graphed(object):
    def __init__(self):
        self.graph = None
        self.static_in = None
        self.static_out = None
        self.keep = None
        self.warm = 0


_CACHE = {}


def clear():
    _CACHE.clear()
    ops.release_capture_workspaces()          # the capture streams are gone with the graphs


def _force_repack(net, layers):
    """The derived backward-data weights must be re-packed on EVERY replay: make them stale so that the
    pack kernels are part of the captured sequence."""
    net._derived_ver = None


def _d_active_conv(D, depth):
    return [m for m in engine.d_active_params(D, depth, 1.0) if m.kind == 'conv']


def _g_active_conv(G, depth):
    ls = [G.block0.c1, G.block0.c2]
    for i in range(depth):
        ls += [G.blocks[i].c1, G.blocks[i].c2]
    return ls


def d_step(D, G, real, latents, mix, lam, eps, target):
    """Graphed ``d_loss_forward`` + ``d_loss_backward``.  Returns (d_cost, d_real_loss, d_fake_loss)."""
    key = ('D', D._flat_param.data_ptr(), G._flat_param.data_ptr(), int(D.depth), tuple(real.shape), tuple(latents.shape), float(lam), float(eps), float(target))
    g = _CACHE.get(key)
    if g is None:
        g = _CACHE[key] = _Graphed()
        g.static_in = (torch.empty_like(real), torch.empty_like(latents), torch.empty_like(mix))
    for dst, src in zip(g.static_in, (real, latents, mix)):
        dst.copy_(src)
    if g.graph is None:
        def body():
            c, rl, fl, state = engine.d_loss_forward(D, G, g.static_in[0], g.static_in[1], g.static_in[2], lam, eps, target)
            engine.d_loss_backward(state)
            return c, rl, fl
        if g.warm < 2:                       # eager warm-up (sets kernel attributes, fills caches, allocator pools)
            g.warm += 1
            return body()
        torch.cuda.synchronize()
        _force_repack(D, _d_active_conv(D, int(D.depth)))
        _force_repack(G, _g_active_conv(G, int(G.depth)))     # the captured body runs G(z) on G's derived (Winograd) weights
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            g.static_out = body()
        g.graph = graph
        # the three-pass forward's activations live in the network's single-slot arena (engine._three_pass_buffers), allocated during the
        # eager warm-up, i.e. OUTSIDE the graph's private pool: the graph bakes their addresses, so it keeps them alive -- a D step of
        # another batch shape at this depth replaces D._early_buffers, and a replay of this graph must not touch freed memory
        g.keep = D.__dict__.get('_early_buffers')
    g.graph.replay()
    engine._assign_grads(D, engine.d_active_params(D, int(D.depth), 1.0), linear=True)
    # the static outputs are overwritten by the next replay: hand out copies (a plugin may keep loss tensors)
    return tuple(t.clone() for t in g.static_out)


def g_step(G, D, latents):
    """Graphed ``g_loss_forward`` + ``g_loss_backward``.  Returns g_cost."""
    key = ('G', D._flat_param.data_ptr(), G._flat_param.data_ptr(), int(G.depth), tuple(latents.shape))
    g = _CACHE.get(key)
    if g is None:
        g = _CACHE[key] = _Graphed()
        g.static_in = (torch.empty_like(latents),)
    g.static_in[0].copy_(latents)
    if g.graph is None:
        def body():
            c, state = engine.g_loss_forward(G, D, g.static_in[0])
            engine.g_loss_backward(state)
            return c, state['active_g']
        if g.warm < 2:
            g.warm += 1
            return body()[0]
        torch.cuda.synchronize()
        _force_repack(D, _d_active_conv(D, int(D.depth)))
        _force_repack(G, _g_active_conv(G, int(G.depth)))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            g.static_out = body()
        g.graph = graph
    g.graph.replay()
    engine._assign_grads(G, g.static_out[1])
    return g.static_out[0].clone()
