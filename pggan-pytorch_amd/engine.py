"""Explicit launch schedules of the PGGAN hot path (no autograd graph, no tracing compiler).

Every pass below is a hand-ordered sequence of the C-ABI kernels (``ops``):

  generator_forward / generator_backward            reference network.py:118-139 + its autograd
  discriminator forward (batched [real|fake|mixed])  reference network.py:225-240
  d_backward            first-order adjoint sweep (data-only or data+weights)
  d_tangent_wgrad       forward-mode (tangent) sweep of the gradient-penalty second-order term

The WGAN-GP double backward (wgan_gp_loss.py:25-31 + trainer.py:98) is NOT done with a generic
autograd.  With default flags D is piecewise linear except for minibatch-stddev, so

    d/dtheta  sum_i gp_i / N   =   d/dtheta  < u , grad_x S(x^, theta) >        (u held constant)
                               =   grad_theta of the directional derivative of S along u

which is evaluated as (i) a tangent pass pushing u through the same masked linear maps,
(ii) weight gradients  tangent_input (x) first-backward-adjoint  per layer, and (iii) the
minibatch-stddev Hessian-vector term injected into the ordinary batched backward of the mixed
samples (SURVEY.md §7 "hard parts"; formulas verified against autograd in tests/).
"""
import os
import weakref as _weakref

import torch

from . import ops


# ------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------
def _check_dev(t, what):
    if not torch.is_tensor(t) or not t.is_cuda:
        raise RuntimeError('%s must be a tensor on the MI355X device (no CPU path exists); got %s'
                           % (what, getattr(t, 'device', type(t))))
    if t.dtype != torch.float32:
        raise RuntimeError('%s must be float32' % what)
    return t.contiguous()


# Winograd F(2x2,3x3) (csrc/conv_wino.hip) replaces the direct implicit GEMM where it measured faster (tools/sweeps/sweep_wino.py):
# 3x3 layers with a full 16-cout MFMA tile and enough 64-tile workgroups to fill the chip.  With the second-generation kernel
# (8-channel chunks) that includes the 16-channel layers of the 512^2 stage (1.3-1.55x the direct kernel) and 8->16 at 1024^2
# (1.16x); 16->8 and 8->8 (half of the cout tile empty) stay on the block-MFMA kernels (0.7-1.0x).
import os as _os
USE_WINOGRAD = _os.environ.get('PGGAN_WINOGRAD', '1') != '0'
WINO_MIN_WORKGROUPS = int(_os.environ.get('PGGAN_WINO_MIN_WG', '32'))     # (K is sliced across workgroups below ~432: ops._stream_with_workspace)
WINO_MIN_CHANNELS = int(_os.environ.get('PGGAN_WINO_MIN_C', '8'))
WINO_MIN_COUT = int(_os.environ.get('PGGAN_WINO_MIN_COUT', '16'))
USE_WINOGRAD_WGRAD = USE_WINOGRAD and _os.environ.get('PGGAN_WINOGRAD_WGRAD', '1') != '0'
WINO_WGRAD_MIN_CHANNELS = int(_os.environ.get('PGGAN_WINO_WGRAD_MIN_C', '16'))
# The c2 output of a DBlock is pooled at once (network.py:229,238); at full resolution only its SIGN is ever used again
# (LeakyReLU' in the backward / tangent sweeps), so from 64x64 up it is kept as sign bytes (1 byte per 4 channels)
# instead of fp32: the fp32 write and its re-reads are the largest avoidable HBM traffic of the high-resolution stages.
import collections as _collections
FALLBACKS = _collections.Counter()       # launches that answered PG_E_UNSUP to a sign-byte request and were redone in fp32
USE_SIGN_BYTES = _os.environ.get('PGGAN_SIGN_BYTES', '1') != '0'
SIGN_BYTES_MIN_H = int(_os.environ.get('PGGAN_SIGN_BYTES_MIN_H', '64'))
# With the mask as sign bytes the pool adjoint between two DBlocks can be evaluated in the input gathers of its consumers
# (backward-data conv and weight gradient of the finer block's c2) instead of being written out at the fine resolution.
USE_LAZY_UNPOOL = _os.environ.get('PGGAN_LAZY_UNPOOL', '1') != '0'
USE_LAZY_UNPOOL_FADE = _os.environ.get('PGGAN_LAZY_UNPOOL_FADE', '1') != '0'    # ... also across the fade-in boundary (alpha < 1)


def _live_conv_layers(net):
    """3x3 / 4x4 conv layers that can run at the network's current growth stage (D: blocks[-(depth+1):], network.py:227-238;
    G: block0 and blocks[:depth], network.py:124-130)."""
    depth = int(net.depth)
    if hasattr(net, 'linear'):                                  # Discriminator
        blocks = list(net.blocks)[len(net.blocks) - 1 - depth:]
    else:
        blocks = [net.block0] + list(net.blocks)[:depth]
    return [m for b in blocks for m in (b.c1, b.c2)]


def _derived(net):
    """Derived weight copies of a network, refreshed with three launches per (weight version, growth stage): backward-data
    (flipped / transposed) weights, and the Winograd-domain forward and backward-data weights — of the layers that are live
    at this stage only (at 4x4 that is 2 of the 18 conv layers of a network: re-deriving all 73 MB per update cost ~0.5 ms of
    a 1.6 ms step)."""
    net._ensure_buffers()
    key = (net._param_version, int(net.depth))
    if net._derived_ver == key:
        return
    base = net._flat_param.data_ptr()
    live = set(id(m) for m in _live_conv_layers(net))
    # flipped / transposed copies: only for the layers whose backward-data conv has asked for one (_wt marks them) -- the layers that
    # always run Winograd there get their backward-data form straight from the parameter (transposed=True below)
    todo = [m for m in net._layers() if m.kind == 'conv' and m._wt is not None and id(m) in live and (getattr(m, '_wt_wanted', False) or not WTU_FROM_PARAM)]
    packs = [((m.conv.weight.data_ptr() - base) // 4, m.ksize, m.conv.weight.shape[2], m.conv.weight.shape[3]) for m in todo]
    # ... and of those only the layers the Winograd path has actually asked for (_wino marks them): the 512-channel layers of the
    # 4x4 stage never reach it, yet are a large part of all Winograd-domain bytes
    wl = [(m, woff, uoff) for m, woff, uoff in (net._wino_layers or []) if id(m) in live and getattr(m, '_wino_wanted', False)]
    wl = wl if USE_WINOGRAD else []

    def backward_copies():                       # what only the backward-data convs read: flipped / transposed weights and the Winograd form of those
        if packs:
            ops.pack_dgrad_weights_batched(net._flat_param, net._flat_wt, packs)
        if wl:
            ops.wino_transform_weights_batched(net._flat_param if WTU_FROM_PARAM else net._flat_wt, net._flat_wtu,
                                               [(woff, uoff, m.conv.weight.shape[3], m.conv.weight.shape[2]) for m, woff, uoff in wl],
                                               transposed=WTU_FROM_PARAM)
    dev = torch.cuda.current_device() if net._flat_param.is_cuda else None
    cur = torch.cuda.current_stream(torch._C._cuda_getDevice()) if dev is not None else None
    capturing = dev is not None and torch.cuda.is_current_stream_capturing()
    # Inside a deferred update (defer_to_side: Adam + this refresh on the second stream while the main stream waits for D's weights -- since
    # round 6's early generator pass that wait is exposed, tools/phase_timeline.py) the forward forms go first and close the update for
    # the waiting stream (``_pending`` = the event behind them); the backward-data forms and the flipped copies follow behind that event and
    # are picked up through ``_await_backward_copies`` by their first reader, a forward pass later.
    split = (SPLIT_DEFERRED_DERIVE and bool(net.__dict__.get('_defer_active')) and dev is not None and cur == _SIDE.get(dev) and not capturing)
    one_launch = DERIVED_ONE_LAUNCH and WTU_FROM_PARAM and bool(wl) and net.__dict__.get('_bwd_wanted', False) and not split
    if one_launch:
        # forward AND backward-data Winograd forms in one launch on this stream (the two forms share a buffer, network._flat_wuu): the
        # parameter is read while it is hot, no second launch, no cross-stream hop for it at the iteration boundary
        half = net._flat_wu.numel()
        ops.wino_transform_weights_batched(
            net._flat_param, net._flat_wuu,
            [(woff, uoff, m.conv.weight.shape[2], m.conv.weight.shape[3]) for m, woff, uoff in wl] +
            [(woff, half + uoff, m.conv.weight.shape[3], m.conv.weight.shape[2]) for m, woff, uoff in wl],
            transposed=[False] * len(wl) + [True] * len(wl))
        wl = []                                  # (backward_copies below: only the flipped copies of the direct-conv layers are left)
    elif wl:                                     # the forward convs' Winograd-domain weights: needed by the very next launch
        ops.wino_transform_weights_batched(net._flat_param, net._flat_wu,
                                           [(woff, uoff, m.conv.weight.shape[2], m.conv.weight.shape[3]) for m, woff, uoff in wl])
    # The backward copies are not needed before the network's next backward sweep (milliseconds away): off the critical path, onto
    # the weight-gradient stream, unless this already IS that stream (the deferred D update) or a hipGraph is being captured.
    # A network that has never run a backward sweep (an EMA / evaluation copy) gets none: the first request for one
    # (_wt / _wino(transposed)) invalidates the derived state.
    # The forward forms are on ``cur`` now.  A consumer on ANOTHER stream must be ordered behind them (_await_derived): the three-pass D
    # forward refreshes D's derived weights from inside its real-third pass on the second stream whenever the update did not come through
    # Trainer's deferred update -- the public ``loss.backward(); optimizer.step()`` loop, a foreign optimizer, load_state_dict -- while the
    # mixed third on the main stream was ordered behind the image copy only (docs/experiments_r6.md §1).
    net._derived_ev = None
    net._derived_waited = set()
    if dev is not None and DERIVED_EVENT and not capturing:
        ev = torch.cuda.Event()
        _record_event(ev, cur)
        net._derived_ev = ev
        net._derived_waited = {cur.cuda_stream}
        if split:
            net._pending_ev = ev
    net._derived_bwd_ev = None
    net._derived_bwd_waited = set()
    if not net.__dict__.get('_bwd_wanted', False):
        pass
    elif (ASYNC_WGRAD and ASYNC_DERIVED and dev is not None and cur != _SIDE.get(dev) and not capturing):
        side = _side_stream()
        _wait_stream(side, cur)
        with torch.cuda.stream(side):
            backward_copies()
            ev = torch.cuda.Event()
            _record_event(ev, side)
        for buf in (net._flat_wt, net._flat_wtu):  # written on the side stream: the allocator must not hand the block on before that
            if buf is not None:
                buf.record_stream(side)
        net._derived_bwd_ev = ev
    else:
        backward_copies()                        # inline, on ``cur``: readers on other streams are ordered behind this event
        if dev is not None and DERIVED_EVENT and not capturing:
            ev = torch.cuda.Event()
            _record_event(ev, cur)
            net._derived_bwd_ev = ev
            net._derived_bwd_waited = {cur.cuda_stream}
    net._derived_ver = key
    net._derived_live = live


def _await_derived(net):
    """The current stream waits (once) for the last refresh of the derived weights if that ran on another stream."""
    ev = net.__dict__.get('_derived_ev')
    if ev is not None:
        cur = torch.cuda.current_stream(torch._C._cuda_getDevice())
        waited = net._derived_waited
        if cur.cuda_stream not in waited and not torch.cuda.is_current_stream_capturing():
            cur.wait_event(ev)
            waited.add(cur.cuda_stream)


def order_side_behind_derived(net):
    """plans.py, before a replay: the recorded body's per-launch ``_await_derived`` calls are not part of a plan (they are no-ops in
    the orders a plan is entered with), so the one ordering a replay could miss -- the second stream behind a refresh ``_prologue`` just
    issued on the main stream -- is established here, once per step."""
    ev = net.__dict__.get('_derived_ev')
    if ev is not None and ASYNC_WGRAD:
        side = _side_stream()
        if side.cuda_stream not in net._derived_waited:
            side.wait_event(ev)
            net._derived_waited.add(side.cuda_stream)


def _assert_live(net, layer):
    """The derived copies are refreshed for the layers that are live at the network's current growth stage only: asking for
    any other layer's would silently compute with the weights of an earlier optimizer step."""
    if id(layer) not in net._derived_live:
        raise RuntimeError('derived weights requested for a conv layer that is not live at depth %d' % int(net.depth))


def _await_backward_copies(net):
    """The current stream waits for the side-stream refresh of the backward-only derived weights.  The event stays until the
    next refresh, so that EVERY stream that consumes (or, for the optimizer, overwrites the source of) the copies is ordered
    behind it, each once."""
    ev = net.__dict__.get('_derived_bwd_ev')
    if ev is not None:
        cur = torch.cuda.current_stream(torch._C._cuda_getDevice())
        waited = net.__dict__.setdefault('_derived_bwd_waited', set())
        if cur.cuda_stream not in waited:
            _wait_event(cur, ev)
            waited.add(cur.cuda_stream)


def _want_backward_copies(net):
    if not net.__dict__.get('_bwd_wanted', False):      # first backward sweep of this network: from now on every refresh includes them
        net._bwd_wanted = True
        net._derived_ver = None


def _wt(net, layer):
    """Backward-data (flipped/transposed) copy of a conv layer's weights, refreshed lazily."""
    if not getattr(layer, '_wt_wanted', False):         # first request: from now on the refresh after every update includes this layer
        layer._wt_wanted = True
        net._derived_ver = None
    _want_backward_copies(net)
    _derived(net)
    _assert_live(net, layer)
    _await_derived(net)
    _await_backward_copies(net)
    return layer._wt


def _wino(layer, N, H, cout, transposed=False):
    """Winograd-domain weights of ``layer`` if that path should run for an N x H x H output with ``cout`` channels."""
    if not USE_WINOGRAD or getattr(layer, '_wu', None) is None or H < 8:
        return None
    cin = layer.conv.weight.shape[2] if transposed else layer.conv.weight.shape[3]     # input channels of THIS direction
    if cin < WINO_MIN_CHANNELS or cout < WINO_MIN_COUT:
        return None
    if -(-(N * (H // 2) * (H // 2)) // 64) * -(-cout // 16) < WINO_MIN_WORKGROUPS:
        return None
    net = layer._net()
    if not getattr(layer, '_wino_wanted', False):       # first request: from now on the refresh after every update includes this layer
        layer._wino_wanted = True
        net._derived_ver = None
    if transposed:
        _want_backward_copies(net)
    _derived(net)
    _assert_live(net, layer)
    _await_derived(net)
    if transposed:
        _await_backward_copies(net)
    return layer._wtu if transposed else layer._wu


def _conv(x, layer, N, H, act=True, mask=None, bias=True, ups=False, out=None, signs_out=False):
    """Forward conv (+bias+act) or, with ``mask``, the masked linear map of the tangent pass.  ``signs_out``: returns
    (y, sign bytes of y) -- (y, None) when the launch cannot produce them."""
    def run(mask, signs_out):
        u = _wino(layer, N, H, layer.conv.weight.shape[2])
        if u is not None:
            return ops.conv2d_wino(x, u, layer.conv.bias.data if bias else None, N, H, H, layer.c,
                                   layer.slope if act else 1.0, mask=mask, mask_slope=layer.slope, ups=ups, out=out,
                                   signs_out=signs_out)
        return ops.conv2d(x, layer.conv.weight.data, layer.conv.bias.data if bias else None, N, H, H,
                          layer.ksize, layer.pad, layer.c, layer.slope if act else 1.0,
                          mask=mask, mask_slope=layer.slope, ups=ups, out=out, signs_out=signs_out)
    if signs_out:
        try:
            return run(mask, True)
        except ops.Unsupported:
            FALLBACKS['conv signs_out %dx%d %s' % (H, H, tuple(layer.conv.weight.shape))] += 1
            return run(mask, False), None
    if isinstance(mask, tuple):                    # (fp32 activation, its sign bytes or None): bytes first, fp32 as the fallback
        m32, mb = mask
        if mb is not None:
            try:
                return run(mb, False)
            except ops.Unsupported:
                FALLBACKS['conv masked %dx%d %s' % (H, H, tuple(layer.conv.weight.shape))] += 1
        return run(m32, False)
    return run(mask, False)


def _mask32(m):
    """fp32 view of a LeakyReLU' mask that may be stored as sign bytes (paths without a byte-aware kernel)."""
    return ops.signbytes_to_mask(m) if m is not None and m.dtype == torch.uint8 else m


def _conv_pool(x, layer, N, H, bias=True, mask=None, other=None, a=1.0, b=0.0, pool_only=False, y_bytes=False):
    """conv (+bias+act | mask) followed by the 2x2 average pool / fade-in blend, one launch.  ``y_bytes``: the
    full-resolution output is returned as sign bytes (falls back to fp32 when the launch cannot fuse)."""
    def run(mask, y_bytes):
        u = _wino(layer, N, H, layer.conv.weight.shape[2])
        if u is not None:
            return ops.conv2d_wino(x, u, layer.conv.bias.data if bias else None, N, H, H, layer.c,
                                   layer.slope if mask is None else 1.0, mask=mask, mask_slope=layer.slope,
                                   pool=True, other=other, a=a, b=b, pool_only=pool_only, y_bytes=y_bytes)
        return ops.conv2d_pool(x, layer.conv.weight.data, layer.conv.bias.data if bias else None, N, H, H, layer.ksize, layer.pad,
                               layer.c, layer.slope if mask is None else 1.0, mask=mask, mask_slope=layer.slope,
                               other=other, a=a, b=b, pool_only=pool_only, y_bytes=y_bytes)
    if y_bytes or (mask is not None and mask.dtype == torch.uint8):
        try:
            return run(mask, y_bytes)
        except ops.Unsupported:
            FALLBACKS['conv_pool %dx%d %s' % (H, H, tuple(layer.conv.weight.shape))] += 1
            return run(_mask32(mask), False)
    return run(mask, False)


def _dgrad(net, gz, layer, N, Hout, mask=None, mask_slope=0.2):
    """Adjoint of the conv wrt its input: gz [N,Hout,Hout,Cout] -> [N,Hin,Hin,Cin_store] (* mask)."""
    def run(mask):
        u = _wino(layer, N, Hout, layer.conv.weight.shape[3], transposed=True) if layer.ksize == 3 and layer.pad == 1 else None
        if u is not None:
            return ops.conv2d_wino(gz, u, None, N, Hout, Hout, layer.c, 1.0, mask=mask, mask_slope=mask_slope)
        return ops.conv2d(gz, _wt(net, layer), None, N, Hout, Hout, layer.ksize, layer.ksize - 1 - layer.pad,
                          layer.c, 1.0, mask=mask, mask_slope=mask_slope)
    if isinstance(mask, tuple):                    # (fp32 activation, its sign bytes or None)
        m32, mb = mask
        if mb is not None:
            try:
                return run(mb)
            except ops.Unsupported:
                FALLBACKS['dgrad masked %dx%d %s' % (Hout, Hout, tuple(layer.conv.weight.shape))] += 1
                if m32 is None:                    # (d_forward(keep_input=False) kept the sign bytes only: expand them)
                    m32 = _mask32(mb)
        return run(m32)
    return run(mask)


def _dgrad_pool(net, gz, layer, N, H, other=None, a=1.0, b=0.0, pnb=None):
    """Backward-data conv + 2x2 pooled epilogue (a * mean + b * other); only the pooled result is produced.
    ``pnb`` = (ysaved, r, slope): the adjoint of the coarser block's last (LeakyReLU -> PixelNorm) follows -- in the same launch
    where the Winograd epilogue holds every channel of the pooled pixel.  Returns (result, adjoint applied?)."""
    u = _wino(layer, N, H, layer.conv.weight.shape[3], transposed=True)
    if pnb is not None:
        if u is not None and u.shape[1] <= 32 and pnb[1] is not None:
            return ops.conv2d_wino_pnbwd(gz, u, pnb[0], pnb[1], N, H, H, layer.c, pnb[2], pool=True, other=other, a=a, b=b), True
        return _dgrad_pool(net, gz, layer, N, H, other=other, a=a, b=b), False
    if u is not None:
        return ops.conv2d_wino(gz, u, None, N, H, H, layer.c, 1.0, pool=True, other=other, a=a, b=b, pool_only=True)[1]
    return ops.conv2d_pool(gz, _wt(net, layer), None, N, H, H, layer.ksize, layer.ksize - 1 - layer.pad, layer.c, 1.0,
                           other=other, a=a, b=b, pool_only=True)[1]


def _dgrad_unpool(net, gz, layer, N, H, upmask, mul, mask_slope):
    """Backward-data conv + pool adjoint + LeakyReLU' mask of the finer activation, one launch."""
    u = _wino(layer, N, H, layer.conv.weight.shape[3], transposed=True)
    if u is not None:
        return ops.conv2d_wino(gz, u, None, N, H, H, layer.c, 1.0, mask_slope=mask_slope, unpool=True, upmask=upmask, up_mul=mul)
    try:
        return ops.conv2d_unpool(gz, _wt(net, layer), N, H, H, layer.ksize, layer.ksize - 1 - layer.pad, layer.c,
                                 upmask=upmask, mul=mul, mask_slope=mask_slope)
    except ops.Unsupported:
        if upmask is None or upmask.dtype != torch.uint8:
            raise
        FALLBACKS['dgrad unpool %dx%d %s' % (H, H, tuple(layer.conv.weight.shape))] += 1
        return ops.conv2d_unpool(gz, _wt(net, layer), N, H, H, layer.ksize, layer.ksize - 1 - layer.pad, layer.c,
                                 upmask=_mask32(upmask), mul=mul, mask_slope=mask_slope)


def _dgrad_pnbwd(net, gz, layer, N, H, ysaved, r, slope):
    """Backward-data conv + adjoint of the previous layer's (LeakyReLU -> PixelNorm)."""
    u = _wino(layer, N, H, layer.conv.weight.shape[3], transposed=True)
    if u is not None and u.shape[1] <= 32 and r is not None:      # every channel of a pixel in one workgroup: the adjoint in the Winograd epilogue
        return ops.conv2d_wino_pnbwd(gz, u, ysaved, r, N, H, H, layer.c, slope)
    if u is not None:                              # wide layers: Winograd conv, then the (HBM-bound) adjoint kernel in place
        g = ops.conv2d_wino(gz, u, None, N, H, H, layer.c, 1.0)
        return ops.pixelnorm_lrelu_bwd(g, ysaved, r, slope, inplace=True)
    return ops.conv2d_pnbwd(gz, _wt(net, layer), ysaved, r, N, H, H, layer.ksize, layer.ksize - 1 - layer.pad, layer.c, slope)


# Weight gradients are leaves of the backward sweeps (nothing downstream reads them before the optimizer), so
# they are launched on a second HIP stream and overlap the backward-data chain on the main stream: two
# half-occupancy MFMA kernels share the CUs instead of running back to back.
ASYNC_WGRAD = _os.environ.get('PGGAN_ASYNC_WGRAD', '1') != '0'
ASYNC_DERIVED = _os.environ.get('PGGAN_ASYNC_DERIVED', '1') != '0'
SPLIT_DEFERRED_DERIVE = _os.environ.get('PGGAN_SPLIT_DERIVE', '1') != '0'      # deferred D update: forward forms first, then release the waiting stream
DERIVED_EVENT = _os.environ.get('PGGAN_DERIVED_EVENT', '1') != '0'    # 0: the round-5 behaviour (ablation for tests/test_e2e_gpu.py::test_derived_refresh_ordering only)
DERIVED_ONE_LAUNCH = _os.environ.get('PGGAN_DERIVED_ONE_LAUNCH', '1') != '0'      # forward + backward-data Winograd weights of a network: one launch
# 0: every live layer gets a flipped / transposed copy and the backward-data Winograd form is derived from that copy (round 2)
WTU_FROM_PARAM = _os.environ.get('PGGAN_WTU_FROM_PARAM', '1') != '0'
_SIDE = {}


# Every cross-stream ordering the step schedules issue goes through these three: plans._Recorder swaps them (module attributes of
# THIS module only -- never torch's classes, so stream / event traffic of other threads or libraries cannot leak into a plan).
def _wait_stream(waiter, other):
    waiter.wait_stream(other)


def _record_event(ev, stream):
    ev.record(stream)


def _wait_event(stream, ev):
    stream.wait_event(ev)


# Un-traced phase timeline (tools/phase_timeline.py): with ``PROBES`` set to a dict, the step schedules drop a timing event on the
# current stream at every phase boundary -- through ``_record_event``, so a launch plan re-records the same events on every replay and
# the probes cost ~25 event packets per step instead of the host slow-down of a tracer (rocprofv3 --kernel-trace makes the host the
# bottleneck of the 340-launch 1024^2 step: its "queue waiting" gaps are partly its own).  None (default): nothing is recorded.
PROBES = None


def probe(tag):
    if PROBES is not None:
        ev = torch.cuda.Event(enable_timing=True)
        _record_event(ev, torch.cuda.current_stream(torch._C._cuda_getDevice()))
        PROBES.setdefault(tag, []).append(ev)


def _side_stream():
    dev = torch.cuda.current_device()
    if dev not in _SIDE:
        _SIDE[dev] = torch.cuda.Stream(device=dev)
        ops._no_workspace_streams.add(_SIDE[dev].cuda_stream)      # (ops._stream_with_workspace: no sliced launch on it inside a capture)
    return _SIDE[dev]


class _on_side(object):
    """``with _on_side(t1, t2, ...):`` runs the body on the side stream after everything queued so far on
    the current stream; the tensors are registered with the allocator as in use on that stream."""

    def __init__(self, *tensors):
        self.tensors = [t for t in tensors if t is not None]
        self.active = ASYNC_WGRAD and len(self.tensors) > 0 and self.tensors[0].is_cuda

    def __enter__(self):
        if not self.active:
            return self
        main = torch.cuda.current_stream(torch._C._cuda_getDevice())      # explicit index: skips the slow device lookup
        self.side = _side_stream()
        _wait_stream(self.side, main)
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if not self.active:
            return False
        self.ctx.__exit__(*exc)
        for t in self.tensors:
            t.record_stream(self.side)
        return False


def _join_side():
    """Main stream waits for every weight-gradient launch (call before the optimizer / all-reduce)."""
    if ASYNC_WGRAD and torch.cuda.is_available() and torch.cuda.current_device() in _SIDE:
        _wait_stream(torch.cuda.current_stream(torch._C._cuda_getDevice()), _SIDE[torch.cuda.current_device()])


def defer_to_side(net, fn):
    """Run ``fn()`` — the tail of a network's update: gradient all-reduce, Adam, derived weights — on the side stream,
    after everything queued so far on the current stream.  The current stream picks the result up in ``wait_pending``
    at its next use of the network's weights, so the tail of the D update overlaps the generator forward that opens
    the G step (the G step touches D only after G(z))."""
    if not (ASYNC_WGRAD and net._flat_param.is_cuda):
        fn()
        return
    main = torch.cuda.current_stream(torch._C._cuda_getDevice())
    side = _side_stream()
    side.wait_stream(main)
    net._defer_active = True
    try:
        with torch.cuda.stream(side):
            fn()
            ev = net.__dict__.pop('_pending_ev', None)      # (_derived: the event behind Adam + the forward forms; the rest of the refresh follows it)
            if ev is None:
                ev = torch.cuda.Event()
                ev.record(side)
    finally:
        net._defer_active = False
    net._pending = ev


def wait_pending(net):
    """The current stream waits for a deferred update of ``net`` (no-op when there is none)."""
    ev = getattr(net, '_pending', None)
    if ev is not None:
        torch.cuda.current_stream(torch._C._cuda_getDevice()).wait_event(ev)
        net._pending = None


def _grads_ready(net, layers):
    """Tell the gradient exchange (``parallel.GradExchange``, installed by ``Trainer`` as ``net._grad_hook`` under data
    parallelism) that every weight-gradient launch of ``layers`` has been enqueued: their spans of the flat gradient
    buffer can start travelling while the sweep goes on."""
    hook = getattr(net, '_grad_hook', None)
    if hook is not None:
        for m in layers:                         # "complete" must never precede a deferred contribution that nothing carried
            if m.__dict__.get('_pending_wgrad') is not None:
                _flush_wgrad(m)
        side = None
        if ASYNC_WGRAD and net._flat_param.is_cuda:
            side = _SIDE.get(torch.cuda.current_device())
        hook(layers, side)


# The gradient-penalty tangent term and the batched adjoint sweep both add to the weight gradient of every D layer.  For the
# Winograd layers the first contribution (3 images) is not launched on its own: it waits on the layer and rides in the launch of
# the second (pg_conv2d_wgrad_wino2_nhwc) -- one commit of dW instead of two, and the commit is a third of a 3-image launch.
DEFER_TANGENT_WGRAD = _os.environ.get('PGGAN_DEFER_TANGENT_WGRAD', '1') != '0'


def _wino_wgrad_ok(layer, Hin):
    return (USE_WINOGRAD_WGRAD and layer.ksize == 3 and layer.pad == 1 and Hin >= 16 and not (Hin & (Hin - 1))
            and min(layer._gw.shape[2], layer._gw.shape[3]) >= WINO_WGRAD_MIN_CHANNELS)


# The weight gradients of the entry (finest) block are the last and largest launches of the batched sweep: when the main stream finishes
# the sweep the second stream still has ~0.4 ms of them queued (tools/phase_timeline.py: D.wgrad_end - D.sweep_end at 1024^2), and with the
# G step's generator pass already done (EarlyG) the main stream has nothing to run until D's update is through.  The last
# TAIL_WGRAD_ON_MAIN of them (fromRGB, c1, c2 of the entry block, in that order of preference) are therefore launched on the main stream;
# the deferred update is ordered behind both streams (defer_to_side).  Not under a bucketed gradient exchange (its flushes wait for the
# second stream only).
# Same-box runs with the early generator pass on, ms per step, 0 | 1 | 2 launches on the main stream: 1024^2 10.182 | 10.135 | 10.134 and
# 10.185 | 10.118 | 10.076; 512^2 13.322 | 13.288 | 13.384.  Default (-1): 2 at 1024^2, 1 at 512^2, none below (the 16-image stages keep both
# streams busy to the end).
TAIL_WGRAD_ON_MAIN = int(os.environ.get('PGGAN_TAIL_WGRAD_MAIN', '-1'))


def _tail_wgrad_on_main(H):
    if TAIL_WGRAD_ON_MAIN >= 0:
        return TAIL_WGRAD_ON_MAIN
    return 2 if H >= 1024 else 1 if H >= 512 else 0


class _on_main(object):
    def __init__(self, *tensors):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _wgrad(x, gz, layer, N, Hin, bias=True, ups=False, defer=False, on_main=False):
    """Weight (and bias) gradient of one conv layer on the side stream.  The wide 3x3 layers take the Winograd form
    (2.25x fewer MFMAs, 1.3-1.7x faster from 16x16 up, 16-channel sides included); the 8-channel layers and the
    4x4 / 8x8 maps keep the direct kernels.  ``defer``: a bias-free contribution that a later ``_wgrad`` of the same layer
    (same step, same input geometry) will carry along."""
    wino = _wino_wgrad_ok(layer, Hin)
    if defer and wino and DEFER_TANGENT_WGRAD and not bias and not ups:
        layer._pending_wgrad = (x, gz, N, Hin)
        return
    pend = layer.__dict__.get('_pending_wgrad')
    if pend is not None and not (wino and pend[3] == Hin and not ups):
        _flush_wgrad(layer)
        pend = None
    layer._pending_wgrad = None
    with (_on_main if on_main else _on_side)(x, gz, *(pend[:2] if pend is not None else ())):
        if wino:
            ops.conv2d_wgrad_wino(x, gz, layer._gw, layer._gb if bias else None, N, Hin, Hin, layer.c, ups=ups,
                                  second=(pend[0], pend[1], pend[2], False) if pend is not None else None)
            return
        ops.conv2d_wgrad(x, gz, layer._gw, layer._gb if bias else None, N, Hin, Hin, layer.ksize, layer.pad,
                         layer.c, ups=ups)


def _flush_wgrad(layer):
    """Launch a deferred contribution on its own (nothing came to carry it)."""
    pend = layer.__dict__.get('_pending_wgrad')
    if pend is None:
        return
    layer._pending_wgrad = None
    x, gz, N, Hin = pend
    with _on_side(x, gz):
        ops.conv2d_wgrad_wino(x, gz, layer._gw, None, N, Hin, Hin, layer.c)


_ONES = {}


def _ones(n, device):
    key = (n, str(device))
    if key not in _ONES:
        _ONES[key] = torch.ones(n, device=device, dtype=torch.float32)
    return _ONES[key]


# ------------------------------------------------------------------------------------------
# Generator
# ------------------------------------------------------------------------------------------
def generator_forward(G, z, save=False, out=None, pair_out=None):
    """reference network.py:118-139.  z [N,latent] -> NCHW image [N,C,r,r] (written into ``out``).  ``pair_out`` = (outA, outB or None):
    the two halves of the batch go to two image buffers (the paired D-step / G-step pass, ``_generator_pair``); returns (imgA, imgB)."""
    wait_pending(G)
    ops.require_gpu()
    G._sync_version()                 # also notice torch-side updates (torch.optim.*, load_state_dict) before any derived weight is used
    z = _check_dev(z, 'latents')
    N, L = z.shape
    if L != G.latent_size or L % 4:
        raise ValueError('latent size %d (expected %d, multiple of 4)' % (L, G.latent_size))
    depth, alpha = int(G.depth), float(G.alpha)
    C = G.num_channels
    b0 = G.block0
    G._ensure_buffers()
    ctx = dict(N=N, depth=depth, alpha=alpha, recs=[])
    if G.normalize_latents:
        zn, _ = ops.pixelnorm_fwd(z, G.eps)                                   # :120-123
    else:
        zn = z
    ctx['zn'] = zn

    def layer(x, lay, H, ups=False):
        if lay.pixelnorm:
            u = _wino(lay, N, H, lay.conv.weight.shape[2]) if lay.ksize == 3 else None
            if u is not None and u.shape[1] <= 32:   # all couts of a pixel in one workgroup: PixelNorm in the Winograd epilogue
                return ops.conv2d_wino_pixelnorm(x, u, lay.conv.bias.data, N, H, H, lay.c, lay.slope, lay.eps, ups=ups)
            if u is not None:                     # wide layers: Winograd conv, PixelNorm as its own (HBM-bound) pass
                y = ops.conv2d_wino(x, u, lay.conv.bias.data, N, H, H, lay.c, lay.slope, ups=ups)
                return ops.pixelnorm_fwd(y, lay.eps, inplace=True)
            # conv + bias + act + PixelNorm in one launch where the tile allows it
            return ops.conv2d_pixelnorm(x, lay.conv.weight.data, lay.conv.bias.data, N, H, H, lay.ksize, lay.pad, lay.c,
                                        lay.slope, lay.eps, ups=ups)
        return _conv(x, lay, N, H, ups=ups), None

    y1, r1 = layer(zn.view(N, 1, 1, L), b0.c1, 1)                            # 4x4 conv pad 3 on 1x1
    y2, r2 = layer(y1, b0.c2, 4)
    ctx.update(y1=y1, r1=r1, y2=y2, r2=r2)
    def to_rgb(h, t, H, out_mul=1.0, prev=None, prev_mul=0.0):
        if pair_out is None:
            return ops.torgb_fwd(h, t.conv.weight.data, t.conv.bias.data, N, C, H, H, t.c, out_mul=out_mul, prev=prev, prev_mul=prev_mul, out=out)
        n2 = N // 2
        return tuple(ops.torgb_fwd(h[a:b], t.conv.weight.data, t.conv.bias.data, n2, C, H, H, t.c, out_mul=out_mul,
                                   prev=prev[a:b] if prev is not None else None, prev_mul=prev_mul, out=o)
                     for (a, b), o in zip(((0, n2), (n2, N)), pair_out))
    if depth == 0:
        img = to_rgb(y2, b0.toRGB, 4)                                          # :55-56
        return (img, ctx) if save else img
    h, H = y2, 4
    for i in range(depth):                                                    # :126-130
        blk = G.blocks[i]
        H *= 2
        a1, ra1 = layer(h, blk.c1, H, ups=True)                               # upsample fused into the conv
        img = None
        if (i == depth - 1 and FUSE_TORGB and alpha >= 1.0 and pair_out is None and blk.c2.pixelnorm and blk.c2.ksize == 3 and a1.is_cuda
                and _wino(blk.c2, N, H, blk.c2.conv.weight.shape[2]) is None):
            t = blk.toRGB                          # the last conv writes the image too (toRGB in its epilogue: the 1024^2 stage)
            try:
                a2, ra2, img = ops.conv2d_pixelnorm_torgb(a1, blk.c2.conv.weight.data, blk.c2.conv.bias.data, t.conv.weight.data, t.conv.bias.data,
                                                          N, C, H, H, blk.c2.c, blk.c2.slope, t.c, blk.c2.eps, out=out)
            except ops.Unsupported:
                FALLBACKS['toRGB in the epilogue %dx%d' % (H, H)] += 1
                img = None
        if img is None:
            a2, ra2 = layer(a1, blk.c2, H)
        ctx['recs'].append(dict(blk=blk, inp=h, a1=a1, r1=ra1, a2=a2, r2=ra2, H=H))
        hprev, h = h, a2
    if img is not None:
        return (img, ctx) if save else img
    t = G.blocks[depth - 1].toRGB
    prev = None
    if alpha < 1.0:                                                           # :131-135
        pt = G.blocks[depth - 2].toRGB if depth > 1 else b0.toRGB
        prev = ops.torgb_fwd(hprev, pt.conv.weight.data, pt.conv.bias.data, N, C, H // 2, H // 2, pt.c)
    img = to_rgb(h, t, H, out_mul=alpha, prev=prev, prev_mul=1.0 - alpha)    # :138
    return (img, ctx) if save else img


def generator_backward(G, ctx, g_out):
    """Adjoint sweep of generator_forward: accumulates into G's flat gradient buffer."""
    wait_pending(G)
    G._ensure_buffers()
    N, depth, alpha = ctx['N'], ctx['depth'], ctx['alpha']
    C = G.num_channels
    b0 = G.block0
    active = [b0.c1, b0.c2]
    g_extra = None
    top_done = False
    if depth == 0:
        t = b0.toRGB
        with _on_side(g_out, ctx['y2']):
            ops.torgb_wgrad(g_out, ctx['y2'], t._gw, t._gb, N, C, 4, 4, t.c, 1.0)
        g = ops.torgb_bwd_data(g_out, t.conv.weight.data, N, C, 4, 4, t.c)
        active.append(t)
    else:
        rec = ctx['recs'][-1]
        H = rec['H']
        t = rec['blk'].toRGB
        with _on_side(g_out, rec['a2']):
            ops.torgb_wgrad(g_out, rec['a2'], t._gw, t._gb, N, C, H, H, alpha * t.c, alpha)
        # toRGB's adjoint + the adjoint of the top block's last (LeakyReLU -> PixelNorm) in one launch (network.py:138, :44-52)
        if rec['r2'] is not None:
            g = ops.torgb_bwd_data_pnbwd(g_out, t.conv.weight.data, rec['a2'], rec['r2'], N, C, H, H, alpha * t.c, rec['blk'].c2.slope)
            top_done = True
        else:                                         # (a generator built without PixelNorm: the adjoint is LeakyReLU' alone)
            g = ops.torgb_bwd_data(g_out, t.conv.weight.data, N, C, H, H, alpha * t.c)
        active.append(t)
        if alpha < 1.0:
            pt = G.blocks[depth - 2].toRGB if depth > 1 else b0.toRGB
            with _on_side(g_out, rec['inp']):
                ops.torgb_wgrad(g_out, rec['inp'], pt._gw, pt._gb, N, C, H // 2, H // 2, (1 - alpha) * pt.c,
                                1 - alpha, down=True)
            g_extra = ops.torgb_bwd_data(g_out, pt.conv.weight.data, N, C, H // 2, H // 2, (1 - alpha) * pt.c, down=True)
            active.append(pt)
    _grads_ready(G, active[2:])
    recs = ctx['recs']
    for k in range(len(recs) - 1, -1, -1):
        rec = recs[k]
        blk, H = rec['blk'], rec['H']
        c1, c2 = blk.c1, blk.c2
        if top_done:                                  # (the top block's adjoint came with toRGB's)
            gz2, top_done = g, False
        else:
            gz2 = ops.pixelnorm_lrelu_bwd(g, rec['a2'], rec['r2'], c2.slope, inplace=True)
        _wgrad(rec['a1'], gz2, c2, N, H)
        # backward-data conv of c2 + adjoint of c1's (LeakyReLU -> PixelNorm) in one launch
        gz1 = _dgrad_pnbwd(G, gz2, c2, N, H, rec['a1'], rec['r1'], c1.slope)
        _wgrad(rec['inp'], gz1, c1, N, H, ups=True)
        # backward-data conv of c1 + adjoint of the nearest x2 upsample (sum over 2x2 = 4 * average pool, exact in fp32)
        # + the fade-in branch's gradient, all in the conv epilogue
        # ... and, where it fits, the adjoint of the coarser block's last (LeakyReLU -> PixelNorm) as well
        prev = (recs[k - 1]['a2'], recs[k - 1]['r2'], recs[k - 1]['blk'].c2.slope) if k > 0 else (ctx['y2'], ctx['r2'], b0.c2.slope)
        g, top_done = _dgrad_pool(G, gz1, c1, N, H, other=g_extra, a=4.0, b=1.0, pnb=prev)
        g_extra = None
        active += [c1, c2]
        _grads_ready(G, [c1, c2])
    gz2 = g if top_done else ops.pixelnorm_lrelu_bwd(g, ctx['y2'], ctx['r2'], b0.c2.slope, inplace=True)
    _wgrad(ctx['y1'], gz2, b0.c2, N, 4)
    g1 = _dgrad(G, gz2, b0.c2, N, 4)
    gz1 = ops.pixelnorm_lrelu_bwd(g1, ctx['y1'], ctx['r1'], b0.c1.slope, inplace=True)
    _wgrad(ctx['zn'].view(N, 1, 1, -1), gz1, b0.c1, N, 1)
    _grads_ready(G, [b0.c1, b0.c2])
    return active


# ------------------------------------------------------------------------------------------
# Minibatch stddev: local-shard (default) or exact-global under data parallelism
# ------------------------------------------------------------------------------------------
# Default under data parallelism: every rank evaluates reference network.py:174-187 on its OWN minibatch (per-rank parity
# with the single-GPU reference at batch mb is exact).  ``Trainer(parallel=dp, global_stddev=True)`` switches to the exact-global mode:
# the one scalar a group's statistic is (and the three scalars its adjoint / Hessian-vector term need: G_sigma, <v, x - mu>, mean v) are
# reduced over all ranks, so world x mb equals one process at batch world * mb.  The collectives are a few floats each on the main
# stream, in the same order on every rank; steps are launched eagerly in this mode (wgan_gp_loss: no hipGraph / launch plan).
def _mb_dp(D):
    return D.__dict__.get('_global_stddev')


def _mbstd_fwd(D, x, groups, cp):
    dp = _mb_dp(D)
    if dp is None:
        return ops.mbstd_fwd(x, groups, cp)
    # the split kernels take the element count of the whole group as local count x ranks (csrc/elementwise.hip mbstd_write_kernel):
    # equal per-rank batches are part of the mode's contract -- checked here, once per batch shape (it was only claimed)
    if D.__dict__.get('_gs_checked') != (int(x.shape[0]), groups):
        dp.assert_same_on_all_ranks(int(x.shape[0]) // groups, 'exact-global minibatch stddev: the per-rank minibatch size')
        D._gs_checked = (int(x.shape[0]), groups)
    st = ops.mbstd_stats(x, groups)
    return ops.mbstd_write(x, st, dp.all_gather_rows(st), cp)


def _mbstd_tangent(D, x, tx, stats, cp):
    dp = _mb_dp(D)
    if dp is None:
        return ops.mbstd_tangent(x, tx, stats, cp)
    ts = ops.mbstd_tangent_stats(x, tx, stats)
    return ops.mbstd_tangent_write(tx, ts, dp.all_gather_rows(ts), stats, cp)


def _mbstd_bwd(D, gy, x, stats, cp, apply_mask, mask_slope, tx=None, tstats=None, gy_first=None, out=None):
    dp = _mb_dp(D)
    if dp is None:
        return ops.mbstd_bwd(gy, x, stats, cp, apply_mask, mask_slope, tx=tx, tstats=tstats, gy_first=gy_first, out=out)
    gs = ops.mbstd_gsum(gy, gy_first if tx is not None else None, stats.shape[0], tuple(x.shape), cp)
    dp.all_reduce_flat(gs.view(-1))
    return ops.mbstd_bwd_global(gy, x, stats, cp, apply_mask, gs, dp.world_size, mask_slope, tx=tx, tstats=tstats, gy_first=gy_first, out=out)


# ------------------------------------------------------------------------------------------
# Discriminator
# ------------------------------------------------------------------------------------------
# The G step's pass through D is followed by a backward-data sweep only (no weight gradients of D): the fp32 output of the entry block's
# fromRGB layer is then needed by nobody but the block's first conv, which evaluates it in its gather from the image
# (ops.conv2d_fromrgb: 12 B of image per pixel instead of 32 B written by one launch and read by the next; the 1024^2 stage).
FUSE_FROMRGB = _os.environ.get('PGGAN_FUSE_FROMRGB', '1') != '0'
FUSE_FROMRGB_BWD = _os.environ.get('PGGAN_FUSE_FROMRGB_BWD', '1') != '0'      # fromRGB's backward-data in the epilogue of the entry block's backward-data conv (ops.conv2d_masked_fromrgb_bwd)
FUSE_FROMRGB_WGRAD = _os.environ.get('PGGAN_FUSE_FROMRGB_WGRAD', '1') != '0'  # ... and fromRGB's weight gradient (the batched adjoint sweep: the 8-channel gradient is then never written)
FUSE_TORGB = _os.environ.get('PGGAN_FUSE_TORGB', '1') != '0'     # the generator's last conv writes the image in its epilogue (ops.conv2d_pixelnorm_torgb)


def d_forward(D, x, groups=1, keep_input=True):
    """reference network.py:225-240 on a batch of ``groups`` independent minibatches stacked along
    N (minibatch-stddev is evaluated per group).  Returns (scores [NB], ctx with every activation).
    ``keep_input=False``: the caller will not ask for D's weight gradients (``d_backward(full=False)``) -- the entry block's fromRGB
    output may stay unmaterialised (``rec['inp']`` is None then; its sign bytes ``rec['inpb']`` are always there)."""
    wait_pending(D)
    D._sync_version()
    NB, C, r, _ = x.shape
    depth, alpha = int(D.depth), float(D.alpha)
    if r != 4 * 2 ** depth or C != D.num_channels:
        raise ValueError('input %s does not match depth %d / %d channels' % (tuple(x.shape), depth, D.num_channels))
    nb = len(D.blocks)
    e = nb - 1 - depth                                                        # blocks[-(depth+1)]  (:227)
    D._ensure_buffers()
    ctx = dict(NB=NB, groups=groups, depth=depth, alpha=alpha, x=x, recs=[])
    pn = bool(getattr(D, 'pixelnorm', False))
    fr = D.blocks[e].fromRGB
    sb = USE_SIGN_BYTES and not pn                                            # byte copies of the fp32 activations (masks)
    curb = None
    fused = None                                 # (a1, a1b) of the entry block when its c1 ran with fromRGB in the gather
    if not keep_input and FUSE_FROMRGB and sb and r >= SIGN_BYTES_MIN_H and e < nb - 1 and x.is_cuda:
        c1 = D.blocks[e].c1
        if c1.ksize == 3 and c1.pad == 1 and _wino(c1, NB, r, c1.conv.weight.shape[2]) is None:
            try:
                a1, a1b, curb = ops.conv2d_fromrgb(x, fr.conv.weight.data, fr.conv.bias.data, fr.c, fr.slope, c1.conv.weight.data,
                                                   c1.conv.bias.data, NB, C, r, r, c1.c, c1.slope)
                fused, cur = (a1, a1b), None
            except ops.Unsupported:
                FALLBACKS['fromRGB in the gather %dx%d' % (r, r)] += 1
    if fused is not None:
        pass
    elif sb and r >= SIGN_BYTES_MIN_H:
        cur, curb = ops.fromrgb_fwd(x, fr.conv.weight.data, fr.conv.bias.data, NB, C, r, r, fr.c, fr.slope, signs_out=True)
    else:
        cur = ops.fromrgb_fwd(x, fr.conv.weight.data, fr.conv.bias.data, NB, C, r, r, fr.c, fr.slope)
    H = r
    a2 = None
    for k, j in enumerate(range(e, nb)):
        blk = D.blocks[j]
        last = (j == nb - 1)
        rec = dict(blk=blk, inp=cur, H=H, first=(k == 0), last=last)
        if k == 0 and curb is not None:
            rec['inpb'] = curb
        if last:
            mb, stats = _mbstd_fwd(D, cur, groups, blk.c1.cin_store)          # :168
            a1 = _conv(mb, blk.c1, NB, H)
            if pn:
                a1, rec['r1'] = ops.pixelnorm_fwd(a1, inplace=True)
            a2 = _conv(a1, blk.c2, NB, H)                                     # 4x4 pad 0 -> 1x1
            if pn:
                a2, rec['r2'] = ops.pixelnorm_fwd(a2, inplace=True)
            rec.update(mb=mb, stats=stats, a1=a1, a2=a2)
        else:
            if k == 0 and fused is not None:
                a1, rec['a1b'] = fused
            elif sb and H >= SIGN_BYTES_MIN_H:
                a1, a1b = _conv(cur, blk.c1, NB, H, signs_out=True)
                if a1b is not None:
                    rec['a1b'] = a1b
            else:
                a1 = _conv(cur, blk.c1, NB, H)
            if pn:                                                            # a1/a2 hold the NORMALISED outputs
                a1, rec['r1'] = ops.pixelnorm_fwd(a1, inplace=True)
            pf = None
            if k == 0 and alpha < 1.0:                                        # :230-233
                nfr = D.blocks[j + 1].fromRGB
                pf = ops.fromrgb_fwd(x, nfr.conv.weight.data, nfr.conv.bias.data, NB, C, H // 2, H // 2,
                                     nfr.c, nfr.slope, pool=True)
                rec['pf'] = pf
            pa, pb = (alpha, 1.0 - alpha) if pf is not None else (1.0, 0.0)
            if pn:
                a2 = _conv(a1, blk.c2, NB, H)
                a2, rec['r2'] = ops.pixelnorm_fwd(a2, inplace=True)
                cur = ops.avgpool2_fwd(a2, pf, pa, pb)                        # :229,238
            else:                                                             # pool (+ fade-in blend) in the conv epilogue
                a2, cur = _conv_pool(a1, blk.c2, NB, H, other=pf, a=pa, b=pb, y_bytes=USE_SIGN_BYTES and H >= SIGN_BYTES_MIN_H)
            rec.update(a1=a1, a2=a2)
            H //= 2
        ctx['recs'].append(rec)
    s = ops.linear1_fwd(a2, D.linear.weight.data, D.linear.bias.data)         # :239
    return s, ctx


# ---- split forward: the real third of a D step's [real | fake | mixed] batch ahead of the other two thirds ------------------------
# The real images' pass through D needs D's weights only -- final since the D update of the PREVIOUS iteration -- while the fake and mixed
# thirds need G's update.  Trainer therefore evaluates the real third of iteration i + 1 on the second stream under iteration i's G step
# (3 images, latency-bound launches: the G step leaves the chip mostly idle), and the D step then runs its forward on the other two thirds
# only.  Both passes write into ONE set of batched activation tensors (ops.Arena), so the adjoint sweeps stay batched over all 3N images.
class EarlyReal(object):
    """State of a real-third pass: the image buffer [3N,C,r,r] (rows [0,N) filled), the arena with the batched activations, the first
    pass's context, what it was computed with (weights version, stage) and the event that closes it."""

    def __init__(self):
        self.arena = ops.Arena()
        self.x3 = None
        self.key = None
        self.real = None
        self.ctx = self.scores = self.event = self.stamp = None
        self.owner = None                   # weakref to the _ArenaUse token of the D-loss state whose activations live in these buffers


class _ArenaUse(object):
    """Token a D-loss state holds while its saved activations alias the per-network arena: as long as the state is alive and its
    ``backward()`` has not run, another D-loss forward on the same network must not write into those buffers (it allocates fresh
    tensors instead -- two losses then backward, a validation loss between forward and backward)."""
    __slots__ = ('consumed', '__weakref__')

    def __init__(self):
        self.consumed = False


def _arena_free(st):
    tok = st.owner() if st.owner is not None else None
    return tok is None or tok.consumed


EARLY_STATS = {'passes': 0, 'used': 0, 'dropped': 0}


def d_forward_real_third(D, real):
    """First pass (current stream): D on the real images, outputs in rows [0, N) of the batched tensors.  Buffers are kept per (network,
    stage, shape): launch plans bake their addresses, and every reader of the previous iteration is ordered before this writer (the
    caller enqueues it behind D's update, which is behind every weight gradient of that iteration)."""
    ops.require_gpu()
    given = real
    real = _check_dev(real, 'real images')
    N = real.shape[0]
    key = (int(D.depth), tuple(real.shape))
    st = D.__dict__.get('_early_buffers')                   # (owned by the network: one stage at a time, a step's worth of activations)
    if st is None or st.key != key:
        st = D._early_buffers = EarlyReal()
        st.key = key
        st.x3 = torch.empty((3 * N,) + tuple(real.shape[1:]), device=real.device, dtype=torch.float32)
    ops.axpby_mask(real, a=1.0, out=st.x3[:N])
    with st.arena.pass_(0):
        st.scores, st.ctx = d_forward(D, st.x3[:N], groups=1)
    EARLY_STATS['passes'] += 1
    st.real = given if given.is_contiguous() else None       # (identity of the caller's tensor: what the D step will be handed)
    st.stamp = (D._param_version, int(D.depth), float(D.alpha))
    return st


def early_real_on_side(D, real):
    """Enqueue the real-third pass of the NEXT D step on the second stream, behind everything queued there (D's deferred update) and
    behind the upload of ``real`` (the current stream has waited for it).  Leaves the state in ``D._early_real``."""
    main = torch.cuda.current_stream(torch._C._cuda_getDevice())
    side = _side_stream()
    _wait_stream(side, main)
    pend, D._pending = D.__dict__.get('_pending'), None      # (the deferred update's event stays for the MAIN stream: this pass is behind the update in stream order)
    try:
        with torch.cuda.stream(side):
            st = d_forward_real_third(D, real)
            st.event = torch.cuda.Event()
            _record_event(st.event, side)
    finally:
        D._pending = pend
    real.record_stream(side)
    D._early_real = st
    return st


def _merge_ctx(D, first, rest, x3, N, third=None):
    """The context of the whole batch from the contexts of the two (three) passes: every tensor of a pass is a row range of a batched tensor."""
    def whole(a, *bs):
        base = a._base if a._base is not None else a
        if all(b._base is not None and b._base is base for b in bs) and base.shape[0] == a.shape[0] + sum(b.shape[0] for b in bs):
            return base
        raise RuntimeError('split D forward: the passes did not write into one tensor')
    ctx = dict(NB=3 * N, groups=3, depth=first['depth'], alpha=first['alpha'], x=x3, recs=[])
    others = [rest['recs']] + ([third['recs']] if third is not None else [])
    for i, ra in enumerate(first['recs']):
        rec = {}
        for k, v in ra.items():
            rec[k] = whole(v, *[o[i][k] for o in others]) if torch.is_tensor(v) else v
        ctx['recs'].append(rec)
    return ctx


def _slice_ctx(ctx, a, b, g0, g1):
    """View of a batched context restricted to images [a,b) == groups [g0,g1)."""
    sub = dict(NB=b - a, groups=g1 - g0, depth=ctx['depth'], alpha=ctx['alpha'], x=ctx['x'][a:b], recs=[])
    for rec in ctx['recs']:
        r2 = dict(rec)
        for k in ('inp', 'a1', 'a2', 'mb', 'pf', 'inpb', 'a1b'):
            if k in rec:
                r2[k] = rec[k][a:b]
        for k in ('r1', 'r2'):                       # PixelNorm scales, one per pixel: [NB*h*w]
            if k in rec:
                per = rec[k].numel() // ctx['NB']
                r2[k] = rec[k][a * per:b * per]
        if 'stats' in rec:
            r2['stats'] = rec['stats'][g0:g1]
        sub['recs'].append(r2)
    return sub


def d_backward(D, ctx, gscore, full, want_gimg, save_adjoints=False, hvp=None):
    """First-order adjoint sweep through D for the batch in ``ctx``.

    gscore [NB] : d loss / d score.   full: also accumulate weight/bias gradients.
    want_gimg   : return d loss / d input image (NCHW).
    hvp         : (n_head, tx, tstats, gy_first) — the last ``NB-n_head`` images form one extra group
                  whose score gradient is zero and which only receives the minibatch-stddev
                  Hessian-vector injection; ``gscore`` then has n_head entries."""
    wait_pending(D)
    if getattr(D, 'pixelnorm', False):
        return _d_backward_pn(D, ctx, gscore, full, want_gimg, save_adjoints, hvp)
    D._ensure_buffers()
    NB, alpha = ctx['NB'], ctx['alpha']
    x = ctx['x']
    C = D.num_channels
    recs = ctx['recs']
    adj = [dict() for _ in recs]
    lastrec = recs[-1]
    nh = NB if hvp is None else hvp[0]
    a2 = lastrec['a2']
    lc2 = lastrec['blk'].c2
    tail = _tail_wgrad_on_main(recs[0]['H']) if (full and ASYNC_WGRAD and getattr(D, '_grad_hook', None) is None and len(recs) > 1) else 0
    if full:
        ops.linear1_wgrad(gscore, a2[:nh], D._lin_gw, D._lin_gb)
    g = ops.linear1_bwd_data(gscore, D.linear.weight.data, a2[:nh], (nh,) + tuple(a2.shape[1:]), lc2.slope)
    gimg = None
    gimg_fused = fw_fused = False                # the entry block's backward-data conv wrote the image gradient / accumulated fromRGB's weight gradient itself
    pending_prev = None
    carry = None
    for idx in range(len(recs) - 1, -1, -1):
        rec = recs[idx]
        blk, H = rec['blk'], rec['H']
        c1, c2 = blk.c1, blk.c2
        fr_slope = blk.fromRGB.slope
        if rec['last']:
            gz2 = g                                                           # [nh,1,1,C]
            a1, mb, inp = rec['a1'], rec['mb'], rec['inp']
            if full:
                _wgrad(a1[:nh], gz2, c2, nh, H)
            gz1 = _dgrad(D, gz2, c2, nh, 1, mask=a1[:nh], mask_slope=c1.slope)
            if full:
                _wgrad(mb[:nh], gz1, c1, nh, H)
            gmb = _dgrad(D, gz1, c1, nh, H)                                   # [nh,4,4,CP]
            cp = c1.cin_store
            if hvp is None:
                gin = _mbstd_bwd(D, gmb, inp, rec['stats'], cp, rec['first'], fr_slope)
            else:
                _, tx, tstats, gy_first = hvp
                gin = torch.empty_like(inp)
                ng = rec['stats'].shape[0] - 1
                _mbstd_bwd(D, gmb, inp[:nh], rec['stats'][:ng], cp, rec['first'], fr_slope, out=gin[:nh])
                _mbstd_bwd(D, None, inp[nh:], rec['stats'][ng:], cp, rec['first'], fr_slope,
                           tx=tx, tstats=tstats, gy_first=gy_first, out=gin[nh:])
            if save_adjoints:
                adj[idx].update(gz2=gz2, gz1=gz1, gmb=gmb)
        else:
            gz2 = g
            if carry is not None:                  # gz2 = pool adjoint of the coarser gradient, evaluated inside the two consumers
                gc, gb, gmul, gsl = carry
                carry = None
                try:
                    gz1 = ops.conv2d_unpooled(gc, _wt(D, c2), gb, gmul, gsl, NB, H, H, c2.c,
                                              mask=rec.get('a1b') if rec.get('a1b') is not None else rec['a1'], mask_slope=c1.slope)
                    if full:
                        _flush_wgrad(c2)           # (this launch does not go through _wgrad: a deferred tangent term rides nowhere)
                        with (_on_main if rec['first'] and tail >= 3 else _on_side)(rec['a1'], gc, gb):
                            ops.conv2d_wgrad_unpooled(rec['a1'], gc, gb, gmul, gsl, c2._gw, c2._gb, NB, H, H, c2.c)
                except ops.Unsupported:                    # (shape checks are identical for both entry points: nothing was accumulated)
                    FALLBACKS['lazy unpool %dx%d' % (H, H)] += 1
                    gz2 = ops.avgpool2_bwd(gc, _mask32(gb), 4.0 * gmul, gsl)
                    if full:
                        _wgrad(rec['a1'], gz2, c2, NB, H)
                    gz1 = _dgrad(D, gz2, c2, NB, H, mask=(rec['a1'], rec.get('a1b')), mask_slope=c1.slope)
            else:
                if full:
                    _wgrad(rec['a1'], gz2, c2, NB, H)
                gz1 = _dgrad(D, gz2, c2, NB, H, mask=(rec['a1'], rec.get('a1b')), mask_slope=c1.slope)
            if full:
                _wgrad(rec['inp'], gz1, c1, NB, H, on_main=rec['first'] and tail >= 2)
            g_fused = None
            if not rec['first'] and not (recs[idx - 1]['first'] and alpha < 1.0):
                pv = recs[idx - 1]
                pc2w = pv['blk'].c2.conv.weight.shape
                if (USE_LAZY_UNPOOL and not save_adjoints and pv['a2'].dtype == torch.uint8 and pc2w[3] == 8 and pc2w[2] in (8, 16)
                        and pv['blk'].c2.ksize == 3 and pv['H'] % 32 == 0):
                    # plain coarse gradient; the finer block's c2 consumers apply the pool adjoint in their gathers
                    carry = (_dgrad(D, gz1, c1, NB, H), pv['a2'], 0.25, pv['blk'].c2.slope)
                else:
                    # backward-data conv + pool adjoint + LeakyReLU' of the finer block's output in one kernel
                    g_fused = _dgrad_unpool(D, gz1, c1, NB, H, pv['a2'], 1.0, pv['blk'].c2.slope)
                gin = None
            else:
                gin = None
                # fromRGB's weight gradient rides in the same epilogue when the sweep asks for weight gradients -- not under a bucketed
                # gradient exchange, whose flushes wait for the weight-gradient stream only (as the tail launches on the main stream)
                fw = full and FUSE_FROMRGB_WGRAD and getattr(D, '_grad_hook', None) is None
                if (rec['first'] and (want_gimg or fw) and FUSE_FROMRGB_BWD and rec.get('inpb') is not None and x.is_cuda and c1.ksize == 3 and c1.pad == 1
                        and _wino(c1, NB, H, c1.conv.weight.shape[3], transposed=True) is None):
                    # the entry block's backward-data conv hands the IMAGE gradient on as well (fromRGB's backward-data in its epilogue); the
                    # 8-channel gradient itself is written only when somebody reads it afterwards (the tangent term of the gradient penalty)
                    fr0 = blk.fromRGB
                    try:
                        gin, gi = ops.conv2d_masked_fromrgb_bwd(gz1, _wt(D, c1), rec['inpb'], fr_slope, fr0.conv.weight.data, fr0.c, NB, C, H, H, c1.c,
                                                                keep_gf=save_adjoints or (full and not fw), want_gimg=want_gimg,
                                                                img=x if fw else None, rgb_dw=fr0._gw if fw else None, rgb_db=fr0._gb if fw else None)
                        if want_gimg:
                            gimg = gi
                        gimg_fused, fw_fused = True, fw
                    except ops.Unsupported:
                        FALLBACKS['fromRGB adjoint in the epilogue %dx%d' % (H, H)] += 1
                if not gimg_fused:
                    gin = _dgrad(D, gz1, c1, NB, H, mask=(rec['inp'], rec.get('inpb')) if rec['first'] else None, mask_slope=fr_slope)
            if save_adjoints:
                adj[idx].update(gz2=gz2, gz1=gz1)
        if rec['first']:
            gf = gin                                                          # adjoint of fromRGB pre-activation
            fr = blk.fromRGB
            if save_adjoints:
                adj[idx]['gf'] = gf
            if full and not fw_fused:
                with (_on_main if tail >= 1 else _on_side)(gf, x):
                    ops.fromrgb_wgrad(gf, x, fr._gw, fr._gb, NB, C, H, H, fr.c)
            if want_gimg:
                if not gimg_fused:
                    gimg = torch.empty_like(x)
                    ops.fromrgb_bwd_data(gf, fr.conv.weight.data, gimg, NB, C, H, H, fr.c)
                if pending_prev is not None:
                    gpf, pfr = pending_prev
                    ops.fromrgb_bwd_data(gpf, pfr.conv.weight.data, gimg, NB, C, H // 2, H // 2, pfr.c,
                                         pool=True, accumulate=True)
        else:
            prev = recs[idx - 1]
            pc2 = prev['blk'].c2
            if prev['first'] and alpha < 1.0:
                pc2w = pc2.conv.weight.shape
                if (USE_LAZY_UNPOOL and USE_LAZY_UNPOOL_FADE and not save_adjoints and prev['a2'].dtype == torch.uint8 and pc2w[3] == 8 and pc2w[2] in (8, 16)
                        and pc2.ksize == 3 and prev['H'] % 32 == 0):
                    # fade-in at the 1024^2 stage: the same lazy pool adjoint as in the fully grown stage (x alpha); the entry block's c2
                    # consumers evaluate it in their gathers instead of reading a 604 MB fine-resolution gradient (round 4)
                    carry = (gin, prev['a2'], 0.25 * alpha, pc2.slope)
                    g = None
                else:
                    g = ops.avgpool2_bwd(gin, _mask32(prev['a2']), alpha, pc2.slope)
                pfr = blk.fromRGB                                             # the block whose fromRGB fed the fade-in
                gpf = ops.axpby_mask(gin, mask=prev['pf'], a=1.0 - alpha, mask_slope=pfr.slope)
                if save_adjoints:
                    adj[idx - 1]['gpf'] = gpf
                if full:
                    with _on_side(gpf, x):
                        ops.fromrgb_wgrad(gpf, x, pfr._gw, pfr._gb, NB, C, H, H, pfr.c, pool=True)
                pending_prev = (gpf, pfr)
            elif not rec['last'] and g_fused is not None:
                g = g_fused
            elif carry is not None:
                g = None                            # consumed through ``carry`` by the next (finer) block
            else:
                g = ops.avgpool2_bwd(gin, _mask32(prev['a2']), 1.0, pc2.slope)
        if full:
            _grads_ready(D, _d_block_done(D, recs, idx, alpha))
    return gimg, adj


def _d_block_done(D, recs, idx, alpha):
    """Layers whose weight gradients are complete once block ``idx`` of the batched backward sweep has been processed
    (the tangent pass ran before the sweep, so nothing else accumulates into them)."""
    rec = recs[idx]
    blk = rec['blk']
    done = [blk.c1, blk.c2]
    if rec['last']:
        done.append(D._lin_layer)
    if rec['first'] or (recs[idx - 1]['first'] and alpha < 1.0):
        done.append(blk.fromRGB)                 # the entry block's fromRGB / the fade-in branch's fromRGB
    return done


def _pn_bwd(gy, y, r, slope, nh, inj):
    """(LeakyReLU -> PixelNorm) adjoint on a batch whose images [nh:] additionally receive the
    gradient-penalty Hessian-vector injection ``inj`` (None: plain adjoint on the whole batch)."""
    if inj is None:
        return ops.pixelnorm_lrelu_bwd(gy, y, r, slope, inplace=True)
    per = r.numel() // y.shape[0]
    if nh > 0:
        ops.pixelnorm_lrelu_bwd(gy[:nh], y[:nh], r[:nh * per], slope, inplace=True)
    ops.pixelnorm_lrelu_bwd(gy[nh:], y[nh:], r[nh * per:], slope, inj=inj, out=gy[nh:])
    return gy


def _d_backward_pn(D, ctx, gscore, full, want_gimg, save_adjoints=False, hvp=None):
    """``d_backward`` for Discriminator(pixelnorm=True): every c1/c2 is conv -> LeakyReLU -> PixelNorm
    (network.py:32-41), so the masks can no longer be fused into the backward-data convs; the adjoint of each
    (LeakyReLU, PixelNorm) pair is one ``pixelnorm_lrelu_bwd`` launch.  With ``hvp`` the mixed images [nh:]
    take the minibatch-stddev AND the per-layer PixelNorm Hessian-vector injections (hvp[4] = list of dicts)."""
    D._ensure_buffers()
    NB, alpha = ctx['NB'], ctx['alpha']
    x = ctx['x']
    C = D.num_channels
    recs = ctx['recs']
    adj = [dict() for _ in recs]
    lastrec = recs[-1]
    nh = NB if hvp is None else hvp[0]
    injs = None if hvp is None else hvp[4]
    a2 = lastrec['a2']
    gtop = ops.linear1_bwd_data(gscore, D.linear.weight.data, None, (nh,) + tuple(a2.shape[1:]))
    if nh < NB:                                      # mixed images: zero score gradient, injections only
        g = torch.empty_like(a2)                     # (C-ABI launches, not ATen ops: a launch plan replays only those)
        ops.zero_(g[nh:])
        ops.axpby_mask(gtop, a=1.0, out=g[:nh])
    else:
        g = gtop
    if full:
        ops.linear1_wgrad(gscore, a2[:nh], D._lin_gw, D._lin_gb)
    gimg = None
    pending_prev = None
    for idx in range(len(recs) - 1, -1, -1):
        rec = recs[idx]
        blk, H = rec['blk'], rec['H']
        c1, c2 = blk.c1, blk.c2
        fr_slope = blk.fromRGB.slope
        inj = injs[idx] if injs is not None else {}
        if save_adjoints:
            adj[idx]['gy2'] = ops.axpby_mask(g, a=1.0)          # device copy through the C-ABI (recorded by a launch plan)
        gz2 = _pn_bwd(g, rec['a2'], rec['r2'], c2.slope, nh, inj.get('inj2'))
        Hc2 = 1 if rec['last'] else H
        if full:
            _wgrad(rec['a1'], gz2, c2, NB, H)
        gy1 = _dgrad(D, gz2, c2, NB, Hc2)
        if save_adjoints:
            adj[idx]['gy1'] = ops.axpby_mask(gy1, a=1.0)
        gz1 = _pn_bwd(gy1, rec['a1'], rec['r1'], c1.slope, nh, inj.get('inj1'))
        if rec['last']:
            mb, inp = rec['mb'], rec['inp']
            if full:
                _wgrad(mb, gz1, c1, NB, H)
            gmb = _dgrad(D, gz1, c1, NB, H)                                   # [NB,4,4,CP]
            cp = c1.cin_store
            if hvp is None:
                gin = _mbstd_bwd(D, gmb, inp, rec['stats'], cp, rec['first'], fr_slope)
            else:
                _, tx, tstats, gy_first = hvp[:4]
                gin = torch.empty_like(inp)
                ng = rec['stats'].shape[0] - 1
                _mbstd_bwd(D, gmb[:nh], inp[:nh], rec['stats'][:ng], cp, rec['first'], fr_slope, out=gin[:nh])
                _mbstd_bwd(D, gmb[nh:], inp[nh:], rec['stats'][ng:], cp, rec['first'], fr_slope,
                           tx=tx, tstats=tstats, gy_first=gy_first, out=gin[nh:])
            if save_adjoints:
                adj[idx].update(gz2=gz2, gz1=gz1, gmb=gmb)
        else:
            if full:
                _wgrad(rec['inp'], gz1, c1, NB, H)
            gin = _dgrad(D, gz1, c1, NB, H, mask=rec['inp'] if rec['first'] else None, mask_slope=fr_slope)
            if save_adjoints:
                adj[idx].update(gz2=gz2, gz1=gz1)
        if rec['first']:
            gf = gin
            fr = blk.fromRGB
            if save_adjoints:
                adj[idx]['gf'] = gf
            if full:
                with _on_side(gf, x):
                    ops.fromrgb_wgrad(gf, x, fr._gw, fr._gb, NB, C, H, H, fr.c)
            if want_gimg:
                gimg = torch.empty_like(x)
                ops.fromrgb_bwd_data(gf, fr.conv.weight.data, gimg, NB, C, H, H, fr.c)
                if pending_prev is not None:
                    gpf, pfr = pending_prev
                    ops.fromrgb_bwd_data(gpf, pfr.conv.weight.data, gimg, NB, C, H // 2, H // 2, pfr.c,
                                         pool=True, accumulate=True)
        else:
            prev = recs[idx - 1]
            if prev['first'] and alpha < 1.0:
                g = ops.avgpool2_bwd(gin, None, alpha)                        # adjoint wrt the NORMALISED a2
                pfr = blk.fromRGB
                gpf = ops.axpby_mask(gin, mask=prev['pf'], a=1.0 - alpha, mask_slope=pfr.slope)
                if save_adjoints:
                    adj[idx - 1]['gpf'] = gpf
                if full:
                    with _on_side(gpf, x):
                        ops.fromrgb_wgrad(gpf, x, pfr._gw, pfr._gb, NB, C, H, H, pfr.c, pool=True)
                pending_prev = (gpf, pfr)
            else:
                g = ops.avgpool2_bwd(gin, None, 1.0)
        if full:
            _grads_ready(D, _d_block_done(D, recs, idx, alpha))
    return gimg, adj


def d_tangent_wgrad(D, sub, adj, u):
    """Gradient-penalty second-order term, steps (i)+(ii): push the seed ``u`` (NCHW, same shape as the
    mixed batch) through the masked linear maps of D and accumulate, per layer,
    dW += c * wgrad(tangent_input, first_backward_adjoint).  Returns the minibatch-stddev HVP inputs."""
    wait_pending(D)
    D._ensure_buffers()
    N, alpha = sub['NB'], sub['alpha']
    C = D.num_channels
    recs = sub['recs']
    rec0 = recs[0]
    fr = rec0['blk'].fromRGB
    H = rec0['H']
    with _on_side(adj[0]['gf'], u):
        ops.fromrgb_wgrad(adj[0]['gf'], u, fr._gw, None, N, C, H, H, fr.c)
    cur = None
    if rec0.get('inpb') is not None:
        try:
            cur = ops.fromrgb_fwd(u, fr.conv.weight.data, None, N, C, H, H, fr.c, 1.0, mask=rec0['inpb'], mask_slope=fr.slope)
        except ops.Unsupported:
            cur = None
    if cur is None:
        cur = ops.fromrgb_fwd(u, fr.conv.weight.data, None, N, C, H, H, fr.c, 1.0, mask=rec0['inp'], mask_slope=fr.slope)
    hvp = None
    t2 = None
    pn = bool(getattr(D, 'pixelnorm', False))
    injs = [dict() for _ in recs] if pn else None
    for idx, rec in enumerate(recs):
        blk, H = rec['blk'], rec['H']
        c1, c2 = blk.c1, blk.c2
        if rec['last']:
            tmb, tstats = _mbstd_tangent(D, rec['inp'], cur, rec['stats'], c1.cin_store)
            hvp = (cur, tstats, adj[idx]['gmb'])
            _wgrad(tmb, adj[idx]['gz1'], c1, N, H, bias=False, defer=True)
            t1 = _conv(tmb, c1, N, H, mask=rec['a1'], bias=False)
            if pn:
                t1, injs[idx]['inj1'] = ops.pixelnorm_tangent(t1, rec['a1'], rec['r1'], adj[idx]['gy1'])
            _wgrad(t1, adj[idx]['gz2'], c2, N, H, bias=False, defer=True)
            t2 = _conv(t1, c2, N, H, mask=rec['a2'], bias=False)
            if pn:
                t2, injs[idx]['inj2'] = ops.pixelnorm_tangent(t2, rec['a2'], rec['r2'], adj[idx]['gy2'])
        else:
            _wgrad(cur, adj[idx]['gz1'], c1, N, H, bias=False, defer=True)
            t1 = _conv(cur, c1, N, H, mask=(rec['a1'], rec.get('a1b')), bias=False)
            if pn:
                t1, injs[idx]['inj1'] = ops.pixelnorm_tangent(t1, rec['a1'], rec['r1'], adj[idx]['gy1'])
            _wgrad(t1, adj[idx]['gz2'], c2, N, H, bias=False, defer=True)
            tpf = None
            if rec['first'] and alpha < 1.0:
                nfr = recs[idx + 1]['blk'].fromRGB
                with _on_side(adj[idx]['gpf'], u):
                    ops.fromrgb_wgrad(adj[idx]['gpf'], u, nfr._gw, None, N, C, H // 2, H // 2, nfr.c, pool=True)
                tpf = ops.fromrgb_fwd(u, nfr.conv.weight.data, None, N, C, H // 2, H // 2, nfr.c, 1.0,
                                      pool=True, mask=rec['pf'], mask_slope=nfr.slope)
            pa, pb = (alpha, 1.0 - alpha) if tpf is not None else (1.0, 0.0)
            if pn:
                t2 = _conv(t1, c2, N, H, mask=rec['a2'], bias=False)
                t2, injs[idx]['inj2'] = ops.pixelnorm_tangent(t2, rec['a2'], rec['r2'], adj[idx]['gy2'])
                cur = ops.avgpool2_fwd(t2, tpf, pa, pb)
            else:                                                 # only the pooled tangent is needed downstream
                t2, cur = _conv_pool(t1, c2, N, H, bias=False, mask=rec['a2'], other=tpf, a=pa, b=pb, pool_only=True)
    # Linear: d/dw <ones, w . t2> = sum_n t2[n]
    ops.linear1_wgrad(_ones(N, u.device), t2, D._lin_gw, None)
    return hvp + (injs,) if pn else hvp


def d_active_params(D, depth, alpha):
    """Parameters that take part at (depth, alpha) — the ones autograd would give a gradient."""
    nb = len(D.blocks)
    e = nb - 1 - depth
    layers = [D.blocks[e].fromRGB]
    if depth > 0 and alpha < 1.0:
        layers.append(D.blocks[e + 1].fromRGB)
    for j in range(e, nb):
        layers += [D.blocks[j].c1, D.blocks[j].c2]
    return layers


def d_exchange_layers(D, depth, alpha):
    """Everything that receives a gradient in a D step at (depth, alpha): the active conv / fromRGB layers + linear."""
    D._ensure_buffers()
    return d_active_params(D, depth, alpha) + [D._lin_layer]


def g_exchange_layers(G, depth, alpha):
    """Layers that receive a gradient in a G step at (depth, alpha) (the ``active`` list of generator_backward)."""
    G._ensure_buffers()
    layers = [G.block0.c1, G.block0.c2]
    for i in range(depth):
        layers += [G.blocks[i].c1, G.blocks[i].c2]
    layers.append(G.blocks[depth - 1].toRGB if depth > 0 else G.block0.toRGB)
    if depth > 0 and alpha < 1.0:
        layers.append(G.blocks[depth - 2].toRGB if depth > 1 else G.block0.toRGB)
    return layers


def discriminator_forward(D, x):
    """Plain ``D(x)`` (network.py:225-240): NCHW image batch -> scores [N,1]."""
    ops.require_gpu()
    x = _check_dev(x, 'D input')
    s, _ = d_forward(D, x, 1)
    return s.view(-1, 1)


# ------------------------------------------------------------------------------------------
# WGAN-GP steps
# ------------------------------------------------------------------------------------------
def _assign_grads(net, layers, linear=False):
    for m in layers:
        m.conv.weight.grad = m._gw
        m.conv.bias.grad = m._gb
    if linear:
        net.linear.weight.grad = net._lin_gw
        net.linear.bias.grad = net._lin_gb


# With the real third already through D (EarlyReal), the fake third of the D step's forward runs on the second stream next to the mixed
# third's forward and the first backward of the gradient penalty (3-image launches that leave the chip partly idle); its scores are
# needed by d_loss only.  Same-box pairs, ms per step off | on: 1024^2 10.52 | 10.47, 512^2 13.82 | 13.74, 256^2 23.42 | 23.39, 128^2 equal,
# 32^2 10.11 | 10.06.  PGGAN_FAKE_SIDE=0: one [fake | mixed] pass on the main stream.
FAKE_THIRD_ON_SIDE = os.environ.get('PGGAN_FAKE_SIDE', '1') == '1'
# The real third on the second stream next to the generator's forward of the SAME step (3-image launches with nothing beside them), the
# fake third behind it on that stream, the mixed third on the main stream: the step time of the look-ahead form (EarlyReal through
# Trainer: 10.36 | 10.36 ms at 1024^2, 13.41 | 13.41 at 512^2, 23.1 | 23.1 at 256^2, 14.57 | 14.46 at 64^2) without drawing the next batch
# early, and for every caller of wgan_gp_D_loss: the D step + gradient penalty alone 7.32 -> 7.21 ms at 1024^2, 9.58 -> 9.34 at 512^2,
# 16.63 -> 16.32 at 256^2.  Default; PGGAN_REAL_SIDE=0 restores the whole-batch forward (and Trainer(early_real_forward=True) /
# PGGAN_EARLY_REAL=1 the look-ahead pass, which takes precedence when one is pending).
REAL_THIRD_IN_STEP = os.environ.get('PGGAN_REAL_SIDE', '1') == '1'


def d_loss_forward(D, G, real, latents, mix, iwass_lambda, iwass_epsilon, iwass_target):
    """Forward half of wgan_gp_D_loss (wgan_gp_loss.py:36-65): three D passes batched as
    [real | fake | mixed], G without graph, and the first backward of the gradient penalty."""
    ops.require_gpu()
    real = _check_dev(real, 'real images')
    latents = _check_dev(latents, 'latents')
    mix = _check_dev(mix, 'mixing factors').view(-1)
    N = real.shape[0]
    D._sync_version()
    early = take_early_real(D, real)
    fake_done = None
    arena_st = early
    if early is not None:
        # the real third is already through D (Trainer, under the previous G step): the other two thirds follow into the same tensors
        x3 = early.x3
        d_step_generator(D, G, latents, x3[N:2 * N])                        # :51-52  (no graph kept)
        ops.gp_mix(x3[:N], x3[N:2 * N], mix, out=x3[2 * N:])                  # :19
        if FAKE_THIRD_ON_SIDE:
            main = torch.cuda.current_stream(torch._C._cuda_getDevice())
            side = _side_stream()
            _wait_stream(side, main)
            with torch.cuda.stream(side):
                with early.arena.pass_(2):
                    s_f, ctx_f = d_forward(D, x3[N:2 * N], groups=1)          # :54
                fake_done = torch.cuda.Event()
                _record_event(fake_done, side)
            _early_g_on_side(D, main, side)
            with early.arena.pass_(3):
                s_m, ctx_m = d_forward(D, x3[2 * N:], groups=1)               # :20
            ctx = _merge_ctx(D, early.ctx, ctx_f, x3, N, third=ctx_m)
        else:
            with early.arena.pass_(1):
                s_rest, ctx_rest = d_forward(D, x3[N:], groups=2)             # :54,20
            ctx = _merge_ctx(D, early.ctx, ctx_rest, x3, N)
        s = early.scores._base if early.scores._base is not None else early.scores
    elif (REAL_THIRD_IN_STEP and float(D.alpha) >= 1.0 and not getattr(D, 'pixelnorm', False) and ASYNC_WGRAD and real.is_cuda
          and hasattr(ops, 'Arena') and D.__dict__.get('_global_stddev') is None     # (exact-global stddev: its collectives stay on one stream, one pass; host tests run the schedules on a CPU emulation of ops: one pass there)
          and _three_pass_buffers(D, real) is not None):
        # the three thirds as three passes into one set of batched tensors (see REAL_THIRD_IN_STEP)
        st = arena_st = D._early_buffers
        x3 = st.x3
        main = torch.cuda.current_stream(torch._C._cuda_getDevice())
        side = _side_stream()
        wait_pending(D)                                                       # on the MAIN stream, before the fork: the side pass's own wait_pending would consume the event there and leave the mixed third un-ordered
        step_start = torch.cuda.Event()
        _record_event(step_start, main)                                       # (everything the previous step left on the main stream)
        probe('D.start')
        d_step_generator(D, G, latents, x3[N:2 * N])                        # :51-52  -- issued first: the host feeds the critical path before the side work
        probe('D.g_fwd_end')
        with torch.cuda.stream(side):
            _wait_event(side, step_start)
            probe('D.side_start')
            ops.axpby_mask(real, a=1.0, out=x3[:N])                           # (the copy into the batched image buffer too, so that the generator starts at once: -0.03 ms)
            real_copied = torch.cuda.Event()
            _record_event(real_copied, side)
            with st.arena.pass_(0):
                s_r, ctx_r = d_forward(D, x3[:N], groups=1)                   # :47
            probe('D.real_end')
        real.record_stream(side)
        _wait_stream(side, main)
        with torch.cuda.stream(side):
            probe('D.fake_start')
            with st.arena.pass_(2):
                s_f, ctx_f = d_forward(D, x3[N:2 * N], groups=1)              # :54
            fake_done = torch.cuda.Event()
            _record_event(fake_done, side)
            probe('D.fake_end')
        _early_g_on_side(D, main, side)
        _wait_event(main, real_copied)
        ops.gp_mix(x3[:N], x3[N:2 * N], mix, out=x3[2 * N:])                  # :19
        probe('D.mix_end')
        with st.arena.pass_(3):
            s_m, ctx_m = d_forward(D, x3[2 * N:], groups=1)                   # :20
        probe('D.mixed_fwd_end')
        ctx = _merge_ctx(D, ctx_r, ctx_f, x3, N, third=ctx_m)
        s = s_r._base if s_r._base is not None else s_r
    else:
        x3 = torch.empty((3 * N,) + tuple(real.shape[1:]), device=real.device, dtype=torch.float32)
        ops.axpby_mask(real, a=1.0, out=x3[:N])                               # device copy (plumbing; a C-ABI launch so that plans.py records it)
        d_step_generator(D, G, latents, x3[N:2 * N])                        # :51-52  (no graph kept)
        ops.gp_mix(x3[:N], x3[N:2 * N], mix, out=x3[2 * N:])                  # :19
        s, ctx = d_forward(D, x3, groups=3)                                   # :47,54,20
    sub = _slice_ctx(ctx, 2 * N, 3 * N, 2, 3)
    gimg, adj = d_backward(D, sub, _ones(N, real.device), full=False, want_gimg=True, save_adjoints=True)  # :25-28
    ss = ops.row_sumsq(gimg)
    gp, u = ops.gp_seed(gimg, ss, iwass_lambda, iwass_target, 1.0 / N)        # :29-31
    probe('D.gp_bwd_end')
    if fake_done is not None:
        _wait_event(torch.cuda.current_stream(torch._C._cuda_getDevice()), fake_done)
    d_cost, d_real_loss, d_fake_loss, gscore = ops.d_loss(s, gp, N, iwass_epsilon)   # :48,55,62
    probe('D.loss_end')
    state = dict(D=D, ctx=ctx, sub=sub, adj=adj, u=u, gscore=gscore, N=N, scores=s, gp=gp)
    if arena_st is not None:
        tok = state['arena_use'] = _ArenaUse()
        state['arena_st'] = arena_st
        arena_st.owner = _weakref.ref(tok)
    return d_cost, d_real_loss, d_fake_loss, state


def _three_pass_buffers(D, real):
    """The per-(network, stage, shape) arena of the three-pass forward, or None when the activations of an earlier D loss of this
    network still live in it (that loss is alive and has not been back-propagated): the caller then takes the allocating one-pass form."""
    key = (int(D.depth), tuple(real.shape))
    st = D.__dict__.get('_early_buffers')
    if st is None or st.key != key:
        st = D._early_buffers = EarlyReal()
        st.key = key
        st.x3 = torch.empty((3 * real.shape[0],) + tuple(real.shape[1:]), device=real.device, dtype=torch.float32)
        return st
    return st if _arena_free(st) else None


def take_early_real(D, real):
    """The real-third pass Trainer left for THIS real batch, if it is still valid (same tensor, same weights, same stage); the current
    stream is ordered behind it.  None: the caller runs the whole batch."""
    st = D.__dict__.pop('_early_real', None)
    if st is None:
        return None
    if (st.real is not real or st.stamp != (D._param_version, int(D.depth), float(D.alpha)) or float(D.alpha) < 1.0
            or getattr(D, 'pixelnorm', False)):
        EARLY_STATS['dropped'] += 1
        return None
    if st.event is not None:
        _wait_event(torch.cuda.current_stream(torch._C._cuda_getDevice()), st.event)
    EARLY_STATS['used'] += 1
    return st


# ---- the G step's generator forward, ahead of time -------------------------------------------------------------------------------------
# The un-traced phase timeline of the 1024^2 step (tools/phase_timeline.py, round 6) shows the main stream busy from the first launch to the
# last -- it IS the critical path -- while the second stream idles under the gradient penalty's first backward and the tangent pass (1.6 ms
# of 3-image launches: the weight gradients of the tangent pass ride in the sweep's launches) and again under the whole first half of the G
# step.  The G step opens with G(z') on fresh latents (trainer.py:103-105) with the SAME generator weights the D step used (G is updated at the
# end of the G step only), so that pass does not have to wait for anything the D step computes: Trainer hands the latents to the D step
# (``request_early_g``), the D step enqueues G(z') -- with its activations kept for the backward -- on the second stream behind the fake
# third, and the G step starts from the finished pass with D's forward.  Same kernels, same inputs, same results; PGGAN_EARLY_G=0 turns it off.
# Same-box pairs, ms per step off | on (round 6): 1024^2 10.344 | 10.263, 10.358 | 10.242, 10.363 | 10.219; 512^2 13.52 | 13.39, 13.53 | 13.37;
# 256^2 23.04 | 22.83; 64^2 (minibatch 16: the chip is full) 14.54 | 14.66; 16^2 5.25 | 5.34 -- on from 256^2 up (the stages whose minibatch
# leaves the chip partly idle); PGGAN_EARLY_G=0 off, =2 at every stage.
EARLY_G_FORWARD = os.environ.get('PGGAN_EARLY_G', '1') != '0'
EARLY_G_MIN_RES = 4 if os.environ.get('PGGAN_EARLY_G', '1') == '2' else int(os.environ.get('PGGAN_EARLY_G_MIN_RES', '256'))
# Two forms: 'side' = on the second stream behind the fake third (the stages whose minibatch leaves the chip partly idle: from 256^2 up);
# 'batched' = ONE generator pass over [z | z'] in place of the D step's G(z) (the launch-bound 4x4 stage: half the generator launches per
# iteration, 1.166 -> 0.997 ms per step).  Same-box pairs, ms per step side | batched: 1024^2 10.31 | 10.55, 10.28 | 10.48, 10.33 | 10.46 (a
# 6-image pass takes 1.50 ms against 1.02 and holds the fake third back); 512^2 13.40 | 13.51; 256^2 23.25 | 23.29; off | batched: 128^2
# 20.05 | 20.23, 64^2 14.66 | 14.77, 16^2 5.24 | 5.42.  PGGAN_EARLY_G_MODE=side / batched forces one form wherever the pass runs at all.
EARLY_G_MODE = os.environ.get('PGGAN_EARLY_G_MODE', 'auto')
EARLY_G_BATCHED_MAX_RES = int(os.environ.get('PGGAN_EARLY_G_BATCHED_MAX_RES', '4'))


def early_g_mode(depth):
    """'side' | 'batched' | None (the G step runs its own generator pass) for a growth stage."""
    if not EARLY_G_FORWARD:
        return None
    res = 4 * 2 ** int(depth)
    if EARLY_G_MODE == 'auto':
        return 'batched' if res <= EARLY_G_BATCHED_MAX_RES else 'side' if res >= EARLY_G_MIN_RES else None
    return EARLY_G_MODE if (res >= EARLY_G_MIN_RES or res <= EARLY_G_BATCHED_MAX_RES) else None
EARLY_G_STATS = {'passes': 0, 'used': 0, 'dropped': 0}


class EarlyG(object):
    """A finished (enqueued) ``generator_forward(G, latents, save=True)``: output image, context, the latents tensor it was computed from
    (identity: what the G loss will be handed), what it was computed with, and the event that closes it on the second stream."""
    __slots__ = ('fake', 'ctx', 'latents', 'stamp', 'event')


def _slice_gctx(ctx, a, b):
    """Images [a, b) of a generator context (every saved tensor is batch-major: activations [n,h,w,c], PixelNorm scales [n*h*w])."""
    n = ctx['N']

    def sl(t):
        if t is None:
            return None
        per = t.shape[0] // n
        return t[a * per:b * per]
    out = dict(N=b - a, depth=ctx['depth'], alpha=ctx['alpha'], recs=[])
    for k in ('zn', 'y1', 'r1', 'y2', 'r2'):
        out[k] = sl(ctx.get(k))
    for rec in ctx['recs']:
        out['recs'].append(dict(blk=rec['blk'], H=rec['H'], inp=sl(rec['inp']), a1=sl(rec['a1']), r1=sl(rec['r1']), a2=sl(rec['a2']), r2=sl(rec['r2'])))
    return out


def d_step_generator(D, G, latents, out):
    """The D step's generator pass G(z) -> ``out`` (wgan_gp_loss.py:51-52, no graph kept).  EARLY_G_MODE 'batched': when Trainer has
    announced the G step's latents z' (``request_early_g``) the two passes run as ONE pass over [z | z'] -- the same weights, twice the
    images per launch, and the <= 64x64 layers of a 3-image pass are latency-bound launches that take 6 images in nearly the same time --
    and the second half (with its activations) is left for the G step as an ``EarlyG``."""
    req = D.__dict__.get('_early_g_request') if early_g_mode(G.depth) == 'batched' else None
    if req is None or req[0] is not G or tuple(req[1].shape) != tuple(latents.shape):
        return generator_forward(G, latents, out=out)
    D.__dict__.pop('_early_g_request', None)
    zg = _check_dev(req[1], 'latents')
    N = latents.shape[0]
    z2 = torch.empty((2 * N,) + tuple(latents.shape[1:]), device=latents.device, dtype=torch.float32)
    ops.axpby_mask(latents, a=1.0, out=z2[:N])
    ops.axpby_mask(zg, a=1.0, out=z2[N:])
    eg = EarlyG()
    eg.fake = torch.empty_like(out)
    (img, _), ctx = generator_forward(G, z2, save=True, pair_out=(out, eg.fake))
    eg.ctx = _slice_gctx(ctx, N, 2 * N)
    eg.event = torch.cuda.Event()
    _record_event(eg.event, torch.cuda.current_stream(torch._C._cuda_getDevice()))
    eg.latents = zg
    eg.stamp = (G._param_version, int(G.depth), float(G.alpha))
    G._early_fwd = eg
    EARLY_G_STATS['passes'] += 1
    return img


def request_early_g(D, G, latents):
    """Trainer, before the D loss: the next G loss of this iteration will be ``wgan_gp_G_loss(G, D, latents)``."""
    D._early_g_request = (G, latents)


def _early_g_on_side(D, main, side):
    """Inside ``d_loss_forward``, right after the fake third was queued on the second stream: the requested generator pass behind it."""
    req = D.__dict__.pop('_early_g_request', None)
    if req is None or early_g_mode(req[0].depth) != 'side':
        return
    G, z = req
    z = _check_dev(z, 'latents')
    with torch.cuda.stream(side):
        probe('D.early_g_start')
        eg = EarlyG()
        eg.fake, eg.ctx = generator_forward(G, z, save=True)
        eg.event = torch.cuda.Event()
        _record_event(eg.event, side)
        probe('D.early_g_end')
    # allocated under the second stream's context, consumed (and eventually freed) on the main stream
    for t in [eg.fake, eg.ctx['zn'], eg.ctx['y1'], eg.ctx['y2'], eg.ctx.get('r1'), eg.ctx.get('r2')] + \
            [r[k] for r in eg.ctx['recs'] for k in ('a1', 'a2', 'r1', 'r2')]:
        if torch.is_tensor(t):
            t.record_stream(main)
    eg.latents = z
    eg.stamp = (G._param_version, int(G.depth), float(G.alpha))
    G._early_fwd = eg
    EARLY_G_STATS['passes'] += 1


def take_early_g(G, latents):
    """The generator pass the D step left for exactly these latents, if it is still valid (same tensor, same weights, same stage); the
    current stream is ordered behind it.  None: the caller runs the pass itself."""
    eg = G.__dict__.pop('_early_fwd', None)
    if eg is None:
        return None
    if eg.latents is not latents or eg.stamp != (G._param_version, int(G.depth), float(G.alpha)):
        EARLY_G_STATS['dropped'] += 1
        return None
    _wait_event(torch.cuda.current_stream(torch._C._cuda_getDevice()), eg.event)
    EARLY_G_STATS['used'] += 1
    return eg


def d_loss_backward(state, scale=1.0):
    """``D_cost.backward()`` (trainer.py:98): tangent pass + batched [real|fake|mixed] adjoint sweep."""
    D, ctx, N = state['D'], state['ctx'], state['N']
    D._ensure_buffers()
    tok = state.get('arena_use')
    if tok is not None:
        if state['arena_st'].owner() is not tok:     # (second line of defence: a retained loss back-propagated again after a later forward)
            raise RuntimeError('the activations of this D loss were overwritten by a later D-loss forward on the same network')
        tok.consumed = True
    ops.zero_(D._flat_grad)
    if scale != 1.0:
        D._grad_hook = None                      # the gradients are rescaled after the sweep: nothing may travel early
    for lay in _live_conv_layers(D):             # (nothing of an aborted earlier step may ride along)
        lay._pending_wgrad = None
    hvp = d_tangent_wgrad(D, state['sub'], state['adj'], state['u'])
    probe('D.tangent_end')
    gs = state['gscore'][:2 * N]
    d_backward(D, ctx, gs, full=True, want_gimg=False, hvp=(2 * N,) + hvp)
    for lay in _live_conv_layers(D):             # (a deferred tangent contribution that no launch of the sweep carried)
        _flush_wgrad(lay)
    probe('D.sweep_end')
    if PROBES is not None and ASYNC_WGRAD and D._flat_param.is_cuda:
        with torch.cuda.stream(_side_stream()):
            probe('D.wgrad_end')
    if not (getattr(D, '_skip_join', False) and scale == 1.0):
        _join_side()         # default: the gradients are complete for whatever the caller does next on this stream
    # (Trainer sets _skip_join when the whole D update follows on the second stream, in order behind the weight
    #  gradients: the main stream then goes straight on to the G step instead of idling under the last, largest
    #  weight-gradient launches of the 1024^2 layers)
    if scale != 1.0:
        ops.axpby_mask(D._flat_grad, a=scale, out=D._flat_grad)
    _assign_grads(D, d_active_params(D, ctx['depth'], ctx['alpha']), linear=True)


def g_loss_forward(G, D, latents):
    """wgan_gp_G_loss (wgan_gp_loss.py:68-74)."""
    ops.require_gpu()
    latents = _check_dev(latents, 'latents')
    D._sync_version()
    G._sync_version()
    probe('G.start')
    eg = take_early_g(G, latents)
    if eg is not None:
        fake, gctx = eg.fake, eg.ctx
    else:
        fake, gctx = generator_forward(G, latents, save=True)
    probe('G.g_fwd_end')
    s, dctx = d_forward(D, fake, 1, keep_input=False)
    g_cost, gscore = ops.g_loss(s)
    probe('G.d_fwd_end')
    return g_cost, dict(G=G, D=D, gctx=gctx, dctx=dctx, gscore=gscore)


def g_loss_backward(state, scale=1.0):
    """``G_cost.backward()`` (trainer.py:111).  D's weight gradients, which the reference computes here
    and throws away at its next D.zero_grad(), are not computed."""
    G, D = state['G'], state['D']
    G._ensure_buffers()
    ops.zero_(G._flat_grad)
    if scale != 1.0:
        G._grad_hook = None
    gimg, _ = d_backward(D, state['dctx'], state['gscore'], full=False, want_gimg=True)
    probe('G.d_bwd_end')
    active = generator_backward(G, state['gctx'], gimg)
    probe('G.g_bwd_end')
    if PROBES is not None and ASYNC_WGRAD and G._flat_param.is_cuda:
        with torch.cuda.stream(_side_stream()):
            probe('G.wgrad_end')
    _join_side()
    if scale != 1.0:
        ops.axpby_mask(G._flat_grad, a=scale, out=G._flat_grad)
    _assign_grads(G, active)
    state['active_g'] = active
