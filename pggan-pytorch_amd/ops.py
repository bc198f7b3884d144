"""Tensor-level wrappers of the C-ABI kernels (include/pggan_hip.h).

PyTorch is used only as the device allocator and stream provider: every function takes fp32
CUDA(HIP) tensors, allocates its output with ``torch.empty`` and launches a hand-written HIP
kernel on the current stream.  No ATen arithmetic, no CPU fallback.
Feature tensors are NHWC ``[N,H,W,C]``; image tensors are NCHW ``[N,C,H,W]``."""
import os as _os

import ctypes

import torch

from . import _lib


def _stream():
    """Raw HIP stream handle of torch's current stream.  torch.cuda.current_stream() costs ~17 us per call (device
    index resolution through is_available() / os.environ) and this is called once per kernel launch (~400 per step)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


# Scratch of the launches that slice K across workgroups (include/pggan_hip.h: pg_set_workspace), one per (device, stream),
# zero-filled, registered with the library the first time a conv is launched on that stream and kept for good.  Streams that are
# being captured into a hipGraph (a new stream per capture) all share ONE more buffer per device, allocated eagerly the first time
# any eager stream gets its own: the engine replays its graphs one after the other, and a buffer allocated inside a capture would
# belong to that graph's memory pool.
WORKSPACE_BYTES = int(_os.environ.get('PGGAN_WORKSPACE_MB', '32')) << 20
_workspaces = {}
_capture_workspace = {}
_no_workspace_streams = set()     # raw handles of streams that must never carry a scratch launch while a graph is captured (engine's side stream)


def workspace_bytes(kind, N, H, W, cin, cout):
    """Scratch the launch of that shape uses when at least as much is registered (pg_workspace_bytes; 0: it never slices).
    kind 0: conv2d_wino, kind 1: the 4x4 valid conv on a 4x4 map."""
    n = ctypes.c_size_t(0)
    _lib.call('pg_workspace_bytes', kind, N, H, W, cin, cout, ctypes.addressof(n))
    return int(n.value)


def _stream_with_workspace():
    """Raw handle of the current stream, with a scratch registered for it (pg_set_workspace).  The scratch carries self-resetting
    tickets, so it belongs to ONE stream; the streams of hipGraph captures share one more per device, which is correct because a
    captured graph of this package has a single branch that launches scratch kernels (the main stream: weight gradients, Adam and
    derived weights on the forked side stream never slice) and graphs replay one after the other -- the first condition is
    enforced here, the second by the engine replaying on one stream."""
    dev = torch._C._cuda_getDevice()
    s = torch._C._cuda_getCurrentRawStream(dev)
    if (dev, s) not in _workspaces:
        ws = None
        capturing = torch.cuda.is_current_stream_capturing()
        if WORKSPACE_BYTES > 0:
            if capturing:
                if s in _no_workspace_streams:
                    raise RuntimeError('a scratch (K-sliced) launch on the forked side stream of a hipGraph capture: the capture '
                                       'streams of a device share one scratch, so only one branch of a graph may slice')
                ws = _capture_workspace.get(dev)             # (None before the first eager launch: this launch runs unsplit)
            else:
                ws = torch.zeros(WORKSPACE_BYTES, dtype=torch.uint8, device='cuda:%d' % dev)
                if dev not in _capture_workspace:
                    _capture_workspace[dev] = torch.zeros(WORKSPACE_BYTES, dtype=torch.uint8, device='cuda:%d' % dev)
        if ws is not None:
            _lib.call('pg_set_workspace', s, ws.data_ptr(), ws.numel())
        if ws is None and capturing:
            return s                                         # not cached: a later capture on a stream with this handle gets the scratch
        _workspaces[(dev, s)] = ws
    return s


def release_capture_workspaces():
    """Forget (and unregister) the scratch entries of capture streams: called when the captured graphs are dropped
    (graphs.clear), whose streams are gone -- a later stream may reuse the raw handle."""
    shared = set(id(t) for t in _capture_workspace.values())
    for (dev, s), ws in list(_workspaces.items()):
        if ws is not None and id(ws) in shared:
            with torch.cuda.device(dev):
                _lib.call('pg_set_workspace', s, None, 0)
            del _workspaces[(dev, s)]


_F32, _U8 = torch.float32, torch.uint8


def _p(t):
    """Device pointer of a checked tensor (None -> NULL)."""
    if t is None:
        return None
    dt = t.dtype
    if not t.is_cuda or (dt is not _F32 and dt is not _U8) or not t.is_contiguous():
        raise ValueError('expected a contiguous fp32 (or sign-byte uint8) device tensor, got %s %s contiguous=%s on %s'
                         % (tuple(t.shape), t.dtype, t.is_contiguous(), t.device))
    return t.data_ptr()


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError('pggan-pytorch_amd needs an MI355X (gfx950) device: the hot path has no CPU fallback')
    _lib.load()


# ------------------------------------------------------------------------------------ output allocation
# engine.d_forward can run as TWO passes over disjoint image ranges that must leave their outputs in ONE set of batched tensors (the
# real third of a D step's [real | fake | mixed] batch is evaluated ahead of the other two thirds): inside an ``Arena`` context the
# forward ops below take their outputs from the arena instead of the allocator.  The first pass (leading extent n) creates every buffer
# with leading extent 3n and gets rows [0, n); the second pass (leading extent 2n) walks the same buffers in the same order and gets rows
# [n, 3n).  Buffers are kept for the next iteration (the launch plans bake their addresses).  Outside a context: torch.empty.
_ARENA = None


def _empty(shape, device=None, dtype=torch.float32):
    a = _ARENA
    if a is None:
        return torch.empty(shape, device=device, dtype=dtype)
    return a.take(tuple(shape), device, dtype)


class Arena(object):
    def __init__(self):
        self.bufs = []
        self.i = 0
        self.part = None

    def take(self, shape, device, dtype):
        lead, rest = shape[0], shape[1:]
        if self.part == 0:
            if self.i == len(self.bufs):
                self.bufs.append(torch.empty((3 * lead,) + rest, device=device, dtype=dtype))
            big = self.bufs[self.i]
            unit = lead
        else:
            if self.i >= len(self.bufs) or (self.part == 1 and lead % 2):
                raise RuntimeError('split D forward: the second pass asks for an output the first pass did not create')
            big = self.bufs[self.i]
            unit = lead // 2 if self.part == 1 else lead
        if tuple(big.shape) != (3 * unit,) + rest or big.dtype != dtype:
            raise RuntimeError('split D forward: output %d differs between the passes (%s %s vs %s x3)' % (self.i, tuple(big.shape), big.dtype, shape))
        self.i += 1
        # part 0: first third | 1: second and third thirds in one pass | 2 / 3: the second / the third third alone
        return big[:unit] if self.part == 0 else big[unit:] if self.part == 1 else big[unit:2 * unit] if self.part == 2 else big[2 * unit:]

    def pass_(self, part):
        return _ArenaPass(self, part)


class _ArenaPass(object):
    def __init__(self, arena, part):
        self.arena, self.part = arena, part

    def __enter__(self):
        global _ARENA
        if _ARENA is not None:
            raise RuntimeError('nested output arenas')
        self.arena.part, self.arena.i = self.part, 0
        _ARENA = self.arena
        return self.arena

    def __exit__(self, *exc):
        global _ARENA
        _ARENA = None
        if exc[0] is None and self.arena.i != len(self.arena.bufs):
            raise RuntimeError('split D forward: the passes created a different number of outputs (%d of %d)' % (self.arena.i, len(self.arena.bufs)))
        return False


# ------------------------------------------------------------------------------------ conv
def _is_bytes(t):
    return t is not None and t.dtype == torch.uint8


FLAG_UPSAMPLE, FLAG_MASK_BYTES, FLAG_Y_BYTES, FLAG_SIGNS_OUT = 1, 2, 4, 8     # PG_FLAG_* of include/pggan_hip.h
Unsupported = _lib.Unsupported


def conv2d(x, w, bias, N, Hin, Win, ks, pad, scale, slope=1.0, mask=None, mask_slope=0.2, ups=False,
           out=None, signs_out=False):
    """x: [N,Hin(/2),Win(/2),Cin]; w packed [ks,ks,Cout,Cin] -> y [N,Hout,Wout,Cout].  A uint8 ``mask`` holds sign bytes;
    ``signs_out`` (forward mode) additionally returns the sign bytes of y: (y, bytes).  Both may raise ops.Unsupported."""
    cout, cin = w.shape[2], w.shape[3]
    ho, wo = Hin + 2 * pad - ks + 1, Win + 2 * pad - ks + 1
    y = out if out is not None else _empty((N, ho, wo, cout), device=x.device, dtype=torch.float32)
    flags = (FLAG_UPSAMPLE if ups else 0) | (FLAG_MASK_BYTES if _is_bytes(mask) else 0)
    sb = None
    if signs_out:
        sb = _empty((N, ho, wo, cout // 4), device=x.device, dtype=torch.uint8)
        mask, flags = sb, flags | FLAG_SIGNS_OUT
    _lib.call('pg_conv2d_nhwc', _p(x), _p(w), _p(bias), _p(mask), _p(y), N, Hin, Win, cin, cout, ks, pad,
              flags, scale, slope, mask_slope, _stream_with_workspace() if ks == 4 else _stream())
    return (y, sb) if signs_out else y


def wino_unpack(u):
    """Device layout of the Winograd-domain weights (8-channel packs, [Cin/8][16][Cout][8], csrc/conv_wino.hip::wino_u_index)
    -> plain [16, Cout, Cin] (tests / diagnostics)."""
    _, cout, cin = u.shape
    return u.reshape(cin // 8, 16, cout, 8).permute(1, 2, 0, 3).reshape(16, cout, cin)


def wino_transform_weights(w, u=None):
    """w packed [3,3,Cout,Cin] -> Winograd-domain weights u, nominal shape [16,Cout,Cin], stored in 8-channel packs (``wino_unpack``)."""
    ks, _, cout, cin = w.shape
    assert ks == 3
    if u is None:
        u = torch.empty((16, cout, cin), device=w.device, dtype=torch.float32)
    _lib.call('pg_wino_transform_weights', _p(w), _p(u), cout, cin, _stream())
    return u


def wino_transform_weights_batched(flat_w, flat_u, layers, transposed=False):
    """layers: [(w element offset in flat_w, u element offset in flat_u, cout, cin), ...]; one launch.  ``transposed``: every
    layer is the backward-data form of a forward layer, computed straight from that layer's parameter (cout / cin are those of
    the backward-data conv, i.e. the forward layer's cin / cout)."""
    import ctypes
    n = len(layers)
    if n == 0:
        return
    woff = (ctypes.c_int64 * n)(*[l[0] for l in layers])
    uoff = (ctypes.c_int64 * n)(*[l[1] for l in layers])
    co = (ctypes.c_int * n)(*[l[2] for l in layers])
    ci = (ctypes.c_int * n)(*[l[3] for l in layers])
    if isinstance(transposed, (list, tuple)):          # per-layer flags: forward and backward-data forms of a network in one launch
        tr = (ctypes.c_int * n)(*[1 if t else 0 for t in transposed]) if any(transposed) else None
    else:
        tr = (ctypes.c_int * n)(*([1] * n)) if transposed else None
    _lib.call('pg_wino_transform_weights_batched', _p(flat_w), _p(flat_u), n, ctypes.cast(woff, ctypes.c_void_p),
              ctypes.cast(uoff, ctypes.c_void_p), ctypes.cast(co, ctypes.c_void_p), ctypes.cast(ci, ctypes.c_void_p),
              ctypes.cast(tr, ctypes.c_void_p) if tr is not None else None, _stream())


def signbytes_to_mask(b):
    """uint8 sign bytes [..., C/4] -> fp32 +1/-1 mask [..., C] (fallback when an entry point does not take sign bytes)."""
    m = torch.empty(tuple(b.shape[:-1]) + (4 * b.shape[-1],), device=b.device, dtype=torch.float32)
    _lib.call('pg_signbytes_to_mask', _p(b), _p(m), b.numel(), _stream())
    return m


def conv2d_wino(x, u, bias, N, H, W, scale, slope=1.0, mask=None, mask_slope=0.2, ups=False, out=None,
                pool=False, other=None, a=1.0, b=0.0, pool_only=False, unpool=False, upmask=None, up_mul=1.0, y_bytes=False,
                signs_out=False):
    """3x3 pad-1 conv on Winograd-domain weights (+ the fused pool / unpool epilogues).  Returns y, (y, ypool) or yup.
    uint8 ``mask`` / ``upmask`` are sign bytes; ``y_bytes`` (with ``pool``) returns the sign bytes of y instead of y."""
    cout, cin = u.shape[1], u.shape[2]
    flags = (FLAG_UPSAMPLE if ups else 0) | (FLAG_MASK_BYTES if _is_bytes(mask) or _is_bytes(upmask) else 0) | (FLAG_Y_BYTES if y_bytes else 0)
    if y_bytes:
        y = _empty((N, H, W, cout // 4), device=x.device, dtype=torch.uint8)
    else:
        y = out if out is not None else _empty((N, H, W, cout), device=x.device, dtype=torch.float32)
    yp = _empty((N, H // 2, W // 2, cout), device=x.device, dtype=torch.float32) if pool else None
    yu = _empty((N, 2 * H, 2 * W, cout), device=x.device, dtype=torch.float32) if unpool else None
    sb = None
    if signs_out:                                  # plain forward launch: (y, sign bytes of y)
        sb = _empty((N, H, W, cout // 4), device=x.device, dtype=torch.uint8)
        mask, flags = sb, flags | FLAG_SIGNS_OUT
    _lib.call('pg_conv2d_wino_nhwc', _p(x), _p(u), _p(bias), _p(mask), _p(y), _p(yp), _p(other), a, b, 1 if pool_only else 0,
              _p(yu), _p(upmask), up_mul, N, H, W, cin, cout, flags, scale, slope, mask_slope, _stream_with_workspace())
    if signs_out:
        return y, sb
    if pool:
        return y, yp
    if unpool:
        return yu
    return y


def conv2d_wino_pixelnorm(x, u, bias, N, H, W, scale, slope, eps=1e-8, ups=False):
    """conv2d_pixelnorm on Winograd-domain weights (3x3 pad 1, at most 32 couts: ops.Unsupported otherwise).  Returns (y, r)."""
    cout, cin = u.shape[1], u.shape[2]
    y = torch.empty((N, H, W, cout), device=x.device, dtype=torch.float32)
    r = torch.empty((N * H * W,), device=x.device, dtype=torch.float32)
    _lib.call('pg_conv2d_wino_pixelnorm_nhwc', _p(x), _p(u), _p(bias), _p(y), _p(r), N, H, W, cin, cout, 1 if ups else 0,
              scale, slope, eps, _stream())
    return y, r


def conv2d_wino_pnbwd(x, u, ysaved, r, N, H, W, scale, slope, pool=False, other=None, a=1.0, b=0.0):
    """Backward-data conv on Winograd-domain weights (+ 2x2 pool blend a * pool + b * other) + the adjoint of the previous layer's
    (LeakyReLU -> PixelNorm), one launch; at most 32 couts (ops.Unsupported otherwise)."""
    cout, cin = u.shape[1], u.shape[2]
    y = torch.empty((N, H // 2, W // 2, cout) if pool else (N, H, W, cout), device=x.device, dtype=torch.float32)
    _lib.call('pg_conv2d_wino_pnbwd_nhwc', _p(x), _p(u), _p(ysaved), _p(r), _p(y), 1 if pool else 0, _p(other), a, b,
              N, H, W, cin, cout, scale, slope, _stream())
    return y


def conv2d_pool(x, w, bias, N, Hin, Win, ks, pad, scale, slope=1.0, mask=None, mask_slope=0.2, other=None, a=1.0, b=0.0,
                pool_only=False, y_bytes=False):
    """conv2d with the following 2x2 average pool (+ fade-in blend a*pool + b*other) fused into the epilogue.
    Returns (y, ypool); with ``pool_only`` the full-resolution y may be left unwritten (do not read it).  A uint8
    ``mask`` holds sign bytes; ``y_bytes`` returns the sign bytes of y instead of y (raises ops.Unsupported when the
    launch cannot fuse -- redo with fp32)."""
    cout, cin = w.shape[2], w.shape[3]
    ho, wo = Hin + 2 * pad - ks + 1, Win + 2 * pad - ks + 1
    flags = (FLAG_MASK_BYTES if _is_bytes(mask) else 0) | (FLAG_Y_BYTES if y_bytes else 0)
    if y_bytes:
        y = _empty((N, ho, wo, cout // 4), device=x.device, dtype=torch.uint8)
    else:
        y = _empty((N, ho, wo, cout), device=x.device, dtype=torch.float32)
    yp = _empty((N, ho // 2, wo // 2, cout), device=x.device, dtype=torch.float32)
    _lib.call('pg_conv2d_pool_nhwc', _p(x), _p(w), _p(bias), _p(mask), _p(y), _p(yp), _p(other), a, b, 1 if pool_only else 0,
              N, Hin, Win, cin, cout, ks, pad, flags, scale, slope, mask_slope, _stream())
    return y, yp


def conv2d_pixelnorm(x, w, bias, N, Hin, Win, ks, pad, scale, slope, eps=1e-8, ups=False):
    """conv -> bias -> LeakyReLU -> PixelNorm in one launch where the tile shape allows it.  Returns (y, r)."""
    cout, cin = w.shape[2], w.shape[3]
    ho, wo = Hin + 2 * pad - ks + 1, Win + 2 * pad - ks + 1
    y = torch.empty((N, ho, wo, cout), device=x.device, dtype=torch.float32)
    r = torch.empty((N * ho * wo,), device=x.device, dtype=torch.float32)
    _lib.call('pg_conv2d_pixelnorm_nhwc', _p(x), _p(w), _p(bias), _p(y), _p(r), N, Hin, Win, cin, cout, ks, pad,
              1 if ups else 0, scale, slope, eps, _stream())
    return y, r


def conv2d_pixelnorm_torgb(x, w, bias, t_w, t_b, N, C, H, W, scale, slope, t_scale, eps=1e-8, out=None):
    """conv2d_pixelnorm (3x3 pad 1) with the block's toRGB layer in the same epilogue.  t_w [C,Cout(,1,1)].  Returns (y, r, img
    [N,C,H,W]); raises ops.Unsupported outside the 8 -> 8 layer of the 1024^2 stage."""
    cout, cin = w.shape[2], w.shape[3]
    y = torch.empty((N, H, W, cout), device=x.device, dtype=torch.float32)
    r = torch.empty((N * H * W,), device=x.device, dtype=torch.float32)
    if out is None:
        out = torch.empty((N, C, H, W), device=x.device, dtype=torch.float32)
    _lib.call('pg_conv2d_pixelnorm_torgb_nhwc', _p(x), _p(w), _p(bias), _p(y), _p(r), _p(t_w), _p(t_b), t_scale, _p(out),
              N, C, H, W, cin, cout, scale, slope, eps, _stream())
    return y, r, out


def conv2d_pnbwd(x, w, ysaved, r, N, Hin, Win, ks, pad, scale, slope):
    """Backward-data conv + adjoint of the previous layer's (LeakyReLU -> PixelNorm) in one launch where possible."""
    cout, cin = w.shape[2], w.shape[3]
    ho, wo = Hin + 2 * pad - ks + 1, Win + 2 * pad - ks + 1
    y = torch.empty((N, ho, wo, cout), device=x.device, dtype=torch.float32)
    _lib.call('pg_conv2d_pnbwd_nhwc', _p(x), _p(w), _p(ysaved), _p(r), _p(y), N, Hin, Win, cin, cout, ks, pad, scale, slope, _stream())
    return y


def conv2d_unpool(x, w, N, Hin, Win, ks, pad, scale, upmask=None, mul=1.0, mask_slope=0.2):
    """Backward-data conv followed by the adjoint of the 2x2 average pool (x0.25*mul, nearest x2) and the
    LeakyReLU' mask of the finer activation, fused.  Returns the fine-resolution gradient [N,2Ho,2Wo,Cout]."""
    cout, cin = w.shape[2], w.shape[3]
    ho, wo = Hin + 2 * pad - ks + 1, Win + 2 * pad - ks + 1
    y = torch.empty((N, ho, wo, cout), device=x.device, dtype=torch.float32)          # scratch (unfused fallback)
    yup = torch.empty((N, 2 * ho, 2 * wo, cout), device=x.device, dtype=torch.float32)
    _lib.call('pg_conv2d_unpool_nhwc', _p(x), _p(w), _p(upmask), _p(y), _p(yup), N, Hin, Win, cin, cout, ks, pad,
              FLAG_MASK_BYTES if _is_bytes(upmask) else 0, scale, mul, mask_slope, _stream())
    return yup


def conv2d_unpooled(g, w, gbytes, gmul, gslope, N, Hin, Win, scale, mask=None, mask_slope=0.2):
    """Backward-data 3x3 conv whose input is the pool adjoint of ``g`` [N,Hin/2,Win/2,Cin] (x gmul, x LeakyReLU' from the sign
    bytes ``gbytes`` [N,Hin,Win,Cin/4]) evaluated in the gather.  w packed [3,3,Cout,Cin]; raises ops.Unsupported."""
    cout, cin = w.shape[2], w.shape[3]
    y = torch.empty((N, Hin, Win, cout), device=g.device, dtype=torch.float32)
    _lib.call('pg_conv2d_unpooled_nhwc', _p(g), _p(w), _p(gbytes), gmul, gslope, _p(mask), _p(y), N, Hin, Win, cin, cout,
              FLAG_MASK_BYTES if _is_bytes(mask) else 0, scale, mask_slope, _stream())
    return y


def conv2d_fromrgb(img, rgb_w, rgb_b, rgb_scale, rgb_slope, w, bias, N, C, H, W, scale, slope, signs_out=True):
    """c1(fromRGB(img)) of a DBlock in one launch, fromRGB evaluated in the conv's gather (its fp32 output is never written).
    img [N,C,H,W], rgb_w [Cmid,C,1,1] / [Cmid,C], w packed [3,3,Cout,Cmid].  Returns (y, sign bytes of y, sign bytes of fromRGB's
    output); raises ops.Unsupported outside the 8 -> 8 layer of the 1024^2 stage."""
    cout, cmid = w.shape[2], w.shape[3]
    y = _empty((N, H, W, cout), device=img.device, dtype=torch.float32)
    yb = _empty((N, H, W, cout // 4), device=img.device, dtype=torch.uint8) if signs_out else None
    xb = _empty((N, H, W, cmid // 4), device=img.device, dtype=torch.uint8)
    _lib.call('pg_conv2d_fromrgb_nhwc', _p(img), _p(rgb_w), _p(rgb_b), rgb_scale, rgb_slope, _p(xb), _p(w), _p(bias), _p(y), _p(yb),
              N, C, H, W, cmid, cout, scale, slope, _stream())
    return y, yb, xb


def conv2d_masked_fromrgb_bwd(gz, wt, mask_bytes, mask_slope, rgb_w, rgb_scale, N, C, H, W, scale, keep_gf=True, gimg=None, want_gimg=True,
                              img=None, rgb_dw=None, rgb_db=None):
    """Backward-data conv of a DBlock's c1 (x LeakyReLU' from the sign bytes of fromRGB's output) with fromRGB's backward-data
    (``want_gimg``) and / or fromRGB's weight + bias gradient (``img``, ``rgb_dw``, ``rgb_db``: accumulated into) in the epilogue.
    wt: flipped / transposed weights [3,3,Cin',Cout'].  Returns (gf [N,H,W,8] or None, gimg [N,C,H,W] or None); raises ops.Unsupported
    outside the 8 -> 8 layer of the 1024^2 stage."""
    cout, cin = wt.shape[2], wt.shape[3]
    gf = _empty((N, H, W, cout), device=gz.device, dtype=torch.float32) if keep_gf else None
    if gimg is None and want_gimg:
        gimg = torch.empty((N, C, H, W), device=gz.device, dtype=torch.float32)
    _lib.call('pg_conv2d_masked_fromrgb_bwd_nhwc', _p(gz), _p(wt), _p(mask_bytes), mask_slope, _p(gf), _p(rgb_w), rgb_scale, _p(gimg),
              _p(img), _p(rgb_dw), _p(rgb_db), N, C, H, W, cin, cout, scale, _stream())
    return gf, gimg



def conv2d_wgrad_unpooled(x, g, gbytes, gmul, gslope, dw, db, N, Hin, Win, scale):
    """Weight gradient with gz = pool adjoint of ``g`` evaluated in the gather (see conv2d_unpooled)."""
    cout, cin = dw.shape[2], dw.shape[3]
    _lib.call('pg_conv2d_wgrad_unpooled_nhwc', _p(x), _p(g), _p(gbytes), gmul, gslope, _p(dw), _p(db), N, Hin, Win, cin, cout,
              scale, _stream())


def conv2d_wgrad(x, gz, dw, db, N, Hin, Win, ks, pad, scale, ups=False):
    """Accumulates into dw [ks,ks,Cout,Cin] (and db [Cout] if given)."""
    cout, cin = dw.shape[2], dw.shape[3]
    _lib.call('pg_conv2d_wgrad_nhwc', _p(x), _p(gz), _p(dw), _p(db), N, Hin, Win, cin, cout, ks, pad,
              1 if ups else 0, scale, _stream())


def conv2d_wgrad_wino(x, gz, dw, db, N, H, W, scale, ups=False, second=None):
    """Winograd form of conv2d_wgrad for 3x3 pad-1 layers: accumulates into dw [3,3,Cout,Cin] (and db).
    ``second`` = (x2, gz2, N2, bias2): another batch of the same layer summed in the same launch (one commit of dW);
    its gz joins db when ``bias2``."""
    cout, cin = dw.shape[2], dw.shape[3]
    if second is None:
        _lib.call('pg_conv2d_wgrad_wino_nhwc', _p(x), _p(gz), _p(dw), _p(db), N, H, W, cin, cout, 1 if ups else 0, scale, _stream())
        return
    x2, gz2, n2, bias2 = second
    _lib.call('pg_conv2d_wgrad_wino2_nhwc', _p(x), _p(gz), N, _p(x2), _p(gz2), n2, _p(dw), _p(db),
              (1 if db is not None else 0) | (2 if (bias2 and db is not None) else 0), H, W, cin, cout, 1 if ups else 0, scale, _stream())


def pack_dgrad_weights(w, wt):
    ks, _, cout, cin = w.shape
    _lib.call('pg_pack_dgrad_weights', _p(w), _p(wt), ks, cout, cin, _stream())
    return wt


def pack_dgrad_weights_batched(flat_w, flat_wt, layers):
    """layers: [(element offset, ks, cout, cin), ...] inside the flat weight buffer / its mirror; one launch."""
    import ctypes
    n = len(layers)
    off = (ctypes.c_int64 * n)(*[l[0] for l in layers])
    ks = (ctypes.c_int * n)(*[l[1] for l in layers])
    co = (ctypes.c_int * n)(*[l[2] for l in layers])
    ci = (ctypes.c_int * n)(*[l[3] for l in layers])
    _lib.call('pg_pack_dgrad_weights_batched', _p(flat_w), _p(flat_wt), n, ctypes.cast(off, ctypes.c_void_p),
              ctypes.cast(ks, ctypes.c_void_p), ctypes.cast(co, ctypes.c_void_p), ctypes.cast(ci, ctypes.c_void_p), _stream())


# --------------------------------------------------------------------------------- from/toRGB
def fromrgb_fwd(img, w, bias, N, C, H, W, scale, slope, pool=False, mask=None, mask_slope=0.2, signs_out=False):
    cout = w.shape[0]
    y = _empty((N, H, W, cout), device=img.device, dtype=torch.float32)
    flags = (1 if pool else 0) | (FLAG_MASK_BYTES if _is_bytes(mask) else 0)
    sb = None
    if signs_out:
        sb = _empty((N, H, W, cout // 4), device=img.device, dtype=torch.uint8)
        mask, flags = sb, flags | FLAG_SIGNS_OUT
    _lib.call('pg_fromrgb_fwd', _p(img), _p(w), _p(bias), _p(mask), _p(y), N, C, H, W, cout, flags,
              scale, slope, mask_slope, _stream())
    return (y, sb) if signs_out else y


def fromrgb_bwd_data(gz, w, gimg, N, C, H, W, scale, pool=False, accumulate=False):
    cout = w.shape[0]
    _lib.call('pg_fromrgb_bwd_data', _p(gz), _p(w), _p(gimg), N, C, H, W, cout, 1 if pool else 0,
              1 if accumulate else 0, scale, _stream())


def fromrgb_wgrad(gz, img, dw, db, N, C, H, W, scale, pool=False):
    cout = dw.shape[0]
    _lib.call('pg_fromrgb_wgrad', _p(gz), _p(img), _p(dw), _p(db), N, C, H, W, cout, 1 if pool else 0, scale, _stream())


def torgb_fwd(x, w, bias, N, C, H, W, scale, out_mul=1.0, prev=None, prev_mul=0.0, out=None):
    cin = w.shape[1]
    if out is None:
        out = torch.empty((N, C, H, W), device=x.device, dtype=torch.float32)
    _lib.call('pg_torgb_fwd', _p(x), _p(w), _p(bias), _p(prev), _p(out), N, C, H, W, cin, scale, out_mul, prev_mul, _stream())
    return out


def torgb_bwd_data(g, w, N, C, H, W, mul_scale, down=False):
    cin = w.shape[1]
    gx = torch.empty((N, H, W, cin), device=g.device, dtype=torch.float32)
    _lib.call('pg_torgb_bwd_data', _p(g), _p(w), _p(gx), N, C, H, W, cin, 1 if down else 0, mul_scale, _stream())
    return gx


def torgb_bwd_data_pnbwd(g, w, ysaved, r, N, C, H, W, mul_scale, slope):
    """torgb_bwd_data + the adjoint of the block's (LeakyReLU -> PixelNorm), one launch where the kernel exists, two otherwise."""
    cin = w.shape[1]
    gx = torch.empty((N, H, W, cin), device=g.device, dtype=torch.float32)
    try:
        _lib.call('pg_torgb_bwd_data_pnbwd', _p(g), _p(w), _p(ysaved), _p(r), _p(gx), N, C, H, W, cin, mul_scale, slope, _stream())
        return gx
    except Unsupported:
        return pixelnorm_lrelu_bwd(torgb_bwd_data(g, w, N, C, H, W, mul_scale), ysaved, r, slope, inplace=True)


def torgb_wgrad(g, x, dw, db, N, C, H, W, mul_scale, mul, down=False):
    cin = dw.shape[1]
    _lib.call('pg_torgb_wgrad', _p(g), _p(x), _p(dw), _p(db), N, C, H, W, cin, 1 if down else 0, mul_scale, mul, _stream())


# --------------------------------------------------------------------------- pool / upsample
def avgpool2_fwd(x, other=None, a=1.0, b=0.0):
    N, H2, W2, C = x.shape
    y = torch.empty((N, H2 // 2, W2 // 2, C), device=x.device, dtype=torch.float32)
    _lib.call('pg_avgpool2_fwd', _p(x), _p(other), _p(y), N, H2 // 2, W2 // 2, C, a, b, _stream())
    return y


def avgpool2_bwd(gy, mask=None, mul=1.0, mask_slope=0.2):
    N, H, W, C = gy.shape
    gx = torch.empty((N, 2 * H, 2 * W, C), device=gy.device, dtype=torch.float32)
    _lib.call('pg_avgpool2_bwd', _p(gy), _p(mask), _p(gx), N, H, W, C, mul, mask_slope, _stream())
    return gx


def upsample2_bwd(g):
    N, H2, W2, C = g.shape
    gx = torch.empty((N, H2 // 2, W2 // 2, C), device=g.device, dtype=torch.float32)
    _lib.call('pg_upsample2_bwd', _p(g), _p(gx), N, H2 // 2, W2 // 2, C, _stream())
    return gx


def axpby_mask(x, other=None, mask=None, a=1.0, b=0.0, mask_slope=0.2, out=None):
    y = out if out is not None else torch.empty_like(x)
    _lib.call('pg_axpby_mask', _p(x), _p(other), _p(mask), _p(y), x.numel(), a, b, mask_slope, _stream())
    return y


# ---------------------------------------------------------------------------------- pixelnorm
def pixelnorm_fwd(x, eps=1e-8, inplace=False):
    C = x.shape[-1]
    P = x.numel() // C
    y = x if inplace else torch.empty_like(x)
    r = torch.empty((P,), device=x.device, dtype=torch.float32)
    _lib.call('pg_pixelnorm_fwd', _p(x), _p(y), _p(r), P, C, eps, _stream())
    return y, r


def pixelnorm_lrelu_bwd(gy, y, r, slope, inplace=False, inj=None, out=None):
    C = y.shape[-1]
    P = y.numel() // C
    gz = out if out is not None else (gy if inplace else torch.empty_like(gy))
    if inj is None:
        _lib.call('pg_pixelnorm_lrelu_bwd', _p(gy), _p(y), _p(r), _p(gz), P, C, slope, _stream())
    else:
        _lib.call('pg_pixelnorm_lrelu_bwd_inj', _p(gy), _p(y), _p(r), _p(inj), _p(gz), P, C, slope, _stream())
    return gz


def pixelnorm_tangent(t, y, r, a):
    """(ty, inj) of pg_pixelnorm_tangent: tangent through PixelNorm and its Hessian-vector injection."""
    C = y.shape[-1]
    P = y.numel() // C
    ty, inj = torch.empty_like(t), torch.empty_like(t)
    _lib.call('pg_pixelnorm_tangent', _p(t), _p(y), _p(r), _p(a), _p(ty), _p(inj), P, C, _stream())
    return ty, inj


# -------------------------------------------------------------------------------------- mbstd
MBSTD_STATS_STRIDE = 136        # PG_MBSTD_STATS_STRIDE (include/pggan_hip.h): [mu, sigma, reduction workspace]


def mbstd_fwd(x, groups, cp):
    NB, H, W, C = x.shape
    y = _empty((NB, H, W, cp), device=x.device, dtype=torch.float32)
    stats = _empty((groups, MBSTD_STATS_STRIDE), device=x.device, dtype=torch.float32)
    _lib.call('pg_mbstd_fwd', _p(x), _p(y), _p(stats), groups, NB // groups, H * W, C, cp, _stream())
    return y, stats


def mbstd_tangent(x, tx, stats, cp):
    NB, H, W, C = x.shape
    groups = stats.shape[0]
    ty = torch.empty((NB, H, W, cp), device=x.device, dtype=torch.float32)
    tstats = torch.empty((groups, MBSTD_STATS_STRIDE), device=x.device, dtype=torch.float32)
    _lib.call('pg_mbstd_tangent', _p(x), _p(tx), _p(stats), _p(ty), _p(tstats), groups, NB // groups, H * W, C, cp, _stream())
    return ty, tstats


def mbstd_bwd(gy, x, stats, cp, apply_mask, mask_slope=0.2, tx=None, tstats=None, gy_first=None, out=None):
    NB, H, W, C = x.shape
    groups = stats.shape[0]
    gx = out if out is not None else torch.empty_like(x)
    _lib.call('pg_mbstd_bwd', _p(gy), _p(x), _p(stats), _p(tx), _p(tstats), _p(gy_first), _p(gx), groups,
              NB // groups, H * W, C, cp, 1 if apply_mask else 0, mask_slope, _stream())
    return gx


# exact-global mode under data parallelism (engine._mbstd_*: the partial rows / Gs sums travel between these calls)
def mbstd_stats(x, groups):
    NB, H, W, C = x.shape
    stats = torch.empty((groups, MBSTD_STATS_STRIDE), device=x.device, dtype=torch.float32)
    _lib.call('pg_mbstd_stats', _p(x), _p(stats), groups, NB // groups, H * W, C, _stream())
    return stats


def mbstd_write(x, stats, gathered, cp):
    """``gathered``: [nranks, groups, MBSTD_STATS_STRIDE] partial rows of all ranks (rank-major).  Returns (y, stats) with mu / sigma of the
    whole group (all shards) in stats[:, 0:2]."""
    NB, H, W, C = x.shape
    groups = stats.shape[0]
    y = torch.empty((NB, H, W, cp), device=x.device, dtype=torch.float32)
    _lib.call('pg_mbstd_write', _p(x), _p(y), _p(stats), _p(gathered), gathered.shape[0], groups, NB // groups, H * W, C, cp, _stream())
    return y, stats


def mbstd_tangent_stats(x, tx, stats):
    NB, H, W, C = x.shape
    groups = stats.shape[0]
    tstats = torch.empty((groups, MBSTD_STATS_STRIDE), device=x.device, dtype=torch.float32)
    _lib.call('pg_mbstd_tangent_stats', _p(x), _p(tx), _p(stats), _p(tstats), groups, NB // groups, H * W, C, _stream())
    return tstats


def mbstd_tangent_write(tx, tstats, gathered, stats, cp):
    NB, H, W, C = tx.shape
    groups = stats.shape[0]
    ty = torch.empty((NB, H, W, cp), device=tx.device, dtype=torch.float32)
    _lib.call('pg_mbstd_tangent_write', _p(tx), _p(ty), _p(tstats), _p(stats), _p(gathered), gathered.shape[0], groups, NB // groups, H * W, C, cp,
              _stream())
    return ty, tstats


def mbstd_gsum(gy, gy_first, groups, shape, cp):
    """[groups, 2]: this shard's sums of the stddev channel of ``gy`` / ``gy_first`` ([NB,H,W,cp]; either may be None)."""
    NB, H, W, C = shape
    ref = gy if gy is not None else gy_first
    out = torch.zeros((groups, 2), device=ref.device, dtype=torch.float32)
    _lib.call('pg_mbstd_gsum', _p(gy), _p(gy_first), _p(out), groups, NB // groups, H * W, C, cp, _stream())
    return out


def mbstd_bwd_global(gy, x, stats, cp, apply_mask, gsums, nranks, mask_slope=0.2, tx=None, tstats=None, gy_first=None, out=None):
    NB, H, W, C = x.shape
    groups = stats.shape[0]
    gx = out if out is not None else torch.empty_like(x)
    _lib.call('pg_mbstd_bwd_global', _p(gy), _p(x), _p(stats), _p(tx), _p(tstats), _p(gy_first), _p(gx), _p(gsums), nranks, groups,
              NB // groups, H * W, C, cp, 1 if apply_mask else 0, mask_slope, _stream())
    return gx


# ------------------------------------------------------------------------------------- linear
def linear1_fwd(h, w, b):
    N, C = h.shape[0], h.numel() // h.shape[0]
    s = _empty((N,), device=h.device, dtype=torch.float32)
    _lib.call('pg_linear1_fwd', _p(h), _p(w), _p(b), _p(s), N, C, _stream())
    return s


def linear1_bwd_data(gs, w, mask, shape, mask_slope=0.2):
    N = gs.shape[0]
    C = w.numel()
    gh = torch.empty(shape, device=gs.device, dtype=torch.float32)
    _lib.call('pg_linear1_bwd_data', _p(gs), _p(w), _p(mask), _p(gh), N, C, mask_slope, _stream())
    return gh


def linear1_wgrad(gs, h, dw, db):
    N = gs.shape[0]
    C = dw.numel()
    _lib.call('pg_linear1_wgrad', _p(gs), _p(h), _p(dw), _p(db), N, C, _stream())


# ------------------------------------------------------------------------------------ WGAN-GP
def gp_mix(real, fake, m, out=None):
    N = real.shape[0]
    E = real.numel() // N
    if out is None:
        out = torch.empty_like(real)
    _lib.call('pg_gp_mix', _p(real), _p(fake), _p(m), _p(out), N, E, _stream())
    return out


def row_sumsq(g):
    N = g.shape[0]
    E = g.numel() // N
    ss = torch.empty((N,), device=g.device, dtype=torch.float32)
    zero_(ss)
    _lib.call('pg_row_sumsq', _p(g), _p(ss), N, E, _stream())
    return ss


def gp_seed(g, ss, lam, target, inv_n):
    N = g.shape[0]
    E = g.numel() // N
    gp = torch.empty((N,), device=g.device, dtype=torch.float32)
    u = torch.empty_like(g)
    _lib.call('pg_gp_seed', _p(g), _p(ss), _p(gp), _p(u), N, E, lam, target, inv_n, _stream())
    return gp, u


def d_loss(scores, gp, N, eps):
    dev = scores.device
    out = torch.empty((1 + 2 * N,), device=dev, dtype=torch.float32)     # one buffer: a replayed step hands out ONE copy of it (plans.d_step)
    d_cost, d_real_loss, d_fake_loss = out[0], out[1:1 + N].view(N, 1), out[1 + N:].view(N, 1)
    gscore = torch.empty((3 * N,), device=dev, dtype=torch.float32)
    _lib.call('pg_d_loss', _p(scores), _p(gp), _p(d_cost), _p(d_real_loss), _p(d_fake_loss), _p(gscore), N, eps, _stream())
    return d_cost, d_real_loss, d_fake_loss, gscore


def g_loss(scores):
    N = scores.shape[0]
    g_cost = torch.empty((), device=scores.device, dtype=torch.float32)
    gscore = torch.empty((N,), device=scores.device, dtype=torch.float32)
    _lib.call('pg_g_loss', _p(scores), _p(g_cost), _p(gscore), N, _stream())
    return g_cost, gscore


# --------------------------------------------------------------------------------------- misc
def adam(p, g, m, v, lr, beta1, beta2, eps, bc1, bc2_sqrt, grad_scale=1.0):
    _lib.call('pg_adam', _p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, bc1, bc2_sqrt, grad_scale, _stream())


def uniform_(t, seed, offset):
    """Fill the fp32 device tensor with U[0,1) draws: element i = Philox4x32-10(seed, offset, i) (wgan_gp_loss.py:15-17)."""
    require_gpu()
    _lib.call('pg_uniform_f32', _p(t), t.numel(), int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), _stream())
    return t


def zero_(t):
    _lib.call('pg_zero', _p(t), t.numel() * 4, _stream())
    return t


# ------------------------------------------------------------------------- input / output steps
def real_prepare_u8(x_u8, alpha, range_in=(0, 255), range_out=(-1, 1)):
    """uint8 device batch [N,C,H,W] -> faded + range-adjusted fp32 batch (dataset.py:54-67 on the device)."""
    if not x_u8.is_cuda or x_u8.dtype != torch.uint8 or not x_u8.is_contiguous():
        raise ValueError('expected a contiguous uint8 device tensor')
    N, C, H, W = x_u8.shape
    out = torch.empty((N, C, H, W), device=x_u8.device, dtype=torch.float32)
    _lib.call('pg_real_prepare_u8', x_u8.data_ptr(), _p(out), N * C, H, W, float(alpha), float(range_in[0]),
              float(range_in[1]), float(range_out[0]), float(range_out[1]), _stream())
    return out


SOUND_MODES = {'abslog': 0, 'reallog': 1}


def spectrogram_u8(signal, n_fft, hop_length, max_out=255.0, img_mode='abslog'):
    """fp32 device waveform [nsamp] or [nsamp, channels] -> uint8 image (SoundImageDataset.load_file, dataset.py:285-300):
    'abslog' / 'reallog' spectrogram [1, n_fft/2, n_fft/2], or 'raw' [1, 2^s, 2^s] with 2^s = the largest power of two
    <= sqrt(nsamp) (dataset.py:289-291)."""
    require_gpu()
    if not signal.is_cuda or signal.dtype != torch.float32 or not signal.is_contiguous() or signal.dim() not in (1, 2):
        raise ValueError('expected a contiguous float32 device tensor [nsamp] or [nsamp, channels]')
    nsamp = signal.shape[0]
    ch = 1 if signal.dim() == 1 else signal.shape[1]
    if img_mode == 'raw':
        side = 1
        while (2 * side) * (2 * side) <= nsamp:
            side *= 2
        mag = torch.empty((side, side), device=signal.device, dtype=torch.float32)
        _lib.call('pg_mono_f32', _p(signal), nsamp, ch, _p(mag), side * side, _stream())
    else:
        if img_mode not in SOUND_MODES:
            raise ValueError("img_mode must be 'abslog', 'reallog' or 'raw' (got %r)" % (img_mode,))
        side = n_fft // 2
        if 1 + nsamp // hop_length < side:
            raise ValueError('%d samples give %d frames, the %dx%d image needs %d' % (nsamp, 1 + nsamp // hop_length, side, side, side))
        mag = torch.empty((side, side), device=signal.device, dtype=torch.float32)
        _lib.call('pg_stft_image', _p(signal), nsamp, ch, _p(mag), n_fft, hop_length, side, side, SOUND_MODES[img_mode], _stream())
    lohi = torch.empty(2, device=signal.device, dtype=torch.float32)
    _lib.call('pg_minmax_f32', _p(mag), mag.numel(), _p(lohi), _stream())
    out = torch.empty((1, side, side), device=signal.device, dtype=torch.uint8)
    _lib.call('pg_stretch_to_u8', _p(mag), out.data_ptr(), mag.numel(), _p(lohi), float(max_out), _stream())
    return out


def image_grid_u8(images, drange=(-1, 1), up=1):
    """fp32 device images [n,C,h,w] -> uint8 HWC grid (output_postprocess.py:35-62 up to the PIL hand-off)."""
    n, C, h, w = images.shape
    gw = 1
    while gw * gw < n:
        gw += 1
    gh = (n - 1) // gw + 1
    grid = torch.empty((gh * h * up, gw * w * up, C), device=images.device, dtype=torch.uint8)
    _lib.call('pg_image_grid_u8', _p(images), grid.data_ptr(), n, C, h, w, up, float(drange[0]), float(drange[1]), _stream())
    return grid


def pyramid_level_u8(batch_u8, depthdiff, range_in=(0, 255)):
    """uint8 device batch [..., H, W] -> the pyramid level ``depthdiff`` below it (dataset.py:243-250)."""
    require_gpu()
    if batch_u8.dtype != torch.uint8 or not batch_u8.is_cuda:
        raise RuntimeError('pyramid_level_u8 expects a uint8 tensor on the device')
    x = batch_u8.contiguous()
    H, W = x.shape[-2], x.shape[-1]
    st = 2 ** int(depthdiff)
    out = torch.empty(tuple(x.shape[:-2]) + (H // st, W // st), device=x.device, dtype=torch.uint8)
    _lib.call('pg_pyramid_level_u8', x.data_ptr(), out.data_ptr(), x.numel() // (H * W), H, W, int(depthdiff),
              float(range_in[0]), float(range_in[1]), _stream())
    return out
