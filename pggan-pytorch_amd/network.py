"""Generator / Discriminator with the reference's Python surface, executed by HIP kernels.

Mirrors /root/reference/network.py: same class names, constructor arguments, mutable ``.depth`` /
``.alpha``, ``.max_depth``, ``G.latent_size``, module tree and state_dict key names
(``block0.c1.conv.weight`` ...).  What differs is *where the numbers live and who computes*:

  * every parameter of a network is a view into ONE flat fp32 device buffer (``_flat_param``), with
    a parallel flat gradient buffer -> one RCCL all-reduce and one fused Adam launch per segment;
  * conv weights are stored in the kernels' native packed layout ``[KH][KW][Cout][Cin]`` (the
    minibatch-stddev layer's 513 input channels are padded to 528 with zero weights); use
    ``reference_state_dict()`` / ``load_reference_state_dict()`` to exchange tensors in the
    reference's ``[Cout,Cin,KH,KW]`` layout;
  * ``forward`` launches the hand-written kernels of ``csrc/`` through ``engine`` — there is no
    ATen arithmetic and no CPU fallback (a missing extension or a CPU tensor raises).
"""
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

import weakref

MBSTD_PAD = 16        # the 513-channel input of DLastBlock.c1 is stored with 512+16 channels
_FLAT_REGISTRY = {}   # flat parameter buffer base pointer -> weakref(network)  (lets FusedAdam find its net)


def network_of_flat_ptr(ptr):
    ref = _FLAT_REGISTRY.get(ptr)
    return ref() if ref is not None else None


class _ConvParams(nn.Module):
    """Holds ``weight`` / ``bias`` so that state_dict keys read ``<layer>.conv.weight`` (network.py:16)."""

    def __init__(self, weight, bias):
        super(_ConvParams, self).__init__()
        self.weight = nn.Parameter(weight)
        self.bias = nn.Parameter(bias)

    def extra_repr(self):
        return 'packed_weight=%s' % (tuple(self.weight.shape),)


class PGConv2d(nn.Module):
    """Equalized-lr conv + activation + PixelNorm.  reference network.py:7-41.

    ``kind``: 'conv' (NHWC->NHWC implicit-GEMM MFMA kernel), 'fromrgb' (NCHW image -> NHWC),
    'torgb' (NHWC -> NCHW image).  ``cin_store`` >= ch_in is the stored (zero-padded) channel count."""

    def __init__(self, ch_in, ch_out, ksize=3, stride=1, pad=1,
                 pixelnorm=True, wscale=True, act='lrelu', kind='conv', cin_store=None):
        super(PGConv2d, self).__init__()
        assert stride == 1
        conv = nn.Conv2d(ch_in, ch_out, ksize, stride, pad)        # default init first: RNG order of network.py:16
        if wscale:
            nn.init.kaiming_normal_(conv.weight)                    # network.py:13,17
            c = torch.sqrt(torch.mean(conv.weight.data ** 2))       # network.py:19
            conv.weight.data /= c                                   # network.py:20
            self.c = float(c)
        else:
            self.c = 1.0                                            # network.py:22
        self.eps = 1e-8
        self.kind, self.ksize, self.pad = kind, ksize, pad
        self.ch_in, self.ch_out = ch_in, ch_out
        self.cin_store = ch_in if cin_store is None else cin_store
        self.pixelnorm = pixelnorm
        self.act = act
        self.slope = 1.0 if act is None else (0.2 if act == 'lrelu' else 0.0)   # network.py:26-29
        self.conv = _ConvParams(self._pack(conv.weight.data), conv.bias.data.clone())
        if kind == 'conv' and (self.cin_store % 4 or ch_out % 4):
            raise ValueError('channel counts must be multiples of 4 for the MFMA conv kernels '
                             '(got %d -> %d)' % (ch_in, ch_out))
        self._gw = self._gb = self._wt = None

    # ---- layout conversion (host-side utilities, not on the hot path) ----
    def _pack(self, w_ref):
        """reference [Cout,Cin,KH,KW] -> stored layout."""
        if self.kind in ('fromrgb', 'torgb'):
            return w_ref.reshape(w_ref.shape[0], w_ref.shape[1]).contiguous().clone()
        w = w_ref.permute(2, 3, 0, 1)                               # [KH,KW,Cout,Cin]
        if self.cin_store != self.ch_in:
            wp = torch.zeros(w.shape[0], w.shape[1], w.shape[2], self.cin_store, dtype=w.dtype, device=w.device)
            wp[..., :self.ch_in] = w
            return wp
        return w.contiguous().clone()

    def reference_weight(self):
        """Stored layout -> reference [Cout,Cin,KH,KW] (a copy)."""
        w = self.conv.weight.data
        if self.kind in ('fromrgb', 'torgb'):
            return w.reshape(self.ch_out, self.ch_in, 1, 1).clone()
        return w[..., :self.ch_in].permute(2, 3, 0, 1).contiguous()

    def load_reference_weight(self, w_ref, bias, c=None):
        with torch.no_grad():
            self.conv.weight.data.copy_(self._pack(w_ref.to(self.conv.weight.device)))
            self.conv.bias.data.copy_(bias)
        if c is not None:
            self.c = float(c)

    def extra_repr(self):
        return '%d, %d, kernel_size=%d, padding=%d, c=%.6g, act=%s, pixelnorm=%s, kind=%s' % (
            self.ch_in, self.ch_out, self.ksize, self.pad, self.c, self.act, self.pixelnorm, self.kind)

    def __getstate__(self):
        d = self.__dict__.copy()
        d['_gw'] = d['_gb'] = d['_wt'] = None
        d['_wt_ver'] = None
        d['_wu'] = d['_wtu'] = None
        d['_net'] = None
        d['_pending_wgrad'] = None
        return d


class GFirstBlock(nn.Module):
    """reference network.py:44-57."""

    def __init__(self, ch_in, ch_out, num_channels, **layer_settings):
        super(GFirstBlock, self).__init__()
        self.c1 = PGConv2d(ch_in, ch_out, 4, 1, 3, **layer_settings)
        self.c2 = PGConv2d(ch_out, ch_out, **layer_settings)
        self.toRGB = PGConv2d(ch_out, num_channels, ksize=1, pad=0, pixelnorm=False, act=None, kind='torgb')


class GBlock(nn.Module):
    """reference network.py:60-72."""

    def __init__(self, ch_in, ch_out, num_channels, **layer_settings):
        super(GBlock, self).__init__()
        self.c1 = PGConv2d(ch_in, ch_out, **layer_settings)
        self.c2 = PGConv2d(ch_out, ch_out, **layer_settings)
        self.toRGB = PGConv2d(ch_out, num_channels, ksize=1, pad=0, pixelnorm=False, act=None, kind='torgb')


class DBlock(nn.Module):
    """reference network.py:142-154."""

    def __init__(self, ch_in, ch_out, num_channels, **layer_settings):
        super(DBlock, self).__init__()
        # NB (network.py:145): built WITHOUT layer_settings -> always wscale + LeakyReLU(0.2), no pixelnorm
        self.fromRGB = PGConv2d(num_channels, ch_in, ksize=1, pad=0, pixelnorm=False, kind='fromrgb')
        self.c1 = PGConv2d(ch_in, ch_in, **layer_settings)
        self.c2 = PGConv2d(ch_in, ch_out, **layer_settings)


class MinibatchStddev(nn.Module):
    """reference network.py:178-187 (one global scalar; parameter-free; executed by pg_mbstd_*)."""

    def __init__(self):
        super(MinibatchStddev, self).__init__()
        self.eps = 1.0


class DLastBlock(nn.Module):
    """reference network.py:157-171."""

    def __init__(self, ch_in, ch_out, num_channels, **layer_settings):
        super(DLastBlock, self).__init__()
        self.fromRGB = PGConv2d(num_channels, ch_in, ksize=1, pad=0, pixelnorm=False, kind='fromrgb')
        self.stddev = MinibatchStddev()
        self.c1 = PGConv2d(ch_in + 1, ch_in, cin_store=ch_in + MBSTD_PAD, **layer_settings)
        self.c2 = PGConv2d(ch_in, ch_out, 4, 1, 0, **layer_settings)


class _GradViews(object):
    """(_gw, _gb) pair of a parameter group that is not a PGConv2d (the final nn.Linear): lets the gradient exchange
    treat it like a layer."""

    def __init__(self, gw, gb):
        self._gw, self._gb = gw, gb


class _FlatParamsMixin(object):
    """All parameters of the network live in one flat fp32 buffer (+ one flat gradient buffer)."""

    def _layers(self):
        ls = self.__dict__.get('_layer_list')             # (walking the module tree costs ~0.2 ms per call at depth 8; the tree is fixed after construction)
        if ls is None:
            ls = self.__dict__['_layer_list'] = [m for m in self.modules() if isinstance(m, PGConv2d)]
        return ls

    def _flatten(self):
        params = list(self.parameters())
        if not params:
            return
        dev = params[0].device
        # Layout: 3x3/4x4 conv layers (+ linear) in module order first, the to/fromRGB layers last.  At any depth the
        # active conv layers are then ONE contiguous run of the buffer (blocks are ordered by resolution), so
        # FusedAdam updates them with a single launch instead of one per gap left by an inactive RGB layer.
        rgb = set()
        for m in self.modules():
            if isinstance(m, PGConv2d) and m.kind != 'conv':
                rgb.update((id(m.conv.weight), id(m.conv.bias)))
        order = [i for i, p in enumerate(params) if id(p) not in rgb] + [i for i, p in enumerate(params) if id(p) in rgb]
        offs, total = [0] * len(params), 0
        for i in order:
            offs[i] = total
            total += (params[i].numel() + 3) // 4 * 4       # keep every view 16-byte aligned
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(params, offs):
                v = flat[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                p.grad = None
        self._flat_param = flat
        self._flat_offsets = offs
        self._flat_grad = None
        self._flat_wt = None
        for m in self._layers():
            m._gw = m._gb = m._wt = None
            m._wu = m._wtu = None
        self._flat_wu = self._flat_wtu = self._flat_wuu = None
        self._derived_ver = None
        self._param_version = getattr(self, '_param_version', 0) + 1
        for m in self._layers():
            m._wt_ver = None
        _FLAT_REGISTRY[flat.data_ptr()] = weakref.ref(self)

    def _apply(self, fn, recurse=True):
        super(_FlatParamsMixin, self)._apply(fn, recurse)
        self._flatten()
        self.__dict__['_plist'] = None
        return self

    def _ensure_buffers(self):
        """Lazily create the flat gradient buffer and the backward-data weight copies."""
        if self._flat_grad is not None:
            return
        flat = self._flat_param
        self._flat_grad = torch.zeros_like(flat)
        self._flat_wt = torch.zeros_like(flat)
        byptr = {}
        for p, o in zip(self.parameters(), self._flat_offsets):
            byptr[id(p)] = o
        for m in self._layers():
            ow, ob = byptr[id(m.conv.weight)], byptr[id(m.conv.bias)]
            w, b = m.conv.weight, m.conv.bias
            m._gw = self._flat_grad[ow:ow + w.numel()].view(w.shape)
            m._gb = self._flat_grad[ob:ob + b.numel()].view(b.shape)
            if m.kind == 'conv':
                ks, _, co, ci = w.shape
                m._wt = self._flat_wt[ow:ow + w.numel()].view(ks, ks, ci, co)
        # Winograd-domain copies (forward and backward-data) of the wide 3x3 layers: 16/9 of their weights each
        wl = [m for m in self._layers() if m.kind == 'conv' and m.ksize == 3 and m.pad == 1 and
              m.conv.weight.shape[2] % 8 == 0 and m.conv.weight.shape[3] % 8 == 0 and
              max(m.conv.weight.shape[2], m.conv.weight.shape[3]) >= 16]     # a direction needs >= 16 OUTPUT channels (engine._wino)
        total = sum(16 * m.conv.weight.shape[2] * m.conv.weight.shape[3] for m in wl)
        # forward and backward-data forms in ONE buffer (two halves): a single batched launch can derive both (engine._derived)
        self._flat_wuu = torch.zeros(2 * max(total, 4), dtype=torch.float32, device=flat.device)
        self._flat_wu = self._flat_wuu[:max(total, 4)]
        self._flat_wtu = self._flat_wuu[max(total, 4):]
        self._wino_layers = []
        off = 0
        for m in self._layers():
            m._wu = m._wtu = None
            m._net = weakref.ref(self)
        for m in wl:
            co, ci = m.conv.weight.shape[2], m.conv.weight.shape[3]
            n = 16 * co * ci
            m._wu = self._flat_wu[off:off + n].view(16, co, ci)
            m._wtu = self._flat_wtu[off:off + n].view(16, ci, co)
            self._wino_layers.append((m, byptr[id(m.conv.weight)], off))
            off += n
        self._derived_ver = None
        if hasattr(self, 'linear'):
            ow, ob = byptr[id(self.linear.weight)], byptr[id(self.linear.bias)]
            self._lin_gw = self._flat_grad[ow:ow + self.linear.weight.numel()].view(self.linear.weight.shape)
            self._lin_gb = self._flat_grad[ob:ob + 1].view(1)
            self._lin_layer = _GradViews(self._lin_gw, self._lin_gb)

    def zero_grad(self, set_to_none=True):
        """torch>=2 semantics (grads -> None) so that inactive parameters are skipped by Adam, as
        in the reference under the same torch; the flat gradient buffer is cleared by the step."""
        plist = self.__dict__.get('_plist')
        if plist is None:
            plist = self.__dict__['_plist'] = list(self.parameters())
        for p in plist:
            p.grad = None

    def mark_params_changed(self):
        """Called by FusedAdam (raw-pointer updates) so derived weight copies are re-packed."""
        self._param_version += 1

    def _sync_version(self):
        """Also notice parameter updates done by torch itself (torch.optim.*, load_state_dict, ...)."""
        plist = self.__dict__.get('_plist')
        if plist is None:                        # (walking the module tree on every forward is host time the 4x4 .. 32x32 stages feel)
            plist = self.__dict__['_plist'] = list(self.parameters())
        v = sum(p._version for p in plist)
        if v != getattr(self, '_torch_versions_seen', None):
            self._torch_versions_seen = v
            self._param_version += 1

    def __getstate__(self):
        d = self.__dict__.copy()
        d['_flat_grad'] = None
        d['_flat_wt'] = None
        d['_flat_wu'] = d['_flat_wtu'] = d['_flat_wuu'] = None
        d['_wino_layers'] = None
        d['_derived_ver'] = None
        d['_pending'] = None
        d['_skip_join'] = False
        d['_lin_gw'] = d['_lin_gb'] = d['_lin_layer'] = None
        d['_grad_hook'] = d['_grad_exchange'] = None
        d['_global_stddev'] = None               # (a process-group handle: a reloaded network starts in the local-shard mode)
        d['_gs_checked'] = None
        d['_plan_unjoined'] = False
        d['_early_real'] = d['_early_buffers'] = None      # (device buffers / events of the real-third pass: rebuilt on demand)
        d['_early_fwd'] = d['_early_g_request'] = None     # (a generator pass left for the G step / the request for one: engine.EarlyG)
        d['_plist'] = None
        d['_layer_list'] = None
        d['_derived_bwd_ev'] = None
        d['_derived_bwd_waited'] = set()
        d['_derived_ev'] = d['_pending_ev'] = None
        d['_defer_active'] = False
        d['_derived_waited'] = set()
        d['_bwd_wanted'] = False
        return d

    def __setstate__(self, state):
        super(_FlatParamsMixin, self).__setstate__(state)
        if getattr(self, '_flat_param', None) is not None:
            _FLAT_REGISTRY[self._flat_param.data_ptr()] = weakref.ref(self)

    # ---- exchange with the reference layout ----
    def reference_state_dict(self):
        """Parameters in the reference's layout/names plus '<layer>.c' floats."""
        out = OrderedDict()
        for name, m in self.named_modules():
            if isinstance(m, PGConv2d):
                out[name + '.conv.weight'] = m.reference_weight()
                out[name + '.conv.bias'] = m.conv.bias.data.clone()
                out[name + '.c'] = m.c
        if hasattr(self, 'linear'):
            out['linear.weight'] = self.linear.weight.data.clone()
            out['linear.bias'] = self.linear.bias.data.clone()
        return out

    def load_reference_state_dict(self, sd):
        """Inverse of reference_state_dict (values may be numpy arrays or tensors on any device)."""
        def t(v):
            return torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v, dtype=torch.float32)
        for name, m in self.named_modules():
            if isinstance(m, PGConv2d):
                m.load_reference_weight(t(sd[name + '.conv.weight']), t(sd[name + '.conv.bias']),
                                        sd.get(name + '.c'))
        if hasattr(self, 'linear'):
            with torch.no_grad():
                self.linear.weight.data.copy_(t(sd['linear.weight']))
                self.linear.bias.data.copy_(t(sd['linear.bias']))
        self.mark_params_changed()


class Generator(_FlatParamsMixin, nn.Module):
    """reference network.py:75-139."""

    def __init__(self,
                 dataset_shape,          # Overriden based on the dataset
                 fmap_base=4096,
                 fmap_decay=1.0,
                 fmap_max=512,
                 latent_size=512,
                 normalize_latents=True,
                 wscale=True,
                 pixelnorm=True,
                 leakyrelu=True):
        super(Generator, self).__init__()
        resolution = dataset_shape[-1]
        num_channels = dataset_shape[1]
        R = int(np.log2(resolution))
        assert resolution == 2 ** R and resolution >= 4

        def nf(stage):
            return min(int(fmap_base / (2.0 ** (stage * fmap_decay))), fmap_max)

        if latent_size is None:
            latent_size = nf(0)
        self.normalize_latents = normalize_latents
        layer_settings = {
            'wscale': wscale,
            'pixelnorm': pixelnorm,
            'act': 'lrelu' if leakyrelu else 'relu'
        }
        self.block0 = GFirstBlock(latent_size, nf(1), num_channels, **layer_settings)
        self.blocks = nn.ModuleList([
            GBlock(nf(i - 1), nf(i), num_channels, **layer_settings)
            for i in range(2, R)
        ])
        self.depth = 0
        self.alpha = 1.0
        self.eps = 1e-8
        self.latent_size = latent_size
        self.max_depth = len(self.blocks)
        self.num_channels = num_channels
        self._flatten()

    def forward(self, x):
        from . import engine
        return engine.generator_forward(self, x)


class Discriminator(_FlatParamsMixin, nn.Module):
    """reference network.py:190-240."""

    def __init__(self,
                 dataset_shape,          # Overriden based on dataset
                 fmap_base=4096,
                 fmap_decay=1.0,
                 fmap_max=512,
                 wscale=True,
                 pixelnorm=False,
                 leakyrelu=True):
        super(Discriminator, self).__init__()
        resolution = dataset_shape[-1]
        num_channels = dataset_shape[1]
        R = int(np.log2(resolution))
        assert resolution == 2 ** R and resolution >= 4
        self.R = R
        self.pixelnorm = bool(pixelnorm)    # engine: PixelNorm after c1/c2 + its gradient-penalty Hessian-vector terms

        def nf(stage):
            return min(int(fmap_base / (2.0 ** (stage * fmap_decay))), fmap_max)
        layer_settings = {
            'wscale': wscale,
            'pixelnorm': pixelnorm,
            'act': 'lrelu' if leakyrelu else 'relu'
        }
        self.blocks = nn.ModuleList([
            DBlock(nf(i), nf(i - 1), num_channels, **layer_settings)
            for i in range(R - 1, 1, -1)
        ] + [DLastBlock(nf(1), nf(0), num_channels, **layer_settings)])
        self.linear = nn.Linear(nf(0), 1)
        self.depth = 0
        self.alpha = 1.0
        self.eps = 1e-8
        self.max_depth = len(self.blocks) - 1
        self.num_channels = num_channels
        self.slope = 0.2 if leakyrelu else 0.0
        self._flatten()

    def forward(self, x):
        from . import engine
        return engine.discriminator_forward(self, x)
