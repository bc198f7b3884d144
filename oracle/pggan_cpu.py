"""CPU oracle for the PGGAN hot path.  TEST INFRASTRUCTURE ONLY.

This file is a functional, torch-CPU (fp32) restatement of the algorithm the reference
implements on its hot path.  It is the *checker* for the HIP path: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.  The
product package (``pggan-pytorch_amd/``) never imports it and has no CPU fallback.

Pinning: every function here is checked against golden vectors exported from the reference
itself (``tests/golden/make_golden.py`` imports ``/root/reference`` in the build container and
writes ``tests/golden/*.npz|json``; ``tests/test_oracle_golden.py`` replays them).  The
reference has no tests or golden vectors of its own, so the pin is
"reference Python + torch 2.10 CPU".

The arithmetic primitives (conv2d, avg_pool2d, nearest upsample, autograd, Adam) live in
PyTorch, a third-party dependency of the reference pinned at ``torch==0.2.0.post3``
(/root/reference/requirements.txt:6) and not vendored; they are used here through the
installed torch 2.10 CPU build, exactly as the reference uses them.

Parameters are plain dicts keyed by the reference's ``state_dict`` names (e.g.
``block0.c1.conv.weight``) plus one float ``<layer>.c`` per PGConv2d (the equalized-lr
constant, reference network.py:19, which is *not* in the state_dict).
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-8


# ----------------------------------------------------------------------------------------
# configuration helpers
# ----------------------------------------------------------------------------------------
def nf(stage, fmap_base=4096, fmap_decay=1.0, fmap_max=512):
    """Feature-map count of a stage.  reference network.py:94-95, 207-208."""
    return min(int(fmap_base / (2.0 ** (stage * fmap_decay))), fmap_max)


class NetCfg(object):
    """Static architecture description shared by G and D (reference network.py:76-116, 191-223)."""

    def __init__(self, resolution, num_channels, fmap_base=4096, fmap_decay=1.0, fmap_max=512,
                 latent_size=512, normalize_latents=True, wscale=True, g_pixelnorm=True,
                 d_pixelnorm=False, leakyrelu=True):
        R = int(np.log2(resolution))
        assert resolution == 2 ** R and resolution >= 4          # network.py:92, 204
        self.resolution, self.R, self.num_channels = resolution, R, num_channels
        self.fmap = (fmap_base, fmap_decay, fmap_max)
        self.latent_size = nf(0, *self.fmap) if latent_size is None else latent_size  # :97-98
        self.normalize_latents = normalize_latents
        self.wscale = wscale
        self.g_pixelnorm, self.d_pixelnorm = g_pixelnorm, d_pixelnorm
        self.slope = 0.2 if leakyrelu else 0.0                     # network.py:27 (ReLU == slope 0)
        self.max_depth = R - 2                                     # network.py:116, 223

    def nf(self, stage):
        return nf(stage, *self.fmap)


# ----------------------------------------------------------------------------------------
# initialisation (RNG-order faithful).  reference network.py:8-30
# ----------------------------------------------------------------------------------------
def _init_pgconv(params, name, ch_in, ch_out, ksize, pad, wscale):
    conv = torch.nn.Conv2d(ch_in, ch_out, ksize, 1, pad)           # network.py:16 (default init first)
    if wscale:
        torch.nn.init.kaiming_normal_(conv.weight)                  # network.py:13,17
        c = torch.sqrt(torch.mean(conv.weight.data ** 2))           # network.py:19 (fp32, empirical)
        conv.weight.data /= c                                       # network.py:20
        c = float(c)
    else:
        c = 1.0                                                     # network.py:22
    params[name + '.conv.weight'] = conv.weight.data.clone()
    params[name + '.conv.bias'] = conv.bias.data.clone()
    params[name + '.c'] = c


def init_generator(cfg):
    """Build G's parameters in the reference's construction order (network.py:99-110)."""
    p = OrderedDict()
    C = cfg.num_channels
    _init_pgconv(p, 'block0.c1', cfg.latent_size, cfg.nf(1), 4, 3, cfg.wscale)   # :47
    _init_pgconv(p, 'block0.c2', cfg.nf(1), cfg.nf(1), 3, 1, cfg.wscale)         # :48
    _init_pgconv(p, 'block0.toRGB', cfg.nf(1), C, 1, 0, True)      # :49 (built without layer_settings: wscale always)
    for j, i in enumerate(range(2, cfg.R)):                                      # :107-110
        _init_pgconv(p, 'blocks.%d.c1' % j, cfg.nf(i - 1), cfg.nf(i), 3, 1, cfg.wscale)
        _init_pgconv(p, 'blocks.%d.c2' % j, cfg.nf(i), cfg.nf(i), 3, 1, cfg.wscale)
        _init_pgconv(p, 'blocks.%d.toRGB' % j, cfg.nf(i), C, 1, 0, True)
    return p


def init_discriminator(cfg):
    """Build D's parameters in the reference's construction order (network.py:214-219)."""
    p = OrderedDict()
    C = cfg.num_channels
    j = 0
    for i in range(cfg.R - 1, 1, -1):                                            # :214-216
        _init_pgconv(p, 'blocks.%d.fromRGB' % j, C, cfg.nf(i), 1, 0, True)        # :145 (wscale always)
        _init_pgconv(p, 'blocks.%d.c1' % j, cfg.nf(i), cfg.nf(i), 3, 1, cfg.wscale)
        _init_pgconv(p, 'blocks.%d.c2' % j, cfg.nf(i), cfg.nf(i - 1), 3, 1, cfg.wscale)
        j += 1
    _init_pgconv(p, 'blocks.%d.fromRGB' % j, C, cfg.nf(1), 1, 0, True)            # :160
    _init_pgconv(p, 'blocks.%d.c1' % j, cfg.nf(1) + 1, cfg.nf(1), 3, 1, cfg.wscale)  # :162
    _init_pgconv(p, 'blocks.%d.c2' % j, cfg.nf(1), cfg.nf(0), 4, 0, cfg.wscale)   # :163
    lin = torch.nn.Linear(cfg.nf(0), 1)                                           # :219 (plain init)
    p['linear.weight'] = lin.weight.data.clone()
    p['linear.bias'] = lin.bias.data.clone()
    return p


def tensor_names(params):
    return [k for k, v in params.items() if torch.is_tensor(v)]


# ----------------------------------------------------------------------------------------
# forward passes
# ----------------------------------------------------------------------------------------
def pgconv(x, p, name, pad, slope, pixelnorm):
    """Equalized-lr conv -> act -> pixelnorm.  reference network.py:32-41 (order conv,act,norm)."""
    h = x * p[name + '.c']                                                        # :33
    h = F.conv2d(h, p[name + '.conv.weight'], p[name + '.conv.bias'], 1, pad)      # :34
    if slope is not None:
        h = _activation(h, slope)                                                 # :35-36
    if pixelnorm:
        h = h * torch.rsqrt(torch.mean(h * h, 1, keepdim=True) + EPS)             # :37-40
    return h


_FORCED = None          # adjudication aid, see ``forced_signs``


def _activation(h, slope):
    """LeakyReLU / ReLU of the reference (network.py:26-29,35-36).  Under ``forced_signs`` the branch of every element is taken
    from a given sign pattern instead of from ``h`` itself."""
    if _FORCED is None:
        return F.leaky_relu(h, slope) if slope != 0.0 else F.relu(h)
    m = _FORCED['masks'].pop(0)
    pos = m.to(h.device) > 0
    _FORCED['flips'] += int((pos != (h > 0)).sum())
    _FORCED['elements'] += h.numel()
    return h * torch.where(pos, torch.ones((), dtype=h.dtype), torch.full((), float(slope), dtype=h.dtype))


class forced_signs(object):
    """Adjudication aid (tests/test_fp64_adjudicator.py): evaluate the networks on the LINEAR PIECE selected by a given list of
    activation sign patterns (one tensor per activation call, in call order, NCHW) instead of the piece the evaluation itself
    would select.  The networks are piecewise linear in their LeakyReLU branches, so two fp32 evaluations that disagree on a
    single branch differ by O(1e-4..1e-3) in their gradients; conditioning an fp64 evaluation on the other party's branches
    gives the exact gradient OF THAT PIECE, against which pure arithmetic error can be measured.  ``.flips`` counts the branches
    that differ from the ones the evaluation would have chosen itself."""

    def __init__(self, masks):
        self.state = dict(masks=list(masks), flips=0, elements=0)

    def __enter__(self):
        global _FORCED
        _FORCED = self.state
        return self

    def __exit__(self, *exc):
        global _FORCED
        _FORCED = None
        if exc[0] is None and self.state['masks']:
            raise RuntimeError('%d sign patterns were not consumed' % len(self.state['masks']))
        return False

    @property
    def flips(self):
        return self.state['flips']

    @property
    def elements(self):
        return self.state['elements']


def generator_forward(p, cfg, z, depth, alpha):
    """reference network.py:118-139."""
    pn, sl = cfg.g_pixelnorm, cfg.slope
    h = z.unsqueeze(2).unsqueeze(3)                                               # :119
    if cfg.normalize_latents:
        h = h * torch.rsqrt(torch.mean(h * h, 1, keepdim=True) + EPS)             # :120-123
    h = pgconv(h, p, 'block0.c1', 3, sl, pn)                                      # :53
    h = pgconv(h, p, 'block0.c2', 1, sl, pn)                                      # :54
    if depth == 0:
        return pgconv(h, p, 'block0.toRGB', 0, None, False)                       # :55-56
    for i in range(depth - 1):                                                    # :126-128
        h = F.interpolate(h, scale_factor=2, mode='nearest')
        h = pgconv(h, p, 'blocks.%d.c1' % i, 1, sl, pn)
        h = pgconv(h, p, 'blocks.%d.c2' % i, 1, sl, pn)
    h = F.interpolate(h, scale_factor=2, mode='nearest')                          # :129
    u = pgconv(h, p, 'blocks.%d.c1' % (depth - 1), 1, sl, pn)
    u = pgconv(u, p, 'blocks.%d.c2' % (depth - 1), 1, sl, pn)
    ult = pgconv(u, p, 'blocks.%d.toRGB' % (depth - 1), 0, None, False)           # :130
    if alpha < 1.0:                                                               # :131-135
        prev = 'blocks.%d.toRGB' % (depth - 2) if depth > 1 else 'block0.toRGB'
        preult = pgconv(h, p, prev, 0, None, False)
    else:
        preult = 0                                                                # :137
    return preult * (1 - alpha) + ult * alpha                                     # :138


def tstdeps(val):
    """One global scalar over the whole tensor.  reference network.py:174-175."""
    return torch.sqrt(((val - val.mean()) ** 2).mean() + 1.0e-8)


def minibatch_stddev(x):
    """reference network.py:183-187."""
    s = tstdeps(x)
    return torch.cat((x, s.expand(x.size(0), 1, x.size(2), x.size(3))), dim=1)


def _dblock(h, p, cfg, j, first, x=None):
    """DBlock / DLastBlock forward.  reference network.py:149-154, 165-171."""
    pn, sl = cfg.d_pixelnorm, cfg.slope
    last = (j == cfg.max_depth)
    name = 'blocks.%d' % j
    if first:
        h = pgconv(x, p, name + '.fromRGB', 0, 0.2, False)   # :145,160 — built WITHOUT layer_settings: always LeakyReLU(0.2), no pn
    if last:
        h = minibatch_stddev(h)                                                   # :168
        h = pgconv(h, p, name + '.c1', 1, sl, pn)
        h = pgconv(h, p, name + '.c2', 0, sl, pn)                                 # 4x4 pad 0
    else:
        h = pgconv(h, p, name + '.c1', 1, sl, pn)
        h = pgconv(h, p, name + '.c2', 1, sl, pn)
    return h


def discriminator_forward(p, cfg, x, depth, alpha):
    """reference network.py:225-240.  blocks[-k] == index (max_depth+1-k)."""
    nb = cfg.max_depth + 1
    h = _dblock(None, p, cfg, nb - (depth + 1), True, x)                          # :227
    if depth > 0:
        h = F.avg_pool2d(h, 2)                                                    # :229
        if alpha < 1.0:
            xlow = F.avg_pool2d(x, 2)                                             # :231
            pre = pgconv(xlow, p, 'blocks.%d.fromRGB' % (nb - depth), 0, 0.2, False)   # :232 (LeakyReLU(0.2) always)
            h = h * alpha + (1 - alpha) * pre                                     # :233
    for i in range(depth, 0, -1):                                                 # :235-238
        h = _dblock(h, p, cfg, nb - i, False)
        if i > 1:
            h = F.avg_pool2d(h, 2)
    h = h.squeeze(-1).squeeze(-1)
    return F.linear(h, p['linear.weight'], p['linear.bias'])                      # :239


# ----------------------------------------------------------------------------------------
# WGAN-GP losses.  reference wgan_gp_loss.py
# ----------------------------------------------------------------------------------------
def _leafify(p, names=None):
    q = OrderedDict()
    for k, v in p.items():
        if torch.is_tensor(v):
            q[k] = v.detach().clone().requires_grad_(True)
        else:
            q[k] = v
    return q


def gradient_penalty(dp, cfg, real, fake, mix, depth, alpha, iwass_lambda, iwass_target, mixed=None):
    """reference wgan_gp_loss.py:13-33.  ``mix`` [N,1] is the U[0,1) draw of :15-17 (weights *fake*).
    (``mixed`` given: adjudication aid — the interpolated samples are taken as they are instead of being re-derived.)"""
    n = real.size(0)
    if mixed is None:
        mixed = (real.reshape(n, -1) * (1 - mix) + fake.reshape(n, -1) * mix).reshape(real.shape)  # :8-10,19
    mixed = mixed.detach().requires_grad_(True)
    scores = discriminator_forward(dp, cfg, mixed, depth, alpha)                   # :20
    g = torch.autograd.grad(scores, mixed, torch.ones_like(scores),
                            create_graph=True, retain_graph=True)[0]               # :25-28
    g = g.reshape(n, -1)
    return ((g.norm(2, dim=1) - iwass_target) ** 2) * iwass_lambda / (iwass_target ** 2)   # :31


def d_loss_and_grads(dparams, gparams, cfg, real, latents, mix, depth, alpha,
                     iwass_lambda=10.0, iwass_epsilon=0.001, iwass_target=1.0, fake=None, mixed=None):
    """``wgan_gp_D_loss`` + ``D_cost.backward()``  (wgan_gp_loss.py:36-65, trainer.py:95-98).

    Returns dict(D_cost, D_real_loss[N,1], D_fake_loss[N,1], gp[N], fake, grads{name: tensor}).
    Parameters that do not take part at this depth/alpha get no entry in ``grads`` (autograd
    leaves ``.grad`` None, which is what makes Adam skip them)."""
    dp = _leafify(dparams)
    d_real = discriminator_forward(dp, cfg, real, depth, alpha)                   # :47
    d_real_loss = -d_real + d_real ** 2 * iwass_epsilon                           # :48
    if fake is None:
        with torch.no_grad():
            fake = generator_forward(gparams, cfg, latents, depth, alpha)         # :51-52 (no graph)
    # (``fake`` / ``mixed`` given: adjudication aid of tests/test_fp64_adjudicator.py — the D step as a function of FIXED images, so
    #  that an fp64 evaluation differs from an fp32 one only by D's own arithmetic and not by G's forward rounding, which the
    #  discontinuous LeakyReLU' amplifies)
    d_fake = discriminator_forward(dp, cfg, fake, depth, alpha)                   # :54
    d_fake_loss = d_fake                                                          # :55
    gp = gradient_penalty(dp, cfg, real, fake, mix, depth, alpha, iwass_lambda, iwass_target, mixed=mixed)  # :58
    d_cost = (d_fake_loss + d_real_loss + gp).mean()                              # :62 ([N,1]+[N] -> [N,N])
    names = tensor_names(dp)
    gr = torch.autograd.grad(d_cost, [dp[k] for k in names], allow_unused=True)
    grads = OrderedDict((k, g) for k, g in zip(names, gr) if g is not None)
    return dict(D_cost=d_cost.detach(), D_real_loss=d_real_loss.detach(),
                D_fake_loss=d_fake_loss.detach(), gp=gp.detach(), fake=fake, grads=grads,
                D_real=d_real.detach(), D_fake=d_fake.detach())


def g_loss_and_grads(gparams, dparams, cfg, latents, depth, alpha):
    """``wgan_gp_G_loss`` + ``.backward()`` (wgan_gp_loss.py:68-74, trainer.py:105-111).  Only G's
    gradients are returned: D's are produced by the reference too but never used (its next
    ``D.zero_grad()`` at wgan_gp_loss.py:42 discards them)."""
    gp_ = _leafify(gparams)
    g_new = generator_forward(gp_, cfg, latents, depth, alpha)                    # :71
    g_cost = (-discriminator_forward(dparams, cfg, g_new, depth, alpha)).mean()   # :72-73
    names = tensor_names(gp_)
    gr = torch.autograd.grad(g_cost, [gp_[k] for k in names], allow_unused=True)
    grads = OrderedDict((k, g) for k, g in zip(names, gr) if g is not None)
    return dict(G_cost=g_cost.detach(), fake=g_new.detach(), grads=grads)


def d_input_gradient(dparams, cfg, x, depth, alpha):
    """Adjudication aid: gradient of wgan_gp_G_loss's ``(-D(x)).mean()`` (wgan_gp_loss.py:72-73) with respect to the images x."""
    x = x.detach().clone().requires_grad_(True)
    cost = (-discriminator_forward(dparams, cfg, x, depth, alpha)).mean()
    return torch.autograd.grad(cost, x)[0]


def g_grads_given_output_gradient(gparams, cfg, latents, depth, alpha, grad_out):
    """Adjudication aid: G's parameter gradients for a GIVEN d(cost)/d(G output) (the vector-Jacobian product of
    generator_forward), i.e. trainer.py:111 with the discriminator half of the chain held fixed."""
    gp_ = _leafify(gparams)
    out = generator_forward(gp_, cfg, latents, depth, alpha)
    names = tensor_names(gp_)
    gr = torch.autograd.grad(out, [gp_[k] for k in names], grad_outputs=grad_out.to(out.dtype), allow_unused=True)
    return OrderedDict((k, g) for k, g in zip(names, gr) if g is not None)


# ----------------------------------------------------------------------------------------
# optimiser + schedules
# ----------------------------------------------------------------------------------------
class AdamState(object):
    """torch.optim.Adam as configured by reference train.py:148-149,195 (betas (0,0.99), eps 1e-8,
    no weight decay), in the torch-2.10 form (denom = sqrt(v)/sqrt(bc2) + eps).  Per-parameter
    step counters: a parameter without a gradient is skipped entirely."""

    def __init__(self, betas=(0.0, 0.99), eps=1e-8):
        self.b1, self.b2, self.eps = betas[0], betas[1], eps
        self.m, self.v, self.t = {}, {}, {}

    def step(self, params, grads, lr):
        for k, g in grads.items():
            if k not in self.m:
                self.m[k] = torch.zeros_like(params[k])
                self.v[k] = torch.zeros_like(params[k])
                self.t[k] = 0
            self.t[k] += 1
            t = self.t[k]
            self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            bc1 = 1 - self.b1 ** t
            bc2 = 1 - self.b2 ** t
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            params[k].addcdiv_(self.m[k], denom, value=-lr / bc1)


def rampup(cur_nimg, lr_rampup_kimg=40):
    """reference train.py:151-156."""
    if cur_nimg < lr_rampup_kimg * 1000:
        p = max(0.0, 1 - cur_nimg / (lr_rampup_kimg * 1000))
        return float(np.exp(-p * p * 5.0))
    return 1.0


def depth_schedule(cur_nimg, max_depth, lod_training_nimg=100 * 1000, lod_transition_nimg=100 * 1000,
                   minibatch_default=16, minibatch_overrides=None, tick_kimg_default=20,
                   tick_kimg_overrides=None):
    """Pure function of cur_nimg.  reference plugins.py:57-74.  Returns (depth, alpha, minibatch,
    tick_duration_nimg).  Integer math + one IEEE double division: must be bit-exact."""
    if minibatch_overrides is None:
        minibatch_overrides = {6: 14, 7: 6, 8: 3}                                 # plugins.py:20
    if tick_kimg_overrides is None:
        tick_kimg_overrides = {3: 10, 4: 10, 5: 5, 6: 2, 7: 2, 8: 1}              # plugins.py:22
    full, rem = divmod(cur_nimg, lod_training_nimg + lod_transition_nimg)        # :59
    tp, rem = divmod(rem, lod_training_nimg)                                      # :60
    depth = min(max_depth, full + tp)                                             # :61
    alpha = rem / lod_transition_nimg if (tp > 0 and full + tp == depth) else 1.0  # :62-63
    mb = minibatch_overrides.get(depth, minibatch_default)                        # :68
    tick = tick_kimg_overrides.get(depth, tick_kimg_default) * 1000               # :72-73
    return depth, alpha, mb, tick


# ----------------------------------------------------------------------------------------
# one Trainer.train() iteration (reference trainer.py:85-115), D_training_repeats = 1
# ----------------------------------------------------------------------------------------
def train_iteration(gparams, dparams, cfg, opt_g, opt_d, real, latents_d, latents_g, mix,
                    depth, alpha, lr_d, lr_g, iwass_lambda=10.0, iwass_epsilon=0.001,
                    iwass_target=1.0):
    d = d_loss_and_grads(dparams, gparams, cfg, real, latents_d, mix, depth, alpha,
                         iwass_lambda, iwass_epsilon, iwass_target)               # :95-98
    opt_d.step(dparams, d['grads'], lr_d)                                         # :100
    g = g_loss_and_grads(gparams, dparams, cfg, latents_g, depth, alpha)          # :105-111
    opt_g.step(gparams, g['grads'], lr_g)                                         # :112
    return d, g


# ----------------------------------------------------------------------------------------
# synthetic inputs shared by tests / bench
# ----------------------------------------------------------------------------------------
def synthetic_batch(seed, n, num_channels, res, latent_size):
    """Seeded synthetic (real, z_d, z_g, mix) using numpy's legacy RandomState (stable stream)."""
    rs = np.random.RandomState(seed)
    real = (rs.rand(n, num_channels, res, res).astype(np.float32) * 2 - 1)
    z_d = rs.randn(n, latent_size).astype(np.float32)
    z_g = rs.randn(n, latent_size).astype(np.float32)
    mix = rs.rand(n, 1).astype(np.float32)
    return tuple(torch.from_numpy(a) for a in (real, z_d, z_g, mix))
