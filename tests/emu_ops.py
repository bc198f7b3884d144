"""TEST INFRASTRUCTURE: a torch-CPU emulation of ``pggan-pytorch_amd/ops.py`` (same signatures).

Purpose: run the *host-side* launch schedules of ``engine.py`` (the hand-derived WGAN-GP
double-backward, the batched [real|fake|mixed] sweep, the flat-buffer optimizer) on a machine
without a GPU and compare them with the autograd-based CPU oracle.  It is installed only by the
``emu`` pytest fixture (monkeypatch) and is never imported by the product: the product path has no
CPU fallback.  Each function states the arithmetic contract of the kernel it stands in for, i.e.
the same contract written in include/pggan_hip.h."""
import torch
import torch.nn.functional as F


def require_gpu():
    return None


def _nchw(x):
    return x.permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _ret(y, out):
    if out is not None:
        out.copy_(y)
        return out
    return y


class Unsupported(RuntimeError):
    pass


def signbytes_of(y):
    """fp32 [..., C] -> uint8 sign bytes [..., C/4] (bit j of a byte = channel 4q+j > 0): the product's format."""
    b = (y > 0).to(torch.uint8).reshape(tuple(y.shape[:-1]) + (y.shape[-1] // 4, 4))
    return (b[..., 0] | (b[..., 1] << 1) | (b[..., 2] << 2) | (b[..., 3] << 3)).contiguous()


def signbytes_to_mask(b):
    bits = torch.stack([(b >> j) & 1 for j in range(4)], dim=-1).reshape(tuple(b.shape[:-1]) + (4 * b.shape[-1],))
    return bits.to(torch.float32) * 2 - 1


def _maskmul(v, mask, slope):
    if mask.dtype == torch.uint8:
        mask = signbytes_to_mask(mask)
    return v * torch.where(mask > 0, torch.ones_like(mask), torch.full_like(mask, slope))


def _lrelu(v, slope):
    return torch.where(v > 0, v, v * slope)


def _wref(w):
    return w.permute(2, 3, 0, 1)            # [ks,ks,co,ci] -> [co,ci,ks,ks]


def conv2d(x, w, bias, N, Hin, Win, ks, pad, scale, slope=1.0, mask=None, mask_slope=0.2, ups=False, out=None, signs_out=False):
    xi = _nchw(x)
    if ups:
        xi = F.interpolate(xi, scale_factor=2, mode='nearest')
    assert xi.shape[2] == Hin and xi.shape[3] == Win and xi.shape[0] == N
    z = F.conv2d(xi, _wref(w), None, 1, pad) * scale
    z = _nhwc(z)
    if mask is not None:
        y = _maskmul(z, mask, mask_slope)
    else:
        if bias is not None:
            z = z + bias
        y = _lrelu(z, slope)
    y = _ret(y, out)
    return (y, signbytes_of(y)) if signs_out else y


def _unpooled(g, gbytes, gmul, gslope):
    return avgpool2_bwd(g, gbytes, 4.0 * gmul, gslope)


def conv2d_unpooled(g, w, gbytes, gmul, gslope, N, Hin, Win, scale, mask=None, mask_slope=0.2):
    return conv2d(_unpooled(g, gbytes, gmul, gslope), w, None, N, Hin, Win, 3, 1, scale, mask=mask, mask_slope=mask_slope)


def conv2d_wgrad_unpooled(x, g, gbytes, gmul, gslope, dw, db, N, Hin, Win, scale):
    conv2d_wgrad(x, _unpooled(g, gbytes, gmul, gslope), dw, db, N, Hin, Win, 3, 1, scale)


def conv2d_wgrad_wino(x, gz, dw, db, N, H, W, scale, ups=False, second=None):
    conv2d_wgrad(x, gz, dw, db, N, H, W, 3, 1, scale, ups=ups)
    if second is not None:
        x2, gz2, n2, bias2 = second
        conv2d_wgrad(x2, gz2, dw, db if bias2 else None, n2, H, W, 3, 1, scale, ups=ups)


def pack_dgrad_weights_batched(flat_w, flat_wt, layers):
    for off, ks, co, ci in layers:
        w = flat_w[off:off + ks * ks * co * ci].view(ks, ks, co, ci)
        pack_dgrad_weights(w, flat_wt[off:off + ks * ks * co * ci].view(ks, ks, ci, co))


class _WinoWeights(torch.Tensor):
    """Emulated Winograd-domain weights: a [16,Cout,Cin] tensor that remembers the packed weights it came from."""


def _wino_of(w):
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=w.dtype)
    u = torch.einsum('ia,abok,jb->ijok', G, w, G).reshape(16, w.shape[2], w.shape[3])
    return u


def wino_transform_weights(w, u=None):
    out = _wino_of(w)
    if u is not None:
        u.copy_(out)
        out = u
    return out


def wino_transform_weights_batched(flat_w, flat_u, layers, transposed=False):
    flags = list(transposed) if isinstance(transposed, (list, tuple)) else [bool(transposed)] * len(layers)
    for (woff, uoff, co, ci), tr in zip(layers, flags):
        if tr:                                           # backward-data form straight from the forward parameter [3][3][ci][co]
            w = flat_w[woff:woff + 9 * co * ci].view(3, 3, ci, co).flip(0, 1).permute(0, 1, 3, 2).contiguous()
        else:
            w = flat_w[woff:woff + 9 * co * ci].view(3, 3, co, ci)
        flat_u[uoff:uoff + 16 * co * ci].view(16, co, ci).copy_(_wino_of(w))


def _unwino(u):
    """Inverse of the weight transform on the 3x3 support (least squares; exact because G has full column rank)."""
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=u.dtype)
    Gp = torch.linalg.pinv(G)
    return torch.einsum('ai,ijok,bj->abok', Gp, u.view(4, 4, u.shape[1], u.shape[2]), Gp).contiguous()


def conv2d_wino(x, u, bias, N, H, W, scale, slope=1.0, mask=None, mask_slope=0.2, ups=False, out=None,
                pool=False, other=None, a=1.0, b=0.0, pool_only=False, unpool=False, upmask=None, up_mul=1.0, y_bytes=False,
                signs_out=False):
    w = _unwino(u)
    y = conv2d(x, w, bias, N, H, W, 3, 1, scale, slope=slope, mask=mask, mask_slope=mask_slope, ups=ups)
    if out is not None:
        out.copy_(y)
        y = out
    if pool:
        return (signbytes_of(y) if y_bytes else y), avgpool2_fwd(y, other, a, b)
    if unpool:
        return avgpool2_bwd(y, upmask, up_mul, mask_slope)
    return (y, signbytes_of(y)) if signs_out else y


def conv2d_wino_pixelnorm(x, u, bias, N, H, W, scale, slope, eps=1e-8, ups=False):
    if u.shape[1] > 32:
        raise Unsupported('PixelNorm epilogue: at most 32 couts')
    return pixelnorm_fwd(conv2d(x, _unwino(u), bias, N, H, W, 3, 1, scale, slope=slope, ups=ups), eps)


def conv2d_wino_pnbwd(x, u, ysaved, r, N, H, W, scale, slope, pool=False, other=None, a=1.0, b=0.0):
    if u.shape[1] > 32:
        raise Unsupported('PixelNorm adjoint epilogue: at most 32 couts')
    g = conv2d(x, _unwino(u), None, N, H, W, 3, 1, scale)
    if pool:
        g = avgpool2_fwd(g, other, a, b)
    return pixelnorm_lrelu_bwd(g, ysaved, r, slope)


def conv2d_pool(x, w, bias, N, Hin, Win, ks, pad, scale, slope=1.0, mask=None, mask_slope=0.2, other=None, a=1.0, b=0.0,
                pool_only=False, y_bytes=False):
    y = conv2d(x, w, bias, N, Hin, Win, ks, pad, scale, slope=slope, mask=mask, mask_slope=mask_slope)
    return (signbytes_of(y) if y_bytes else y), avgpool2_fwd(y, other, a, b)


def conv2d_pixelnorm(x, w, bias, N, Hin, Win, ks, pad, scale, slope, eps=1e-8, ups=False):
    return pixelnorm_fwd(conv2d(x, w, bias, N, Hin, Win, ks, pad, scale, slope=slope, ups=ups), eps)


def conv2d_pnbwd(x, w, ysaved, r, N, Hin, Win, ks, pad, scale, slope):
    return pixelnorm_lrelu_bwd(conv2d(x, w, None, N, Hin, Win, ks, pad, scale), ysaved, r, slope)


def conv2d_unpool(x, w, N, Hin, Win, ks, pad, scale, upmask=None, mul=1.0, mask_slope=0.2):
    return avgpool2_bwd(conv2d(x, w, None, N, Hin, Win, ks, pad, scale), upmask, mul, mask_slope)


def conv2d_wgrad(x, gz, dw, db, N, Hin, Win, ks, pad, scale, ups=False):
    xi = _nchw(x)
    if ups:
        xi = F.interpolate(xi, scale_factor=2, mode='nearest')
    g = _nchw(gz)
    co, ci = dw.shape[2], dw.shape[3]
    gw = torch.nn.grad.conv2d_weight(xi.contiguous(), (co, ci, ks, ks), g.contiguous(), stride=1, padding=pad)
    dw += gw.permute(2, 3, 0, 1) * scale
    if db is not None:
        db += gz.sum(dim=(0, 1, 2))


def pack_dgrad_weights(w, wt):
    wt.copy_(w.flip(0, 1).permute(0, 1, 3, 2))
    return wt


def _img(img, pool):
    return F.avg_pool2d(img, 2) if pool else img


def fromrgb_fwd(img, w, bias, N, C, H, W, scale, slope, pool=False, mask=None, mask_slope=0.2, signs_out=False):
    xi = _img(img, pool)
    z = torch.einsum('nchw,oc->nhwo', xi, w) * scale
    if mask is not None:
        return _maskmul(z, mask, mask_slope).contiguous()
    if bias is not None:
        z = z + bias
    y = _lrelu(z, slope).contiguous()
    return (y, signbytes_of(y)) if signs_out else y


def fromrgb_bwd_data(gz, w, gimg, N, C, H, W, scale, pool=False, accumulate=False):
    g = torch.einsum('nhwo,oc->nchw', gz, w) * scale
    if pool:
        g = F.interpolate(g, scale_factor=2, mode='nearest') * 0.25
    if accumulate:
        gimg += g
    else:
        gimg.copy_(g)


def fromrgb_wgrad(gz, img, dw, db, N, C, H, W, scale, pool=False):
    xi = _img(img, pool)
    dw += torch.einsum('nhwo,nchw->oc', gz, xi) * scale
    if db is not None:
        db += gz.sum(dim=(0, 1, 2))


def torgb_fwd(x, w, bias, N, C, H, W, scale, out_mul=1.0, prev=None, prev_mul=0.0, out=None):
    y = torch.einsum('nhwi,ci->nchw', x, w) * scale
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    y = y * out_mul
    if prev is not None:
        y = y + prev_mul * F.interpolate(prev, scale_factor=2, mode='nearest')
    return _ret(y.contiguous(), out)


def torgb_bwd_data(g, w, N, C, H, W, mul_scale, down=False):
    gg = F.avg_pool2d(g, 2) * 4 if down else g
    return (torch.einsum('nchw,ci->nhwi', gg, w) * mul_scale).contiguous()


def torgb_bwd_data_pnbwd(g, w, ysaved, r, N, C, H, W, mul_scale, slope):
    return pixelnorm_lrelu_bwd(torgb_bwd_data(g, w, N, C, H, W, mul_scale), ysaved, r, slope)


def torgb_wgrad(g, x, dw, db, N, C, H, W, mul_scale, mul, down=False):
    gg = F.avg_pool2d(g, 2) * 4 if down else g
    dw += torch.einsum('nchw,nhwi->ci', gg, x) * mul_scale
    if db is not None:
        db += gg.sum(dim=(0, 2, 3)) * mul


def avgpool2_fwd(x, other=None, a=1.0, b=0.0):
    y = _nhwc(F.avg_pool2d(_nchw(x), 2))
    if other is not None:
        return y * a + b * other
    return y * a


def avgpool2_bwd(gy, mask=None, mul=1.0, mask_slope=0.2):
    g = _nhwc(F.interpolate(_nchw(gy), scale_factor=2, mode='nearest')) * (0.25 * mul)
    if mask is not None:
        g = _maskmul(g, mask, mask_slope)
    return g.contiguous()


def upsample2_bwd(g):
    return _nhwc(F.avg_pool2d(_nchw(g), 2) * 4)


def axpby_mask(x, other=None, mask=None, a=1.0, b=0.0, mask_slope=0.2, out=None):
    y = x * a
    if other is not None:
        y = y + b * other
    if mask is not None:
        y = _maskmul(y, mask, mask_slope)
    return _ret(y, out)


def pixelnorm_fwd(x, eps=1e-8, inplace=False):
    C = x.shape[-1]
    r = torch.rsqrt((x * x).mean(dim=-1) + eps)
    y = x * r.unsqueeze(-1)
    if inplace:
        x.copy_(y)
        y = x
    return y, r.reshape(-1)


def pixelnorm_lrelu_bwd(gy, y, r, slope, inplace=False, inj=None, out=None):
    if r is not None:
        rr = r.view(y.shape[:-1]).unsqueeze(-1)
        gh = rr * (gy - y * (gy * y).mean(dim=-1, keepdim=True))
    else:
        gh = gy
    if inj is not None:
        gh = gh + inj
    gz = _maskmul(gh, y, slope)
    if out is not None:
        out.copy_(gz)
        return out
    if inplace:
        gy.copy_(gz)
        return gy
    return gz


def pixelnorm_tangent(t, y, r, a):
    C = y.shape[-1]
    rr = r.view(y.shape[:-1]).unsqueeze(-1)
    ty_, ay_, ta_ = [(u * v).sum(dim=-1, keepdim=True) for u, v in ((t, y), (a, y), (t, a))]

    def proj(v):
        return rr * (v - y * (y * v).sum(dim=-1, keepdim=True) / C)
    S = ta_ - ty_ * ay_ / C
    inj = -(rr * rr * S / C) * y - (rr / C) * proj(ay_ * t + ty_ * a)
    return proj(t), inj


def mbstd_fwd(x, groups, cp):
    NB, H, W, C = x.shape
    n = NB // groups
    y = torch.zeros((NB, H, W, cp))
    y[..., :C] = x
    stats = torch.zeros((groups, 2))
    for g in range(groups):
        xg = x[g * n:(g + 1) * n]
        mu = xg.mean()
        sigma = torch.sqrt(((xg - mu) ** 2).mean() + 1e-8)
        stats[g, 0], stats[g, 1] = mu, sigma
        y[g * n:(g + 1) * n, :, :, C] = sigma
    return y, stats


# ---- exact-global mode (the host side's view of pg_mbstd_stats / _write / _tangent_stats / _tangent_write / _gsum / _bwd_global): a
# partial row is [mu, sigma, count, sum, sum of squares] resp. [mean tx, dot, sum tx, dot] -- the emulation keeps the interface, not the layout
def mbstd_stats(x, groups):
    NB = x.shape[0]
    n = NB // groups
    st = torch.zeros((groups, 8), dtype=torch.float64)
    for g in range(groups):
        xg = x[g * n:(g + 1) * n].double()
        st[g, 2], st[g, 3], st[g, 4] = xg.numel(), xg.sum(), (xg * xg).sum()
    return st


def mbstd_write(x, stats, gathered, cp):
    NB, H, W, C = x.shape
    groups = stats.shape[0]
    n = NB // groups
    y = torch.zeros((NB, H, W, cp))
    y[..., :C] = x
    tot = gathered.double().sum(dim=0)
    out = torch.zeros((groups, 8), dtype=torch.float64)
    for g in range(groups):
        M, sx, sxx = tot[g, 2], tot[g, 3], tot[g, 4]
        mu = sx / M
        sigma = torch.sqrt(torch.clamp(sxx / M - mu * mu, min=0.0) + 1e-8)
        out[g, 0], out[g, 1], out[g, 2] = mu, sigma, M
        y[g * n:(g + 1) * n, :, :, C] = sigma.float()
    return y, out


def mbstd_tangent_stats(x, tx, stats):
    NB = x.shape[0]
    groups = stats.shape[0]
    n = NB // groups
    ts = torch.zeros((groups, 8), dtype=torch.float64)
    for g in range(groups):
        xg, tg = x[g * n:(g + 1) * n].double(), tx[g * n:(g + 1) * n].double()
        ts[g, 2], ts[g, 3] = tg.sum(), ((xg - stats[g, 0]) * tg).sum()
    return ts


def mbstd_tangent_write(tx, tstats, gathered, stats, cp):
    NB, H, W, C = tx.shape
    groups = stats.shape[0]
    n = NB // groups
    ty = torch.zeros((NB, H, W, cp))
    ty[..., :C] = tx
    tot = gathered.double().sum(dim=0)
    out = torch.zeros((groups, 8), dtype=torch.float64)
    for g in range(groups):
        M, sigma = stats[g, 2], stats[g, 1]
        out[g, 0], out[g, 1] = tot[g, 2] / M, tot[g, 3]
        ty[g * n:(g + 1) * n, :, :, C] = (tot[g, 3] / (M * sigma)).float()
    return ty, out


def mbstd_gsum(gy, gy_first, groups, shape, cp):
    NB, H, W, C = shape
    n = NB // groups
    out = torch.zeros((groups, 2), dtype=torch.float64)
    for g in range(groups):
        if gy is not None:
            out[g, 0] = gy[g * n:(g + 1) * n][..., C].double().sum()
        if gy_first is not None:
            out[g, 1] = gy_first[g * n:(g + 1) * n][..., C].double().sum()
    return out


def mbstd_bwd_global(gy, x, stats, cp, apply_mask, gsums, nranks, mask_slope=0.2, tx=None, tstats=None, gy_first=None, out=None):
    NB, H, W, C = x.shape
    groups = stats.shape[0]
    n = NB // groups
    gx = torch.zeros_like(x)
    for g in range(groups):
        sl = slice(g * n, (g + 1) * n)
        xg = x[sl].double()
        mu, sigma, M = stats[g, 0], stats[g, 1], stats[g, 2]
        v = torch.zeros_like(xg)
        if gy is not None:
            v = gy[sl][..., :C].double() + gsums[g, 0] * (xg - mu) / (M * sigma)
        if tx is not None:
            tmean, dot = tstats[g, 0], tstats[g, 1]
            v = v + gsums[g, 1] / (M * sigma) * ((tx[sl].double() - tmean) - (xg - mu) * dot / (M * sigma * sigma))
        v = v.float()
        if apply_mask:
            v = _maskmul(v, x[sl], mask_slope)
        gx[sl] = v
    return _ret(gx, out)


def mbstd_tangent(x, tx, stats, cp):
    NB, H, W, C = x.shape
    groups = stats.shape[0]
    n = NB // groups
    ty = torch.zeros((NB, H, W, cp))
    ty[..., :C] = tx
    tstats = torch.zeros((groups, 2))
    for g in range(groups):
        xg, tg = x[g * n:(g + 1) * n], tx[g * n:(g + 1) * n]
        mu, sigma = stats[g, 0], stats[g, 1]
        M = xg.numel()
        dot = ((xg - mu) * tg).sum()
        tstats[g, 0], tstats[g, 1] = tg.mean(), dot
        ty[g * n:(g + 1) * n, :, :, C] = dot / (M * sigma)
    return ty, tstats


def mbstd_bwd(gy, x, stats, cp, apply_mask, mask_slope=0.2, tx=None, tstats=None, gy_first=None, out=None):
    NB, H, W, C = x.shape
    groups = stats.shape[0]
    n = NB // groups
    gx = torch.zeros_like(x)
    for g in range(groups):
        sl = slice(g * n, (g + 1) * n)
        xg = x[sl]
        mu, sigma = stats[g, 0], stats[g, 1]
        M = xg.numel()
        v = torch.zeros_like(xg)
        if gy is not None:
            Gs = gy[sl][..., C].sum()
            v = gy[sl][..., :C] + Gs * (xg - mu) / (M * sigma)
        if tx is not None:
            Gs1 = gy_first[sl][..., C].sum()
            tmean, dot = tstats[g, 0], tstats[g, 1]
            v = v + Gs1 / (M * sigma) * ((tx[sl] - tmean) - (xg - mu) * dot / (M * sigma * sigma))
        if apply_mask:
            v = _maskmul(v, xg, mask_slope)
        gx[sl] = v
    return _ret(gx, out)


def linear1_fwd(h, w, b):
    return h.reshape(h.shape[0], -1) @ w.reshape(-1) + (b if b is not None else 0.0)


def linear1_bwd_data(gs, w, mask, shape, mask_slope=0.2):
    gh = (gs.view(-1, 1) * w.reshape(1, -1)).view(shape)
    if mask is not None:
        gh = _maskmul(gh, mask.reshape(shape), mask_slope)
    return gh.contiguous()


def linear1_wgrad(gs, h, dw, db):
    dw += (gs.view(-1, 1) * h.reshape(h.shape[0], -1)).sum(0).view(dw.shape)
    if db is not None:
        db += gs.sum()


def gp_mix(real, fake, m, out=None):
    mm = m.view(-1, 1, 1, 1)
    return _ret(real * (1 - mm) + fake * mm, out)


def row_sumsq(g):
    return (g.reshape(g.shape[0], -1) ** 2).sum(1)


def gp_seed(g, ss, lam, target, inv_n):
    norm = ss.sqrt()
    gp = (norm - target) ** 2 * lam / target ** 2
    coef = inv_n * 2 * lam * (norm - target) / (target ** 2 * norm)
    return gp, g * coef.view(-1, 1, 1, 1)


def d_loss(scores, gp, N, eps):
    sr, sf = scores[:N], scores[N:2 * N]
    rl = -sr + sr * sr * eps
    d_cost = (sf + rl + gp).mean()
    gscore = torch.cat([(-1 + 2 * eps * sr) / N, torch.full((N,), 1.0 / N), torch.zeros(N)])
    return d_cost, rl.view(N, 1).clone(), sf.view(N, 1).clone(), gscore


def g_loss(scores):
    N = scores.shape[0]
    return (-scores).mean(), torch.full((N,), -1.0 / N)


def adam(p, g, m, v, lr, beta1, beta2, eps, bc1, bc2_sqrt, grad_scale=1.0):
    gg = g * grad_scale
    m.mul_(beta1).add_(gg, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(gg, gg, value=1 - beta2)
    p.sub_((lr / bc1) * m / (v.sqrt() / bc2_sqrt + eps))


def zero_(t):
    return t.zero_()
