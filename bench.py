#!/usr/bin/env python
"""bench.py — images/sec of the PGGAN train step on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--depth D] [--no-cpu] [--no-per-depth]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one Trainer.train() iteration (reference trainer.py:85-115): wgan_gp_D_loss + backward
(3 D passes batched, G forward, gradient-penalty double backward) + Adam(D), then wgan_gp_G_loss +
backward + Adam(G), on one synthetic minibatch already resident in HBM.  Headline workload: the
1024x1024 growth stage (depth 8) of the default-width (fmap_base 4096) CelebA-HQ-shape network with
the reference's per-depth minibatch (3 per GPU, plugins.py:20), fp32.  Data-parallel: one process
per GPU, minibatch per rank fixed (weak scaling), one RCCL sum-all-reduce of each network's flat
gradient buffer per iteration.  Rank 0 prints ONE JSON line (the last line of stdout).
Setup before the W warmup steps: --prime (default 50) untimed steps that load the code objects, grow the
caching allocator and let the clocks settle; the count is reported as "priming_steps".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_F32_PEAK = 157.3e12          # gfx950 f32-input MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
REF_MINIBATCH = {6: 14, 7: 6, 8: 3}          # reference plugins.py:19-20 (default 16)
# per-image forward FLOPs (2*MAC, conv+linear), fmap_base 4096, alpha 1 — SURVEY.md §8a / BASELINE.md §4
F_D = [0.0841, 0.6882, 3.1047, 6.7294, 10.3548, 13.9819, 17.6120, 21.2485, 24.8975]


def conv_flops(n, hout, wout, ks, pad, c_a, c_b):
    """Algorithmic FLOPs of one conv launch (forward, backward-data, tangent or weight gradient — all the
    same count): 2*N*Hout*Wout*Cout*Cin*taps with the reference's channel counts (513, not the stored 528) and
    one live tap per output pixel for the 1x1 -> 4x4 first layer of G / its transpose."""
    c_a = 513 if c_a == 528 else c_a
    c_b = 513 if c_b == 528 else c_b
    taps = ks * ks
    if ks == 4 and pad == 3:
        taps = 1
    return 2.0 * n * hout * wout * c_a * c_b * taps


class KernelTimer(object):
    """HIP-event timing of every MFMA conv launch of a step, recorded around the C-ABI call on the stream the
    kernel is launched on (weight gradients run on the second stream, so wrapping happens at ``ops`` level, inside
    the stream context).  The durations are the ones inside the two-stream step — what rocprofv3 --kernel-trace
    reports for the same command; ``--serial-kernel-timing`` switches the second stream off for isolated numbers."""

    def __init__(self, pg):
        self.pg, self.rec, self.saved = pg, [], {}

    def _wrap(self, name, describe):
        ops = self.pg.ops
        orig = getattr(ops, name)
        self.saved[name] = orig
        lib = self.pg._lib.load()

        def wrapped(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig(*a, **k)
            e1.record()
            sym = (lib.pg_debug_last_wino_kernel() if name == 'conv2d_wino' else
                   lib.pg_debug_last_wino_wgrad_kernel() if name == 'conv2d_wgrad_wino' else
                   lib.pg_debug_last_conv_kernel()).decode()
            fl, tag = describe(a, k)
            self.rec.append((sym, fl, e0, e1, '%s %s' % (tag, sym.replace('conv_', '').replace('_kernel', ''))))
            return out
        setattr(ops, name, wrapped)

    def __enter__(self):
        def conv_desc(a, k):          # conv2d(x, w, bias, N, Hin, Win, ks, pad, scale, ...)
            w, n, hin, win, ks, pad = a[1], a[3], a[4], a[5], a[6], a[7]
            ho, wo = hin + 2 * pad - ks + 1, win + 2 * pad - ks + 1
            return (conv_flops(n, ho, wo, ks, pad, w.shape[2], w.shape[3]),
                    'conv %d->%d k%d @%d n%d%s' % (w.shape[3], w.shape[2], ks, ho, n, ' masked' if k.get('mask') is not None else ''))

        def wgrad_desc(a, k):         # conv2d_wgrad(x, gz, dw, db, N, Hin, Win, ks, pad, scale, ups=)
            dw, n, hin, win, ks, pad = a[2], a[4], a[5], a[6], a[7], a[8]
            ho, wo = hin + 2 * pad - ks + 1, win + 2 * pad - ks + 1
            return (conv_flops(n, ho, wo, ks, pad, dw.shape[2], dw.shape[3]),
                    'wgrad %d->%d k%d @%d n%d' % (dw.shape[3], dw.shape[2], ks, ho, n))
        def wino_wgrad_desc(a, k):    # conv2d_wgrad_wino(x, gz, dw, db, N, H, W, scale, ups=)
            dw, n, h = a[2], a[4], a[5]
            return (conv_flops(n, h, h, 3, 1, dw.shape[2], dw.shape[3]),
                    'wgrad %d->%d k3 @%d n%d winograd' % (dw.shape[3], dw.shape[2], h, n))
        def pool_desc(a, k):          # conv2d_pool(x, w, bias, N, Hin, Win, ks, pad, scale, ...): the conv of conv_desc + pooled output
            fl, tag = conv_desc(a, k)
            return fl, tag + ' +pool'
        def generic(w_i, n_i, suffix):  # ops whose signature is (x, w, ..., N, Hin, Win, ks, pad, ...) with N at position n_i
            def desc(a, k):
                w, n, hin, win, ks, pad = a[w_i], a[n_i], a[n_i + 1], a[n_i + 2], a[n_i + 3], a[n_i + 4]
                ho, wo = hin + 2 * pad - ks + 1, win + 2 * pad - ks + 1
                return (conv_flops(n, ho, wo, ks, pad, w.shape[2], w.shape[3]),
                        'conv %d->%d k%d @%d n%d %s' % (w.shape[3], w.shape[2], ks, ho, n, suffix))
            return desc
        def wino_desc(a, k):          # conv2d_wino(x, u, bias, N, H, W, scale, ...): 3x3 pad 1 on Winograd-domain weights
            u, n, h = a[1], a[3], a[4]
            return (conv_flops(n, h, h, 3, 1, u.shape[1], u.shape[2]),
                    'conv %d->%d k3 @%d n%d winograd%s' % (u.shape[2], u.shape[1], h, n, ' masked' if k.get('mask') is not None else ''))
        self._wrap('conv2d', conv_desc)
        self._wrap('conv2d_wino', wino_desc)
        self._wrap('conv2d_pool', pool_desc)
        self._wrap('conv2d_pixelnorm', generic(1, 3, '+pixelnorm'))
        self._wrap('conv2d_unpool', generic(1, 2, '+unpool'))
        self._wrap('conv2d_pnbwd', generic(1, 4, '+pn adjoint'))
        self._wrap('conv2d_wgrad', wgrad_desc)
        self._wrap('conv2d_wgrad_wino', wino_wgrad_desc)

        def unpooled_desc(a, k):      # conv2d_unpooled(g, w, gbytes, gmul, gslope, N, Hin, Win, scale, ...)
            w, n, h = a[1], a[5], a[6]
            return (conv_flops(n, h, h, 3, 1, w.shape[2], w.shape[3]), 'conv %d->%d k3 @%d n%d pool adjoint in the gather' % (w.shape[3], w.shape[2], h, n))

        def wgrad_unpooled_desc(a, k):   # conv2d_wgrad_unpooled(x, g, gbytes, gmul, gslope, dw, db, N, Hin, Win, scale)
            dw, n, h = a[5], a[7], a[8]
            return (conv_flops(n, h, h, 3, 1, dw.shape[2], dw.shape[3]), 'wgrad %d->%d k3 @%d n%d pool adjoint in the gather' % (dw.shape[3], dw.shape[2], h, n))
        self._wrap('conv2d_unpooled', unpooled_desc)
        self._wrap('conv2d_wgrad_unpooled', wgrad_unpooled_desc)
        return self

    def __exit__(self, *exc):
        for name, orig in self.saved.items():
            setattr(self.pg.ops, name, orig)

    def summary(self, nsteps):
        torch.cuda.synchronize()
        fam = {}
        self.table = {}
        per_tag = {}
        for family, fl, e0, e1, tag in self.rec:
            per_tag.setdefault(tag, []).append(e0.elapsed_time(e1))
        med = {tag: sorted(v)[len(v) // 2] for tag, v in per_tag.items()}
        for family, fl, e0, e1, tag in self.rec:
            ms = e0.elapsed_time(e1)
            if ms > 10.0 * med[tag]:                       # a one-off stall (allocator / first touch) is not the kernel
                ms = med[tag]
            t = self.table.setdefault(tag, dict(flops=0.0, ms=0.0, launches=0))
            t['flops'] += fl
            t['ms'] += ms
            t['launches'] += 1
            d = fam.setdefault(family, dict(flops=0.0, ms=0.0, launches=0))
            d['flops'] += fl
            d['ms'] += ms
            d['launches'] += 1
        for d in fam.values():
            d['flops_per_step'] = d['flops'] / nsteps
            d['ms_per_step'] = d['ms'] / nsteps
            d['launches_per_step'] = d['launches'] / nsteps
            d['avg_launch_us'] = 1e3 * d['ms'] / max(1, d['launches'])
            d['tflops'] = d['flops'] / (d['ms'] * 1e-3) / 1e12 if d['ms'] > 0 else 0.0
        return fam


def make_trainer(pg, res, depth, alpha, mb, seed, dp, fmap_base=4096):
    torch.manual_seed(1337)                       # same weights on every rank (train.py:21)
    shape = (1, 3, res, res)
    G = pg.Generator(shape, fmap_base=fmap_base).cuda()
    D = pg.Discriminator(shape, fmap_base=fmap_base).cuda()
    G.depth = D.depth = depth
    G.alpha = D.alpha = alpha
    gs = 1.0 if dp is None else dp.grad_scale
    opt_g = pg.FusedAdam(G.parameters(), 0.001, betas=(0.0, 0.99), grad_scale=gs)
    opt_d = pg.FusedAdam(D.parameters(), 0.001, betas=(0.0, 0.99), grad_scale=gs)
    ds = pg.utils.SyntheticDataset(res, 3, seed=seed)
    ds.model_depth = depth
    pg.wgan_gp_loss.manual_seed(seed)
    tr = pg.Trainer(D, G, pg.wgan_gp_D_loss, pg.wgan_gp_G_loss, opt_d, opt_g, ds, ds.loader(mb),
                    pg.utils.device_latents(mb, 512, seed=seed + 7), parallel=dp)
    return tr


# A cold box needs ~1 s of work before the step time settles (first launches load code objects, the caching allocator
# grows, the clocks ramp): measured 17.2 ms on the first 30 steps of a fresh box vs 16.0 ms afterwards.  These setup
# steps run before the W warmup steps of the contract and are reported as "priming_steps" in the JSON line.
PRIME_STEPS = 50


def timed_steps(tr, steps, warmup, dp):
    for _ in range(warmup):
        tr.train()
    if dp is not None:
        dp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train()
    torch.cuda.synchronize()
    if dp is not None:
        dp.barrier()
    dt = time.perf_counter() - t0
    if dp is not None:
        t = torch.tensor([dt], device='cuda', dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    return dt


def d_step_ms(tr, steps):
    """per-depth 'D+GP ms' = wgan_gp_D_loss + backward + Adam(D)."""
    real = next(tr.dataiter)
    z = tr.random_latents_generator()
    def one():
        c = tr.D_loss(tr.D, tr.G, real, z)[0]
        c.backward()
        if tr.parallel is not None:                          # the D-step of Trainer.train(): gradients summed over ranks
            tr.parallel.all_reduce_grads(tr.D)
        tr.optimizer_d.step()
    for _ in range(2):
        one()
    if tr.parallel is not None:
        tr.parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if tr.parallel is not None:
        t = torch.tensor([dt], device='cuda', dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    return 1e3 * dt / steps


def cpu_baseline(depth, mb):
    """The CPU oracle (a restatement of the reference's op sequence, pinned to the reference by the
    golden fixtures) timed on this host: ONE full train iteration of the same workload (bounded sample,
    ~10 s; torch-CPU/oneDNN degrades with >32 threads on these convolutions, so at most 32 threads are
    used and that is the core count reported)."""
    from oracle import pggan_cpu as oc
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 32)
    torch.set_num_threads(cores)
    res = 4 * 2 ** depth
    cfg = oc.NetCfg(1024, 3)
    torch.manual_seed(1337)
    gp, dp_ = oc.init_generator(cfg), oc.init_discriminator(cfg)
    real, z_d, z_g, mix = oc.synthetic_batch(1337, mb, 3, res, 512)
    og, od = oc.AdamState(), oc.AdamState()
    t0 = time.perf_counter()
    oc.train_iteration(gp, dp_, cfg, og, od, real, z_d, z_g, mix, depth, 1.0, 1e-3, 1e-3)
    dt = time.perf_counter() - t0
    return dict(value=mb / dt, unit='images/sec', cores=cores, kind='port',
                sample='1 full train iteration (D+GP step, G step, Adam) at depth %d (%dx%d), minibatch %d, '
                       'torch-CPU fp32 oracle, %d threads, %.1f s' % (depth, res, res, mb, cores, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--prime', type=int, default=PRIME_STEPS, help='untimed setup steps before the warmup (see PRIME_STEPS)')
    ap.add_argument('--depth', type=int, default=8)
    ap.add_argument('--alpha', type=float, default=1.0)
    ap.add_argument('--minibatch', type=int, default=0, help='per-GPU minibatch (default: reference schedule)')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-per-depth', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--graphs', action='store_true', help='replay every stage from captured hipGraphs (graphs.py); default: only the launch-bound 4x4 stage, eager two-stream launching elsewhere (measured faster)')
    ap.add_argument('--serial-kernel-timing', action='store_true', help='instrumented passes with the weight-gradient stream off '
                    '(isolated per-kernel durations instead of the durations inside the two-stream step)')
    ap.add_argument('--kernel-table', action='store_true', help='per-layer conv timing table on stderr')
    args = ap.parse_args()

    import pggan_amd as pg
    if os.environ.get('PGGAN_TUNE'):                       # kernel A/B aid: "key=value,..." -> pg_debug_set_tuning
        for kv in os.environ['PGGAN_TUNE'].split(','):
            pg._lib.load().pg_debug_set_tuning(*[int(v) for v in kv.split('=')])
    pg.wgan_gp_loss.enable_graphs(True if args.graphs else 'auto')   # 'auto': hipGraph replay only at the launch-bound 4x4 stage
    world = int(os.environ.get('WORLD_SIZE', '1'))
    force_dp = os.environ.get('PGGAN_FORCE_DP', '') == '1'      # one-rank RCCL group: smoke test of the DP code path
    dp = pg.parallel.DataParallel.from_env(force=force_dp) if (world > 1 or force_dp) else None
    rank = 0 if dp is None else dp.rank
    if dp is None:
        torch.cuda.set_device(0)
    n_gpus = world
    depth = args.depth
    res = 4 * 2 ** depth
    mb = args.minibatch or REF_MINIBATCH.get(depth, 16)

    tr = make_trainer(pg, 1024, depth, args.alpha, mb, pg.parallel.shard_seed(1337, rank), dp)
    if dp is not None:
        dp.broadcast_params(tr.G, tr.D)
    for _ in range(args.prime):           # setup, untimed and reported: code-object loading, allocator growth, clock ramp
        tr.train()
    dt = timed_steps(tr, args.steps, args.warmup, dp)
    ms_per_step = 1e3 * dt / args.steps
    value = n_gpus * mb * args.steps / dt

    out = {
        'metric': 'images/sec, PGGAN full train step (D+GP step + G step + Adam) at %dx%d' % (res, res),
        'value': value, 'unit': 'images/sec', 'n_gpus': n_gpus, 'steps': args.steps, 'warmup': args.warmup, 'priming_steps': args.prime,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'PGGAN (default widths fmap_base=4096, C=3, latent 512) growth stage depth %d = %dx%d, '
                               'alpha %.2f, minibatch %d per GPU (reference per-depth schedule), Trainer.train() with '
                               'WGAN-GP (lambda 10) and Adam(0,0.99), fp32' % (depth, res, res, args.alpha, mb),
                   'resolution': res, 'depth': depth, 'minibatch_per_gpu': mb, 'global_batch': mb * n_gpus,
                   'parallelism': 'dp%d' % n_gpus},
    }
    W = (14 * F_D[depth] + 4 * F_D[depth]) * 1e9 if depth < len(F_D) else None     # F_G ~= F_D
    if W:
        out['step_algorithmic_gflop_per_image'] = W / 1e9
        out['step_mfma_frac'] = W * (value / n_gpus) / MFMA_F32_PEAK

    out['config']['hip_graphs'] = bool((args.graphs or depth == 0) and args.alpha >= 1.0)
    if rank == 0 and not args.no_kernel_timing:
        psteps = 3
        pg.wgan_gp_loss.enable_graphs(False)               # per-launch HIP events need eager launches
        async_wgrad = pg.engine.ASYNC_WGRAD
        if args.serial_kernel_timing:
            pg.engine.ASYNC_WGRAD = False
        try:
            with KernelTimer(pg) as kt:
                for _ in range(psteps):
                    tr.train()
                fam = kt.summary(psteps)
        finally:
            pg.engine.ASYNC_WGRAD = async_wgrad
        if args.kernel_table:
            for tag, t in sorted(kt.table.items(), key=lambda kv: -kv[1]['ms']):
                sys.stderr.write('%-72s calls/step %5.1f  ms/step %8.3f  TFLOP/s %7.2f\n' % (
                    tag, t['launches'] / psteps, t['ms'] / psteps, t['flops'] / (t['ms'] * 1e-3) / 1e12))
        dom = max(fam, key=lambda k: fam[k]['ms'])
        d = fam[dom]
        traffic = None
        try:                                   # HBM bytes per launch of that symbol from the committed PMC summary
            with open(os.path.join(ROOT, 'profiles', 'r01_roofline.json')) as f:
                traffic = json.load(f)['per_kernel'][dom]['hbm_bytes_per_launch']
        except Exception:
            pass
        out['roofline'] = {'bound': 'mfma', 'kernel': dom, 'achieved': d['tflops'], 'peak': MFMA_F32_PEAK / 1e12,
                           'unit': 'TFLOP/s', 'frac': d['tflops'] * 1e12 / MFMA_F32_PEAK, 'traffic': traffic,
                           'avg_launch_us': d['avg_launch_us'], 'launches_per_step': d['launches_per_step'],
                           'ms_per_step_in_kernel': d['ms_per_step'],
                           'algorithmic_gflop_per_step': d['flops_per_step'] / 1e9}
        out['kernels'] = {k: {'tflops': v['tflops'], 'ms_per_step': v['ms_per_step'],
                              'launches_per_step': v['launches_per_step'], 'avg_launch_us': v['avg_launch_us']}
                          for k, v in fam.items()}
    elif dp is not None and not args.no_kernel_timing:
        pg.wgan_gp_loss.enable_graphs(False)
        for _ in range(3):                                  # keep collectives matched with rank 0
            tr.train()
    pg.wgan_gp_loss.enable_graphs(True if args.graphs else 'auto')

    if not args.no_per_depth:
        per = []
        del tr
        torch.cuda.empty_cache()
        for d in range(0, 9):
            m = REF_MINIBATCH.get(d, 16)
            t = make_trainer(pg, 1024, d, 1.0, m, pg.parallel.shard_seed(1337, rank), dp)
            if dp is not None:
                dp.broadcast_params(t.G, t.D)
            k = 100 if d <= 1 else (40 if d <= 3 else (20 if d <= 5 else (8 if d <= 7 else 5)))
            tt = timed_steps(t, k, 5 if d <= 3 else 3, dp)            # max over ranks; minibatch m PER RANK (weak scaling)
            dms = d_step_ms(t, k)
            Wd = 18 * F_D[d] * 1e9
            per.append({'depth': d, 'res': 4 * 2 ** d, 'minibatch': m, 'images_per_sec': n_gpus * m * k / tt,
                        'ms_per_step': 1e3 * tt / k, 'd_step_gp_ms': dms,
                        'mfma_frac': Wd * (m * k / tt) / MFMA_F32_PEAK})
            del t
            torch.cuda.empty_cache()
        out['per_depth'] = per

    if rank == 0:
        if not args.no_cpu and n_gpus == 1:
            out['cpu_baseline'] = cpu_baseline(depth, mb)
        else:
            out['cpu_baseline'] = None
    if dp is not None:
        dp.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: RCCL prints its banner through C stdio, which is still buffered here
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
