#!/usr/bin/env python
"""bench.py — images/sec of the PGGAN train step on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--depth D] [--alpha A] [--fmap-base F] [--config {2,3,4,5}]
                    [--no-cpu] [--no-per-depth] [--no-configs]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one Trainer.train() iteration (reference trainer.py:85-115): wgan_gp_D_loss + backward
(3 D passes batched, G forward, gradient-penalty double backward) + Adam(D), then wgan_gp_G_loss +
backward + Adam(G), on one synthetic minibatch already resident in HBM.  Headline workload (BASELINE.json config 5):
the 1024x1024 growth stage (depth 8) of the default-width (fmap_base 4096) CelebA-HQ-shape network with
the reference's per-depth minibatch (3 per GPU, plugins.py:20), fp32.  Data-parallel: one process
per GPU, minibatch per rank fixed (weak scaling), RCCL sum-all-reduce of each network's flat gradient buffer
(bucketed, overlapped with the backward sweeps) through the library's C-ABI.  ``--gpus N`` with N > 1 and no
WORLD_SIZE in the environment launches the N ranks itself (torch.distributed.run) and fails loudly when fewer than N
devices are visible.  Rank 0 prints ONE JSON line (the last line of stdout), at most 6 KB: the contract keys, ``roofline``,
``cpu_baseline``, the D+GP window and one row per growth stage.  Every table behind it (per-kernel timings, secondary workloads, CPU
per-depth numbers, timing windows) is written to ``bench_detail.json`` next to this file (a short digest goes to stderr).

Fractions (all <= 1 by construction; Winograd F(2x2,3x3) launches are credited 16/36 of the 2*MAC count, peak = the nominal 157.3 TF of
fp32 MFMA): ``roofline.frac`` = MFMA FLOP the launches of the dominant conv symbol EXECUTE / their HIP-event time (per-launch event
pairs on the launch's stream, inside the step as it is issued in the timed loop: ``_lib.CALL_HOOK`` also runs on launch-plan replay);
``executed_mfma_frac`` / per-depth ``executed_frac`` / ``d_step_gp.executed_frac`` = MFMA FLOP all conv launches of a step (window)
execute / the step's (window's) wall time -- the chip's utilisation over the step, indifferent to how launches overlap on the streams
(``conv_kernel_executed_*`` in bench_detail.json: the same FLOP / the launches' summed time, which counts concurrent launches twice).
The algorithmic 2*MAC work of the reference's convolutions is reported as a RATE (``algorithmic_tflops``; it can exceed
what the matrix cores execute, so it is never called a fraction outside ``roofline.algorithmic_frac`` of the dominant kernel);
``mfma_busy_pct`` = time-weighted SQ_VALU_MFMA_BUSY_CYCLES of the conv kernels from the committed PMC pass
(``mfma_busy_source``).  Setup before the W warmup steps: --prime (default 50) untimed steps (code objects, allocator
growth, clock ramp), reported as "priming_steps".
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# Data-parallel runs only: with the library's RCCL communicator alive the step uses three HIP streams (main, weight gradients,
# gradient exchange) next to RCCL's own; under ROCm's default of 4 hardware queues two of them share a queue and serialise
# (measured with a one-rank communicator: 184 vs 218 img/s; the control-plane backend makes no difference).  Single-GPU runs
# keep the default: 8 queues slow the hipGraph replay of the 4x4 stage down (3.8 vs 2.3 ms per step).  Must be set before the
# HIP runtime is loaded.
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')    # kernel arguments in device memory (the runtime's default on this image; =0 costs 0.33 ms per 1024^2 step: 318 launches)
if int(os.environ.get('WORLD_SIZE', '1')) > 1 or os.environ.get('PGGAN_FORCE_DP', '') == '1':
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import torch  # noqa: E402

MFMA_F32_PEAK = 157.3e12          # gfx950 f32-input MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
REF_MINIBATCH = {6: 14, 7: 6, 8: 3}          # reference plugins.py:19-20 (default 16)
def _latest_profile_tag():
    """profiles/<tag>_roofline.json of the newest round: the PMC pass the traffic / MFMA-busy figures are quoted from.  A round may
    commit several sets of the same workload from different boxes of the pool (``r06``, ``r06b``, ...: the counter figure's denominator is
    kernel time x the NOMINAL clock, so a box that clocks lower reads lower): the line quotes the WORSE one -- the set whose
    ``d_step_gp_window.mfma_busy_pct_all_kernels`` is smallest -- and lists every set's figure in ``d_step_gp.boxes``."""
    import glob
    tags = sorted(os.path.basename(f)[:-len('_roofline.json')] for f in glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]*_roofline.json'))
                  if len(os.path.basename(f)) <= len('r00x_roofline.json'))
    if not tags:
        return 'none', {}
    rnd = tags[-1][:3]
    figures = {}
    for t in tags:
        if t[:3] != rnd:
            continue
        try:
            with open(os.path.join(ROOT, 'profiles', t + '_roofline.json')) as f:
                figures[t] = json.load(f).get('d_step_gp_window', {}).get('mfma_busy_pct_all_kernels')
        except Exception:
            figures[t] = None
    known = {t: v for t, v in figures.items() if v is not None}
    return (min(known, key=known.get) if known else rnd), figures


PROFILE_TAG, PROFILE_BOXES = _latest_profile_tag()


def _rocprof_avg_us(symbol):
    """Average duration (us) of a conv symbol in profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the bench command)."""
    import csv
    try:
        with open(os.path.join(ROOT, 'profiles', PROFILE_TAG + '_kernel_stats.csv')) as f:
            for row in csv.DictReader(f):
                name = row['Name']
                if symbol in name and name.split(symbol, 1)[1][:1] in ('(', ''):
                    return float(row['AverageNs']) / 1e3
    except Exception:
        pass
    return None


def forward_flops(G, D, depth, alpha):
    """Per-image forward FLOPs (2*MAC of every conv + the final linear; elementwise work excluded) of G and D at a
    growth stage, with the reference's channel counts (513, not the stored 528) and ONE live tap per output pixel for
    the 4x4 pad-3 conv on the 1x1 latent (depth 8, fmap_base 4096 -> F_G 24.8974, F_D 24.8975 GFLOP)."""
    C = D.num_channels
    r = 4 * 2 ** depth

    def conv(m, h, taps=None):
        return 2.0 * h * h * m.ch_in * m.ch_out * (m.ksize * m.ksize if taps is None else taps)
    b0 = G.block0
    fg = conv(b0.c1, 4, taps=1) + conv(b0.c2, 4)
    h = 4
    for i in range(depth):
        h *= 2
        fg += conv(G.blocks[i].c1, h) + conv(G.blocks[i].c2, h)
    fg += conv(G.blocks[depth - 1].toRGB if depth > 0 else b0.toRGB, r)
    if depth > 0 and alpha < 1.0:
        fg += conv(G.blocks[depth - 2].toRGB if depth > 1 else b0.toRGB, r // 2)
    nb = len(D.blocks)
    e = nb - 1 - depth
    fd = conv(D.blocks[e].fromRGB, r)
    if depth > 0 and alpha < 1.0:
        fd += conv(D.blocks[e + 1].fromRGB, r // 2)
    h = r
    for j in range(e, nb - 1):
        fd += conv(D.blocks[j].c1, h) + conv(D.blocks[j].c2, h)
        h //= 2
    last = D.blocks[nb - 1]
    fd += conv(last.c1, 4) + 2.0 * 16 * last.c2.ch_in * last.c2.ch_out + 2.0 * D.linear.in_features
    return fg, fd


def step_flops(G, D, depth, alpha):
    """Algorithmic FLOP per image of the D-step + GP (12 F_D + F_G) and of the full iteration (14 F_D + 4 F_G),
    SURVEY.md §8d."""
    fg, fd = forward_flops(G, D, depth, alpha)
    return 12 * fd + fg, 14 * fd + 4 * fg


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


# C-ABI entry points that launch an MFMA conv kernel: positions of (N, H, W, Cin, Cout, KS, pad) in their argument lists
# (include/pggan_hip.h; KS / pad None = 3 / 1), and which thread-local "last kernel" query names the symbol that ran
CONV_ENTRY = {
    'pg_conv2d_nhwc': (5, 6, 7, 8, 9, 10, 11, 'conv', ''),
    'pg_conv2d_wgrad_nhwc': (4, 5, 6, 7, 8, 9, 10, 'conv', 'wgrad'),
    'pg_conv2d_pool_nhwc': (10, 11, 12, 13, 14, 15, 16, 'conv', '+pool'),
    'pg_conv2d_pixelnorm_nhwc': (5, 6, 7, 8, 9, 10, 11, 'conv', '+pixelnorm'),
    'pg_conv2d_pnbwd_nhwc': (5, 6, 7, 8, 9, 10, 11, 'conv', '+pn adjoint'),
    'pg_conv2d_unpool_nhwc': (5, 6, 7, 8, 9, 10, 11, 'conv', '+unpool'),
    'pg_conv2d_unpooled_nhwc': (7, 8, 9, 10, 11, None, None, 'conv', 'pool adjoint in the gather'),
    'pg_conv2d_wgrad_unpooled_nhwc': (7, 8, 9, 10, 11, None, None, 'conv', 'wgrad, pool adjoint in the gather'),
    'pg_conv2d_pixelnorm_torgb_nhwc': (9, 11, 12, 13, 14, None, None, 'conv', '+pixelnorm +toRGB'),
    'pg_conv2d_masked_fromrgb_bwd_nhwc': (11, 13, 14, 15, 16, None, None, 'conv', 'masked +fromRGB adjoint / wgrad'),
    'pg_conv2d_fromrgb_nhwc': (10, 12, 13, 14, 15, None, None, 'conv', 'fromRGB in the gather'),
    'pg_conv2d_wino_nhwc': (13, 14, 15, 16, 17, None, None, 'wino', 'winograd'),
    'pg_conv2d_wino_pixelnorm_nhwc': (5, 6, 7, 8, 9, None, None, 'wino', 'winograd +pixelnorm'),
    'pg_conv2d_wino_pnbwd_nhwc': (9, 10, 11, 12, 13, None, None, 'wino', 'winograd +pn adjoint'),
    'pg_conv2d_wgrad_wino_nhwc': (4, 5, 6, 7, 8, None, None, 'wwino', 'wgrad winograd'),
    'pg_conv2d_wgrad_wino2_nhwc': (None, 9, 10, 11, 12, None, None, 'wwino', 'wgrad winograd'),      # N = args[2] + args[5]
}


class KernelTimer(object):
    """HIP-event timing of every MFMA conv launch of a step, recorded around the C-ABI call on the stream the kernel is launched on
    (``_lib.CALL_HOOK``: eager calls and calls replayed from a launch plan alike, so the timed steps are issued exactly as the
    measured ones -- the host stays ahead of the device and the event pair brackets the kernel, not a wait for the host; an eager
    instrumented pass is host-bound on the 3-image launches and read 74 us where rocprofv3 has 58).  The durations are the ones
    inside the multi-stream step -- what rocprofv3 --kernel-trace reports for the same command."""

    def __init__(self, pg):
        self.pg, self.rec, self.streams = pg, [], {}

    def _stream(self, handle):
        h = int(handle or 0)
        s = self.streams.get(h)
        if s is None:
            s = self.streams[h] = torch.cuda.default_stream() if h == 0 else torch.cuda.ExternalStream(h)
        return s

    def hook(self, fn, args, name):
        spec = CONV_ENTRY.get(name)
        if spec is None:
            return fn(*args)
        s = self._stream(args[-1])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        rc = fn(*args)
        e1.record(s)
        if rc:
            return rc
        lib = self.lib
        sym = (lib.pg_debug_last_wino_kernel() if spec[7] == 'wino' else lib.pg_debug_last_wino_wgrad_kernel() if spec[7] == 'wwino'
               else lib.pg_debug_last_conv_kernel()).decode()
        n = args[2] + args[5] if spec[0] is None else args[spec[0]]
        h, w, cin, cout = args[spec[1]], args[spec[2]], args[spec[3]], args[spec[4]]
        ks = 3 if spec[5] is None else args[spec[5]]
        pad = 1 if spec[6] is None else args[spec[6]]
        ho, wo = h + 2 * pad - ks + 1, w + 2 * pad - ks + 1
        fl = conv_flops(n, ho, wo, ks, pad, cout, cin)
        self.rec.append((sym, fl, e0, e1, '%s %d->%d k%d @%d n%d %s' % (spec[8], cin, cout, ks, ho, n, sym.replace('conv_', '').replace('_kernel', ''))))
        return 0

    def __enter__(self):
        self.lib = self.pg._lib.load()
        # What an event pair adds to the kernel between its records (the two packets' processing + the dispatch of the kernel behind
        # the first): calibrated as the pair around a 16 K-element add (whose own ~2 us are left in: the correction never exceeds
        # the overhead) and subtracted from every launch, so that the figure is the kernel's duration as rocprofv3 --kernel-trace
        # reports it rather than kernel + event packets
        torch.cuda.synchronize()
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
        buf = torch.zeros(1 << 14, device='cuda')
        for a, b in pairs:
            buf.add_(1.0)                                   # (a kernel ahead of the pair, as in the step: the pair never starts on an idle queue)
            a.record()
            buf.add_(1.0)
            b.record()
        torch.cuda.synchronize()
        self.pair_ms = max(0.0, sorted(a.elapsed_time(b) for a, b in pairs)[len(pairs) // 2] - 0.002)
        self.saved = self.pg._lib.CALL_HOOK
        self.pg._lib.CALL_HOOK = self.hook
        return self

    def __exit__(self, *exc):
        self.pg._lib.CALL_HOOK = self.saved

    def summary(self, nsteps):
        torch.cuda.synchronize()
        fam = {}
        self.table = {}
        per_tag = {}
        raw = [max(e0.elapsed_time(e1) - self.pair_ms, 0.5 * e0.elapsed_time(e1)) for _, _, e0, e1, _ in self.rec]
        for (family, fl, e0, e1, tag), ms in zip(self.rec, raw):
            per_tag.setdefault(tag, []).append(ms)
        med = {tag: sorted(v)[len(v) // 2] for tag, v in per_tag.items()}
        for (family, fl, e0, e1, tag), ms in zip(self.rec, raw):
            if ms > 10.0 * med[tag]:                       # a one-off stall (allocator / first touch) is not the kernel
                ms = med[tag]
            t = self.table.setdefault(tag, dict(flops=0.0, ms=0.0, launches=0))
            t['flops'] += fl
            t['ms'] += ms
            t['launches'] += 1
            d = fam.setdefault(family, dict(flops=0.0, exec_flops=0.0, ms=0.0, launches=0))
            d['flops'] += fl
            d['exec_flops'] += fl * (16.0 / 36.0 if 'wino' in family else 1.0)     # F(2x2,3x3) issues 16 of the 36 multiplies
            d['ms'] += ms
            d['launches'] += 1
        for d in fam.values():
            d['flops_per_step'] = d['flops'] / nsteps
            d['ms_per_step'] = d['ms'] / nsteps
            d['launches_per_step'] = d['launches'] / nsteps
            d['avg_launch_us'] = 1e3 * d['ms'] / max(1, d['launches'])
            d['tflops'] = d['flops'] / (d['ms'] * 1e-3) / 1e12 if d['ms'] > 0 else 0.0
            d['exec_tflops'] = d['exec_flops'] / (d['ms'] * 1e-3) / 1e12 if d['ms'] > 0 else 0.0
        return fam



def make_trainer(pg, res, depth, alpha, mb, seed, dp, fmap_base=4096, channels=3, ring=8, host_data=False, lookahead=False):
    torch.manual_seed(1337)                       # same weights on every rank (train.py:21)
    shape = (1, channels, res, res)
    G = pg.Generator(shape, fmap_base=fmap_base).cuda()
    D = pg.Discriminator(shape, fmap_base=fmap_base).cuda()
    G.depth = D.depth = depth
    G.alpha = D.alpha = alpha
    opt_g = pg.FusedAdam(G.parameters(), 0.001, betas=(0.0, 0.99))     # 1/world gradient pre-scale: set by Trainer(parallel=...)
    opt_d = pg.FusedAdam(D.parameters(), 0.001, betas=(0.0, 0.99))
    ds = pg.utils.SyntheticDataset(res, channels, seed=seed, ring=ring, host=host_data)   # pre-generated ring: no RNG kernels in the timed steps
    ds.model_depth = depth
    pg.wgan_gp_loss.manual_seed(seed)
    tr = pg.Trainer(D, G, pg.wgan_gp_D_loss, pg.wgan_gp_G_loss, opt_d, opt_g, ds, ds.loader(mb),
                    pg.utils.device_latents(mb, G.latent_size, seed=seed + 7, ring=2 * ring), parallel=dp, prefetch_inputs=lookahead)
    if dp is not None:
        dp.broadcast_params(tr.G, tr.D)
    return tr


# A cold box needs ~1 s of work before the step time settles (first launches load code objects, the caching allocator
# grows, the clocks ramp): measured 17.2 ms on the first 30 steps of a fresh box vs 16.0 ms afterwards.  These setup
# steps run before the W warmup steps of the contract and are reported as "priming_steps" in the JSON line.
PRIME_STEPS = 50


def _max_over_ranks(dt, dp):
    if dp is None:
        return dt
    t = torch.tensor([dt], device='cuda' if torch.distributed.get_backend() == 'nccl' else 'cpu', dtype=torch.float64)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t)


def _window_marker():
    """One launch of a kernel no train step uses (minmax_kernel of sound.hip) on the current stream: delimits the timed region in
    a rocprofv3 counter collection, so that tools/summarize_profile.py sums the kernels of the window and not those of the setup."""
    import pggan_amd as pg
    buf = _window_marker.__dict__.setdefault('buf', torch.zeros(4, device='cuda'))
    pg._lib.call('pg_minmax_f32', buf.data_ptr(), 2, buf.data_ptr() + 8, torch.cuda.current_stream().cuda_stream)


def timed_steps(tr, steps, warmup, dp, fn=None, markers=False):
    fn = tr.train if fn is None else fn
    for _ in range(warmup):
        fn()
    if dp is not None:
        dp.barrier()
    torch.cuda.synchronize()
    if markers:
        _window_marker()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    stamps = []
    for _ in range(steps):
        fn()
        stamps.append(time.perf_counter())
    HOST_ENQUEUE['ms'] = 1e3 * (stamps[-1] - t0) / steps           # host time to ENQUEUE a step, averaged over the whole loop: includes the time the
                                                                    # runtime holds the host back once it is a queue's depth ahead of the device
    head = [b - a for a, b in zip([t0] + stamps[:9], stamps[:10])]
    HOST_ENQUEUE['free_ms'] = 1e3 * sorted(head)[len(head) // 2]   # median of the first ten steps after the synchronisation: the host running free
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    if markers:
        _window_marker()
        torch.cuda.synchronize()
    if dp is not None:
        dp.barrier()
        dt_local = time.perf_counter() - t0
    return _max_over_ranks(dt_local, dp)


HOST_ENQUEUE = {'ms': None, 'free_ms': None}


def robust_ms(tr, dp, fn=None, prime=20, window_s=0.35, windows=3):
    """ms per call of ``fn`` (default: a train step), robust against one-off stalls: ``prime`` untimed calls, then the
    MEDIAN of ``windows`` timed windows of >= ``window_s`` seconds each (the call count per window is derived from a
    short probe and agreed across ranks)."""
    fn = tr.train if fn is None else fn
    probe = timed_steps(tr, 3, prime, dp, fn) / 3
    k = max(3, int(window_s / max(probe, 1e-5)) + 1)
    if dp is not None:
        t = torch.tensor([k], device='cuda' if torch.distributed.get_backend() == 'nccl' else 'cpu', dtype=torch.int64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        k = int(t)
    ms = sorted(1e3 * timed_steps(tr, k, 0, dp, fn) / k for _ in range(windows))
    return ms[len(ms) // 2], k, ms


def d_step_fn(tr):
    """per-depth 'D+GP ms' = wgan_gp_D_loss + backward + gradient exchange + Adam(D)."""
    real = next(tr.dataiter)
    z = tr.random_latents_generator()

    def one():
        c = tr.D_loss(tr.D, tr.G, real, z)[0]
        c.backward()
        tr._exchange(tr.D)
        tr.optimizer_d.step()
    return one


CPU_SAMPLE_MAX_MB = 4


def cpu_baseline(depth, mb, per_depth=True):
    """The CPU oracle (a restatement of the reference's op sequence, pinned to the reference by the golden fixtures) timed on
    this host's cores, as BASELINE.md §3 specifies: per growth stage 1 warm-up + 2 timed full train iterations (the warm-up
    absorbs oneDNN primitive creation); ``value`` is the headline stage.  Bounded sample: the CPU minibatch is
    min(reference minibatch, 4) -- torch-CPU convolution throughput per image is flat in N there -- and at most 32 threads are
    used (torch-CPU/oneDNN degrades beyond that on these convolutions); that is the core count reported."""
    from oracle import pggan_cpu as oc
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 32)
    torch.set_num_threads(cores)
    cfg = oc.NetCfg(1024, 3)

    def one_depth(d, m):
        res = 4 * 2 ** d
        torch.manual_seed(1337)
        gp, dp_ = oc.init_generator(cfg), oc.init_discriminator(cfg)
        real, z_d, z_g, mix = oc.synthetic_batch(1337, m, 3, res, 512)
        og, od = oc.AdamState(), oc.AdamState()
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            oc.train_iteration(gp, dp_, cfg, og, od, real, z_d, z_g, mix, d, 1.0, 1e-3, 1e-3)
            times.append(time.perf_counter() - t0)
        dt = 0.5 * (times[1] + times[2])
        return {'depth': d, 'res': res, 'minibatch': m, 'images_per_sec': m / dt, 'ms_per_step': 1e3 * dt,
                'warmup_iteration_s': times[0], 'timed_iterations_s': times[1:]}
    t_all = time.perf_counter()
    stages = []
    for d in (range(0, 9) if per_depth else [depth]):
        m = mb if d == depth else REF_MINIBATCH.get(d, 16)
        stages.append(one_depth(d, min(m, CPU_SAMPLE_MAX_MB)))
    head = [e for e in stages if e['depth'] == depth][0]
    out = dict(value=head['images_per_sec'], unit='images/sec', cores=cores, kind='port',
               sample='per growth stage: 1 warm-up + 2 timed full train iterations (D+GP step, G step, Adam) of the torch-CPU fp32 oracle, '
                      'minibatch min(reference, %d), %d threads; value = depth %d (%dx%d, minibatch %d); %.0f s in all'
                      % (CPU_SAMPLE_MAX_MB, cores, depth, head['res'], head['res'], head['minibatch'], time.perf_counter() - t_all))
    if per_depth:
        out['per_depth'] = stages
    return out


def relaunch(n):
    """``python bench.py --gpus N`` without a torchrun environment: start the N ranks here."""
    have = torch.cuda.device_count()
    if have < n and os.environ.get('PGGAN_DP_SHARE_GPU', '') != '1':       # (test aid: all ranks on device 0, see parallel.from_env; its numbers mean nothing)
        sys.stderr.write('bench.py: --gpus %d requested but only %d GPU(s) are visible on this node; refusing to '
                         'report an N=%d number measured on fewer devices\n' % (n, have, n))
        sys.exit(2)
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def executed_fraction(pg, tr, steps=2):
    """MFMA FLOP the conv / weight-gradient launches of ``steps`` eager train steps EXECUTE (Winograd F(2x2,3x3) launches credited
    16/36 of the algorithmic 2*MAC count) per second of their summed HIP-event time, over the nominal peak: a fraction <= 1 by
    construction, for every growth stage (the algorithmic rate of a Winograd stage can exceed the peak; it is reported as a rate,
    never as a fraction)."""
    mode = pg.wgan_gp_loss._use_graphs
    if pg.wgan_gp_loss._replay_mode(tr.G) == 'graph':  # a hipGraph replay has no per-launch hook: eager launches there (the switch itself: recorded plans stay)
        pg.wgan_gp_loss._use_graphs = False
    try:
        with KernelTimer(pg) as kt:
            for _ in range(steps):
                tr.train()
            fam = kt.summary(steps)
    finally:
        pg.wgan_gp_loss._use_graphs = mode
    tot = sum(v['ms'] for v in fam.values())
    if not tot:
        return None, None, None
    return (sum(v['exec_flops'] for v in fam.values()) / (tot * 1e-3) / MFMA_F32_PEAK, tot / steps, sum(v['exec_flops'] for v in fam.values()) / steps)


def stage_entry(pg, tr, dp, n_gpus, mb, depth, alpha, extra=None, executed=True):
    """images/s, ms per step and D+GP ms of one configured trainer (median of 3 timed windows); the algorithmic work as a RATE
    (TFLOP/s of the reference's 2*MAC count) and the executed-MFMA fraction of the conv launches."""
    ms, k, all_ms = robust_ms(tr, dp)
    dms, _, _ = robust_ms(tr, dp, fn=d_step_fn(tr), prime=3, window_s=0.25)
    w_d, w = step_flops(tr.G, tr.D, depth, alpha)
    e = {'depth': depth, 'res': 4 * 2 ** depth, 'alpha': alpha, 'minibatch': mb, 'images_per_sec': n_gpus * mb / (ms * 1e-3),
         'ms_per_step': ms, 'ms_windows': all_ms, 'steps_per_window': k, 'd_step_gp_ms': dms,
         'algorithmic_gflop_per_image': w / 1e9,
         'algorithmic_tflops_per_gpu': w * (mb / (ms * 1e-3)) / 1e12,
         'd_step_gp_algorithmic_tflops_per_gpu': w_d * (mb / (dms * 1e-3)) / 1e12}
    if executed:               # every rank runs the instrumented steps (collectives stay matched); the line quotes rank 0's
        kf, kms, xfl = executed_fraction(pg, tr)
        # executed_frac: MFMA FLOP the conv launches of a step execute (Winograd credited 16/36) / step time / peak -- a utilisation of
        # the chip over the step, <= 1 by construction and indifferent to how the launches overlap on the streams; the kernel-time
        # form (same FLOP / summed HIP-event time of the launches) next to it counts concurrent launches' time twice
        e['executed_frac'] = xfl / (ms * 1e-3) / MFMA_F32_PEAK if xfl else None
        e['conv_kernel_executed_frac'], e['conv_kernel_ms_per_step'] = kf, kms
    if extra:
        e.update(extra)
    return e


def grow_run(pg, dp, n_gpus, rank):
    """BASELINE.json config 2: the res-32 network grown depth 0 -> 3 with alpha fade-ins, minibatch 64, through the
    product's Trainer + DepthManager + LRScheduler (shortened lod_*_nimg so every stage and every fade occurs)."""
    torch.manual_seed(1337)
    shape = (1, 3, 32, 32)
    G, D = pg.Generator(shape).cuda(), pg.Discriminator(shape).cuda()
    opt_g = pg.FusedAdam(G.parameters(), 0.001, betas=(0.0, 0.99))
    opt_d = pg.FusedAdam(D.parameters(), 0.001, betas=(0.0, 0.99))
    seed = pg.parallel.shard_seed(1337, rank)
    ds = pg.utils.SyntheticDataset(32, 3, seed=seed, ring=8)
    pg.wgan_gp_loss.manual_seed(seed)
    world = 1 if dp is None else dp.world_size
    tr = pg.Trainer(D, G, pg.wgan_gp_D_loss, pg.wgan_gp_G_loss, opt_d, opt_g, ds, None, None, parallel=dp)
    if dp is not None:
        dp.broadcast_params(G, D)
    lod = 64 * 40 * world                                   # 40 iterations per stabilisation / fade span
    dm = pg.DepthManager(ds.loader, lambda n: pg.utils.device_latents(n, 512, seed=seed + 7, ring=16), 3,
                         minibatch_default=64, lod_training_nimg=lod, lod_transition_nimg=lod)
    tr.register_plugin(dm)
    tr.register_plugin(pg.LRScheduler(pg.RampupLR(opt_d, pg.utils.rampup), pg.RampupLR(opt_g, pg.utils.rampup)))
    total = 7 * lod                                         # stages 0,1,2,3 + three fades
    stages = []                                             # [depth, fading, iterations, t_start, t_end], closed with a device sync

    def key_now():
        return (int(tr.G.depth), float(tr.G.alpha) < 1.0)

    class Clock(pg.Plugin):
        """Registered last: runs after DepthManager has prepared the NEXT iteration; a stage closes (device synchronised, so the
        host clock is the GPU's) when the (depth, fading) pair of the next iteration differs from the current stage's."""

        def __init__(self):
            super(Clock, self).__init__([(1, 'iteration')])

        def register(self, trainer):
            pass

        def iteration(self, *a):
            stages[-1][2] += 1
            if key_now() != (stages[-1][0], stages[-1][1]):
                torch.cuda.synchronize()
                now = time.perf_counter()
                stages[-1][4] = now
                stages.append([key_now()[0], key_now()[1], 0, now, None])
    tr.register_plugin(Clock())
    stages.append([0, False, 0, None, None])
    for _ in range(10):                                     # untimed: code objects of the 4x4 stage
        tr.train()
    tr.cur_nimg = 0
    dm.depth = dm.alpha = -1
    dm.iteration()
    if dp is not None:
        dp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    del stages[:]
    stages.append([key_now()[0], key_now()[1], 0, t0, None])
    tr.run(total / 1000.0)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    stages[-1][4] = t1
    dt = _max_over_ranks(t1 - t0, dp)
    stages = [{'depth': d, 'fade_in': f, 'iterations': n, 'ms_per_step': 1e3 * (e - b) / n, 'images_per_sec': n_gpus * 64 * n / (e - b)}
              for d, f, n, b, e in stages if n > 0]
    return {'workload': 'config 2: 32x32 network, grow depth 0->3 with alpha fade-ins, minibatch 64 per GPU, DepthManager + '
                        'LRScheduler, %d iterations' % tr.iterations,
            'images_per_sec': total / dt, 'seconds': dt, 'iterations': tr.iterations, 'stages': stages}


LINE_LIMIT = 6000          # bytes: the driver keeps ~8 KB of stdout; a 20.8 KB line (round 4) could not be parsed


def _r(v, sig=5):
    """floats to ``sig`` significant digits (the line is a record, not a checkpoint)."""
    if isinstance(v, float):
        return float('%.*g' % (sig, v))
    return v


def _pick(d, keys):
    return {k: _r(d[k]) for k in keys if d is not None and k in d and d[k] is not None}


def write_detail(out):
    """Everything measured goes to ``bench_detail.json`` (next to bench.py, and under gpurun_out/ when that exists), a digest to stderr;
    stdout carries only the compact line."""
    text = json.dumps(out, indent=1, sort_keys=True)
    first = None
    for d in (ROOT, os.path.join(ROOT, 'gpurun_out')):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, 'bench_detail.json'), 'w') as f:
                    f.write(text)
                first = first or os.path.relpath(os.path.join(d, 'bench_detail.json'), ROOT)
            except OSError:
                pass
    # stderr: a short human-readable digest (never a JSON object: nothing but the stdout line may look like the record)
    w = sys.stderr.write
    w('[bench] detail file: %s (%d bytes)\n' % (first, len(text)))
    for e in out.get('per_depth') or []:
        w('[bench] depth %d (%4dx%-4d mb %2d): %9.1f img/s  %8.3f ms/step  D+GP %8.3f ms  executed MFMA frac %s\n' % (
            e['depth'], e['res'], e['res'], e['minibatch'], e['images_per_sec'], e['ms_per_step'], e['d_step_gp_ms'],
            '%.3f' % e['executed_frac'] if e.get('executed_frac') is not None else 'n/a'))
    for k, v in sorted((out.get('kernels') or {}).items(), key=lambda kv: -kv[1]['ms_per_step'])[:12]:
        w('[bench] %-58s %6.3f ms/step %5.1f launches %7.1f us  %6.1f TF executed\n' % (k, v['ms_per_step'], v['launches_per_step'], v['avg_launch_us'], v['executed_tflops']))
    sys.stderr.flush()
    return first


def compact_line(out, detail_path=None):
    """The ONE stdout line of the contract, <= LINE_LIMIT bytes: contract keys, ``roofline`` of the dominant kernel, ``cpu_baseline``,
    the D+GP window and a [depth, img/s, ms, D+GP ms] row per growth stage.  Tables (per-kernel, secondary workloads, CPU per depth,
    timing windows) live in the detail file."""
    line = {k: _r(out[k]) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'priming_steps', 'ms_per_step',
                                    'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data') if k in out}
    line['config'] = {k: _r(v) for k, v in out.get('config', {}).items()}
    roof = out.get('roofline')
    if roof:
        line['roofline'] = _pick(roof, ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'algorithmic_frac'))
        line['roofline']['traffic'] = _r(roof.get('traffic'))                        # (null = no PMC pass to quote)
        line['roofline'].update(_pick(roof, ('traffic_source', 'mfma_busy_pct', 'valu_per_mfma', 'avg_launch_us', 'event_pair_us_subtracted', 'avg_launch_us_rocprof',
                                             'frac_rocprof', 'launches_per_step', 'ms_per_step_in_kernel')))
    cpu = out.get('cpu_baseline')
    line['cpu_baseline'] = _pick(cpu, ('value', 'unit', 'cores', 'kind', 'sample', 'measured_at_n_gpus')) if cpu else None
    dwin = out.get('d_step_gp_counters') or {}
    d = {'ms': _r(out.get('d_step_gp_ms')), 'executed_frac': _r(out.get('d_step_gp_executed_mfma_frac')),
         'mfma_busy_pct_conv': _r(dwin.get('mfma_busy_pct_conv_kernels', out.get('d_step_gp_mfma_busy_pct'))),
         'mfma_busy_pct_all': _r(dwin.get('mfma_busy_pct_all_kernels')), 'mfma_busy_wall_pct_hybrid': _r(dwin.get('mfma_busy_wall_pct')),
         'counter_source': dwin.get('source_short'), 'boxes': {k: _r(v) for k, v in (dwin.get('boxes') or {}).items()} or None}
    line['d_step_gp'] = {k: v for k, v in d.items() if v is not None}
    line.update(_pick(out, ('executed_mfma_frac', 'mfma_busy_pct', 'step_issue', 'rccl_ranks', 'allreduce_ms', 'allreduce_bytes_per_step',
                            'exposed_exchange_ms', 'ms_per_step_without_exchange')))
    if out.get('per_depth'):
        line['per_depth_cols'] = ['depth', 'images_per_sec', 'ms_per_step', 'd_step_gp_ms', 'executed_frac']
        line['per_depth'] = [[e['depth'], _r(e['images_per_sec'], 4), _r(e['ms_per_step'], 4), _r(e['d_step_gp_ms'], 4),
                              _r(e.get('executed_frac'), 3)] for e in out['per_depth']]
    if out.get('configs'):
        line['configs'] = {k: [_r(v['images_per_sec'], 4), _r(v.get('ms_per_step', 1e3 * v.get('seconds', 0.0)), 4)] for k, v in out['configs'].items()}
    if detail_path:
        line['detail'] = detail_path
    text = json.dumps(line, separators=(',', ':'))
    for drop in ('configs', 'per_depth', 'per_depth_cols'):      # never exceed the limit: shed the optional tables first
        if len(text) <= LINE_LIMIT:
            break
        line.pop(drop, None)
        text = json.dumps(line, separators=(',', ':'))
    if len(text) > LINE_LIMIT:
        raise RuntimeError('bench.py: compact JSON line is %d bytes (limit %d)' % (len(text), LINE_LIMIT))
    return text


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--prime', type=int, default=PRIME_STEPS, help='untimed setup steps before the warmup (see PRIME_STEPS)')
    ap.add_argument('--depth', type=int, default=8)
    ap.add_argument('--alpha', type=float, default=1.0)
    ap.add_argument('--fmap-base', type=int, default=4096, help='4096: reference default; 8192: the paper\'s widths')
    ap.add_argument('--config', type=int, default=5, choices=[2, 3, 4, 5],
                    help='BASELINE.json config timed as the headline (default 5 = the 1024x1024 stage the metric is quoted on)')
    ap.add_argument('--minibatch', type=int, default=0, help='per-GPU minibatch (default: reference schedule)')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-per-depth', action='store_true')
    ap.add_argument('--no-configs', action='store_true', help='skip the secondary workloads (BASELINE configs 2-4, fmap_base 8192, alpha 0.5)')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-d-step', action='store_true', help='skip the separate D step + gradient penalty timing of the headline stage (profiling runs: the trace then holds the timed steps only)')
    ap.add_argument('--graphs', action='store_true', help='replay every stage from captured hipGraphs (graphs.py); default: only the launch-bound 4x4 stage, eager two-stream launching elsewhere (measured faster)')
    ap.add_argument('--serial-kernel-timing', action='store_true', help='instrumented passes with the weight-gradient stream off '
                    '(isolated per-kernel durations instead of the durations inside the two-stream step)')
    ap.add_argument('--kernel-table', action='store_true', help='per-layer conv timing table on stderr')
    ap.add_argument('--host-data', action='store_true', help='the headline loop is fed from PINNED HOST batches (the input step of reference '
                    'trainer.py:92: the Trainer uploads batch k + 1 on a copy stream under iteration k); the device-resident number stays "value"')
    ap.add_argument('--d-step-only', action='store_true', help='the timed loop runs the D step + gradient penalty + Adam(D) only (profiling aid: '
                    'tools/profile_round.sh collects the MFMA-busy counters of that window; the JSON line says so in "metric")')
    args = ap.parse_args()

    env_world = os.environ.get('WORLD_SIZE')
    if args.gpus > 1 and env_world is None:
        relaunch(args.gpus)
    world = int(env_world or '1')
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        sys.stderr.write('bench.py: --gpus %d but WORLD_SIZE=%d\n' % (args.gpus, world))
        sys.exit(2)

    import pggan_amd as pg
    if os.environ.get('PGGAN_TUNE'):                       # kernel A/B aid: "key=value,..." -> pg_debug_set_tuning
        for kv in os.environ['PGGAN_TUNE'].split(','):
            pg._lib.load().pg_debug_set_tuning(*[int(v) for v in kv.split('=')])
    if os.environ.get('PGGAN_WINO'):                       # kernel A/B aid: Winograd conv generation (pg_debug_set_wino)
        pg._lib.load().pg_debug_set_wino(int(os.environ['PGGAN_WINO']))
    pg.wgan_gp_loss.enable_graphs(True if args.graphs else 'auto')   # 'auto': hipGraph replay only at the launch-bound 4x4 stage
    force_dp = os.environ.get('PGGAN_FORCE_DP', '') == '1'      # one-rank RCCL group: smoke test of the DP code path
    dp = pg.parallel.DataParallel.from_env(force=force_dp) if (world > 1 or force_dp) else None
    rank = 0 if dp is None else dp.rank
    if dp is None:
        torch.cuda.set_device(0)
    n_gpus = world
    rccl = None
    if dp is not None:                                     # prove the library's RCCL communicator spans every rank
        probe = torch.ones(1 << 16, device='cuda', dtype=torch.float32)
        dp.all_reduce_flat(probe)
        torch.cuda.synchronize()
        got = float(probe[0])
        if got != float(world) or (dp.comm is not None and dp.comm_ranks != world):
            raise RuntimeError('RCCL all-reduce over %d ranks returned %r (communicator size %d)' % (world, got, dp.comm_ranks))
        rccl = {'rccl_ranks': dp.comm_ranks, 'allreduce_probe_sum': got}
        dp.stats.update(collectives=0, bytes=0)

    # ---- headline workload
    cfgs = {5: dict(net=1024, depth=8, mb=3, ch=3), 3: dict(net=128, depth=5, mb=16, ch=3), 4: dict(net=256, depth=6, mb=8, ch=1),
            2: dict(net=32, depth=3, mb=64, ch=3)}
    c = cfgs[args.config]
    depth = args.depth if args.config == 5 else c['depth']
    net_res = c['net']
    res = 4 * 2 ** depth
    mb = args.minibatch or (REF_MINIBATCH.get(depth, 16) if args.config == 5 else c['mb'])
    seed = pg.parallel.shard_seed(1337, rank)

    tr = make_trainer(pg, net_res, depth, args.alpha, mb, seed, dp, fmap_base=args.fmap_base, channels=c['ch'])
    for _ in range(args.prime):           # setup, untimed and reported: code-object loading, allocator growth, clock ramp
        tr.train()
    if dp is not None:
        dp.stats.update(collectives=0, bytes=0)
    dt = timed_steps(tr, args.steps, args.warmup, dp, fn=d_step_fn(tr) if args.d_step_only else None, markers=args.d_step_only)
    ms_per_step = 1e3 * dt / args.steps
    host_ms, host_free_ms = HOST_ENQUEUE['ms'], HOST_ENQUEUE['free_ms']
    # D step + gradient penalty + Adam(D) of the headline stage (its own timed loop, after the contract's)
    d_gp_ms = None if (args.no_d_step or args.d_step_only) else robust_ms(tr, dp, fn=d_step_fn(tr), prime=3, window_s=0.25)[0]
    if args.d_step_only:
        d_gp_ms = ms_per_step
    value = n_gpus * mb * args.steps / dt
    w_d, w = step_flops(tr.G, tr.D, depth, args.alpha)

    out = {
        'metric': ('images/sec, PGGAN D step + gradient penalty + Adam(D) ONLY (--d-step-only, profiling aid) at %dx%d' if args.d_step_only else
                   'images/sec, PGGAN full train step (D+GP step + G step + Adam) at %dx%d') % (res, res),
        'value': value, 'unit': 'images/sec', 'n_gpus': n_gpus, 'steps': args.steps, 'warmup': args.warmup, 'priming_steps': args.prime,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic (seeded uniform images / normal latents, ring of 8 device-resident batches)' + (
            ' -- PGGAN_DP_SHARE_GPU=1: ALL RANKS ON ONE DEVICE, a collective-matching smoke test, NOT a measurement' if os.environ.get('PGGAN_DP_SHARE_GPU', '') == '1' else ''),
        'config': {'workload': 'BASELINE config %d: PGGAN %dx%d net (fmap_base %d, C=%d, latent %d), stage depth %d = %dx%d, '
                               'alpha %.2f, minibatch %d/GPU%s, Trainer.train(): WGAN-GP + Adam(0,0.99), fp32'
                               % (args.config, net_res, net_res, args.fmap_base, c['ch'], tr.G.latent_size, depth, res, res, args.alpha, mb,
                                  ' (reference schedule)' if args.config == 5 and not args.minibatch else ''),
                   'resolution': res, 'depth': depth, 'minibatch_per_gpu': mb, 'global_batch': mb * n_gpus,
                   'parallelism': 'dp%d' % n_gpus, 'fmap_base': args.fmap_base},
        'step_algorithmic_gflop_per_image': w / 1e9,
        'step_algorithmic_tflops_per_gpu': w * (value / n_gpus) / 1e12,     # a rate: Winograd stages can exceed the executed peak
        'host_enqueue_ms_per_step': host_free_ms,  # Python + launch time of one step on the host running free (median of the first ten timed steps)
        'host_enqueue_ms_per_step_whole_loop': host_ms,   # ... averaged over all timed steps (includes waiting for queue space behind the device)
        'step_issue': pg.wgan_gp_loss._replay_mode(tr.G) or 'eager',
        'd_step_gp_ms': d_gp_ms,
    }
    out['config']['hip_graphs'] = bool(pg.wgan_gp_loss._replay_mode(tr.G) == 'graph' and args.alpha >= 1.0)
    if args.host_data or (args.config == 5 and not args.no_configs):
        # the same step fed from pinned host memory (37.7 MB per step at 1024x1024): never "value" (inputs resident in HBM is the
        # contract), reported next to it
        th = make_trainer(pg, net_res, depth, args.alpha, mb, seed, dp, fmap_base=args.fmap_base, channels=c['ch'], host_data=True)
        for _ in range(max(10, args.prime // 2)):
            th.train()
        dth = timed_steps(th, args.steps, args.warmup, dp)
        out['host_resident_input'] = {'ms_per_step': 1e3 * dth / args.steps, 'images_per_sec': n_gpus * mb * args.steps / dth,
                                      'vs_device_resident': (dth / args.steps) / (dt / args.steps),
                                      'h2d_bytes_per_step': int(mb * c['ch'] * res * res * 4),
                                      'prefetch_hits': th._inputs.hits, 'prefetch_misses': th._inputs.misses,
                                      'is': 'Trainer.train() fed from a ring of pinned host batches, uploaded on a copy stream where the reference calls .cuda() (the host runs ahead of the device, so the upload overlaps the previous step)'}
        del th
        torch.cuda.empty_cache()
    if rccl is not None:
        out.update(rccl)
        steps_counted = args.steps + args.warmup
        out['allreduce_bytes_per_step'] = dp.stats['bytes'] / steps_counted
        out['allreduce_collectives_per_step'] = dp.stats['collectives'] / steps_counted
        # busy time of the exchange per step (HIP events around every collective on the stream it runs on) and the step
        # time with the collectives left out: their difference to ms_per_step is the EXPOSED exchange
        dp.record_events, dp.events = True, []
        esteps = 5
        for _ in range(esteps):
            tr.train()
        torch.cuda.synchronize()
        dp.record_events = False
        out['allreduce_ms'] = sum(a.elapsed_time(b) for a, b in dp.events) / esteps
        dp.skip_exchange = True
        dt0 = timed_steps(tr, args.steps, 4, dp)          # (a plan of its own: two eager warm-ups + the recording step stay outside the timed steps)
        dp.skip_exchange = False
        dp.broadcast_params(tr.G, tr.D)                        # ranks diverged while nothing was exchanged
        out['ms_per_step_without_exchange'] = 1e3 * dt0 / args.steps
        out['exposed_exchange_ms'] = ms_per_step - out['ms_per_step_without_exchange']

    if rank == 0 and not args.no_kernel_timing:
        psteps = 3
        graph_mode = pg.wgan_gp_loss._replay_mode(tr.G) == 'graph'
        if graph_mode or args.serial_kernel_timing:
            pg.wgan_gp_loss.enable_graphs(False)           # a hipGraph replay has no per-launch hook (launch plans do: _lib.CALL_HOOK)
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
        prof, traffic, busy = None, None, None
        src = os.path.join('profiles', PROFILE_TAG + '_roofline.json')
        try:                                   # HBM bytes per launch / MFMA-busy of that symbol from the committed PMC pass
            with open(os.path.join(ROOT, src)) as f:
                prof = json.load(f)['per_kernel']
        except Exception:
            prof, src = None, None
        if prof is not None and dom in prof:
            traffic = prof[dom]['hbm_bytes_per_launch']
        # rocprofv3's in-step average duration of the same symbol from the committed kernel trace of this round (the live
        # HIP-event bracket also holds the dispatch latency behind a busy second queue and under-reads ``frac`` by 4-8 %)
        rocprof_us = _rocprof_avg_us(dom)
        out['roofline'] = {'bound': 'mfma', 'kernel': dom, 'achieved': d['exec_tflops'], 'peak': MFMA_F32_PEAK / 1e12,
                           'unit': 'TFLOP/s', 'frac': d['exec_tflops'] * 1e12 / MFMA_F32_PEAK,
                           'frac_is': 'MFMA FLOP the launches EXECUTE (Winograd F(2x2,3x3): 16/36 of the algorithmic 2*MAC count) / HIP-event time / nominal peak',
                           'algorithmic_tflops': d['tflops'], 'algorithmic_frac': d['tflops'] * 1e12 / MFMA_F32_PEAK,
                           'algorithmic_frac_is': '2*MAC FLOP of the reference convolution / HIP-event time / nominal peak (what SURVEY.md 8d calls achieved; exceeds the executed fraction by 36/16 on Winograd launches)',
                           'traffic': traffic, 'traffic_source': src if traffic is not None else None,
                           'mfma_busy_pct': prof[dom]['mfma_busy_pct'] if prof and dom in prof else None,
                           'valu_per_mfma': prof[dom].get('valu_per_mfma') if prof and dom in prof else None,
                           'avg_launch_us': d['avg_launch_us'], 'launches_per_step': d['launches_per_step'], 'event_pair_us_subtracted': 1e3 * kt.pair_ms,
                           'avg_launch_us_raw': d['avg_launch_us'] + 1e3 * kt.pair_ms,      # before the calibrated event-pair overhead is taken off
                           'frac_raw_events': d['exec_tflops'] * 1e12 / MFMA_F32_PEAK * d['avg_launch_us'] / (d['avg_launch_us'] + 1e3 * kt.pair_ms),
                           'avg_launch_us_rocprof': rocprof_us,
                           'frac_rocprof': (d['exec_tflops'] * 1e12 / MFMA_F32_PEAK * d['avg_launch_us'] / rocprof_us) if rocprof_us else None,
                           'frac_rocprof_is': 'this run\'s executed FLOP per launch / the in-step average duration of the symbol in profiles/%s_kernel_stats.csv (rocprofv3 --kernel-trace of the same command, builder-side box)' % PROFILE_TAG,
                           'ms_per_step_in_kernel': d['ms_per_step'],
                           'algorithmic_gflop_per_step': d['flops_per_step'] / 1e9}
        tot_ms = sum(v['ms'] for v in fam.values())
        xfl = sum(v['exec_flops'] for v in fam.values()) / psteps
        out['executed_mfma_frac'] = xfl / (ms_per_step * 1e-3) / MFMA_F32_PEAK
        out['executed_mfma_frac_is'] = 'MFMA FLOP the conv launches of a step execute (Winograd credited 16/36) / step time / peak'
        out['conv_kernel_executed_mfma_frac'] = xfl * psteps / (tot_ms * 1e-3) / MFMA_F32_PEAK      # / summed HIP-event time of the launches (concurrent launches count twice)
        if prof:
            num = sum(v['ms'] * prof[k]['mfma_busy_pct'] for k, v in fam.items() if k in prof)
            den = sum(v['ms'] for k, v in fam.items() if k in prof)
            out['mfma_busy_pct'] = num / den if den else None
            out['mfma_busy_source'] = src + ' (SQ_VALU_MFMA_BUSY_CYCLES per kernel symbol, weighted by this run\'s time per symbol)'
        # the north-star window: D step + gradient penalty + Adam(D).  (a) this run's conv launches of that window (HIP events),
        # weighted with the per-symbol SQ_VALU_MFMA_BUSY_CYCLES of the committed PMC pass; (b) the counter figure of a
        # --d-step-only PMC pass over ALL kernels of the window (tools/profile_round.sh, profiles/<tag>_roofline.json)
        if graph_mode or args.serial_kernel_timing:
            pg.wgan_gp_loss.enable_graphs(False)
        try:
            with KernelTimer(pg) as ktd:
                one = d_step_fn(tr)
                for _ in range(psteps):
                    one()
                famd = ktd.summary(psteps)
        finally:
            pg.engine.ASYNC_WGRAD = async_wgrad
        totd = sum(v['ms'] for v in famd.values())
        out['d_step_gp_executed_mfma_frac'] = (sum(v['exec_flops'] for v in famd.values()) / psteps / (d_gp_ms * 1e-3) / MFMA_F32_PEAK
                                               if (totd and d_gp_ms) else None)                 # / the window's wall time
        if prof:
            num = sum(v['ms'] * prof[k]['mfma_busy_pct'] for k, v in famd.items() if k in prof)
            den = sum(v['ms'] for k, v in famd.items() if k in prof)
            out['d_step_gp_mfma_busy_pct'] = num / den if den else None
            out['d_step_gp_mfma_busy_is'] = 'conv / weight-gradient launches of the D step + gradient penalty window, time-weighted ' + out['mfma_busy_source']
        try:
            with open(os.path.join(ROOT, src)) as f:
                dwin = json.load(f).get('d_step_gp_window')
        except Exception:
            dwin = None
        if dwin:
            out['d_step_gp_counters'] = dict(dwin, source=src + ' (rocprofv3 --pmc pass of bench.py --d-step-only: every kernel of the window)', source_short=src,
                                             boxes={t: v for t, v in PROFILE_BOXES.items() if v is not None})
            if d_gp_ms and dwin.get('kernel_time_ms_per_pass') and dwin.get('mfma_busy_pct_all_kernels') is not None:
                # HYBRID, labelled as such: MFMA-busy cycles of the window summed in the SERIALISED counter pass / (THIS run's
                # un-serialised wall time of the window x nominal clock x SIMDs) -- the share of the window's wall time the matrix pipes are
                # busy when the two streams overlap.  The strict figure is mfma_busy_pct_all (busy cycles / serialised kernel time).
                out['d_step_gp_counters']['mfma_busy_wall_pct'] = dwin['kernel_time_ms_per_pass'] * dwin['mfma_busy_pct_all_kernels'] / d_gp_ms
                out['d_step_gp_counters']['mfma_busy_wall_is'] = ('hybrid: busy cycles of the serialised --pmc pass (' + src + ') / this run\'s '
                                                                  'un-serialised D+GP wall time; not a counter figure of one run')
        out['kernels'] = {k: {'tflops': v['tflops'], 'executed_tflops': v['exec_tflops'], 'ms_per_step': v['ms_per_step'],
                              'launches_per_step': v['launches_per_step'], 'avg_launch_us': v['avg_launch_us']}
                          for k, v in fam.items()}
    elif dp is not None and not args.no_kernel_timing:
        pg.wgan_gp_loss.enable_graphs(False)
        for _ in range(3):                                  # keep collectives matched with rank 0: its instrumented train steps ...
            tr.train()
        one = d_step_fn(tr)
        for _ in range(3):                                  # ... and its instrumented D steps (each exchanges D's gradients)
            one()
    pg.wgan_gp_loss.enable_graphs(True if args.graphs else 'auto')
    del tr
    pg.plans.clear()
    torch.cuda.empty_cache()

    if not args.no_per_depth:
        per = []
        for d in range(0, 9):
            m = REF_MINIBATCH.get(d, 16)
            t = make_trainer(pg, 1024, d, 1.0, m, seed, dp, fmap_base=args.fmap_base)
            per.append(stage_entry(pg, t, dp, n_gpus, m, d, 1.0))          # minibatch m PER RANK (weak scaling), max over ranks
            del t
            pg.plans.clear()                                               # (a plan pins its stage's activations)
            pg.graphs.clear()
            torch.cuda.empty_cache()
        out['per_depth'] = per

    if not args.no_configs:
        sec = {}
        t = make_trainer(pg, 1024, 8, 0.5, 3, seed, dp)
        sec['depth8_alpha0.5'] = stage_entry(pg, t, dp, n_gpus, 3, 8, 0.5, {'workload': 'config 5 network, 1024x1024 stage in the middle of its fade-in (alpha 0.5)'})
        del t
        pg.plans.clear()
        torch.cuda.empty_cache()
        t = make_trainer(pg, 1024, 8, 1.0, 3, seed, dp, fmap_base=8192)
        sec['depth8_fmap8192'] = stage_entry(pg, t, dp, n_gpus, 3, 8, 1.0, {'workload': 'paper widths (fmap_base 8192), 1024x1024 stage, minibatch 3 per GPU'})
        del t
        pg.plans.clear()
        torch.cuda.empty_cache()
        t = make_trainer(pg, 128, 5, 1.0, 16, seed, dp)
        sec['config3'] = stage_entry(pg, t, dp, n_gpus, 16, 5, 1.0, {'workload': 'config 3: 128x128 network at depth 5, minibatch 16 per GPU (32 global on 2 GPUs)'})
        del t
        pg.plans.clear()
        torch.cuda.empty_cache()
        t = make_trainer(pg, 256, 6, 1.0, 8, seed, dp, channels=1)
        sec['config4'] = stage_entry(pg, t, dp, n_gpus, 8, 6, 1.0, {'workload': 'config 4: 256x256 C=1 (abslog-spectrogram shape) network at depth 6, minibatch 8 per GPU (32 global on 4 GPUs)'})
        del t
        pg.plans.clear()
        torch.cuda.empty_cache()
        sec['config2'] = grow_run(pg, dp, n_gpus, rank)
        torch.cuda.empty_cache()
        out['configs'] = sec

    if rank == 0:
        # The CPU baseline is a property of the host, measured once at N = 1 (rank 0's cores all to itself).  At N > 1 the
        # line repeats that measurement when an N = 1 run of this box left it behind (the driver runs N = 1, 2, 4, 8 back to
        # back); otherwise rank 0 measures it now, after the timed region, while the other ranks wait at the closing barrier.
        import hashlib
        import socket
        h = hashlib.sha256()
        for fn in (os.path.abspath(__file__), os.path.join(ROOT, 'oracle', 'pggan_cpu.py')):
            with open(fn, 'rb') as f:
                h.update(f.read())
        # (keyed by host and by the code that produced it: a stale file of another build or box is never reported as this box's)
        cache = os.path.join(tempfile.gettempdir(), 'pggan_cpu_baseline_%s_%s_d%d_mb%d.json' % (socket.gethostname(), h.hexdigest()[:12], depth, mb))
        if args.no_cpu or args.config != 5:
            out['cpu_baseline'] = None
        elif n_gpus == 1:
            out['cpu_baseline'] = cpu_baseline(depth, mb)
            try:
                with open(cache, 'w') as f:
                    json.dump(out['cpu_baseline'], f)
            except OSError:
                pass
        else:
            try:
                with open(cache) as f:
                    out['cpu_baseline'] = dict(json.load(f), measured_at_n_gpus=1, source='the N=1 run of this box (%s)' % cache)
            except (OSError, ValueError):
                out['cpu_baseline'] = dict(cpu_baseline(depth, mb), measured_at_n_gpus=n_gpus,
                                           source='rank 0 after the timed region, the other ranks idle at the barrier')
    if dp is not None:
        dp.barrier()
        dp.close()
        torch.distributed.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: RCCL prints its banner through C stdio, which is still buffered here
        import ctypes
        detail_path = write_detail(out)
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(compact_line(out, detail_path), flush=True)


if __name__ == '__main__':
    main()
