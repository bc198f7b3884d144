"""Two-pass (V form) Winograd conv against the one-pass tile kernel, alone on cold rotating inputs (round 5).
   python tools/bench_wino_v.py            # the K-heavy layers of the 1024^2 step
Prints per layer: one-pass us | transform us + V-form conv us (NCB 1 and 2) | speed-up incl. the transform | max rel. difference."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pggan_amd as pg  # noqa: E402

ops = pg.ops
lib = pg._lib.load()
CASES = [  # (N, H, Cin, Cout, epilogue)
    (9, 32, 256, 256, 'bias'), (9, 32, 256, 256, 'unpoolb'), (9, 16, 512, 512, 'mask32'), (9, 16, 512, 512, 'bias+pool'),
    (9, 32, 512, 256, 'mask32'), (9, 32, 256, 512, 'bias+pool'), (9, 64, 128, 256, 'bias+pool'), (9, 64, 256, 128, 'maskb'),
    (9, 64, 128, 128, 'bias'), (3, 16, 512, 512, 'bias'), (3, 32, 256, 256, 'bias'), (3, 32, 512, 256, 'mask32'), (3, 64, 256, 128, 'maskb'),
    (3, 8, 512, 512, 'bias'), (9, 8, 512, 512, 'bias'), (9, 128, 128, 64, 'maskb'), (9, 128, 64, 128, 'bias+pool'),
]
ROT = 6


def timeit(fn, reps=20):
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def main():
    torch.manual_seed(0)
    for N, H, ci, co, epi in CASES:
        xs = [torch.randn(N, H, H, ci, device='cuda') for _ in range(ROT)]
        w = torch.randn(3, 3, co, ci, device='cuda') * 0.1
        u = ops.wino_transform_weights(w)
        b = torch.randn(co, device='cuda')
        m32 = torch.randn(N, H, H, co, device='cuda')
        mb = (torch.rand(N, H, H, co // 4, device='cuda') * 16).to(torch.uint8)
        ub = (torch.rand(N, 2 * H, 2 * H, co // 4, device='cuda') * 16).to(torch.uint8)
        kw = {'bias': dict(slope=0.2), 'mask32': dict(mask=m32), 'maskb': dict(mask=mb), 'bias+pool': dict(slope=0.2, pool=True),
              'unpoolb': dict(unpool=True, upmask=ub)}[epi]
        bias = b if epi.startswith('bias') else None

        def run(i, v_form, v=None):
            return ops.conv2d_wino(xs[i % ROT], u, bias, N, H, H, 0.3, v_form=v_form, v=v, **kw)
        t1 = timeit(lambda i: run(i, False))
        k1 = lib.pg_debug_last_wino_kernel().decode()
        ref = run(0, False)
        ref = ref[0] if isinstance(ref, tuple) else ref
        tt = timeit(lambda i: ops.wino_transform_input(xs[i % ROT], N, H, H, ci))
        vs = [ops.wino_transform_input(x, N, H, H, ci) for x in xs]
        line = '%-28s one-pass %6.1f us (%s) | transform %5.1f us |' % ('n%d %d->%d @%d %s' % (N, ci, co, H, epi), t1, k1[10:], tt)
        for ncb in (1, 2):
            lib.pg_debug_set_wino_v(ncb)
            tv = timeit(lambda i: run(i, True, vs[i % ROT]))
            kv = lib.pg_debug_last_wino_kernel().decode()
            got = run(0, True, vs[0])
            got = got[0] if isinstance(got, tuple) else got
            err = float((got.float() - ref.float()).abs().max() / ref.float().abs().max())
            line += ' NCB%d %6.1f us (%s) x%.2f err %.1e |' % (ncb, tv, kv[18:], t1 / (tv + tt), err)
        lib.pg_debug_set_wino_v(0)
        print(line, flush=True)


if __name__ == '__main__':
    main()
