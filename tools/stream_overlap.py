"""Occupancy of the two HIP streams from a rocprofv3 kernel trace of bench.py (run on the GPU box):

    cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace -d $OUT -o kt --output-format csv -- python bench.py --no-cpu --no-per-depth --no-kernel-timing --prime 10 --steps 10 --warmup 3
    python tools/stream_overlap.py $OUT/**/kt_kernel_trace.csv

Takes the last 10 train steps (delimited by the Adam launches of the generator), and reports per queue: busy time,
share of the wall time; the union of both queues (GPU never idle?) and the sum of kernel durations per step."""
import csv
import glob
import sys
from collections import defaultdict

path = glob.glob(sys.argv[1], recursive=True)[0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', r.get('Stream_Id', '0')), r['Kernel_Name']))
rows.sort()
# the last 10 steps of the timed loop, delimited by the mixing-factor draw that opens every D step (one uniform_kernel per train step at every
# growth stage; the number of Adam launches per step differs between stages)
marks = [i for i, r in enumerate(rows) if 'uniform_kernel' in r[3]]
nsteps = min(10, len(marks) - 1)
first, last = marks[-nsteps - 1], marks[-1] - 1
win = rows[first:last + 1]
# rocprofv3 flushes its buffers every few thousand records: the device then idles for milliseconds inside ONE step.  Steps with a
# hole > 1.5 ms anywhere are the tracer's, not the product's: they are reported and left out of the window statistics.
bounds = [marks[-nsteps - 1 + k] for k in range(nsteps)] + [last + 1]
good = []
for k in range(nsteps):
    st = sorted(rows[bounds[k]:bounds[k + 1]])
    hole, end = 0, st[0][1]
    for s_, e_, _, _ in st[1:]:
        hole = max(hole, s_ - end)
        end = max(end, e_)
    span = max(r[1] for r in st) - st[0][0]
    print('step %d: %.3f ms%s' % (k, span / 1e6, '   (largest hole %.2f ms: tracer flush, excluded)' % (hole / 1e6) if hole > 1.5e6 else ''))
    if hole <= 1.5e6:
        good.append(st)
if not good:
    good = [sorted(win)]
win = [r for st in good for r in st]
nsteps = len(good)
wall = sum(max(r[1] for r in st) - st[0][0] for st in good)          # (sum of the kept steps' spans)


def union(intervals):
    tot, cur_s, cur_e = 0, None, None
    for s, e in sorted(intervals):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


byq = defaultdict(list)
for s, e, q, name in win:
    byq[q].append((s, e))
print('window: %d kernels, %.2f ms wall = %.3f ms per step (%d steps)' % (len(win), wall / 1e6, wall / 1e6 / nsteps, nsteps))
for q, iv in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    busy = sum(union([(s, e) for s, e, qq, _ in st if qq == q]) for st in good)
    print('queue %s: %5d kernels, busy %.2f ms (%.1f %% of wall), sum of durations %.2f ms' % (
        q, len(iv), busy / 1e6, 100.0 * busy / wall, sum(e - s for s, e in iv) / 1e6))
allbusy = sum(union([(s, e) for s, e, _, _ in st]) for st in good)
print('any queue busy: %.1f %% of wall; sum of all kernel durations %.3f ms per step (%.2fx the wall time)' % (
    100.0 * allbusy / wall, sum(e - s for s, e, _, _ in win) / 1e6 / nsteps, sum(e - s for s, e, _, _ in win) / wall))

# where nothing runs: the largest gaps of the union, with the kernels on either side, and the gap histogram
gaps = []
for st in good:
    iv = sorted((s, e, name) for s, e, _, name in st)
    cur_e, cur_name = iv[0][1], iv[0][2]
    for s, e, name in iv[1:]:
        if s > cur_e:
            gaps.append((s - cur_e, cur_name, name))
        if e > cur_e:
            cur_e, cur_name = e, name


def short(n):
    import re
    m = re.search(r'([A-Za-z_0-9]+_kernel(<[^>]*>)?|__amd_rocclr_\w+)', n)
    return m.group(1) if m else n[:40]


tot = sum(g[0] for g in gaps)
print('idle (no queue busy): %.3f ms per step in %d gaps per step; gaps > 20 us: %.3f ms per step' % (
    tot / 1e6 / nsteps, len(gaps) // nsteps, sum(g[0] for g in gaps if g[0] > 20000) / 1e6 / nsteps))
agg = defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    k = short(a) + ' -> ' + short(b)
    agg[k][0] += g
    agg[k][1] += 1
for k, (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:12]:
    print('  %7.1f us per step in %4.1f gaps per step: %s' % (g / 1e3 / nsteps, c / nsteps, k))

# where the MAIN queue waits while another queue works (dependencies on the weight-gradient / update stream): gaps of the main queue
# that are not idle gaps of the whole device, aggregated by the main-queue kernels on either side
mainq = max(byq, key=lambda q: len(byq[q]))
waits = defaultdict(lambda: [0, 0])
for st in good:
    mq = sorted((s, e, name) for s, e, q, name in st if q == mainq)
    others = sorted((s, e) for s, e, q, _ in st if q != mainq)
    for (s0, e0, n0), (s1, e1, n1) in zip(mq, mq[1:]):
        if s1 - e0 < 10000:
            continue
        cover = union([(max(s, e0), min(e, s1)) for s, e in others if e > e0 and s < s1])        # part of the gap in which another queue is busy
        if cover > 5000:
            w = waits[(n0.split('(')[0][-60:], n1.split('(')[0][-60:])]
            w[0] += cover
            w[1] += 1
tot_wait = sum(w[0] for w in waits.values())
print('main queue waiting while another queue is busy: %.3f ms per step' % (tot_wait / 1e6 / nsteps))
for (a, b), (ns, cnt) in sorted(waits.items(), key=lambda kv: -kv[1][0])[:10]:
    print('    %6.1f us per step in %4.1f gaps per step: %s -> %s' % (ns / 1e3 / nsteps, cnt / nsteps, a, b))
