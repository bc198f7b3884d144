"""Timeline of ONE train step from a rocprofv3 kernel trace of bench.py (run on the GPU box):

    cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace -d $OUT -o kt --output-format csv -- python bench.py --no-cpu --no-per-depth --no-configs --no-kernel-timing --prime 10 --steps 10 --warmup 3
    python tools/step_timeline.py "$OUT/**/kt_kernel_trace.csv" [step index from the end, default 3] > timeline.txt

Every kernel of the step in start order: start (us from the step's first kernel), duration, queue (M = the queue with the most kernels,
S = others), workgroups, name; then the same step condensed into PHASES of the main queue (cut at the loss / optimizer kernels): wall
time of the phase, main-queue busy time inside it, other-queue busy time overlapping it."""
import csv
import glob
import re
import sys
from collections import defaultdict

path = glob.glob(sys.argv[1], recursive=True)[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        gs, ws = int(r.get('Grid_Size', 0) or 0), int(r.get('Workgroup_Size', 1) or 1)
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '0'), r['Kernel_Name'], gs // max(ws, 1)))
rows.sort()


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'([A-Za-z_0-9:]+(<[^(]*>)?)', n)
    return (m.group(1) if m else n)[:64]


g_adam = [i for i, r in enumerate(rows) if 'uniform_kernel' in r[3]]            # the mixing-factor draw opens a step
a, b = g_adam[-back - 1], g_adam[-back]
st = rows[a:b]
t0 = st[0][0]
cnt = defaultdict(int)
for r in st:
    cnt[r[2]] += 1
mainq = max(cnt, key=cnt.get)
print('step of %d kernels, %.3f ms; main queue %s (%d kernels), others %s' % (len(st), (max(r[1] for r in st) - t0) / 1e6, mainq, cnt[mainq],
                                                                              {q: c for q, c in cnt.items() if q != mainq}))
for s, e, q, n, wg in st:
    print('%9.1f %8.1f %s %6d  %s' % ((s - t0) / 1e3, (e - s) / 1e3, 'M' if q == mainq else 'S', wg, short(n)))

# phases of the main queue
CUTS = [('gp_mix', 'D step: G forward'), ('row_sumsq', 'D step: D forward [real|fake|mixed] + GP first backward'), ('d_loss_kernel', 'D step: GP seed + losses'),
        ('g_loss_kernel', 'D step: tangent pass + batched backward sweep; G step: G forward + D forward'), ('adam_kernel', None)]
print()
print('phases (main queue):')
mq = [r for r in st if r[2] == mainq]
oq = [r for r in st if r[2] != mainq]
marks = [0]
names = []
for i, r in enumerate(mq):
    for key, label in CUTS[:4]:
        if key in r[3]:
            marks.append(i + 1)
            names.append(label)
marks.append(len(mq))
names.append('G step: backward through D and G (D-only loop: tangent pass + batched backward sweep)')
for k in range(len(marks) - 1):
    seg = mq[marks[k]:marks[k + 1]]
    if not seg:
        continue
    s0, e0 = seg[0][0], max(r[1] for r in seg)
    busy = sum(r[1] - r[0] for r in seg)
    ov = sum(max(0, min(e, e0) - max(s, s0)) for s, e, _, _, _ in oq)
    print('  %-88s %4d kernels  wall %8.1f us  main busy %8.1f us  other queues busy inside %8.1f us' % (
        names[k] if k < len(names) else '?', len(seg), (e0 - s0) / 1e3, busy / 1e3, ov / 1e3))
