"""Markdown tables for DESIGN.md from a bench.py JSON line:  python tools/bench_tables.py gpurun_out/bench.log"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('headline: %.1f img/s, %.3f ms/step, algorithmic_frac %.3f, executed_mfma_frac %s, mfma_busy_pct %s' % (
    d['value'], d['ms_per_step'], d['algorithmic_frac'], d.get('executed_mfma_frac'), d.get('mfma_busy_pct')))
r = d.get('roofline')
if r:
    print('roofline:', json.dumps(r))
print()
print('| growth stage | res | minibatch | images/s | ms / train step (3 windows) | D+GP ms | algorithmic frac (step / D+GP) |')
print('|---|---|---|---|---|---|---|')
for e in d.get('per_depth', []):
    print('| %d | %d | %d | %.0f | %.2f (%s) | %.2f | %.2f / %.2f |' % (e['depth'], e['res'], e['minibatch'], e['images_per_sec'], e['ms_per_step'],
          ' '.join('%.2f' % m for m in e['ms_windows']), e['d_step_gp_ms'], e['algorithmic_frac'], e['d_step_gp_algorithmic_frac']))
print()
print('| workload | images/s | ms / step | D+GP ms | algorithmic frac |')
print('|---|---|---|---|---|')
for k, e in d.get('configs', {}).items():
    if k == 'config2':
        print('| %s | %.0f (whole run, %d iterations, %.1f s) | %s | | |' % (e['workload'], e['images_per_sec'], e['iterations'], e['seconds'],
              '; '.join('d%d%s %.1f' % (s['depth'], ' fade' if s['fade_in'] else '', s['ms_per_step']) for s in e['stages'])))
    else:
        print('| %s | %.0f | %.2f | %.2f | %.2f |' % (e['workload'], e['images_per_sec'], e['ms_per_step'], e['d_step_gp_ms'], e['algorithmic_frac']))
print()
if 'kernels' in d:
    print('| kernel symbol | launches / step | ms / step (in-step HIP events) | algorithmic TF | executed TF |')
    print('|---|---|---|---|---|')
    for k, v in sorted(d['kernels'].items(), key=lambda kv: -kv[1]['ms_per_step']):
        print('| `%s` | %.0f | %.3f | %.1f | %.1f |' % (k, v['launches_per_step'], v['ms_per_step'], v['tflops'], v['executed_tflops']))
print('cpu_baseline:', d.get('cpu_baseline'))
