"""bench.py's stdout contract: ONE JSON line the driver can parse (round 4's 20.8 KB line could not be: BENCH_r04.parsed = null)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _canned():
    kernels = {'conv_wino2_kernel<1, 1, %d, false, %d>' % (i, j): {'tflops': 100.123456789, 'executed_tflops': 44.4999, 'ms_per_step': 1.23456789,
                                                                  'launches_per_step': 32.0, 'avg_launch_us': 70.9123} for i in range(6) for j in range(4)}
    per_depth = [{'depth': d, 'res': 4 * 2 ** d, 'alpha': 1.0, 'minibatch': 16, 'images_per_sec': 1532.9453960974013, 'ms_per_step': 10.437283,
                  'ms_windows': [10.1, 10.2, 10.3], 'steps_per_window': 33, 'd_step_gp_ms': 7.3512345, 'algorithmic_gflop_per_image': 55.884363776,
                  'algorithmic_tflops_per_gpu': 165.84, 'd_step_gp_algorithmic_tflops_per_gpu': 172.5, 'executed_frac': 0.312345678,
                  'conv_kernel_ms_per_step': 8.1} for d in range(9)]
    cpu = {'value': 0.31349930930477765, 'unit': 'images/sec', 'cores': 32, 'kind': 'port', 'sample': 'x' * 300,
           'per_depth': [{'depth': d, 'res': 4 * 2 ** d, 'minibatch': 4, 'images_per_sec': 0.76, 'ms_per_step': 5260.8, 'warmup_iteration_s': 5.3,
                          'timed_iterations_s': [5.30, 5.21]} for d in range(9)]}
    return {
        'metric': 'images/sec, PGGAN full train step (D+GP step + G step + Adam) at 1024x1024', 'value': 278.1234567, 'unit': 'images/sec',
        'n_gpus': 1, 'steps': 20, 'warmup': 5, 'priming_steps': 50, 'ms_per_step': 10.7865432, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic (seeded uniform images / normal latents, ring of 8 device-resident batches)',
        'config': {'workload': 'w' * 220, 'resolution': 1024, 'depth': 8, 'minibatch_per_gpu': 3, 'global_batch': 3, 'parallelism': 'dp1',
                   'fmap_base': 4096, 'hip_graphs': False},
        'roofline': {'bound': 'mfma', 'kernel': 'conv_wino2_kernel<1, 1, 3, false, 0>', 'achieved': 55.2, 'peak': 157.3, 'unit': 'TFLOP/s',
                     'frac': 0.351, 'frac_is': 'y' * 200, 'algorithmic_tflops': 124.2, 'algorithmic_frac': 0.79, 'algorithmic_frac_is': 'z' * 200,
                     'traffic': 74764221.5, 'traffic_source': 'profiles/r04_roofline.json', 'mfma_busy_pct': 42.7, 'valu_per_mfma': 3.7,
                     'avg_launch_us': 73.8, 'launches_per_step': 32.0, 'ms_per_step_in_kernel': 2.36, 'algorithmic_gflop_per_step': 293.4},
        'cpu_baseline': cpu, 'kernels': kernels, 'per_depth': per_depth, 'd_step_gp_ms': 7.35, 'd_step_gp_executed_mfma_frac': 0.32,
        'd_step_gp_counters': {'mfma_busy_pct_all_kernels': 35.4, 'mfma_busy_pct_conv_kernels': 41.4, 'source': 's' * 150,
                               'source_short': 'profiles/r04_roofline.json'},
        'configs': {k: dict(per_depth[0], workload='c' * 120) for k in ('depth8_alpha0.5', 'depth8_fmap8192', 'config3', 'config4')},
        'executed_mfma_frac': 0.33, 'mfma_busy_pct': 40.1, 'step_issue': 'plan', 'host_resident_input': {'is': 'h' * 300},
    }


def test_compact_line_fits_and_round_trips():
    out = _canned()
    assert len(json.dumps(out)) > 8192                    # the full record is what broke the driver's parser
    line = bench.compact_line(out, 'bench_detail.json')
    assert '\n' not in line and len(line) < 6000, len(line)
    back = json.loads(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in back, k
    assert set(('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')) <= set(back['roofline'])
    assert set(('value', 'unit', 'cores', 'kind', 'sample')) <= set(back['cpu_baseline'])
    assert 'per_depth' not in back['cpu_baseline'] and 'kernels' not in back
    assert back['d_step_gp']['mfma_busy_pct_all'] == 35.4 and back['d_step_gp']['ms'] == 7.35
    assert len(back['per_depth']) == 9 and back['per_depth'][3][0] == 3
    assert all(row[4] is None or row[4] <= 1.0 for row in back['per_depth'])           # nothing above 1 is called a fraction
    assert back['detail'] == 'bench_detail.json'


def test_compact_line_sheds_tables_before_failing():
    out = _canned()
    out['config']['workload'] = 'w' * 3000
    out['cpu_baseline']['sample'] = 'x' * 1500
    back = json.loads(bench.compact_line(out))
    assert 'roofline' in back and 'cpu_baseline' in back and 'configs' not in back
    assert len(json.dumps(back)) < 6100


def test_null_traffic_is_kept():
    out = _canned()
    out['roofline']['traffic'] = None
    assert json.loads(bench.compact_line(out))['roofline']['traffic'] is None
