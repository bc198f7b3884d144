set -x
mkdir -p gpurun_out/r6_loop gpurun_out/r6_earlyg
T=tests/test_e2e_gpu.py
python -m pytest -q -m gpu $T tests/test_collective_gpu.py -k "early_g or derived_refresh or two_d_losses or alternating_batch or launch_plan_under or plan_replay or deferred_d_update or three_pass or early_real" 2>&1 | tail -15 | tee gpurun_out/r6_earlyg/tests.txt
for i in 1 2; do for v in 0 1 2; do echo -n "[tail=$v] "; PGGAN_TAIL_WGRAD_MAIN=$v timeout 300 python bench.py --no-cpu --no-per-depth --no-configs --no-kernel-timing --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['d_step_gp']['ms'])"; done; done 2>&1 | tee gpurun_out/r6_earlyg/ab_tail_depth8.txt
for v in 0 1 2; do echo -n "[tail=$v depth 7] "; PGGAN_TAIL_WGRAD_MAIN=$v timeout 300 python bench.py --depth 7 --no-cpu --no-per-depth --no-configs --no-kernel-timing --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['d_step_gp']['ms'])"; done 2>&1 | tee gpurun_out/r6_earlyg/ab_tail_depth7.txt
tools/ab.sh PGGAN_EARLY_G_MIN_RES=256 PGGAN_EARLY_G_MIN_RES=128 2 --depth 5 2>&1 | tee gpurun_out/r6_earlyg/ab_depth5.txt
PGGAN_TAIL_WGRAD_MAIN=2 python tools/phase_timeline.py > gpurun_out/r6_earlyg/phase_timeline_tail2.txt 2>&1; tail -8 gpurun_out/r6_earlyg/phase_timeline_tail2.txt
timeout 1500 tools/loop_tests.sh 150 gpurun_out/r6_loop/new_code_loops_others.txt -x $T::test_launch_plan_replay_matches_eager $T::test_deferred_d_update_matches_inline $T::test_three_pass_d_forward_matches_whole_batch_forward $T::test_trainer_early_g_forward_matches_in_step_forward
