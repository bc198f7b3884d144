set -x
mkdir -p gpurun_out/r6_loop gpurun_out/r6_earlyg
T=tests/test_e2e_gpu.py
python -m pytest -q -m gpu -x $T tests/test_collective_gpu.py -k "early_g or derived_refresh or two_d_losses or alternating_batch or launch_plan_under or plan_replay or deferred_d_update or three_pass" 2>&1 | tail -15 | tee gpurun_out/r6_earlyg/tests.txt
python tools/phase_timeline.py > gpurun_out/r6_earlyg/phase_timeline_early_g.txt 2>&1; tail -40 gpurun_out/r6_earlyg/phase_timeline_early_g.txt
PGGAN_EARLY_G=0 python tools/phase_timeline.py > gpurun_out/r6_earlyg/phase_timeline_off.txt 2>&1; tail -8 gpurun_out/r6_earlyg/phase_timeline_off.txt
tools/ab.sh PGGAN_EARLY_G=0 PGGAN_EARLY_G=1 3 2>&1 | tee gpurun_out/r6_earlyg/ab_depth8.txt
tools/ab.sh PGGAN_EARLY_G=0 PGGAN_EARLY_G=1 2 --depth 7 2>&1 | tee gpurun_out/r6_earlyg/ab_depth7.txt
tools/ab.sh PGGAN_EARLY_G=0 PGGAN_EARLY_G=1 1 --depth 6 2>&1 | tee gpurun_out/r6_earlyg/ab_depth6.txt
tools/ab.sh PGGAN_EARLY_G=0 PGGAN_EARLY_G=1 1 --depth 4 2>&1 | tee gpurun_out/r6_earlyg/ab_depth4.txt
tools/ab.sh PGGAN_EARLY_G=0 PGGAN_EARLY_G=1 1 --depth 2 2>&1 | tee gpurun_out/r6_earlyg/ab_depth2.txt
P=$T::test_plan_replay_public_api_loop
timeout 1500 tools/loop_tests.sh 300 gpurun_out/r6_loop/new_code_loops.txt -x $P $T::test_launch_plan_replay_matches_eager $T::test_deferred_d_update_matches_inline $T::test_three_pass_d_forward_matches_whole_batch_forward
