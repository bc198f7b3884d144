set -x
mkdir -p gpurun_out/r6_loop
T=tests/test_e2e_gpu.py
P=$T::test_plan_replay_public_api_loop
PGGAN_DERIVED_EVENT=0 timeout 1200 tools/loop_tests.sh 300 gpurun_out/r6_loop/old_code_public_loop.txt $P
PGGAN_DERIVED_EVENT=0 AMD_SERIALIZE_KERNEL=3 timeout 1200 tools/loop_tests.sh 100 gpurun_out/r6_loop/old_code_public_loop_serialized.txt $P
timeout 1500 tools/loop_tests.sh 300 gpurun_out/r6_loop/new_code_loops.txt -x $P $T::test_launch_plan_replay_matches_eager $T::test_deferred_d_update_matches_inline $T::test_three_pass_d_forward_matches_whole_batch_forward
