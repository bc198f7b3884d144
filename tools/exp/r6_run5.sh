set -x
mkdir -p gpurun_out/r6_split
T=tests/test_e2e_gpu.py
python -m pytest -q -m gpu $T tests/test_collective_gpu.py tests/test_checkpoint.py -k "early_g or derived_refresh or launch_plan or plan_replay or deferred_d_update or three_pass or early_real or trainer or checkpoint or hipgraph" 2>&1 | tail -8 | tee gpurun_out/r6_split/tests.txt
tools/ab.sh PGGAN_SPLIT_DERIVE=0 PGGAN_SPLIT_DERIVE=1 3 2>&1 | tee gpurun_out/r6_split/ab_split_depth8.txt
tools/ab.sh PGGAN_DERIVED_ONE_LAUNCH=1 PGGAN_DERIVED_ONE_LAUNCH=0 3 2>&1 | tee gpurun_out/r6_split/ab_onelaunch_depth8.txt
tools/ab.sh PGGAN_SPLIT_DERIVE=0 PGGAN_SPLIT_DERIVE=1 2 --depth 6 2>&1 | tee gpurun_out/r6_split/ab_split_depth6.txt
tools/ab.sh PGGAN_SPLIT_DERIVE=0 PGGAN_SPLIT_DERIVE=1 1 --depth 3 2>&1 | tee gpurun_out/r6_split/ab_split_depth3.txt
python tools/phase_timeline.py > gpurun_out/r6_split/phase_timeline.txt 2>&1; tail -30 gpurun_out/r6_split/phase_timeline.txt
