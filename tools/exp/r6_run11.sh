set -x
mkdir -p gpurun_out/r6_batched
python -m pytest -q -m gpu tests/test_e2e_gpu.py tests/test_collective_gpu.py -k "early_g or plan or trainer or deferred" 2>&1 | tail -6 | tee gpurun_out/r6_batched/tests2.txt
tools/ab.sh PGGAN_EARLY_G=0 PGGAN_EARLY_G=1 2 --depth 0 2>&1 | tee gpurun_out/r6_batched/ab2_depth0.txt
tools/ab.sh PGGAN_EARLY_G_BATCHED_MAX_RES=4 PGGAN_EARLY_G_BATCHED_MAX_RES=8 2 --depth 1 2>&1 | tee gpurun_out/r6_batched/ab2_depth1.txt
tools/ab.sh PGGAN_EARLY_G=0 PGGAN_EARLY_G=1 1 2>&1 | tee gpurun_out/r6_batched/ab2_depth8.txt
