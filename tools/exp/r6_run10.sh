set -x
mkdir -p gpurun_out/r6_batched
python -m pytest -q -m gpu tests/test_e2e_gpu.py -k "early_g" 2>&1 | tail -8 | tee gpurun_out/r6_batched/tests.txt
tools/ab.sh PGGAN_EARLY_G_MODE=side PGGAN_EARLY_G_MODE=batched 3 2>&1 | tee gpurun_out/r6_batched/ab_depth8.txt
tools/ab.sh PGGAN_EARLY_G_MODE=side PGGAN_EARLY_G_MODE=batched 2 --depth 7 2>&1 | tee gpurun_out/r6_batched/ab_depth7.txt
tools/ab.sh PGGAN_EARLY_G_MODE=side PGGAN_EARLY_G_MODE=batched 2 --depth 6 2>&1 | tee gpurun_out/r6_batched/ab_depth6.txt
tools/ab.sh "PGGAN_EARLY_G=0" "PGGAN_EARLY_G=2 PGGAN_EARLY_G_MODE=batched" 2 --depth 5 2>&1 | tee gpurun_out/r6_batched/ab_depth5.txt
tools/ab.sh "PGGAN_EARLY_G=0" "PGGAN_EARLY_G=2 PGGAN_EARLY_G_MODE=batched" 1 --depth 4 2>&1 | tee gpurun_out/r6_batched/ab_depth4.txt
tools/ab.sh "PGGAN_EARLY_G=0" "PGGAN_EARLY_G=2 PGGAN_EARLY_G_MODE=batched" 1 --depth 2 2>&1 | tee gpurun_out/r6_batched/ab_depth2.txt
tools/ab.sh "PGGAN_EARLY_G=0" "PGGAN_EARLY_G=2 PGGAN_EARLY_G_MODE=batched" 1 --depth 0 2>&1 | tee gpurun_out/r6_batched/ab_depth0.txt
PGGAN_EARLY_G_MODE=batched python tools/phase_timeline.py > gpurun_out/r6_batched/phase_timeline.txt 2>&1; tail -6 gpurun_out/r6_batched/phase_timeline.txt
