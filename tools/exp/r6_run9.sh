set -x
bash tools/profile_round.sh r06b > gpurun_out/prof_r06b.log 2>&1; tail -2 gpurun_out/prof_r06b.log
mkdir -p gpurun_out/r6_final
python tools/phase_timeline.py > gpurun_out/r6_final/phase_timeline_d8.txt 2>&1; tail -12 gpurun_out/r6_final/phase_timeline_d8.txt
python tools/phase_timeline.py --depth 7 > gpurun_out/r6_final/phase_timeline_d7.txt 2>&1; tail -5 gpurun_out/r6_final/phase_timeline_d7.txt
PGGAN_DP_SHARE_GPU=1 PGGAN_DP_CONTROL=gloo PGGAN_DP_TORCH_ALLREDUCE=1 timeout 900 python bench.py --gpus 2 --no-cpu --no-per-depth --no-configs --steps 10 --warmup 3 > gpurun_out/r6_final/dp2_share.json 2> gpurun_out/r6_final/dp2_share.err; echo rc=$?; tail -c 600 gpurun_out/r6_final/dp2_share.json; tail -3 gpurun_out/r6_final/dp2_share.err
rocm-smi --showclocks 2>/dev/null | head -20
