set -x
mkdir -p gpurun_out/r6_full
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee gpurun_out/r6_full/gputests.txt
python bench.py > gpurun_out/r6_full/bench.json 2> gpurun_out/r6_full/bench.err; cp bench_detail.json gpurun_out/r6_full/bench_detail.json; tail -c 3000 gpurun_out/r6_full/bench.json
PGGAN_FORCE_DP=1 python bench.py --no-cpu --no-configs > gpurun_out/r6_full/bench_dp.json 2> gpurun_out/r6_full/bench_dp.err; cp bench_detail.json gpurun_out/r6_full/bench_dp_detail.json; tail -c 2500 gpurun_out/r6_full/bench_dp.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
