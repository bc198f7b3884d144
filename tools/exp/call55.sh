cd /root/repo
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r2_gputests.log; cat gpurun_out/r2_gputests.log
bash tools/profile_round.sh r02 > gpurun_out/r2_prof.log 2>&1
python bench.py > gpurun_out/r2_bench_full.log 2> gpurun_out/r2_bench_full.err; tail -c 300 gpurun_out/r2_bench_full.err
python bench.py --no-cpu --no-per-depth --no-configs --kernel-table --serial-kernel-timing > gpurun_out/r2_b_kt5.log 2> gpurun_out/r2_kt_serial_final.txt
PGGAN_FORCE_DP=1 python bench.py --no-cpu --no-per-depth --no-configs > gpurun_out/r2_bench_dp1.log 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
