set -x
mkdir -p gpurun_out/r6_loop
for pn in "" "--pixelnorm"; do
  python tools/exp/r6_lockstep_diag.py --reps 120 $pn > gpurun_out/r6_loop/diag_default${pn}.txt 2>&1; tail -12 gpurun_out/r6_loop/diag_default${pn}.txt
  python tools/exp/r6_lockstep_diag.py --reps 120 --no-splitk $pn > gpurun_out/r6_loop/diag_nosplitk${pn}.txt 2>&1; tail -6 gpurun_out/r6_loop/diag_nosplitk${pn}.txt
done
python tools/phase_timeline.py > gpurun_out/r6_loop/phase_timeline.txt 2>&1; tail -45 gpurun_out/r6_loop/phase_timeline.txt
python -m pytest -q -m gpu -x tests/test_collective_gpu.py tests/test_fp64_adjudicator.py -k "launch_plan_under or 32-" 2>&1 | tail -15
