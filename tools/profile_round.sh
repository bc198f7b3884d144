#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats of the default bench command plus
# separate PMC passes (never combined with sys/hip traces).  Outputs under gpurun_out/prof/<tag>/.
# usage: profile_round.sh <tag> [extra bench.py arguments, e.g. --config 3]   (extra arguments: no per-layer table, that is a 1024^2 tool)
TAG=${1:-r05}
shift
EXTRA="$@"
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py $EXTRA --no-cpu --no-per-depth --no-configs --no-d-step --prime 10 --steps 10 --warmup 3 > $OUT/bench_kernels.json 2> $OUT/bench_kernels.err; cp $R/bench_detail.json $OUT/bench_detail.json      # un-traced, FIRST (after PMC passes the clocks stay low for a while): its per-symbol launch counts are checked against the trace
BENCH="python $R/bench.py $EXTRA --no-cpu --no-per-depth --no-configs --no-kernel-timing --no-d-step --prime 10 --steps 10 --warmup 3"      # (no instrumented passes after the timed loop: the trace ends with the timed steps)
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- $BENCH > $OUT/bench_kt.log 2>&1
PMCB="python $R/bench.py $EXTRA --no-cpu --no-per-depth --no-kernel-timing --no-configs --no-d-step --prime 0 --steps 2 --warmup 1"
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p --output-format csv -- $PMCB > $OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o p --output-format csv -- $PMCB > $OUT/pmc_write.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA -d $OUT/pmc_sq -o p --output-format csv -- $PMCB > $OUT/pmc_sq.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_lds -o p --output-format csv -- $PMCB > $OUT/pmc_lds.log 2>&1
# the north-star window by counter: D step + gradient penalty + Adam(D) only (3 passes)
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d $OUT/pmc_sq_dstep -o p --output-format csv -- $PMCB --d-step-only --warmup 3 --steps 3 > $OUT/pmc_sq_dstep.log 2>&1      # (argparse: the last --warmup / --steps win; the timed passes replay the launch plan and sit between two marker launches)
# per-layer table of the Winograd conv launches of one step, each alone on cold inputs (tools/layer_table.py)
if [ -z "$EXTRA" ]; then
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d $OUT/pmc_layers -o p --output-format csv -- python $R/tools/layer_table.py run $OUT/layers.json > $OUT/pmc_layers.log 2>&1
python $R/tools/layer_table.py table $OUT/layers.json $(find $OUT/pmc_layers -name p_counter_collection.csv | head -1) $OUT/${TAG}_wino_layer_table.csv >> $OUT/summary_layers.log 2>&1
fi
# keep only the small summaries (the raw traces are large)
python $R/tools/summarize_profile.py $OUT $TAG > $OUT/summary.log 2>&1
python $R/tools/stream_overlap.py "$OUT/kt/**/kt_kernel_trace.csv" > $OUT/${TAG}_stream_overlap.txt 2>&1
python $R/tools/step_timeline.py "$OUT/kt/**/kt_kernel_trace.csv" 3 > $OUT/${TAG}_step_timeline.txt 2>&1
rm -f $OUT/*/p_kernel_trace.csv $OUT/kt/kt_kernel_trace.csv $OUT/*/p_counter_collection.csv $OUT/*/*_agent_info.csv
find $OUT -name '*.csv' -size +2M -delete      # gpurun merges at most 64 MiB back
ls -la $OUT $OUT/* | head -40
