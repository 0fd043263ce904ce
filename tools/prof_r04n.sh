#!/bin/bash
# dataset-scale step: tests of the touched paths, per-step wall time, ordered kernel trace of one graphed Cora-shaped training step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
if [ -n "$TESTS" ]; then python -m pytest $TESTS -x -q -m gpu 2>&1 | tail -15; fi
python tools/small_graph_step.py 2>&1 | grep -v Warn | tee gpurun_out/r04n_small_graph_step.txt
cd /tmp
rm -rf $R/gpurun_out/prof_small_trace; mkdir -p $R/gpurun_out/prof_small_trace
REPLAYS=3 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_small_trace -- python $R/tools/small_graph_kernels.py ${CASE:-cora_ds_add} > $R/gpurun_out/prof_small_trace.log 2>&1
tail -2 $R/gpurun_out/prof_small_trace.log
