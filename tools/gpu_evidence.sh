#!/bin/bash
# round evidence: Flow bench line, default bench line, rocprofv3 kernel stats of the default bench, PMC passes
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out/ev; export TMPDIR=/tmp
timeout 600 python bench.py --modality Flow --cpu-baseline-videos 0 > gpurun_out/ev/bench_flow.log 2>&1; echo "rc=$?" >> gpurun_out/ev/bench_flow.log
timeout 900 python bench.py > gpurun_out/ev/bench_default.log 2>&1; echo "rc=$?" >> gpurun_out/ev/bench_default.log
tail -2 gpurun_out/ev/bench_default.log | cut -c1-400
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/ev/prof" -o run -- python "$R/bench.py" --cpu-baseline-videos 0 > "$R/gpurun_out/ev/prof.log" 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$R/gpurun_out/ev/pmc$i" -o p -- python "$R/bench.py" --steps 3 --warmup 1 --cpu-baseline-videos 0 --no-graph --no-kernel-events > "$R/gpurun_out/ev/pmc$i.log" 2>&1
  echo "pmc $i rc=$?"
done
cd "$R"; find gpurun_out/ev -name "*kernel_trace.csv" -delete; du -sh gpurun_out/ev
