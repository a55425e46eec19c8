#!/bin/bash
# PMC passes (separate runs, --kernel-trace only) of a short eager bench -> per-family summary JSON (tools/pmc_summary.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$R/gpurun_out/pmc/pmc$i" -o p -- python "$R/bench.py" --steps 2 --warmup 1 --cpu-baseline-videos 0 --no-graph --single-stream --no-kernel-events > "$R/gpurun_out/pmc/pmc$i.log" 2>&1
  echo "pmc $i rc=$?"
done
cd "$R"
python tools/pmc_summary.py gpurun_out/pmc gpurun_out/pmc/summary.json > gpurun_out/pmc/summary.log 2>&1; tail -3 gpurun_out/pmc/summary.log
find gpurun_out/pmc -name "*kernel_trace.csv" -delete; du -sh gpurun_out/pmc
