#!/bin/bash
# PMC passes (each its own rocprofv3 run, kernel-trace only) over the backbone fwd+bwd micro-workload
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > "$R/gpurun_out/pmc/counters.txt" 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$R/gpurun_out/pmc/p$i" -o p -- python "$R/tools/layer_table.py" > "$R/gpurun_out/pmc/run$i.log" 2>&1
  echo "pass $i rc=$?" >> "$R/gpurun_out/pmc/run$i.log"
  tail -2 "$R/gpurun_out/pmc/run$i.log"
done
cd "$R"; find gpurun_out/pmc -name "*.csv" | xargs ls -la | head -20
