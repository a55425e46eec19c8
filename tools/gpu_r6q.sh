#!/bin/bash
# call Q: over-fetch of the one-tap weight-gradient family, problem by problem (events, then FETCH_SIZE)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
timeout 600 python tools/pmc_wgrad_alone.py run gpurun_out/r6/q_alone > gpurun_out/r6/q_alone.txt 2> gpurun_out/r6/q_alone.err; echo "events rc=$?"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/r6/q_pmc" -o p -- python "$R/tools/pmc_wgrad_alone.py" run "$R/gpurun_out/r6/q_pmcrun" > "$R/gpurun_out/r6/q_pmc.log" 2>&1; echo "pmc rc=$?"
cd "$R"
python tools/pmc_wgrad_alone.py parse gpurun_out/r6/q_pmcrun gpurun_out/r6/q_pmc > gpurun_out/r6/q_overfetch.txt 2>&1
find gpurun_out/r6/q_pmc -name "*kernel_trace.csv" -delete
cat gpurun_out/r6/q_alone.txt gpurun_out/r6/q_overfetch.txt | cut -c1-200
