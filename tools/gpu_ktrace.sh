mkdir -p gpurun_out/r3
R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3/ktrace -o graph -- python $R/bench.py --steps 4 --warmup 2 --cpu-baseline-videos 0 --no-kernel-events > $R/gpurun_out/r3/ktrace.log 2>&1
cd $R; f=$(find gpurun_out/r3/ktrace -name "*kernel_trace.csv" | head -1); ls -la $f; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), rows[0].keys())
# keep the last ~1/6 of the rows (one step), columns trimmed
keep = ['Kernel_Name','Start_Timestamp','End_Timestamp','Queue_Id','Stream_Id','Grid_Size_X','Workgroup_Size_X','LDS_Block_Size','VGPR_Count']
keep = [k for k in keep if k in rows[0]]
import gzip
with gzip.open('gpurun_out/r3/ktrace_graph_small.csv.gz','wt') as f:
    w = csv.writer(f); w.writerow(keep)
    for r in rows[-6000:]:
        w.writerow([r[k][:60] if k=='Kernel_Name' else r[k] for k in keep])
PY
find gpurun_out/r3/ktrace -type f -delete
