#!/bin/bash
# Round 4, call G: where the non-convolution time of the Inception-v3 dense test goes (rocprofv3 kernel stats of one video)
O=gpurun_out/r4g; mkdir -p $O
R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/dense3 -o d -- python $R/bench.py --mode dense-test --arch InceptionV3 --steps 1 --warmup 1 --cpu-baseline-videos 0 > $R/$O/dense3.log 2>&1
cd $R; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
f=$(find $O/dense3 -name "*kernel_stats.csv" | head -1); cp "$f" $O/dense3_kernel_stats.csv; head -30 $O/dense3_kernel_stats.csv | cut -c1-150
