#!/bin/bash
# Per-bounce record of the staged pipeline with and without the packet walks of the primary rays and mirror-floor bounces
# (run through gpurun):  tools/gpu_packet.sh TAG [WORKLOAD] [FRAMES]  -> gpurun_out/TAG_per_bounce.txt
set -u
TAG=$1; WL=${2:-soup1m_4k_d4}; FRAMES=${3:-6}; export TMPDIR=/tmp RT_TRAVERSAL=3
DEPTH=$(python3 -c "import bench; print(bench.WORKLOADS['$WL'][4])")
: > gpurun_out/${TAG}_per_bounce.txt
for P in 0 1; do
    export RT_WF_PACKET=$P
    OUT=gpurun_out/${TAG}_packet$P; mkdir -p $OUT
    timeout -k 10 300 python3 tools/profile_workload.py $WL $FRAMES > $OUT/run_unprofiled.txt 2> $OUT/run_unprofiled.err || { echo plain run failed; exit 1; }
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/profile_workload.py $WL $FRAMES > $OUT/run.txt 2> $OUT/trace.err || { echo trace failed; exit 1; }
    timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc_sq -- python3 tools/profile_workload.py $WL $FRAMES > $OUT/pmc_sq.txt 2> $OUT/pmc_sq.err || { echo sq failed; exit 1; }
    { echo "== RT_WF_PACKET=$P  ($(cat $OUT/run_unprofiled.txt))"; python3 tools/per_bounce.py $DEPTH $OUT/trace $OUT/pmc_sq; echo; } >> gpurun_out/${TAG}_per_bounce.txt
done
cat gpurun_out/${TAG}_per_bounce.txt
