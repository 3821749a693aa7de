#!/bin/bash
# Profile one workload with tools/quick_bench.py: kernel stats + cache / memory PMC passes.
#   tools/gpu_prof.sh TAG TRAVERSAL WORKLOAD   (TRAVERSAL: 0 auto 1 wave 2 bvh 3 wavefront)
set -u
TAG=$1; export RT_TRAVERSAL=$2; WL=$3
OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 tools/quick_bench.py $WL > "$OUT/bench.txt" 2> "$OUT/prof.err" || { echo rocprof failed; tail -5 "$OUT/prof.err"; exit 1; }
cat "$OUT/bench.txt"; cut -c1-140 "$OUT"/prof/*/*kernel_stats.csv | head -8
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
            "tcp:TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
            "ta:TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
            "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_VMEM" \
            "sq2:SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    timeout -s KILL 200 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/pmc_$name" -- python3 tools/quick_bench.py $WL > "$OUT/pmc_$name.txt" 2> "$OUT/pmc_$name.err" || { echo "pmc $name failed"; tail -5 "$OUT/pmc_$name.err"; exit 1; }
done
python tools/pmc_summary.py "$OUT"/pmc_* > "$OUT/pmc_summary.txt"
grep -E "k_wf|k_render" "$OUT/pmc_summary.txt" | grep -v "k_render<true>" | cut -c13-
