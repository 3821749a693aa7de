#!/bin/bash
# Per-workload profile on the GPU box (run through gpurun from the repo root):
#   tools/gpu_pmc.sh WORKLOAD TAG [FRAMES]
# rocprofv3 kernel stats, then one PMC pass per counter group (separate runs: a pass holds at most
# 8 SQ / 4 TCC counters; FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2), each under its own time limit, all on
# `python3 tools/profile_workload.py WORKLOAD FRAMES`.  Results land under gpurun_out/TAG/; fold them into
# profiles/ with tools/update_traffic.py.  Stops at the first failing step.
set -u
WL=$1; TAG=$2; FRAMES=${3:-10}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
# the same program without a profiler first: what the profiler itself costs the kernels (clock, serialisation) is stated, not hidden
timeout -k 10 300 python3 tools/profile_workload.py $WL $FRAMES > "$OUT/run_unprofiled.txt" 2> "$OUT/run_unprofiled.err" || { echo plain run failed; tail -5 "$OUT/run_unprofiled.err"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 tools/profile_workload.py $WL $FRAMES > "$OUT/run.txt" 2> "$OUT/prof.err" || { echo rocprof failed; tail -5 "$OUT/prof.err"; exit 1; }
cat "$OUT/run.txt"
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
            "ta:TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
            "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_VMEM" \
            "sq2:SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH GRBM_GUI_ACTIVE"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    timeout -s KILL 240 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/pmc_$name" -- python3 tools/profile_workload.py $WL $FRAMES > "$OUT/pmc_$name.txt" 2> "$OUT/pmc_$name.err" || { echo "pmc $name failed"; tail -5 "$OUT/pmc_$name.err"; exit 1; }
done
python tools/pmc_summary.py "$OUT"/pmc_* > "$OUT/pmc_summary.txt"
python tools/update_traffic.py --dry-run $WL "$OUT" $FRAMES > "$OUT/fold_preview.json" 2>&1 || true
