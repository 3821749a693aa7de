#!/bin/bash
# One measurement round on the GPU box (run through gpurun from the repo root):
#   tools/gpu_measure.sh TAG [tests]
# GPU parity tests (optional), bench.py, rocprofv3 kernel stats of the same bench command and the
# three PMC passes (fetch, write, SQ), each step under its own time limit; everything lands under
# gpurun_out/TAG/.  Stops at the first failing step.
set -u
TAG=${1:-meas}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="bench.py --no-cpu --no-extras --steps 50 --warmup 5"

if [ "${2:-}" = "tests" ]; then
    timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$OUT/gpu_tests.log" 2>&1
    rc=$?; tail -8 "$OUT/gpu_tests.log"; [ $rc -eq 0 ] || exit $rc
fi
timeout -k 10 900 python -u bench.py ${BENCH_FLAGS:-} > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "bench failed"; tail -5 "$OUT/bench.err"; exit 1; }
tail -c 3000 "$OUT/bench.json"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 $BENCH > "$OUT/prof_bench.json" 2> "$OUT/prof.err" || { echo "rocprof failed"; tail -5 "$OUT/prof.err"; exit 1; }
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" \
            "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_SMEM" \
            "sq2:SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/pmc_$name" -- python3 bench.py --no-cpu --no-extras --steps 10 --warmup 2 > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err" || { echo "pmc $name failed"; tail -5 "$OUT/pmc_$name.err"; exit 1; }
done
python tools/pmc_summary.py "$OUT"/pmc_* > "$OUT/pmc_summary.txt"
grep k_render "$OUT/pmc_summary.txt" | grep -E "FETCH|WRITE"
find "$OUT/prof" -name "*kernel_stats.csv" -exec head -4 {} \;
