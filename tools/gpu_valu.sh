#!/bin/bash
# VALU issue calibration on the GPU box (through gpurun from the repo root):  tools/gpu_valu.sh TAG
# tools/valu_rate.hip (prebuilt: build/valu_rate) and tools/class_probe.py, each once plain (timings) and once under a
# single --pmc pass (what the counters say about the same kernels).  Results under gpurun_out/TAG/.
set -u
TAG=${1:-valu}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
CTRS="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU"
timeout -k 10 60 ./build/valu_rate > "$OUT/valu_rate.txt" 2>&1 || { echo valu_rate failed; tail -5 "$OUT/valu_rate.txt"; exit 1; }
cat "$OUT/valu_rate.txt"
timeout -s KILL 120 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/pmc_valu_rate" -- ./build/valu_rate > "$OUT/pmc_valu_rate.txt" 2> "$OUT/pmc_valu_rate.err" || { echo "pmc valu_rate failed"; tail -5 "$OUT/pmc_valu_rate.err"; exit 1; }
python tools/pmc_summary.py "$OUT/pmc_valu_rate" > "$OUT/valu_rate_pmc_summary.txt"
timeout -k 10 200 python3 tools/class_probe.py 6 > "$OUT/class_probe.txt" 2> "$OUT/class_probe.err" || { echo class_probe failed; tail -5 "$OUT/class_probe.err"; exit 1; }
cat "$OUT/class_probe.txt"
timeout -s KILL 240 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/pmc_class" -- python3 tools/class_probe.py 6 > "$OUT/pmc_class.txt" 2> "$OUT/pmc_class.err" || { echo "pmc class failed"; tail -5 "$OUT/pmc_class.err"; exit 1; }
python tools/class_probe_join.py "$OUT/class_probe.txt" "$OUT/pmc_class" > "$OUT/class_probe_pmc.txt"
cat "$OUT/class_probe_pmc.txt"
