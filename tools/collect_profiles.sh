#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + stats of bench.py, then separate PMC passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950; counters are collected with --kernel-trace only).
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/kt" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/$OUT/fetch" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/$OUT/write" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/write.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$R/$OUT/sq" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/sq.log" 2>&1
python3 tools/summarize_profiles.py "$OUT" "$TAG"
# the bench line of the PROFILED run (its hipEvent durations and the trace's come out of one process)
grep '^{"metric"' "$OUT/kt.log" > "gpurun_out/profiles_$TAG/${TAG}_bench_under_rocprofv3_kernel_trace.json" || true
# the bench line quotes the PMC traffic of THIS build: make the fresh summary the newest one it finds
cp "gpurun_out/profiles_$TAG/${TAG}_pmc_traffic_by_kernel_family.json" profiles/
python3 bench.py --steps 20 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
cp "$OUT/bench.json" "gpurun_out/profiles_$TAG/${TAG}_bench.json"
