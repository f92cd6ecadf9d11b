#!/bin/bash
# Everything the round's profiles/ directory is built from, in one GPU session:  tools/collect_round.sh r03
# (rocprofv3 kernel trace + stats and the separate PMC passes of bench.py via collect_profiles.sh, then the secondary lines:
#  config 5, likelihood-only, the shard emulation, per-dispatch traces, the end-to-end breakdown).
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
TAG=${1:-r03}
P=gpurun_out/profiles_$TAG
bash tools/collect_profiles.sh "$TAG"
echo "== secondary lines" 
python3 bench.py --likelihood-only --no-cpu-baseline --steps 20 --warmup 3 > "$P/${TAG}_bench_likelihood_only.json" 2> "$P/lik.err"
python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline > "$P/${TAG}_bench_c5.json" 2> "$P/c5.err"
python3 bench.py --config c5 --likelihood-only --steps 5 --warmup 2 --no-cpu-baseline > "$P/${TAG}_bench_c5_likelihood_only.json" 2> "$P/c5l.err"
python3 tools/shard_timing.py 1 2 4 8 > "$P/${TAG}_shard_emulation_2_4_8.txt" 2>&1
python3 tools/shard_timing.py --config c5 1 2 4 8 > "$P/${TAG}_c5_shard_emulation_2_4_8.txt" 2>&1
python3 tools/shard_timing.py --config c5 --likelihood-only 1 8 > "$P/${TAG}_c5_shard_emulation_likelihood_only_1_8.txt" 2>&1
MRA_TRACE_PLAN=1 MRA_TRACE_REPLAY=1 python3 tools/e2e_breakdown.py > "$P/${TAG}_e2e_breakdown.txt" 2>&1
# the N > 1 paths, rehearsed on this one GPU (every rank on device 0, the front exchange through gloo): bench.py starts its own ranks
MRA_BENCH_SINGLE_DEVICE=1 python3 bench.py --gpus 2 --exchange gloo --steps 10 --warmup 3 --no-cpu-baseline > "$P/${TAG}_bench_2ranks_one_gpu_gloo_rehearsal.json" 2> "$P/g2.err"
MRA_BENCH_SINGLE_DEVICE=1 python3 bench.py --config c5 --gpus 4 --exchange gloo --mle --steps 3 --warmup 1 --no-cpu-baseline > "$P/${TAG}_bench_c5_4ranks_one_gpu_gloo_rehearsal_mle.json" 2> "$P/c5g4.err"
bash tools/trace_kernels.sh 1gpu -- python3 tools/one_pass.py 1 > "$P/${TAG}_trace_one_pass_1gpu.txt" 2>&1
bash tools/trace_kernels.sh 8way -- python3 tools/one_pass.py 8 > "$P/${TAG}_trace_one_pass_8way_shard_rank0.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_${TAG}_c5" -- python3 bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > "$P/c5_kt.log" 2>&1
cp $(ls -t gpurun_out/prof_${TAG}_c5/*/*_kernel_stats.csv | head -1) "$P/${TAG}_c5_rocprofv3_kernel_stats.csv"
# (gloo prints its rank chatter on stdout: keep the JSON line)
for f in "$P/${TAG}_bench_2ranks_one_gpu_gloo_rehearsal.json" "$P/${TAG}_bench_c5_4ranks_one_gpu_gloo_rehearsal_mle.json"; do grep '^{"metric"' "$f" > "$f.tmp" && mv "$f.tmp" "$f"; done
rm -f "$P"/*.err "$P"/*.log
ls -la "$P"
# SQ counters of a config-5 pass (matrix-pipe busy fraction per kernel): counters in a run of their own, the program itself behind --
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$R/gpurun_out/prof_${TAG}_c5sq" -- python3 tools/shard_timing.py --config c5 1 > "$P/c5_sq.log" 2>&1
python3 tools/summarize_profiles.py --sq "$(ls -t gpurun_out/prof_${TAG}_c5sq/*/*_counter_collection.csv | head -1)" "$P/${TAG}_c5_pmc_sq_summary.json"
rm -f "$P"/*.log
