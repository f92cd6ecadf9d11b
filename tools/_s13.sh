export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/prof_r04_c5sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$R/gpurun_out/prof_r04_c5sq/sq" -- python3 bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --likelihood-only > gpurun_out/prof_r04_c5sq/sq.log 2>&1
echo "sq rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d "$R/gpurun_out/prof_r04_c5sq/sq2" -- python3 bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --likelihood-only > gpurun_out/prof_r04_c5sq/sq2.log 2>&1
echo "sq2 rc=$?"
ls gpurun_out/prof_r04_c5sq/sq/*/ gpurun_out/prof_r04_c5sq/sq2/*/ 2>/dev/null | head
tail -3 gpurun_out/prof_r04_c5sq/sq2.log
