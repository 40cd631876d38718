#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box from the repo root; writes under gpurun_out/):
#   1. kernel trace + stats of the default bench command (30 denoise steps)
#   2. PMC passes, one run each, over a 2-denoise-step run of the same kernels and shapes (a 30-step run under --pmc died inside
#      the profiler with SIGSEGV on its first try this round).  For FETCH_SIZE / WRITE_SIZE tools/pmc_summary.py re-weights the
#      denoise loop's dispatches by 15 so that a kernel class's bytes per launch are averaged over the SAME launch mix as bench.py's
#      algorithmic_bytes_per_launch (UNet : VAE launches = 30 : 1); MFMA busy / LDS figures are per-launch ratios and need no mix.
# Summaries (tools/pmc_summary.py) land in gpurun_out/r3_*.json; copy the ones to keep into profiles/.
# Each pass has its own time limit; a failing pass ends the script.
set -e
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
A="--steps 1 --warmup 0 --denoise-steps 2 --no-cpu-baseline"
F="$A"
# counters are collected on plain launches (SDE_GRAPH=0: the same kernels, one dispatch each, without the hipGraph replay)
export SDE_GRAPH=${SDE_GRAPH_PROFILE:-0}
O=gpurun_out
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_v3 -o k -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_v3_bench.log 2>&1
echo stats-done
timeout -s KILL 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python3 bench.py $F > $O/pmc_f.log 2>&1
echo fetch-done
timeout -s KILL 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python3 bench.py $F > $O/pmc_w.log 2>&1
echo write-done
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE MfmaUtil --kernel-trace --output-format csv -d $O/pmc_m -o m -- python3 bench.py $A > $O/pmc_m.log 2>&1
echo mfma-done
timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_l -o l -- python3 bench.py $A > $O/pmc_l.log 2>&1
echo lds-done
csvof() { find "$1" -name '*counter_collection.csv' | sort | tail -1; }
python3 tools/pmc_summary.py --traffic "$(csvof $O/pmc_f)" "$(csvof $O/pmc_w)" $O/r3_pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --steps 1 --denoise-steps 2, denoise-loop dispatches weighted x15 = the bench's 30-step launch mix, SDE_GRAPH=0 (tools/collect_profiles.sh); round 3" 15
python3 tools/pmc_summary.py "$(csvof $O/pmc_m)" $O/r3_pmc_mfma_util.json
python3 tools/pmc_summary.py "$(csvof $O/pmc_l)" $O/r3_pmc_lds.json
cp "$(find $O/prof_v3 -name '*kernel_stats.csv' | sort | tail -1)" $O/r3_rocprofv3_kernel_stats.csv
grep -h '^{' $O/prof_v3_bench.log | tail -1 > $O/r3_bench_under_rocprof.json
echo summaries-done
