#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box from the repo root; writes under gpurun_out/):
#   1. kernel trace + stats of the default bench command (30 denoise steps)
#   2. PMC passes, one run each (FETCH_SIZE / WRITE_SIZE / MFMA busy / LDS), over a 2-denoise-step run of the same kernels and
#      shapes (a full 30-step run under --pmc is ~10x slower and adds nothing per launch)
# Summaries: tools/pmc_summary.py.  Each pass has its own time limit; a failing pass ends the script.
set -e
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
A="--steps 1 --warmup 0 --denoise-steps 2 --no-cpu-baseline"
# counters are collected on plain launches (SDE_GRAPH=0: the same kernels, one dispatch each, without the hipGraph replay)
export SDE_GRAPH=${SDE_GRAPH_PROFILE:-0}
O=gpurun_out
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_v3 -o k -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_v3_bench.log 2>&1
echo stats-done
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python3 bench.py $A > $O/pmc_f.log 2>&1
echo fetch-done
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python3 bench.py $A > $O/pmc_w.log 2>&1
echo write-done
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE MfmaUtil --kernel-trace --output-format csv -d $O/pmc_m -o m -- python3 bench.py $A > $O/pmc_m.log 2>&1
echo mfma-done
timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_l -o l -- python3 bench.py $A > $O/pmc_l.log 2>&1
echo lds-done
