#!/bin/bash
# Hardware counters of one command, one rocprofv3 --pmc pass per counter group (a pass holds at most 8 SQ, 4 TCC
# (FETCH_SIZE takes 3, WRITE_SIZE 2), 4 TCP and 2 GRBM counters), plus a kernel trace for the durations.
#   tools/pmc_collect.sh NAME python3 tools/closest_probe.py --scene soup
# writes gpurun_out/pmc_NAME_<group>/ ; summarise with tools/pmc_json.py.
export TMPDIR=/tmp
name=$1; shift
run() { g=$1; shift; timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_${name}_$g -- "${CMD[@]}" > gpurun_out/pmc_${name}_$g.log 2>&1 || { echo "pass $g failed"; tail -3 gpurun_out/pmc_${name}_$g.log; exit 1; }; echo "pass $g ok"; }
CMD=("$@")
run fetch FETCH_SIZE && run write WRITE_SIZE && run l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum \
 && run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD \
 && run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum \
 && run grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY \
 && timeout -s KILL 150 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pmc_${name}_trace -- "${CMD[@]}" > gpurun_out/pmc_${name}_trace.log 2>&1 && echo "trace ok"
