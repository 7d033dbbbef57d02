#!/bin/bash
# Derived-metric PMC passes (rocprofv3 --pmc, one small group per pass, --kernel-trace only) over the two network plans of the bench
# step, each kernel alone on the GPU:   tools/pmc_passes.sh <tag>   ->  gpurun_out/<tag>_pmc_<group>/ ; summary: tools/pmc_summary.py
set -e
TAG=${1:-r04}
root=$(pwd)
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "VALUBusy LdsBankConflict LdsUtil" "MemUnitStalled VmemLatency LdsLatency" "VALUUtilization MfmaUtil OccupancyPercent" "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i + 1))
  rm -rf $root/gpurun_out/${TAG}_pmc_$i
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $root/gpurun_out/${TAG}_pmc_$i -- python3 $root/tools/plan_profile.py 256 512 u8 > $root/gpurun_out/${TAG}_pmc_$i.log 2>&1 || echo "pass $i ($grp) failed"
done
cd $root
python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc_1 gpurun_out/${TAG}_pmc_2 gpurun_out/${TAG}_pmc_3 gpurun_out/${TAG}_pmc_4 > gpurun_out/${TAG}_pmc_summary.md 2>&1
tail -30 gpurun_out/${TAG}_pmc_summary.md
