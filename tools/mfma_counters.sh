set -e
TAG=${1:-r03}
root=$(pwd)
export TMPDIR=/tmp
cd /tmp
rm -rf $root/gpurun_out/${TAG}_mfma_a $root/gpurun_out/${TAG}_mfma_b
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $root/gpurun_out/${TAG}_mfma_a -- python3 $root/tools/plan_profile.py 256 512 u8 > $root/gpurun_out/${TAG}_mfma_a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $root/gpurun_out/${TAG}_mfma_b -- python3 $root/tools/plan_profile.py 256 512 u8 > $root/gpurun_out/${TAG}_mfma_b.log 2>&1
cd $root
python3 tools/sq_counters.py gpurun_out/${TAG}_mfma_a gpurun_out/${TAG}_mfma_b > gpurun_out/${TAG}_mfma_counters.txt 2>&1
tail -3 gpurun_out/${TAG}_mfma_a.log
