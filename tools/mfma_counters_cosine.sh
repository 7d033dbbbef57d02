set -e
root=$(pwd)
export TMPDIR=/tmp
cd /tmp
rm -rf $root/gpurun_out/r03_mfma_cos
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $root/gpurun_out/r03_mfma_cos -- python3 $root/tools/config_bench.py cosine > $root/gpurun_out/r03_mfma_cos.log 2>&1
cd $root
tail -2 gpurun_out/r03_mfma_cos.log | cut -c1-200
