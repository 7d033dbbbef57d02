#!/bin/bash
# Collects the rocprofv3 evidence of one command on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh <tag> <python script and args...>
# -> gpurun_out/<tag>_stats (kernel trace + stats), gpurun_out/<tag>_fetch / _write (one PMC counter per pass, as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass), then tools/profile_summary.py
# condenses them into profiles/<tag>_*.  The program after `--` is python3 itself (no shell / env hop under rocprofv3).
set -e
tag=$1; shift
root=$(pwd)
export TMPDIR=/tmp
cd /tmp
rm -rf "$root/gpurun_out/${tag}_stats" "$root/gpurun_out/${tag}_fetch" "$root/gpurun_out/${tag}_write"
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/${tag}_stats" -- python3 "$root/$1" "${@:2}" > "$root/gpurun_out/${tag}_stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$root/gpurun_out/${tag}_fetch" -- python3 "$root/$1" "${@:2}" > "$root/gpurun_out/${tag}_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$root/gpurun_out/${tag}_write" -- python3 "$root/$1" "${@:2}" > "$root/gpurun_out/${tag}_write.log" 2>&1
cd "$root"
python3 tools/profile_summary.py "$tag" "gpurun_out/${tag}_stats" "gpurun_out/${tag}_fetch" "gpurun_out/${tag}_write" "$*" > "gpurun_out/${tag}_summary.log" 2>&1
tail -5 "gpurun_out/${tag}_stats.log"
