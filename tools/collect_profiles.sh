#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + HBM PMC passes of the bench command.
#   tools/collect_profiles.sh r01 [extra bench.py arguments]
# Writes raw output under gpurun_out/prof_<tag>/ ; tools/profile_summary.py turns it into profiles/<tag>_*.
set -u
TAG=${1:-r01}
shift || true
EXTRA="$*"                         # extra bench.py arguments, e.g. --resolution 512 --batch 32, --model recnext_a3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
CMD="python3 bench.py --steps 20 --warmup 10 --no-cpu-baseline $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- $CMD > "$OUT/kt_bench.log" 2>&1
# counters in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $CMD > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $CMD > "$OUT/pmc_write.log" 2>&1
python3 tools/profile_summary.py "$TAG" "$OUT" "$ROOT/gpurun_out/profiles_$TAG"
ls -la "$ROOT/gpurun_out/profiles_$TAG"
rm -rf "$OUT/kt" "$OUT/pmc_fetch" "$OUT/pmc_write"      # the raw traces are tens of MB per configuration: gpurun copies back at most 64 MiB
