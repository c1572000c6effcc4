#!/bin/bash
# profiles/run_pmc_r02.sh <tag> -- HBM traffic of the hot kernels from the PMC counters, collected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), each with --kernel-trace only; the same two passes also over the calibration
# kernels of tools/exp/pmc_calib.hip (a 16 B/lane stream and random 8-byte gathers over a 48.8 MB table) so that the counters can be read
# in this build's own access patterns.  tools/pmc_summary.py turns the CSVs into profiles/<round>/pmc_traffic.json.
set -e
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CALIB=$GRAFT_REPO_ROOT/tools/exp/pmc_calib.bin      # built binaries are not kept in the tree: build the calibration kernels here if needed
[ -x $CALIB ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $GRAFT_REPO_ROOT/tools/exp/pmc_calib.hip -o $CALIB
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$C -o $C -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-second-point --no-fp32-point --no-trajectory-point --roofline-steps 0 --steps 12 --warmup 2 > $OUT/bench_$C.log 2>&1 || true
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_calib_$C -o $C -- $GRAFT_REPO_ROOT/tools/exp/pmc_calib.bin > $OUT/calib_$C.log 2>&1 || true
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmc_${TAG} $OUT
