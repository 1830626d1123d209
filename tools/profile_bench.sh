#!/bin/bash
# (--no-graph: the passes are enqueued launch by launch under the profiler -- the same kernels as the graph replays of a plain run)
# Profiles bench.py on the GPU box: per-kernel durations (rocprofv3 --kernel-trace --stats) and, in separate passes,
# the HBM traffic counters FETCH_SIZE / WRITE_SIZE.  Usage: tools/profile_bench.sh <tag> [bench.py args...]
set -u
TAG=${1:-run}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
# the kernel sources these measurements belong to (bench.py nulls `roofline.traffic` when the tree has moved on)
(cd "$ROOT" && python3 -c "import bench; print(bench.kernel_source_hash())") > "$OUT/kernel_source_hash.txt"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-ragged --no-graph --also "" "$@" > "$OUT/bench_traced.json" 2> "$OUT/trace.err"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-ragged --no-graph --also "" "$@" > "$OUT/bench_pmc_$C.json" 2> "$OUT/pmc_$C.err"
  python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_$C" "$OUT/pmc_${C}_summary.json" > "$OUT/pmc_${C}_summary.txt" 2>&1
done
# matrix-pipe occupancy of every kernel (own pass: SQ counters only, with the GRBM clock counter)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_MFMA" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-ragged --no-graph --also "" "$@" > "$OUT/bench_pmc_MFMA.json" 2> "$OUT/pmc_MFMA.err"
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_MFMA" "$OUT/pmc_MFMA_summary.json" > "$OUT/pmc_MFMA_summary.txt" 2>&1
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
# keep the merge small: drop the raw per-dispatch traces
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*.db" -delete
ls -la "$OUT"
