#!/bin/bash
# Round profile on the GPU box: kernel stats + three PMC passes over bench.py, summaries into gpurun_out/$1/.
#   bash tools/profile_round.sh r2        (run through gpurun; copy the summaries you keep into profiles/)
set -e -o pipefail
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
B="python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-}"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $B > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
cp "$(ls "$OUT"/stats/*/*kernel_stats.csv | head -1)" "$OUT/kernel_stats.csv"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $B > /dev/null 2> "$OUT/pmc_fetch.err"
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $B > /dev/null 2> "$OUT/pmc_write.err"
python3 tools/pmc_summary.py "$(ls "$OUT"/pmc_fetch/*/*counter_collection.csv | head -1)" "$(ls "$OUT"/pmc_write/*/*counter_collection.csv | head -1)" "$OUT/pmc_traffic.json" 510 256 > "$OUT/pmc_traffic.txt"
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- $B > /dev/null 2> "$OUT/pmc_sq.err"
python3 tools/pmc_mfma_summary.py "$(ls "$OUT"/pmc_sq/*/*counter_collection.csv | head -1)" "$OUT/pmc_mfma_util.json" > "$OUT/pmc_mfma_util.txt"
rm -rf "$OUT"/stats "$OUT"/pmc_fetch "$OUT"/pmc_write "$OUT"/pmc_sq      # raw traces are large; the summaries stay
echo "profile summaries in $OUT"
