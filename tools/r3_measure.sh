#!/bin/bash
# (round 3) the measurement batch behind profiles/r03_*: profile set, breakdown bench, C5 kernel mix, exchange-wait bound, batches of 8
set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3m; mkdir -p $O
# 1. the round's profile set over bench.py (kernel stats + PMC passes)
bash tools/profile_round.sh r3prof > $O/profile_round.log 2>&1
# 2. bench with every kernel kind timed: roofline_by_kernel + precision_auto
timeout -k 10 200 python3 bench.py --breakdown > $O/bench_breakdown.json 2> $O/bench_breakdown.err
# 3. kernel stats of a short C5 job (rows up to 512 tokens)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5stats -- python3 -m clip_amd.job --config C5 --rows 300000 --topk 250000 --precision bf16 > $O/job_C5_300k_under_rocprof.json 2> $O/c5stats.err
cp "$(ls $O/c5stats/*/*kernel_stats.csv | head -1)" $O/c5_kernel_stats.csv; rm -rf $O/c5stats
# 4. upper bound of removing the fused-LayerNorm exchange wait (diagnostic library, results void): interleaved A/B
for i in 1; do
  timeout -k 10 120 python3 bench.py --breakdown --no-cpu-baseline > $O/ab_prod_$i.json 2>/dev/null
  CLIP_AMD_LIB=$PWD/gpurun_tmp/libnosync.so timeout -k 10 120 python3 bench.py --breakdown --no-cpu-baseline > $O/ab_nosync_$i.json 2>/dev/null
done
# 5. the reference's batches of 8 through the nn.Module: fused vs unfused LayerNorm flow
PRECISION=bf16 timeout -k 10 200 python3 tools/bs8_latency.py > $O/bs8_fused.log 2>&1
PRECISION=bf16 CLIP_AMD_FUSED_LN=0 timeout -k 10 200 python3 tools/bs8_latency.py > $O/bs8_unfused.log 2>&1
echo done
