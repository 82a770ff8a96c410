#!/bin/bash
# Regenerate the judged artifacts of a round on the GPU box:  bash tools/profile_round.sh <tag>
# (run through gpurun; outputs land in gpurun_out/<tag>/, copy the summaries into profiles/).
set -u
TAG=${1:-r06_end}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/bench.err
DM_PROF_DUMP=$OUT/shapes_raw.txt python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side --no-parity > /dev/null 2>&1
python tools/prof_shapes.py $OUT/shapes_raw.txt 3 > $OUT/shapes.txt; rm -f $OUT/shapes_raw.txt
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side \
    > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    rocprofv3 --pmc $set --kernel-trace -f csv -d $OUT/pmc_$i -o pmc -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-side \
        > $OUT/pmc_$i.json 2> $OUT/pmc_$i.err
    i=$((i+1))
done
python tools/pmc_to_json.py $OUT 2 > $OUT/pmc.json    # a pass runs 2 steps: 1 warm-up + 1 timed (--no-side: no grid-D2H leg since r06)
# the graded line AFTER the counters: bench.py takes roofline.traffic from profiles/<tag>_pmc.json (the same library, the same box)
mkdir -p profiles && cp $OUT/pmc.json profiles/${TAG}_pmc.json      # (bench.py reads the newest tag it knows first: hbm_traffic_per_launch)
python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2>> $OUT/bench.err
# per-shape HBM traffic of the igemm family (cold caches per launch)
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/pmcs_fetch -o pmc -- python tools/pmc_shapes.py run > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $OUT/pmcs_write -o pmc -- python tools/pmc_shapes.py run > /dev/null 2>&1
python tools/pmc_shapes.py parse $OUT/pmcs_fetch $OUT/pmcs_write > $OUT/pmc_shapes.txt 2> $OUT/pmc_shapes.err
rm -rf $OUT/pmcs_fetch $OUT/pmcs_write
# the side workloads (BASELINE configs[3], [4]) with the same keys as the graded line, and the cost of the live events
for w in dift xray; do python bench.py --workload $w --steps 5 --warmup 2 >> $OUT/side_workloads.jsonl 2>> $OUT/bench.err; done
python bench.py --workload dift --dift-dtype f16 --steps 5 --warmup 2 --no-cpu-baseline >> $OUT/side_workloads.jsonl 2>> $OUT/bench.err
# configs[3] in the reference's fp32 (the fp32 net): per-shape table and rocprofv3 kernel stats of the same command
DM_PROF_DUMP=$OUT/dift32_raw.txt python bench.py --workload dift --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python tools/prof_shapes.py $OUT/dift32_raw.txt 3 157.3e12 > $OUT/dift_f32_shapes.txt; rm -f $OUT/dift32_raw.txt
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats32 -o stats -- python bench.py --workload dift --steps 5 --warmup 2 --no-cpu-baseline \
    > $OUT/dift_f32_bench_under_rocprof.json 2>> $OUT/rocprof.err
find $OUT/stats32 -name '*kernel_stats.csv' -exec cp {} $OUT/dift_f32_kernel_stats.csv \; ; rm -rf $OUT/stats32
python tools/t_deviation_gpu.py 16 > $OUT/T_deviation_baseline_size_fp32.txt 2>> $OUT/bench.err
# PMC passes of the fp32 DIFT run (gemm32 / attn32): the same four counter sets, folded per kernel family
mkdir -p $OUT/d32
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    rocprofv3 --pmc $set --kernel-trace -f csv -d $OUT/d32/pmc_$i -o pmc -- python bench.py --workload dift --steps 1 --warmup 1 --no-cpu-baseline \
        > /dev/null 2> $OUT/d32/pmc_$i.err
    i=$((i+1))
done
python tools/pmc_to_json.py $OUT/d32 2 "bench.py --workload dift" > $OUT/dift_f32_pmc.json; rm -rf $OUT/d32
for w in vae pixels; do python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline >> $OUT/side_workloads.jsonl 2>> $OUT/bench.err; done
DM_BENCH_NOPROF=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side > $OUT/bench_noprof.json 2>> $OUT/bench.err
# r06: the 3x3-convolution launches' SQ counters (which unit saturates), the operand-mix probe, the head_dim-160 scan
bash tools/conv_pmc.sh $OUT/conv_pmc > /dev/null 2>&1; cp $OUT/conv_pmc/conv_sq.txt $OUT/conv_sq_counters.txt 2>/dev/null
[ -x tools/probes/bin/probe_mix ] && ./tools/probes/bin/probe_mix > $OUT/probe_mix.txt 2>&1
python tools/attn_d160_scan.py > $OUT/attn_d160_scan.txt 2>&1
DM_BENCH_ITERS=10 python tools/ab_attn.py attn_pipe 9 1 > $OUT/ab_attn_r04_vs_r06.txt 2>&1
# the X-ray step with the r04 attention dispatch (attn_pipe = 9) and the current one (three anti-phase wave sets from 8192 keys), alternating
for i in 1 2; do for a in 9 1; do echo -n "attn_pipe=$a " >> $OUT/ab_xray_attn.txt; DM_ATTN_PIPE=$a python bench.py --workload xray --steps 5 --warmup 2 --no-cpu-baseline 2>> $OUT/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['attention_tflops'])" >> $OUT/ab_xray_attn.txt; done; done
find $OUT/stats -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
# raw traces are large; keep only the summaries
rm -rf $OUT/pmc_[0-9] $OUT/stats
ls -la $OUT
cat $OUT/bench.json
