#!/bin/bash
# r06: the ascending and the snake MFMA order on the bench step in ABBA order, with the clock / power sample of the line beside each run.
#     python tools/build_variant.py nosnake -DDM_MFMA_SNAKE=0 && gpurun -- 'bash tools/ab_snake_clock.sh'      -> profiles/r06_ab_mfma_snake.txt (run 3)
A=diff-mining_amd/lib/libdm_engine_nosnake.so; B=diff-mining_amd/lib/libdm_engine.so
run() { echo -n "$1: "; DM_ENGINE_LIB=$2 python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-side --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%9.3f ms/step  sclk %7.1f MHz (min %6.1f max %6.1f)  power %7.1f W (max %6.1f)  samples %d' % (d['ms_per_step'], r['sclk_mhz_mean'], r['sclk_mhz_min'], r['sclk_mhz_max'], r['power_w_mean'], r['power_w_max'], r['clock_samples']))"; }
for i in 1 2 3; do run "ascending" $A; run "snake    " $B; run "snake    " $B; run "ascending" $A; done
