#!/bin/bash
# sample power / clocks while the bench runs
python bench.py --steps 250 --warmup 3 --no-cpu-baseline > /tmp/bench_out.json 2>/dev/null &
BP=$!
sleep 28
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|GPU use|fclk" | tr '\n' ' '; echo
  sleep 0.7
done
wait $BP
python -c "import json; d=json.load(open('/tmp/bench_out.json')); print(d['value'], d['ms_per_step'])"
rocm-smi --showmaxpower 2>/dev/null | grep -i power
