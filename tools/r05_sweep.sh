run() { "$@" python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-side --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['attention_tflops'], d['scores_checksum'])"; }
for i in 1 2; do
echo -n "base            "; run env
echo -n "attn_pipe=12    "; run env DM_ATTN_PIPE=12
echo -n "attn_pipe=10    "; run env DM_ATTN_PIPE=10
echo -n "DEV_KERNARG=1   "; run env HIP_FORCE_DEV_KERNARG=1
echo -n "tap_reuse=2     "; run env DM_TAP_REUSE=2
done
