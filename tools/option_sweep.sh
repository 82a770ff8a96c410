line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%9.3f ms/step  igemm %7.2f TF/s  attn %6.2f' % (d['ms_per_step'], d['roofline']['achieved'], d['roofline']['attention_tflops']))"; }
for i in 1 2; do
for v in "" "DM_UP_FOLD=0" "DM_TAP_REUSE=2" "DM_GN_FOLD=0" "DM_LN_INKERNEL=2" "DM_LN_INKERNEL=0" "DM_TAP_REUSE=0"; do
  echo -n "[$v] "; env $v python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-parity 2>/dev/null | line
done; done
