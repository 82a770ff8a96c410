for sh in "655360 320 2560 0" "655360 320 2560 1" "655360 320 320 0" "163840 640 5120 1"; do python tools/igemm_timing.py $sh 2>&1 | grep "^M=\|wave 0" | tail -2; done
