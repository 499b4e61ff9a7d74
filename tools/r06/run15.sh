#!/bin/bash
# round 6, GPU run 15: the straddling pair of every skirt shift by one v_pk_mov_b32 (LB_SKIRT_PKMOV): bitwise checks (k_step5, k_deep<6>, <7>,
# k_deep2), A/B against the library before it
set -u
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_pkmov_check.txt
: > $out
for f in "" "--six" "--seven" "--deep2"; do
  echo "== step5_check $f" >> $out
  timeout 300 python3 tools/step5_check.py $f --no-time 2>&1 | tail -2 >> $out
done
ROUNDS=3 timeout 900 bash tools/gpu_ab.sh gpurun_out/r06_pkmov_ab.txt 2d-lb_amd/LB_D2Q9/liblbhip_prev.so 2d-lb_amd/LB_D2Q9/liblbhip.so \
  "--bc periodic --n 8192 --steps 84" "--bc periodic --n 4096 --steps 84" "--bc pipe --n 8192 --steps 84" "--bc pipe --tiff --n 4096 --steps 84" \
  "--bc pipe --cyl --n 3751 --ny 1251 --steps 140" "--bc cavity --n 2048 --steps 200" > /dev/null 2>&1
cat $out
cut -c1-45,170-260 gpurun_out/r06_pkmov_ab.txt.sorted
