#!/bin/bash
# Round 6: every GPU run of the round as one function each (run1 ... run63), in the order they were made; the header comment of a
# function says what it measured, the outputs it names under gpurun_out/ were copied to profiles/ (profiles/r06_experiments.txt cites them).
# Usage (through gpurun, from the repository root):   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/r06/runs.sh 50'
set -u
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
: ${GRAFT_REPO_ROOT:=$(pwd)}
export GRAFT_REPO_ROOT

# round 6, GPU run 1: the relaxation with omega folded into the density (LB_RELAX_FOLD): full GPU suite + A/B against round 5's library
run1() {
timeout 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r06_run1_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_run1_pytest.log
ROUNDS=2 timeout 600 bash tools/gpu_ab.sh gpurun_out/r06_fold_ab.txt 2d-lb_amd/LB_D2Q9/liblbhip_r05.so 2d-lb_amd/LB_D2Q9/liblbhip.so \
  "--bc periodic --n 8192 --steps 84" "--bc periodic --n 4096 --steps 84" "--bc pipe --n 8192 --steps 84" "--bc cavity --n 4096 --steps 84" \
  "--bc pipe --tiff --n 4096 --steps 84" "--bc pipe --cyl --n 3751 --ny 1251 --steps 140" "--bc cavity --n 1024 --steps 400" \
  "--bc periodic --n 8192 --ny 1024 --steps 84" > /dev/null 2>&1
tail -5 gpurun_out/r06_run1_pytest.log
cat gpurun_out/r06_fold_ab.txt.sorted
}

# round 6, GPU run 2: thick edge bands (band_extra), the edge stream at normal priority: full GPU suite, slab proxy A/B, slab stress under contention
run2() {
timeout 1100 python3 -m pytest tests -m gpu -q > gpurun_out/r06_run2_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_run2_pytest.log
P=gpurun_out/r06_slab_proxy_bands.txt
: > $P
for rep in 1 2; do
  echo "== bands as in round 5 (LB_BAND_EXTRA=0)" >> $P
  LB_BAND_EXTRA=0 timeout 300 python3 tools/slab_proxy.py --parts 8,4,2 --steps 140 --variants -1 --transports rccl,peer >> $P 2>&1
  echo "== thick bands (default slack 8)" >> $P
  timeout 300 python3 tools/slab_proxy.py --parts 8,4,2 --steps 140 --variants -1 --transports rccl,peer >> $P 2>&1
done
for sl in 2 5 12 16; do
  echo "== thick bands, LB_BAND_SLACK=$sl" >> $P
  LB_BAND_SLACK=$sl timeout 200 python3 tools/slab_proxy.py --parts 8,4 --steps 140 --variants -1 --transports rccl >> $P 2>&1
done
timeout 420 python3 tools/slab_stress.py 80 3 > gpurun_out/r06_slab_stress_normal_prio.txt 2>&1
echo "stress rc=$?" >> gpurun_out/r06_slab_stress_normal_prio.txt
tail -6 gpurun_out/r06_run2_pytest.log
cat $P
tail -5 gpurun_out/r06_slab_stress_normal_prio.txt
}

# round 6, GPU run 3: k_deep's mask-free march per workgroup (A/B through LB_MASK_CLEAN_PATH), ABI 9 (slab cycle tuning, exchange timing):
# full GPU suite, the masked cases with and without the clean path, bench.py through the slab path on one GPU (both transports)
run3() {
timeout 1100 python3 -m pytest tests -m gpu -q > gpurun_out/r06_run3_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_run3_pytest.log
L=2d-lb_amd/LB_D2Q9/liblbhip.so
AB_ENV_A="LB_MASK_CLEAN_PATH=0" ROUNDS=2 timeout 600 bash tools/gpu_ab.sh gpurun_out/r06_mask_clean_ab.txt $L $L \
  "--bc pipe --cyl --n 3751 --ny 1251 --steps 140" "--bc pipe --cyl --n 4096 --steps 84" "--bc pipe --tiff --n 4096 --steps 84" \
  "--bc periodic --mask --n 8192 --steps 84" "--bc cavity --cyl --n 6144 --steps 84" > /dev/null 2>&1
for t in rccl peer; do
  timeout 300 python3 bench.py --force-slab-path --transport $t --steps 56 --warmup 14 --no-cpu-baseline > gpurun_out/r06_bench_slabpath_$t.json 2> gpurun_out/r06_bench_slabpath_$t.err
  echo "bench $t rc=$?" >> gpurun_out/r06_run3_pytest.log
done
tail -8 gpurun_out/r06_run3_pytest.log
cat gpurun_out/r06_mask_clean_ab.txt.sorted
python3 - <<'PY'
import json
for t in ("rccl","peer"):
    try:
        d=json.loads(open("gpurun_out/r06_bench_slabpath_%s.json"%t).read().strip().splitlines()[-1])
        print(t, d["value"], d["roofline"]["kernel"][:40], d.get("slabs"))
    except Exception as e:
        print(t, "no line:", e); print(open("gpurun_out/r06_bench_slabpath_%s.err"%t).read()[-1500:])
PY
}

# round 6, GPU run 4: the row in flight at a[0:42] (a wave takes ~280 registers, not a SIMD's whole file), split edge bands with the
# exchange on the communication stream: full GPU suite, slab proxy (split on / off, both transports), plain-grid A/B against round 5
run4() {
timeout 1100 python3 -m pytest tests -m gpu -q > gpurun_out/r06_run4_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_run4_pytest.log
P=gpurun_out/r06_slab_proxy_split.txt
: > $P
for rep in 1 2; do
  echo "== split bands (default)" >> $P
  timeout 300 python3 tools/slab_proxy.py --parts 8,4,2 --steps 140 --variants -1 --transports rccl,peer 2>&1 | grep grid >> $P
  echo "== one launch per band (LB_SPLIT_BANDS=0)" >> $P
  LB_SPLIT_BANDS=0 timeout 300 python3 tools/slab_proxy.py --parts 8,4,2 --steps 140 --variants -1 --transports rccl,peer 2>&1 | grep grid >> $P
done
echo "== round-5 bands (LB_BAND_EXTRA=0), split schedule" >> $P
LB_BAND_EXTRA=0 timeout 300 python3 tools/slab_proxy.py --parts 8,4,2 --steps 140 --variants -1 --transports rccl,peer 2>&1 | grep grid >> $P
for sl in 0 6 10; do
  echo "== split bands, LB_BAND_SLACK=$sl" >> $P
  LB_BAND_SLACK=$sl timeout 200 python3 tools/slab_proxy.py --parts 8,4 --steps 140 --variants -1 --transports rccl,peer 2>&1 | grep grid >> $P
done
ROUNDS=2 timeout 400 bash tools/gpu_ab.sh gpurun_out/r06_agpr_low_ab.txt 2d-lb_amd/LB_D2Q9/liblbhip_r05.so 2d-lb_amd/LB_D2Q9/liblbhip.so \
  "--bc periodic --n 8192 --steps 84" "--bc periodic --n 4096 --steps 84" "--bc pipe --n 8192 --steps 84" "--bc pipe --tiff --n 4096 --steps 84" > /dev/null 2>&1
for t in rccl peer; do
  timeout 300 python3 bench.py --force-slab-path --transport $t --steps 56 --warmup 14 --no-cpu-baseline > gpurun_out/r06_bench_slabpath_$t.json 2> gpurun_out/r06_bench_slabpath_$t.err
  echo "bench $t rc=$?" >> gpurun_out/r06_run4_pytest.log
done
tail -6 gpurun_out/r06_run4_pytest.log
cat $P
cut -c1-60,190-260 gpurun_out/r06_agpr_low_ab.txt.sorted
python3 - <<'PY'
import json
for t in ("rccl","peer"):
    try:
        d=json.loads(open("gpurun_out/r06_bench_slabpath_%s.json"%t).read().strip().splitlines()[-1])
        print(t, d["value"], d["roofline"]["kernel"][:40], d.get("slabs",{}).get("per_rank"), d.get("slabs",{}).get("cycle_tuning"))
    except Exception as e:
        print(t, "no line:", e); print(open("gpurun_out/r06_bench_slabpath_%s.err"%t).read()[-1500:])
PY
}

# round 6, GPU run 5: new tests (planar 8192^2 under k_deep, lb_set_slab_cycle / exchange timing), the two-waves-per-SIMD probe, round 5's
# RW = 2 miscompare, slab proxy with exchange times (split under RCCL, one launch per band under the peer transport)
run5() {
timeout 900 python3 -m pytest tests -m gpu -q -k "planar_layout_8192 or slab_cycle_depth or self_ring or peer_transport or slab_schedule" > gpurun_out/r06_run5_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_run5_pytest.log
timeout 400 bash tools/r06/occ2_probe.sh > /dev/null 2>&1
timeout 700 bash tools/r06/rw2_check.sh > /dev/null 2>&1
P=gpurun_out/r06_slab_proxy_policy.txt
: > $P
for rep in 1 2; do
  echo "== default policy" >> $P
  timeout 400 python3 tools/slab_proxy.py --parts 8,4,2 --steps 140 --variants -1 --transports rccl,peer --reps 5 2>&1 | grep grid >> $P
done
echo "== round-5 bands (LB_BAND_EXTRA=0)" >> $P
LB_BAND_EXTRA=0 timeout 400 python3 tools/slab_proxy.py --parts 8,4,2 --steps 140 --variants -1 --transports rccl,peer --reps 5 2>&1 | grep grid >> $P
echo "== RCCL, split, slack 12" >> $P
LB_BAND_SLACK=12 timeout 300 python3 tools/slab_proxy.py --parts 8,4 --steps 140 --variants -1 --transports rccl --reps 5 2>&1 | grep grid >> $P
tail -5 gpurun_out/r06_run5_pytest.log
cat gpurun_out/r06_occ2_probe.txt
cat gpurun_out/r06_rw2_check.txt | tail -40
cat $P
}

# round 6, GPU run 6: k_deep2 (two waves per strip and direction, two waves per SIMD, the row gathered ahead loaded straight into LDS):
# bitwise against the single-step kernel in every family, then timed against k_deep<6> / k_deep<7>
run6() {
timeout 600 python3 tools/step5_check.py --deep2 --sizes 8192,4096 > gpurun_out/r06_deep2_check.txt 2>&1
echo "rc=$?" >> gpurun_out/r06_deep2_check.txt
cat gpurun_out/r06_deep2_check.txt | tail -45
}

# round 6, GPU run 7: k_deep2 with bare barriers (no memory drain) and the roles interleaved over the two workgroups of a CU (against: not interleaved)
run7() {
timeout 600 python3 tools/step5_check.py --deep2 --sizes 8192,4096 > gpurun_out/r06_deep2_check2.txt 2>&1
echo "rc=$?" >> gpurun_out/r06_deep2_check2.txt
echo "== roles not interleaved (LB_DEEP2_SWAP=0)" >> gpurun_out/r06_deep2_check2.txt
LB_LIB=$PWD/2d-lb_amd/LB_D2Q9/liblbhip_d2ns.so timeout 600 python3 tools/step5_check.py --deep2 --sizes 8192,4096 >> gpurun_out/r06_deep2_check2.txt 2>&1
echo "rc=$?" >> gpurun_out/r06_deep2_check2.txt
grep -v "^checked" gpurun_out/r06_deep2_check2.txt | tail -60
}

# round 6, GPU run 8: where k_deep2<7>'s launch goes -- diagnostic build, timing only: everything / no stores / no loads / no global memory /
# neither memory nor arithmetic; k_deep<7> beside it.  Microseconds per launch, 8192^2 periodic.
run8() {
out=gpurun_out/r06_deep2_ablate.txt
: > $out
L=$PWD/2d-lb_amd/LB_D2Q9/liblbhip_diag.so
for rep in 1 2; do
for v in 53601 119137; do
  for diag in 0 4194304 8388608 12582912 12582913 1; do
    r=$(LB_LIB=$L LB_DIAG=$diag python3 tools/run_case.py --bc periodic --n 8192 --steps 70 --variant $v --repeat 3 2>&1 | tail -1)
    us=$(echo "$r" | sed -n 's/.* \([0-9.]*\) us per step.*/\1/p')
    echo "variant $v LB_DIAG=$diag: launch $(python3 -c "print('%.1f' % (7*float('${us:-0}')))") us  [$(echo "$r" | cut -c1-70)]" >> $out
  done
done
done
cat $out
}

# round 6, GPU run 9: full GPU suite on the library with k_deep2 (its variants are in the test matrices now), k_deep2 re-checked after the
# M0 save / restore
run9() {
timeout 300 python3 tools/step5_check.py --deep2 --no-time > gpurun_out/r06_deep2_check3.txt 2>&1
echo "rc=$?" >> gpurun_out/r06_deep2_check3.txt
timeout 1500 python3 -m pytest tests -m gpu -q > gpurun_out/r06_run9_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_run9_pytest.log
tail -3 gpurun_out/r06_deep2_check3.txt
tail -15 gpurun_out/r06_run9_pytest.log
}

# round 6, GPU run 10: the lines and the profiles of the final library -- bench.py (default command and the driver's), rocprofv3 kernel
# trace + FETCH_SIZE / WRITE_SIZE passes of configurations 4, 5, 3, 2 with the tuner's choice pinned (tools/gpu_profile.sh)
run10() {
timeout 600 python3 bench.py > gpurun_out/r06a_bench_default.json 2> gpurun_out/r06a_bench_default.err
echo "bench default rc=$?"
timeout 400 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r06a_bench_steps20.json 2> gpurun_out/r06a_bench_steps20.err
echo "bench steps20 rc=$?"
timeout 500 bash tools/gpu_profile.sh r06c4 > gpurun_out/r06_profile_c4.log 2>&1
timeout 500 bash tools/gpu_profile.sh r06c5 --config 5 > gpurun_out/r06_profile_c5.log 2>&1
timeout 500 bash tools/gpu_profile.sh r06c3 --config 3 > gpurun_out/r06_profile_c3.log 2>&1
timeout 500 bash tools/gpu_profile.sh r06c2 --config 2 > gpurun_out/r06_profile_c2.log 2>&1
python3 - <<'PY'
import json
for f in ("gpurun_out/r06a_bench_default.json","gpurun_out/r06a_bench_steps20.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d["roofline"]
        print(f, d["value"], "MLUPS; launch", r["launch_ms"], "frac", r["frac"], "plain", r.get("frac_plain_launch"), "plan", r.get("block_plan"), "six", (r.get("six_step_kernel") or {}).get("MLUPS"))
        for o in d.get("other_configs",[]): print("   ", o.get("config"), o.get("path",""), o.get("value"), o.get("roofline_frac"), str(o.get("kernel"))[:50], o.get("error",""))
        print("    cpu:", {k:v for k,v in (d.get("cpu_baseline") or {}).items() if k in ("value","cores","kind")})
    except Exception as e:
        print(f, "no line", e)
PY
ls gpurun_out/prof_r06c*/ | head -30
}

# round 6, GPU run 11: full GPU suite on the final library (launchers report an unsupported family), SQ counters of k_deep<7> 8192^2,
# kernel timeline of the slab cycle (one of eight slabs, both transports)
run11() {
timeout 1500 python3 -m pytest tests -m gpu -q > gpurun_out/r06_run11_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_run11_pytest.log
timeout 600 bash tools/gpu_pmc_case.sh r06deep7 --bc periodic --n 8192 --steps 140 > gpurun_out/r06_sq_deep7.txt 2>&1
for t in rccl peer; do
  (cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_$t -- python3 $GRAFT_REPO_ROOT/tools/slab_proxy.py --parts 8 --steps 56 --variants -1 --transports $t --reps 1 > $GRAFT_REPO_ROOT/gpurun_out/tl_$t.log 2>&1)
  python3 tools/timeline.py gpurun_out/tl_$t 40 > gpurun_out/r06_slab_timeline_$t.txt 2>&1
done
tail -4 gpurun_out/r06_run11_pytest.log
grep "k_deep<1, false, false, 7" gpurun_out/r06_sq_deep7.txt | head -40
head -45 gpurun_out/r06_slab_timeline_rccl.txt
}

# round 6, GPU run 12: soak of the final library -- automatic choice against the single-step kernel, bit for bit, alone and beside a second
# process streaming an 8192^2 lattice; the same with k_deep2 forced (LB_VARIANT=119137)
run12() {
out=gpurun_out/r06_soak.txt
echo "== alone" > $out
timeout 900 python3 tools/soak_bitwise.py >> $out 2>&1
echo "== beside a second process streaming an 8192^2 lattice" >> $out
python3 - <<'PY' &
import os, sys
sys.path[:0] = [os.path.join(os.environ["GRAFT_REPO_ROOT"], "2d-lb_amd"), os.environ["GRAFT_REPO_ROOT"]]
from LB_D2Q9.simulation import Simulation
from bench import shear_layer
s = Simulation(8192, 8192, 1.7, bc="periodic"); s.init_equilibrium(*shear_layer(8192, 8192, 0, 8192))
import time
t0 = time.time()
while time.time() - t0 < 420: s.run(140)
PY
NOISE=$!
sleep 20
timeout 600 python3 tools/soak_bitwise.py >> $out 2>&1
kill $NOISE 2>/dev/null; wait $NOISE 2>/dev/null
echo "== k_deep2 forced for the seven-step launches (LB_SOAK_VARIANT=119137)" >> $out
LB_SOAK_VARIANT=119137 timeout 600 python3 tools/soak_bitwise.py >> $out 2>&1
cat $out
}

# round 6, GPU run 13: what k_deep2's two barriers per row cost -- diagnostic build, timing only (races): steady state without barriers
run13() {
out=gpurun_out/r06_deep2_nobarrier.txt
: > $out
L=$PWD/2d-lb_amd/LB_D2Q9/liblbhip_diag.so
for rep in 1 2; do
  for diag in 0 16777216 12582912 29360128 29360129; do
    r=$(LB_LIB=$L LB_DIAG=$diag python3 tools/run_case.py --bc periodic --n 8192 --steps 70 --variant 119137 --repeat 3 2>&1 | tail -1)
    us=$(echo "$r" | sed -n 's/.* \([0-9.]*\) us per step.*/\1/p')
    echo "k_deep2<7> LB_DIAG=$diag: launch $(python3 -c "print('%.1f' % (7*float('${us:-0}')))") us" >> $out
  done
done
cat $out
}

# round 6, GPU run 14: where the waves of the occupancy probe's one-wave-per-SIMD build sat (k_deep<4> behind the six-step launcher, four
# waves per CU) -- per-wave records of one launch (LB_DIAG bit 12), against the product's k_deep<6>
run14() {
out=gpurun_out/r06_occ2_placement.txt
echo "== k_deep<4> behind the six-step launcher, __launch_bounds__(128, 1), four waves per CU (liblbhip_p4a.so)" > $out
LB_TIMELINE_LIB=$PWD/2d-lb_amd/LB_D2Q9/liblbhip_p4a.so LB_TIMELINE_DEPTH=6 timeout 200 python3 tools/wave_timeline.py 8192 4 2>&1 | grep -E "SIMD|wave slot|launch span|residency" >> $out
echo "== k_deep<6> of the diagnostic build, four waves per CU" >> $out
LB_TIMELINE_DEPTH=6 timeout 200 python3 tools/wave_timeline.py 8192 4 2>&1 | grep -E "SIMD|wave slot|launch span|residency" >> $out
cat $out
}

# round 6, GPU run 15: the straddling pair of every skirt shift by one v_pk_mov_b32 (LB_SKIRT_PKMOV): bitwise checks (k_step5, k_deep<6>, <7>,
# k_deep2), A/B against the library before it
run15() {
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
}

# round 6, GPU run 16: more of the randomised checks on the final library -- 150 seeds of tests/test_gpu_random.py (kernel variants incl.
# k_deep2, random slab partitions through lb_run_group with thick bands, RCCL self-rings with split bands), tools/slab_stress.py with 200
# random partitions beside three noise processes, four rank processes on one GPU over the peer transport
run16() {
LB_RANDOM_SEEDS=150 timeout 1500 python3 -m pytest tests/test_gpu_random.py -m gpu -q > gpurun_out/r06_random150.txt 2>&1
echo "rc=$?" >> gpurun_out/r06_random150.txt
timeout 900 python3 tools/slab_stress.py 200 3 > gpurun_out/r06_slab_stress_200.txt 2>&1
echo "rc=$?" >> gpurun_out/r06_slab_stress_200.txt
timeout 600 python3 tools/peer_ranks_check.py --ranks 4 > gpurun_out/r06_peer_ranks4.txt 2>&1
echo "rc=$?" >> gpurun_out/r06_peer_ranks4.txt
tail -4 gpurun_out/r06_random150.txt; tail -3 gpurun_out/r06_slab_stress_200.txt; tail -5 gpurun_out/r06_peer_ranks4.txt
}

# round 6, GPU run 17: the halo cycle replayed from a captured hipGraph (LB_CYCLE_GRAPH=1, peer transport) against eager launches
run17() {
P=gpurun_out/r06_slab_proxy_graph.txt
: > $P
for rep in 1 2 3; do
  echo "== eager" >> $P
  timeout 300 python3 tools/slab_proxy.py --parts 8,4 --steps 280 --variants -1 --transports peer --reps 5 2>&1 | grep grid >> $P
  echo "== LB_CYCLE_GRAPH=1" >> $P
  LB_CYCLE_GRAPH=1 timeout 300 python3 tools/slab_proxy.py --parts 8,4 --steps 280 --variants -1 --transports peer --reps 5 2>&1 | grep grid >> $P
done
cut -c1-160 $P
}

# round 6, GPU run 18: last sanity pass on the final library -- smoke(), bench.py through the slab path (both transports), the driver's command
run18() {
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_final_smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/r06_final_smoke.txt
for t in rccl peer; do
  timeout 300 python3 bench.py --force-slab-path --transport $t --steps 56 --warmup 14 --no-cpu-baseline > gpurun_out/r06b_bench_slabpath_$t.json 2> gpurun_out/r06b_bench_slabpath_$t.err
  echo "bench $t rc=$?" >> gpurun_out/r06_final_smoke.txt
done
timeout 400 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r06b_bench_steps20.json 2> gpurun_out/r06b_bench_steps20.err
echo "bench steps20 rc=$?" >> gpurun_out/r06_final_smoke.txt
cat gpurun_out/r06_final_smoke.txt
python3 - <<'PY'
import json
for f in ("r06b_bench_slabpath_rccl","r06b_bench_slabpath_peer","r06b_bench_steps20"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d["roofline"].get("frac_plain_launch"), (d.get("slabs") or {}).get("per_rank"), [ (o.get("config"), o.get("value")) for o in d.get("other_configs",[])])
    except Exception as e:
        print(f, "no line", e)
PY
}

# round 6, GPU run 19: peer transport, bands in one launch against split bands, ONE box, parts 1 (the whole grid as one slab), 2, 4, 8
run19() {
P=gpurun_out/r06_slab_proxy_peer_split_ab.txt
: > $P
for rep in 1 2 3; do
  for sp in 0 1; do
    echo "== LB_SPLIT_BANDS=$sp" >> $P
    LB_SPLIT_BANDS=$sp timeout 400 python3 tools/slab_proxy.py --parts 1,2,4,8 --steps 140 --variants -1 --transports peer --reps 5 2>&1 | grep grid | cut -c1-140 >> $P
  done
done
cat $P
}

# round 6, GPU run 20: split bands for both transports (the default now): full GPU suite, slab proxy both transports, bench over the slab path
run20() {
timeout 1300 python3 -m pytest tests -m gpu -q > gpurun_out/r06_run20_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_run20_pytest.log
P=gpurun_out/r06c_slab_proxy_final.txt
: > $P
for rep in 1 2; do
  timeout 500 python3 tools/slab_proxy.py --parts 1,2,4,8 --steps 140 --variants -1 --transports rccl,peer --reps 5 2>&1 | grep grid >> $P
done
for t in rccl peer; do
  timeout 300 python3 bench.py --force-slab-path --transport $t --steps 56 --warmup 14 --no-cpu-baseline > gpurun_out/r06c_bench_slabpath_$t.json 2> gpurun_out/r06c_bench_slabpath_$t.err
  echo "bench $t rc=$?" >> gpurun_out/r06_run20_pytest.log
done
timeout 600 python3 tools/peer_ranks_check.py --ranks 4 > gpurun_out/r06c_peer_ranks4.txt 2>&1
echo "peer ranks rc=$?" >> gpurun_out/r06_run20_pytest.log
tail -5 gpurun_out/r06_run20_pytest.log
cut -c1-150 $P
tail -5 gpurun_out/r06c_peer_ranks4.txt
python3 - <<'PY'
import json
for t in ("rccl","peer"):
    try:
        d=json.loads(open("gpurun_out/r06c_bench_slabpath_%s.json"%t).read().strip().splitlines()[-1])
        print(t, d["value"], d.get("slabs",{}).get("per_rank"))
    except Exception as e:
        print(t, "no line:", e)
PY
}

# round 6, GPU run 21: kernel timelines of the slab cycle under RCCL at 4 and 2 slabs (where RCCL trails the peer transport)
run21() {
for parts in 4 2; do
  (cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_rccl_$parts -- python3 $GRAFT_REPO_ROOT/tools/slab_proxy.py --parts $parts --steps 56 --variants -1 --transports rccl --reps 1 > $GRAFT_REPO_ROOT/gpurun_out/tl_rccl_$parts.log 2>&1)
  python3 tools/timeline.py gpurun_out/tl_rccl_$parts 44 > gpurun_out/r06c_slab_timeline_rccl_$parts.txt 2>&1
  rm -rf gpurun_out/tl_rccl_$parts
done
cut -c1-170 gpurun_out/r06c_slab_timeline_rccl_4.txt
}

# round 6, GPU run 22: which RCCL the bench and the proxy load and how many workgroups its send / receive kernel takes; proxy at 4 and 2
# slabs with the channel count capped
run22() {
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_bench -- python3 $GRAFT_REPO_ROOT/bench.py --force-slab-path --transport rccl --steps 56 --warmup 14 --no-cpu-baseline --min-blocks 3 > $GRAFT_REPO_ROOT/gpurun_out/tl_bench.log 2>&1)
python3 tools/timeline.py gpurun_out/tl_bench 400 | grep -v "k_deep" | tail -12 > gpurun_out/r06c_bench_rccl_kernels.txt 2>&1
rm -rf gpurun_out/tl_bench
python3 - > gpurun_out/r06c_rccl_libs.txt 2>&1 <<'PY'
import os, sys
sys.path[:0] = ["2d-lb_amd", "."]
import torch
print("after import torch:", [l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l][:1])
from LB_D2Q9.simulation import Simulation, comm_unique_id
s = Simulation(512, 512, 1.7, bc="periodic", halo=True)
s.comm_init(comm_unique_id(), 0, 1)
print("after lb_comm_init:", sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l)))
PY
P=gpurun_out/r06c_slab_proxy_channels.txt
: > $P
for ch in default 2 4 8 16; do
  echo "== NCCL_MAX_NCHANNELS=$ch" >> $P
  if [ $ch = default ]; then
    timeout 300 python3 tools/slab_proxy.py --parts 4,2 --steps 140 --variants -1 --transports rccl --reps 4 2>&1 | grep grid >> $P
  else
    NCCL_MAX_NCHANNELS=$ch NCCL_MIN_NCHANNELS=1 timeout 300 python3 tools/slab_proxy.py --parts 4,2 --steps 140 --variants -1 --transports rccl --reps 4 2>&1 | grep grid >> $P
  fi
done
cat gpurun_out/r06c_bench_rccl_kernels.txt | cut -c1-170
cat gpurun_out/r06c_rccl_libs.txt
cut -c1-150,230-330 $P
}

# round 6, GPU run 23: the halo communicator capped at 8 channels through ncclConfig_t::maxCTAs (does this RCCL honour it?): the send /
# receive kernel's workgroups in the timeline, proxy capped | uncapped | other caps, bench over the slab path, the RCCL tests
run23() {
for cap in 8 0; do
  (cd /tmp && export TMPDIR=/tmp && LB_RCCL_MAX_CTAS=$cap timeout 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_cap$cap -- python3 $GRAFT_REPO_ROOT/tools/slab_proxy.py --parts 4 --steps 56 --variants -1 --transports rccl --reps 1 > $GRAFT_REPO_ROOT/gpurun_out/tl_cap$cap.log 2>&1)
  python3 tools/timeline.py gpurun_out/tl_cap$cap 44 > gpurun_out/r06c_slab_timeline_rccl_4_cap$cap.txt 2>&1
  rm -rf gpurun_out/tl_cap$cap
done
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_bench -- python3 $GRAFT_REPO_ROOT/bench.py --force-slab-path --transport rccl --steps 56 --warmup 14 --no-cpu-baseline --min-blocks 3 > $GRAFT_REPO_ROOT/gpurun_out/tl_bench.log 2>&1)
python3 tools/timeline.py gpurun_out/tl_bench 4000 | grep -i "nccl" | tail -6 > gpurun_out/r06c_bench_rccl_kernels.txt 2>&1
rm -rf gpurun_out/tl_bench
P=gpurun_out/r06c_slab_proxy_maxctas.txt
: > $P
for rep in 1 2; do
for cap in 8 0 4 16; do
  echo "== LB_RCCL_MAX_CTAS=$cap" >> $P
  LB_RCCL_MAX_CTAS=$cap timeout 300 python3 tools/slab_proxy.py --parts 8,4,2,1 --steps 140 --variants -1 --transports rccl --reps 4 2>&1 | grep grid >> $P
done
done
timeout 900 python3 -m pytest tests -m gpu -q -k "rccl or slab or comm or distributed" > gpurun_out/r06_run23_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_run23_pytest.log
timeout 300 python3 bench.py --force-slab-path --transport rccl --steps 56 --warmup 14 --no-cpu-baseline > gpurun_out/r06d_bench_slabpath_rccl.json 2> gpurun_out/r06d_bench_slabpath_rccl.err
grep -i nccl gpurun_out/r06c_slab_timeline_rccl_4_cap8.txt | tail -2 | cut -c1-150
grep -i nccl gpurun_out/r06c_slab_timeline_rccl_4_cap0.txt | tail -2 | cut -c1-150
cut -c1-150 gpurun_out/r06c_bench_rccl_kernels.txt
cut -c1-150,230-330 $P
tail -3 gpurun_out/r06_run23_pytest.log
python3 -c "
import json
d=json.loads(open('gpurun_out/r06d_bench_slabpath_rccl.json').read().strip().splitlines()[-1]); print(d['value'], d['slabs']['per_rank'])"
}

# round 6, GPU run 24: why bench.py's RCCL self-exchange takes 32 us and the proxy's 250: the bench's timeline around an exchange
run24() {
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_bench -- python3 $GRAFT_REPO_ROOT/bench.py --force-slab-path --transport rccl --steps 56 --warmup 14 --no-cpu-baseline --min-blocks 3 > $GRAFT_REPO_ROOT/gpurun_out/tl_bench.log 2>&1)
python3 tools/timeline.py gpurun_out/tl_bench 36 ncclDevKernel > gpurun_out/r06c_bench_timeline_rccl.txt 2>&1
rm -rf gpurun_out/tl_bench
cut -c1-170 gpurun_out/r06c_bench_timeline_rccl.txt
}

# round 6, GPU run 25: a slab handle's three streams probed onto three hardware queues (separate_queues); RCCL's channel count through
# the environment.  bench.py over the slab path (both transports; RCCL uncapped | 8 channels), its timeline around an exchange, the
# proxy at 8 | 4 | 2 slabs with RCCL uncapped | 4 | 8 | 16 channels, the slab tests
run25() {
export LB_QUEUE_PROBE=2
for ch in default 8; do
  for t in rccl peer; do
    if [ $ch = default ]; then unset NCCL_MAX_NCHANNELS; else export NCCL_MAX_NCHANNELS=$ch; fi
    timeout 300 python3 bench.py --force-slab-path --transport $t --steps 56 --warmup 14 --no-cpu-baseline > gpurun_out/r06e_bench_slabpath_${t}_ch$ch.json 2> gpurun_out/r06e_bench_slabpath_${t}_ch$ch.err
  done
done
export NCCL_MAX_NCHANNELS=8
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_bench -- python3 $GRAFT_REPO_ROOT/bench.py --force-slab-path --transport rccl --steps 56 --warmup 14 --no-cpu-baseline --min-blocks 3 > $GRAFT_REPO_ROOT/gpurun_out/tl_bench.log 2>&1)
python3 tools/timeline.py gpurun_out/tl_bench 30 GenericKernel > gpurun_out/r06e_bench_timeline_rccl.txt 2>&1
rm -rf gpurun_out/tl_bench
unset NCCL_MAX_NCHANNELS
P=gpurun_out/r06e_slab_proxy_channels.txt
: > $P
for rep in 1 2; do
for ch in default 4 8 16; do
  echo "== NCCL_MAX_NCHANNELS=$ch" >> $P
  if [ $ch = default ]; then unset NCCL_MAX_NCHANNELS; else export NCCL_MAX_NCHANNELS=$ch; fi
  timeout 300 python3 tools/slab_proxy.py --parts 8,4,2 --steps 140 --variants -1 --transports rccl --reps 4 2>&1 | grep "grid\|lb_create" >> $P
done
done
unset NCCL_MAX_NCHANNELS
timeout 1200 python3 -m pytest tests -m gpu -q -k "slab or rccl or peer or distributed or random or halo" > gpurun_out/r06_run25_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_run25_pytest.log
for f in gpurun_out/r06e_bench_slabpath_*.json; do
  python3 - $f <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f[31:-5], d["value"], d["slabs"]["per_rank"][0]["exchange_ms_mean"])
except Exception as e:
    print(f, "no line", e)
PY
  grep -h lb_create ${f%.json}.err | sort | uniq -c
done
cut -c1-160 gpurun_out/r06e_bench_timeline_rccl.txt
cut -c1-150,230-330 $P
tail -3 gpurun_out/r06_run25_pytest.log
}

# round 6, GPU run 26: bench.py over the slab path with more hardware queues per process (GPU_MAX_HW_QUEUES): does the communication
# stream get a queue of its own?
run26() {
export LB_QUEUE_PROBE=2
for hq in 8 16; do
  export GPU_MAX_HW_QUEUES=$hq
  for ch in default 8; do
    if [ $ch = default ]; then unset NCCL_MAX_NCHANNELS; else export NCCL_MAX_NCHANNELS=$ch; fi
    for t in rccl peer; do
      timeout 300 python3 bench.py --force-slab-path --transport $t --steps 56 --warmup 14 --no-cpu-baseline > gpurun_out/r06f_bench_slabpath_${t}_ch${ch}_hq$hq.json 2> gpurun_out/r06f_bench_slabpath_${t}_ch${ch}_hq$hq.err
    done
  done
done
export GPU_MAX_HW_QUEUES=8
export NCCL_MAX_NCHANNELS=8
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_bench -- python3 $GRAFT_REPO_ROOT/bench.py --force-slab-path --transport rccl --steps 56 --warmup 14 --no-cpu-baseline --min-blocks 3 > $GRAFT_REPO_ROOT/gpurun_out/tl_bench.log 2>&1)
python3 tools/timeline.py gpurun_out/tl_bench 30 GenericKernel > gpurun_out/r06f_bench_timeline_rccl_hq8.txt 2>&1
rm -rf gpurun_out/tl_bench
for f in gpurun_out/r06f_bench_slabpath_*.json; do
  python3 - $f <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f[31:-5], d["value"], d["slabs"]["per_rank"][0]["exchange_ms_mean"])
except Exception as e:
    print(f, "no line", e)
PY
done
cut -c1-160 gpurun_out/r06f_bench_timeline_rccl_hq8.txt
}

# round 6, GPU run 27: the proxy inside bench.py's process structure (a torch.distributed group first): hardware queues 4 (HIP's default)
# | 8, RCCL's channels uncapped | 8, both transports, 8 | 4 | 2 slabs; two rounds
run27() {
P=gpurun_out/r06f_slab_proxy_queues.txt
: > $P
for rep in 1 2; do
for hq in default 8; do
for ch in default 8; do
  echo "== GPU_MAX_HW_QUEUES=$hq NCCL_MAX_NCHANNELS=$ch" >> $P
  if [ $ch = default ]; then unset NCCL_MAX_NCHANNELS; else export NCCL_MAX_NCHANNELS=$ch; fi
  if [ $hq = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$hq; fi
  timeout 300 python3 tools/slab_proxy.py --torch-dist --parts 8,4,2 --steps 140 --variants -1 --transports rccl,peer --reps 4 2>&1 | grep "grid" >> $P
done
done
done
cut -c1-150,230-330 $P
}

# round 6, GPU run 28: timelines of one slab of eight inside bench.py's process structure: hardware queues 4 | 8, RCCL channels uncapped | 8
run28() {
for hq in default 8; do
for ch in default 8; do
  if [ $ch = default ]; then unset NCCL_MAX_NCHANNELS; else export NCCL_MAX_NCHANNELS=$ch; fi
  if [ $hq = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$hq; fi
  (cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_x -- python3 $GRAFT_REPO_ROOT/tools/slab_proxy.py --torch-dist --parts 8 --steps 56 --variants -1 --transports rccl --reps 1 > $GRAFT_REPO_ROOT/gpurun_out/tl_x.log 2>&1)
  python3 tools/timeline.py gpurun_out/tl_x 30 > gpurun_out/r06f_slab_timeline_rccl_8_hq${hq}_ch${ch}.txt 2>&1
  rm -rf gpurun_out/tl_x
  echo "== hq $hq ch $ch"; cut -c1-150 gpurun_out/r06f_slab_timeline_rccl_8_hq${hq}_ch${ch}.txt
done
done
}

# round 6, GPU run 29: bench.py itself on the slab one of 8 | 4 | 2 ranks would hold (--force-slab-path --slab-rows), hardware queues 4 | 8,
# both transports; and the timeline of the 1024-row case at 4 queues
run29() {
P=gpurun_out/r06g_bench_slab_rows.txt
: > $P
for rep in 1 2; do
for hq in default 8; do
  if [ $hq = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$hq; fi
  for rows in 1024 2048 4096; do
    for t in rccl peer; do
      timeout 200 python3 bench.py --force-slab-path --slab-rows $rows --transport $t --steps 56 --warmup 14 --no-cpu-baseline --no-other-configs > gpurun_out/x.json 2> gpurun_out/x.err
      python3 - $hq $rows $t >> $P <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/x.json").read().strip().splitlines()[-1])
    print("hw queues %-7s rows %5s %-4s: %9.1f MLUPS, exchange %.1f us mean" % (sys.argv[1], sys.argv[2], sys.argv[3], d["value"], 1e3 * d["slabs"]["per_rank"][0]["exchange_ms_mean"]))
except Exception as e:
    print("hw queues %s rows %s %s: no line (%s)" % (sys.argv[1], sys.argv[2], sys.argv[3], e)); print(open("gpurun_out/x.err").read()[-800:])
PY
    done
  done
done
done
unset GPU_MAX_HW_QUEUES
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_bench -- python3 $GRAFT_REPO_ROOT/bench.py --force-slab-path --slab-rows 1024 --transport rccl --steps 56 --warmup 14 --no-cpu-baseline --no-other-configs --min-blocks 3 > $GRAFT_REPO_ROOT/gpurun_out/tl_bench.log 2>&1)
python3 tools/timeline.py gpurun_out/tl_bench 30 GenericKernel > gpurun_out/r06g_bench_timeline_rccl_1024.txt 2>&1
rm -rf gpurun_out/tl_bench gpurun_out/x.json gpurun_out/x.err
cat $P
cut -c1-150 gpurun_out/r06g_bench_timeline_rccl_1024.txt
}

# round 6, GPU run 30: the communication stream at the highest priority (LB_COMM_PRIO=1, a hardware-queue pool of its own): bench.py on the
# slab of 8 | 4 | 2 | 1 ranks in 280-step blocks, both transports, against the default; its timeline at 1024 rows; the slab tests with it
run30() {
P=gpurun_out/r06h_bench_comm_prio.txt
: > $P
for rep in 1 2 3; do
for prio in 0 1; do
  export LB_COMM_PRIO=$prio
  for rows in 1024 2048 4096 8192; do
    for t in rccl peer; do
      timeout 200 python3 bench.py --force-slab-path --slab-rows $rows --transport $t --steps 280 --warmup 28 --no-cpu-baseline --no-other-configs > gpurun_out/x.json 2> gpurun_out/x.err
      python3 - $prio $rows $t >> $P <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/x.json").read().strip().splitlines()[-1])
    print("LB_COMM_PRIO=%s rows %5s %-4s: %9.1f MLUPS, exchange %.1f us mean" % (sys.argv[1], sys.argv[2], sys.argv[3], d["value"], 1e3 * d["slabs"]["per_rank"][0]["exchange_ms_mean"]))
except Exception as e:
    print("LB_COMM_PRIO=%s rows %s %s: no line (%s)" % (sys.argv[1], sys.argv[2], sys.argv[3], e)); print(open("gpurun_out/x.err").read()[-800:])
PY
    done
  done
done
done
export LB_COMM_PRIO=1
for t in rccl peer; do
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_bench -- python3 $GRAFT_REPO_ROOT/bench.py --force-slab-path --slab-rows 1024 --transport $t --steps 56 --warmup 14 --no-cpu-baseline --no-other-configs --min-blocks 3 > $GRAFT_REPO_ROOT/gpurun_out/tl_bench.log 2>&1)
python3 tools/timeline.py gpurun_out/tl_bench 30 k_halo_ > gpurun_out/r06h_bench_timeline_${t}_1024_prio.txt 2>&1
rm -rf gpurun_out/tl_bench
done
timeout 1200 python3 -m pytest tests -m gpu -q -k "slab or rccl or peer or distributed or random or halo" > gpurun_out/r06_run30_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_run30_pytest.log
rm -f gpurun_out/x.json gpurun_out/x.err
sort $P | uniq | cat
cut -c1-150 gpurun_out/r06h_bench_timeline_rccl_1024_prio.txt
tail -3 gpurun_out/r06_run30_pytest.log
}

# round 6, GPU run 31: bench.py's GPU_MAX_HW_QUEUES=8 default against HIP's four: the slab of 8 | 4 | 2 | 1 ranks in 280-step blocks, both
# transports, three rounds on one box; the plain single-GPU line both ways
run31() {
P=gpurun_out/r06h_bench_hw_queues.txt
: > $P
for rep in 1 2 3; do
for hq in 4 8; do
  export GPU_MAX_HW_QUEUES=$hq
  for rows in 1024 2048 4096 8192; do
    for t in rccl peer; do
      timeout 200 python3 bench.py --force-slab-path --slab-rows $rows --transport $t --steps 280 --warmup 28 --no-cpu-baseline --no-other-configs > gpurun_out/x.json 2> gpurun_out/x.err
      python3 - $hq $rows $t >> $P <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/x.json").read().strip().splitlines()[-1])
    print("GPU_MAX_HW_QUEUES=%s rows %5s %-4s: %9.1f MLUPS, exchange %.1f us mean" % (sys.argv[1], sys.argv[2], sys.argv[3], d["value"], 1e3 * d["slabs"]["per_rank"][0]["exchange_ms_mean"]))
except Exception as e:
    print("GPU_MAX_HW_QUEUES=%s rows %s %s: no line (%s)" % (sys.argv[1], sys.argv[2], sys.argv[3], e)); print(open("gpurun_out/x.err").read()[-800:])
PY
    done
  done
  timeout 200 python3 bench.py --steps 84 --warmup 14 --no-cpu-baseline --no-other-configs > gpurun_out/x.json 2> gpurun_out/x.err
  python3 -c "
import json
d=json.loads(open('gpurun_out/x.json').read().strip().splitlines()[-1]); print('GPU_MAX_HW_QUEUES=$hq plain single-GPU line: %9.1f MLUPS' % d['value'])" >> $P
done
done
unset GPU_MAX_HW_QUEUES
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_bench -- python3 $GRAFT_REPO_ROOT/bench.py --force-slab-path --slab-rows 1024 --transport rccl --steps 56 --warmup 14 --no-cpu-baseline --no-other-configs --min-blocks 3 > $GRAFT_REPO_ROOT/gpurun_out/tl_bench.log 2>&1)
python3 tools/timeline.py gpurun_out/tl_bench 30 k_halo_ > gpurun_out/r06h_bench_timeline_rccl_1024_hq8.txt 2>&1
rm -rf gpurun_out/tl_bench gpurun_out/x.json gpurun_out/x.err
sort $P | uniq -c | cat
cut -c1-150 gpurun_out/r06h_bench_timeline_rccl_1024_hq8.txt | tail -22
}

# round 6, GPU run 32: does rocprofv3's PC sampling (beta) work on this box?  What the agent offers; a host-trap sample of k_deep<7> 8192^2
run32() {
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 60 rocprofv3 -L > $R/gpurun_out/r06_rocprof_avail.txt 2>&1
grep -i -B2 -A12 "pc.sampl" $R/gpurun_out/r06_rocprof_avail.txt | head -60
timeout 180 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval 1 --output-format csv -d $R/gpurun_out/pcs -- python3 $R/tools/run_case.py --bc periodic --n 8192 --steps 140 > $R/gpurun_out/pcs.log 2>&1
echo "rc=$?"
tail -5 $R/gpurun_out/pcs.log
find $R/gpurun_out/pcs -type f | head; for f in $(find $R/gpurun_out/pcs -name "*pc_sampling*.csv"); do wc -l $f; head -5 $f; done
}

# round 6, GPU run 33: lb_set_exchange_inline (ABI 10) -- the slab tests (every depth x placement, both transports; random self-rings with
# odd seeds inline; four rank processes over the peer transport, inline), bench.py on the slab of 8 | 4 | 2 | 1 ranks with the collective
# tuner choosing depth AND placement (what did it choose?), proxy A/B of the placement
run33() {
timeout 1500 python3 -m pytest tests -m gpu -q -k "slab or rccl or peer or distributed or random or halo or abi" > gpurun_out/r06_run33_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_run33_pytest.log
timeout 600 python3 tools/peer_ranks_check.py --ranks 4 --inline > gpurun_out/r06i_peer_ranks4_inline.txt 2>&1
echo "peer ranks inline rc=$?" >> gpurun_out/r06_run33_pytest.log
P=gpurun_out/r06i_bench_placement.txt
: > $P
for rep in 1 2; do
  for rows in 1024 2048 4096 8192; do
    for t in rccl peer; do
      timeout 200 python3 bench.py --force-slab-path --slab-rows $rows --transport $t --steps 280 --warmup 28 --no-cpu-baseline --no-other-configs > gpurun_out/x.json 2> gpurun_out/x.err
      python3 - $rows $t >> $P <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/x.json").read().strip().splitlines()[-1])
    c = d["slabs"]["cycle_tuning"]
    print("rows %5s %-4s: %9.1f MLUPS, exchange %.1f us mean; tuner: depth %d, exchange %s; us per step beside %s | between %s" % (
        sys.argv[1], sys.argv[2], d["value"], 1e3 * d["slabs"]["per_rank"][0]["exchange_ms_mean"], c["depth"],
        "between the launches" if c["exchange_inline"] else "beside them",
        {k: round(1e3 * v, 2) for k, v in c["ms_per_step"].items()}, {k: round(1e3 * v, 2) for k, v in c["ms_per_step_inline"].items()}))
except Exception as e:
    print("rows %s %s: no line (%s)" % (sys.argv[1], sys.argv[2], e)); print(open("gpurun_out/x.err").read()[-800:])
PY
    done
  done
done
Q=gpurun_out/r06i_slab_proxy_placement.txt
: > $Q
for rep in 1 2; do
  for inl in "" "--inline"; do
    echo "== placement: ${inl:-beside}" >> $Q
    timeout 400 python3 tools/slab_proxy.py --torch-dist $inl --parts 8,4,2,1 --steps 140 --variants -1 --transports rccl,peer --reps 4 2>&1 | grep grid >> $Q
  done
done
rm -f gpurun_out/x.json gpurun_out/x.err
tail -4 gpurun_out/r06_run33_pytest.log; tail -2 gpurun_out/r06i_peer_ranks4_inline.txt
cat $P
cut -c1-150 $Q
}

# round 6, GPU run 34: the collective tuner with twenty cycles per candidate: what it chooses on the slab of 8 | 4 | 2 | 1 ranks, and
# the rate of the 280-step blocks behind it; three rounds
run34() {
P=gpurun_out/r06j_bench_placement.txt
: > $P
for rep in 1 2 3; do
  for rows in 1024 2048 4096 8192; do
    for t in rccl peer; do
      timeout 200 python3 bench.py --force-slab-path --slab-rows $rows --transport $t --steps 280 --warmup 28 --no-cpu-baseline --no-other-configs > gpurun_out/x.json 2> gpurun_out/x.err
      python3 - $rows $t >> $P <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/x.json").read().strip().splitlines()[-1])
    c = d["slabs"]["cycle_tuning"]
    print("rows %5s %-4s: %9.1f MLUPS, exchange %.1f us mean; tuner: depth %d, exchange %s; us per step beside %s | between %s" % (
        sys.argv[1], sys.argv[2], d["value"], 1e3 * d["slabs"]["per_rank"][0]["exchange_ms_mean"], c["depth"],
        "BETWEEN the launches" if c["exchange_inline"] else "beside them",
        {k: round(1e3 * v, 2) for k, v in c["ms_per_step"].items()}, {k: round(1e3 * v, 2) for k, v in c["ms_per_step_inline"].items()}))
except Exception as e:
    print("rows %s %s: no line (%s)" % (sys.argv[1], sys.argv[2], e)); print(open("gpurun_out/x.err").read()[-800:])
PY
    done
  done
done
rm -f gpurun_out/x.json gpurun_out/x.err
sort $P
}

# round 6, GPU run 35: the library of commit "ABI 10": full GPU suite, smoke, the driver's bench command, bench over the slab path (both
# transports), rocprofv3 kernel-trace summary of the driver's command
run35() {
timeout 1500 python3 -m pytest tests -m gpu -q > gpurun_out/r06k_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06k_pytest_gpu.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06k_smoke.txt 2>&1
echo "smoke rc=$?" >> gpurun_out/r06k_smoke.txt
timeout 600 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r06k_bench_steps20.json 2> gpurun_out/r06k_bench_steps20.err
echo "bench rc=$?" >> gpurun_out/r06k_smoke.txt
for t in rccl peer; do
  timeout 300 python3 bench.py --force-slab-path --transport $t --steps 56 --warmup 14 --no-cpu-baseline > gpurun_out/r06k_bench_slabpath_$t.json 2> gpurun_out/r06k_bench_slabpath_$t.err
  echo "bench $t rc=$?" >> gpurun_out/r06k_smoke.txt
done
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06k_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/r06k_prof.log 2>&1)
f=$(find gpurun_out/r06k_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -8 "$f" > gpurun_out/r06k_rocprof_kernel_stats.csv
rm -rf gpurun_out/r06k_prof
tail -3 gpurun_out/r06k_pytest_gpu.log; cat gpurun_out/r06k_smoke.txt | tail -5
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r06k_bench_steps20.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_plain_launch"), d["roofline"]["kernel"][:60])
print({k: v for k, v in d.items() if k in ("cpu_baseline",)})
print([ (o["config"], o["value"]) for o in d.get("other_configs", [])])
for t in ("rccl","peer"):
    e=json.loads(open("gpurun_out/r06k_bench_slabpath_%s.json"%t).read().strip().splitlines()[-1])
    print(t, e["value"], e["slabs"]["per_rank"], {k: e["slabs"]["cycle_tuning"][k] for k in ("depth","exchange_inline")})
PY
cat gpurun_out/r06k_rocprof_kernel_stats.csv | cut -c1-200
}

# round 6, GPU run 36: the driver's bench command again (the other configurations had failed on an undefined name), and the default run
run36() {
timeout 600 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r06k_bench_steps20.json 2> gpurun_out/r06k_bench_steps20.err
echo "rc=$?"
timeout 600 python3 bench.py > gpurun_out/r06k_bench_default.json 2> gpurun_out/r06k_bench_default.err
echo "rc=$?"
python3 - <<'PY'
import json
for f in ("gpurun_out/r06k_bench_steps20.json", "gpurun_out/r06k_bench_default.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["steps"], d["roofline"]["frac"], d["roofline"].get("frac_plain_launch"), d["roofline"]["launch_ms"])
    print([(o["config"], o.get("path"), o.get("value"), o.get("error")) for o in d.get("other_configs", [])])
PY
}

# round 6, GPU run 37: the reference's case (3751 x 1251 pipe with a disc) over the kernel families, 600 steps each, best of 3
run37() {
P=gpurun_out/r06l_reference_case_variants.txt
: > $P
for v in -1 53601 20833 4449 353 609 119137; do
  timeout 120 python3 tools/run_case.py --bc pipe --cyl --n 3751 --ny 1251 --steps 840 --repeat 3 --variant $v >> $P 2>&1
done
timeout 200 python3 tools/reference_grid_bench.py >> $P 2>&1
cat $P
}

# round 6, GPU run 38: non-temporal stores (variant bit 0) below the 450 MB lattice-pair threshold of round 3 (measured with k_step4 then):
# k_deep<7> / k_deep<6> / k_step5 with plain | non-temporal stores on grids of 1.5 M - 6 M cells; 840 steps, best of 3
run38() {
P=gpurun_out/r06l_nt_stores_midsize.txt
: > $P
run() { timeout 100 python3 tools/run_case.py "$@" --steps 840 --repeat 3 2>&1 | tail -1 | sed -e 's/\[k_\([a-z0-9<>]*\)[^]]*\]/[\1]/' >> $P; }
for v in 53616 53601 20848 20833 4464 4449; do
  run --bc pipe --cyl --n 3751 --ny 1251 --variant $v
done
for n in 1536 2048 2400; do
  for v in 53616 53601 20848 20833 4464 4449; do run --bc periodic --n $n --variant $v; done
done
for n in 2048 2400; do
  for v in 53616 53601 4464 4449; do run --bc pipe --n $n --variant $v; done
  for v in 53616 53601 4464 4449; do run --bc cavity --mask --n $n --variant $v; done
done
cat $P
}

# round 6, GPU run 39: the reference's case: what separates the automatic variant from the forced ones (bits 0 and 4)?
run39() {
P=gpurun_out/r06l_reference_case_bits.txt
: > $P
run() { timeout 100 python3 tools/run_case.py "$@" --steps 840 --repeat 3 2>&1 | tail -1 | sed -e 's/\[k_\([a-z0-9<>]*\)[^]]*\]/[\1]/' >> $P; }
for rep in 1 2; do
for v in -1 53600 53616 53601 53617 20832 20833; do
  run --bc pipe --cyl --n 3751 --ny 1251 --variant $v
done
done
cat $P
}

# round 6, GPU run 40: non-temporal stores from 320 MB per lattice pair: the driver's bench command (reference case behind it), the
# reference-grid tool, twice
run40() {
for rep in 1 2; do
  timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06m_bench_steps20_$rep.json 2> gpurun_out/r06m_bench_steps20_$rep.err
  timeout 200 python3 tools/reference_grid_bench.py > gpurun_out/r06m_reference_grid_$rep.txt 2>&1
done
python3 - <<'PY'
import json
for r in (1, 2):
    d=json.loads(open("gpurun_out/r06m_bench_steps20_%d.json" % r).read().strip().splitlines()[-1])
    print(d["value"], d["roofline"]["frac"], [(o["config"], o.get("path"), o.get("value"), o.get("steps_per_launch")) for o in d.get("other_configs", [])])
PY
cat gpurun_out/r06m_reference_grid_1.txt gpurun_out/r06m_reference_grid_2.txt
}

# round 6, GPU run 41: k_deep2<7> among lb_autotune's candidates in walled boxes / with a mask; non-temporal stores from 320 MB: full GPU
# suite, what the tuner picks across sizes and families, the driver's bench command, the reference-grid tool
run41() {
timeout 1500 python3 -m pytest tests -m gpu -q > gpurun_out/r06n_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06n_pytest_gpu.log
timeout 600 python3 tools/tune_probe.py 2048 3072 4096 8192 2>&1 | cut -c1-150 > gpurun_out/r06n_tune_probe.txt
for rep in 1 2; do
  timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06n_bench_steps20_$rep.json 2> gpurun_out/r06n_bench_steps20_$rep.err
  timeout 200 python3 tools/reference_grid_bench.py > gpurun_out/r06n_reference_grid_$rep.txt 2>&1
done
tail -3 gpurun_out/r06n_pytest_gpu.log
cat gpurun_out/r06n_tune_probe.txt
python3 - <<'PY'
import json
for r in (1, 2):
    d=json.loads(open("gpurun_out/r06n_bench_steps20_%d.json" % r).read().strip().splitlines()[-1])
    print(d["value"], d["roofline"]["frac"], [(o["config"], o.get("path"), o.get("value"), (o.get("kernel") or "")[:12]) for o in d.get("other_configs", [])])
PY
cat gpurun_out/r06n_reference_grid_1.txt gpurun_out/r06n_reference_grid_2.txt | grep opencl
}

# round 6, GPU run 42: profiles of configurations 4 and 5 on the final library (tuner pinned: the summary names the kernel the line names)
run42() {
rm -rf gpurun_out/prof_r06nc4 gpurun_out/prof_r06nc5
timeout 500 bash tools/gpu_profile.sh r06nc4 > gpurun_out/r06n_profile_c4.log 2>&1
timeout 500 bash tools/gpu_profile.sh r06nc5 --config 5 > gpurun_out/r06n_profile_c5.log 2>&1
cat gpurun_out/prof_r06nc5/tune_cache.txt
python3 - <<'PY'
import json
for t in ("r06nc4", "r06nc5"):
    d=json.loads(open("gpurun_out/prof_%s/unprofiled.json" % t).read().strip().splitlines()[-1])
    print(t, d["value"], d["roofline"]["kernel"][:70], d["roofline"]["launch_ms"])
PY
du -sh gpurun_out/prof_r06nc4 gpurun_out/prof_r06nc5
}

# round 6, GPU run 43: walled boxes below k_deep's static threshold (2300^2): k_step5 | k_deep<7> | k_deep2<7>, 840 steps, best of 3
run43() {
P=gpurun_out/r06o_walled_small_sweep.txt
: > $P
run() { timeout 100 python3 tools/run_case.py "$@" --steps 840 --repeat 3 2>&1 | tail -1 | sed -e 's/\[k_\([a-z0-9<>]*\)[^]]*\]/[\1]/' >> $P; }
for n in 1280 1536 1792 2048 2304 2560; do
  if [ $n -ge 2110 ]; then nt=1; else nt=16; fi
  for fam in "--bc pipe" "--bc cavity" "--bc pipe --mask" "--bc periodic --mask"; do
    for v in $((4448 + nt)) $((53600 + nt)) $((53600 + 65536 + nt)); do
      run $fam --n $n --variant $v
    done
  done
done
cat $P
}

# round 6, GPU run 44: the LDS tiles (k_tile4) against the marching kernels in walled boxes around the tiles' static threshold (1850^2)
run44() {
P=gpurun_out/r06o_walled_tile_sweep.txt
: > $P
run() { timeout 100 python3 tools/run_case.py "$@" --steps 840 --repeat 3 2>&1 | tail -1 | sed -e 's/\[k_\([a-z0-9<>]*\)[^]]*\]/[\1]/' >> $P; }
for n in 1280 1536 1664 1792 1920 2048; do
  for fam in "--bc pipe" "--bc cavity" "--bc pipe --mask"; do
    for v in 625 4464 119152; do
      run $fam --n $n --variant $v
    done
  done
done
cat $P
}

# round 6, GPU run 45: the static kernel table after its round-6 revision (walled: tiles below 1450^2, k_step5 to 1700^2, k_deep2 to
# 2900^2, k_deep<7> above; periodic with a mask: k_deep from 1250^2): full GPU suite; static choice against lb_autotune across sizes
run45() {
timeout 1500 python3 -m pytest tests -m gpu -q > gpurun_out/r06p_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06p_pytest_gpu.log
timeout 900 python3 tools/tune_probe.py 1280 1536 1792 2048 2560 3072 4096 2>&1 | cut -c1-120 > gpurun_out/r06p_tune_probe.txt
timeout 200 python3 tools/reference_grid_bench.py > gpurun_out/r06p_reference_grid.txt 2>&1
tail -3 gpurun_out/r06p_pytest_gpu.log
cat gpurun_out/r06p_tune_probe.txt
grep opencl gpurun_out/r06p_reference_grid.txt
}

# round 6, GPU run 46: periodic boxes without a mask around the static thresholds (tiles below 1200^2, k_step5 to 1500^2, k_deep above):
# tiles | k_step5 | k_deep<6> | k_deep<7>; 1680 steps, best of 3 (the clocks have ramped by then)
run46() {
P=gpurun_out/r06q_periodic_small_sweep.txt
: > $P
run() { timeout 100 python3 tools/run_case.py "$@" --steps 1680 --repeat 3 2>&1 | tail -1 | sed -e 's/\[k_\([a-z0-9<>]*\)[^]]*\]/[\1]/' >> $P; }
for n in 1024 1152 1280 1408 1536 1792 2048; do
  for v in 625 4464 20848 53616; do
    run --bc periodic --n $n --variant $v
  done
done
cat $P
}

# round 6, GPU run 47: the static table after the periodic revision (k_deep<6> from 1100^2, k_deep<7> from 1900^2, tiles below 1100^2): GPU
# suite; the automatic variant across sizes and families (1680-step runs, best of 3)
run47() {
timeout 1500 python3 -m pytest tests -m gpu -q > gpurun_out/r06r_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06r_pytest_gpu.log
P=gpurun_out/r06r_static_choice.txt
: > $P
run() { timeout 100 python3 tools/run_case.py "$@" --steps 1680 --repeat 3 2>&1 | tail -1 | cut -c1-150 >> $P; }
for n in 1024 1152 1280 1536 1792 2048 2560; do
  for fam in "--bc periodic" "--bc pipe" "--bc cavity --mask"; do
    run $fam --n $n --variant -1
  done
done
run --bc pipe --cyl --n 3751 --ny 1251 --variant -1
tail -3 gpurun_out/r06r_pytest_gpu.log
cat $P
}

# round 6, GPU run 48: k_deep2<7> against k_deep<7> on the plain grids a slab of 8 | 4 | 2 ranks holds (short segments), periodic and pipe
run48() {
P=gpurun_out/r06s_deep2_slab_shapes.txt
: > $P
run() { timeout 100 python3 tools/run_case.py "$@" --steps 840 --repeat 3 2>&1 | tail -1 | sed -e 's/\[k_\([a-z0-9<>]*\)[^]]*\]/[\1]/' >> $P; }
for rep in 1 2; do
for ny in 1024 2048 4096; do
  for v in 53601 119137 20833; do
    run --bc periodic --n 8192 --ny $ny --variant $v
  done
done
done
cat $P
}

# round 6, GPU run 49: k_deep2<7> inside the slab cycle (variant bit 16 on a slab handle): bitwise? faster?
run49() {
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "slab_cycle_depth" > gpurun_out/r06s_pytest_slab_deep2.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06s_pytest_slab_deep2.log
P=gpurun_out/r06s_slab_proxy_deep2.txt
: > $P
for rep in 1 2; do
  timeout 500 python3 tools/slab_proxy.py --parts 8,4,2 --steps 140 --variants 53601,119137 --transports peer,rccl --reps 4 2>&1 | grep grid | cut -c1-150 >> $P
done
tail -3 gpurun_out/r06s_pytest_slab_deep2.log
cat $P
}

# round 6, GPU run 50: k_deep2<7> in the slab cycle (lb_set_slab_cycle(8); the default under RCCL): full GPU suite (new: one slab of eight
# as a ring of its own at full size, both transports; random self-rings and rank processes with the deep variants), four rank processes
# over the peer transport, the proxy on the automatic variant, bench.py on the slab of 8 | 4 | 2 | 1 ranks with the tuner choosing
run50() {
timeout 1800 python3 -m pytest tests -m gpu -q > gpurun_out/r06t_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06t_pytest_gpu.log
timeout 900 python3 tools/peer_ranks_check.py --ranks 4 > gpurun_out/r06t_peer_ranks4.txt 2>&1
echo "peer ranks rc=$?" >> gpurun_out/r06t_pytest_gpu.log
P=gpurun_out/r06t_slab_proxy_final.txt
: > $P
for rep in 1 2; do
  timeout 500 python3 tools/slab_proxy.py --parts 8,4,2,1 --steps 140 --variants -1 --transports rccl,peer --reps 5 2>&1 | grep grid | cut -c1-260 >> $P
done
Q=gpurun_out/r06t_bench_placement.txt
: > $Q
for rep in 1 2; do
  for rows in 1024 2048 4096 8192; do
    for t in rccl peer; do
      timeout 200 python3 bench.py --force-slab-path --slab-rows $rows --transport $t --steps 280 --warmup 28 --no-cpu-baseline --no-other-configs > gpurun_out/x.json 2> gpurun_out/x.err
      python3 - $rows $t >> $Q <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/x.json").read().strip().splitlines()[-1])
    c = d["slabs"]["cycle_tuning"]
    print("rows %5s %-4s: %9.1f MLUPS, exchange %.1f us mean; tuner: depth %d, exchange %s; us per step beside %s | between %s" % (
        sys.argv[1], sys.argv[2], d["value"], 1e3 * d["slabs"]["per_rank"][0]["exchange_ms_mean"], c["depth"],
        "BETWEEN the launches" if c["exchange_inline"] else "beside them",
        {k: round(1e3 * v, 2) for k, v in c["ms_per_step"].items()}, {k: round(1e3 * v, 2) for k, v in c["ms_per_step_inline"].items()}))
except Exception as e:
    print("rows %s %s: no line (%s)" % (sys.argv[1], sys.argv[2], e)); print(open("gpurun_out/x.err").read()[-800:])
PY
    done
  done
done
rm -f gpurun_out/x.json gpurun_out/x.err
tail -4 gpurun_out/r06t_pytest_gpu.log; tail -2 gpurun_out/r06t_peer_ranks4.txt
cut -c1-150 $P
cat $Q
}

# round 6, GPU run 51: the slab tests again (lb_set_slab_cycle(8) beats the variant's bit)
run51() {
timeout 1800 python3 -m pytest tests -m gpu -q > gpurun_out/r06t_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06t_pytest_gpu.log
tail -4 gpurun_out/r06t_pytest_gpu.log
}

# round 6, GPU run 52: k_deep<7> against k_deep2<7> on the headline grid (8192^2 periodic) and on 4096^2, alternating, three rounds
run52() {
P=gpurun_out/r06u_deep2_headline.txt
: > $P
run() { timeout 100 python3 tools/run_case.py "$@" --steps 840 --repeat 3 2>&1 | tail -1 | sed -e 's/\[k_\([a-z0-9<>]*\)[^]]*\]/[\1]/' >> $P; }
for rep in 1 2 3; do
  for v in 53601 119137; do
    run --bc periodic --n 8192 --variant $v
    run --bc periodic --n 4096 --variant $v
  done
done
sort $P
}

# round 6, GPU run 53: k_deep2<7> among the tuner's candidates in every family: the driver's bench command three times (what does the
# headline pick?), the default command, pinned profiles of configuration 4 by whichever kernel the tuner picks AND by the other one
run53() {
for rep in 1 2 3; do
  timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06v_bench_steps20_$rep.json 2> gpurun_out/r06v_bench_steps20_$rep.err
done
timeout 600 python3 bench.py > gpurun_out/r06v_bench_default.json 2> gpurun_out/r06v_bench_default.err
rm -rf gpurun_out/prof_r06vc4 gpurun_out/prof_r06vc4d2 gpurun_out/prof_r06vc4d1
timeout 500 bash tools/gpu_profile.sh r06vc4 > gpurun_out/r06v_profile_c4.log 2>&1
timeout 500 bash tools/gpu_profile.sh r06vc4d2 --variant 119137 > gpurun_out/r06v_profile_c4d2.log 2>&1
timeout 500 bash tools/gpu_profile.sh r06vc4d1 --variant 53601 > gpurun_out/r06v_profile_c4d1.log 2>&1
python3 - <<'PY'
import json
for f in ["gpurun_out/r06v_bench_steps20_%d.json" % r for r in (1, 2, 3)] + ["gpurun_out/r06v_bench_default.json", "gpurun_out/prof_r06vc4/unprofiled.json", "gpurun_out/prof_r06vc4d2/unprofiled.json", "gpurun_out/prof_r06vc4d1/unprofiled.json"]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f[11:], d["value"], d["roofline"]["frac"], d["roofline"].get("frac_plain_launch"), d["roofline"]["kernel"][:12], [(o["config"], o.get("value"), (o.get("kernel") or "")[:9]) for o in d.get("other_configs", [])])
    except Exception as e:
        print(f, "no line", e)
PY
}

# round 6, GPU run 54: the last library: full GPU suite, smoke, the driver's bench command and the default one, bench over the slab path
run54() {
timeout 1800 python3 -m pytest tests -m gpu -q > gpurun_out/r06w_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06w_pytest_gpu.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06w_smoke.txt 2>&1
echo "smoke rc=$?" >> gpurun_out/r06w_smoke.txt
timeout 600 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r06w_bench_steps20.json 2> gpurun_out/r06w_bench_steps20.err
echo "bench rc=$?" >> gpurun_out/r06w_smoke.txt
timeout 600 python3 bench.py > gpurun_out/r06w_bench_default.json 2> gpurun_out/r06w_bench_default.err
echo "bench default rc=$?" >> gpurun_out/r06w_smoke.txt
for t in rccl peer; do
  timeout 300 python3 bench.py --force-slab-path --transport $t --steps 56 --warmup 14 --no-cpu-baseline > gpurun_out/r06w_bench_slabpath_$t.json 2> gpurun_out/r06w_bench_slabpath_$t.err
  echo "bench $t rc=$?" >> gpurun_out/r06w_smoke.txt
done
timeout 200 python3 bench.py --variant 119137 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/r06w_bench_forced_deep2.json 2> gpurun_out/r06w_bench_forced_deep2.err
tail -3 gpurun_out/r06w_pytest_gpu.log; cat gpurun_out/r06w_smoke.txt | tail -6
python3 - <<'PY'
import json
for f in ("r06w_bench_steps20", "r06w_bench_default", "r06w_bench_forced_deep2"):
    d=json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(f, d["value"], r["frac"], r.get("frac_plain_launch"), r["launch_ms"], r.get("block_plan"), r["kernel"][:12], [(o["config"], o.get("value"), (o.get("kernel") or "")[:9], o.get("error")) for o in d.get("other_configs", [])], {k: v for k, v in (d.get("cpu_baseline") or {}).items() if k in ("value", "kind", "cores")})
for t in ("rccl","peer"):
    e=json.loads(open("gpurun_out/r06w_bench_slabpath_%s.json"%t).read().strip().splitlines()[-1])
    print(t, e["value"], e["slabs"]["per_rank"], {k: e["slabs"]["cycle_tuning"][k] for k in ("depth","exchange_inline")})
PY
}

# round 6, GPU run 55: more of the randomised slab checks on the last library: 96 random RCCL self-rings (every fourth on the deep
# cycles, k_deep2 among them; odd seeds with the exchange between the launches), 120 seeds of the kernel-variant / partition tests
run55() {
LB_RANDOM_RING_SEEDS=96 LB_RANDOM_SEEDS=120 timeout 1700 python3 -m pytest tests/test_gpu_random.py -m gpu -q > gpurun_out/r06x_random.txt 2>&1
echo "rc=$?" >> gpurun_out/r06x_random.txt
tail -4 gpurun_out/r06x_random.txt
}

# round 6, GPU run 56: SQ counters of k_deep2<7> (8192^2 periodic, 140 steps per pass, four --pmc passes), k_deep<7> beside it on the same box
run56() {
timeout 600 bash tools/gpu_pmc_case.sh r06deep2 --bc periodic --n 8192 --steps 140 --variant 119137 > gpurun_out/r06_sq_deep2.txt 2>&1
timeout 600 bash tools/gpu_pmc_case.sh r06deep7b --bc periodic --n 8192 --steps 140 --variant 53601 > gpurun_out/r06_sq_deep7b.txt 2>&1
grep "k_deep2" gpurun_out/r06_sq_deep2.txt | head -40
grep "k_deep<1, false, false, 7" gpurun_out/r06_sq_deep7b.txt | head -40
}

# round 6, GPU run 57: per-wave timeline of one k_deep2<7> launch (diagnostic build, LB_DIAG bit 12), 8192^2 periodic; k_deep<7> beside it
run57() {
LB_TIMELINE_DEEP2=1 LB_TIMELINE_DEPTH=7 timeout 200 python3 tools/wave_timeline.py 8192 > gpurun_out/r06y_wave_timeline_deep2.txt 2>&1
LB_TIMELINE_DEPTH=7 timeout 200 python3 tools/wave_timeline.py 8192 > gpurun_out/r06y_wave_timeline_deep7.txt 2>&1
cat gpurun_out/r06y_wave_timeline_deep2.txt | cut -c1-400
head -12 gpurun_out/r06y_wave_timeline_deep7.txt | cut -c1-300
}

# round 6, GPU run 58: the rebuilt final library (only LB_DIAG-guarded source changed since the last full suite): smoke + the parity files
run58() {
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_health.py -m gpu -q > gpurun_out/r06z_pytest_parity.log 2>&1
echo "rc=$?"; grep -n "passed\|failed" gpurun_out/r06z_pytest_parity.log | tail -2
}

# round 6, GPU run 59: obstacle masks at full size: k_deep<7> | k_deep2<7>, periodic 8192^2 with 1 % random solid cells, pipe + mask 8192^2,
# config 5's image 4096^2, unmasked beside them
run59() {
P=gpurun_out/r06z_mask_deep2.txt
: > $P
run() { timeout 100 python3 tools/run_case.py "$@" --steps 420 --repeat 3 2>&1 | tail -1 | sed -e 's/\[k_\([a-z0-9<>]*\)[^]]*\]/[\1]/' >> $P; }
for v in 53601 119137; do
  run --bc periodic --n 8192 --variant $v
  run --bc periodic --mask --n 8192 --variant $v
  run --bc pipe --n 8192 --variant $v
  run --bc pipe --mask --n 8192 --variant $v
  run --bc pipe --tiff --n 4096 --variant $v
  run --bc cavity --cyl --n 6144 --variant $v
done
sort $P
}

# round 6, GPU run 60: pinned profiles of configurations 2, 3, 5 on the final library (k_deep2 among the tuner's candidates)
run60() {
for c in 2 3 5; do
  rm -rf gpurun_out/prof_r06zc$c
  timeout 500 bash tools/gpu_profile.sh r06zc$c --config $c > gpurun_out/r06z_profile_c$c.log 2>&1
  python3 - $c <<'PY'
import json, sys
d=json.loads(open("gpurun_out/prof_r06zc%s/unprofiled.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("config", sys.argv[1], d["value"], d["roofline"]["kernel"][:60], d["roofline"]["launch_ms"], d["roofline"]["frac"])
PY
done
}

# round 6, GPU run 61: soak of the final library (the automatic choices after the size table's revision; k_deep2 forced): bitwise against the
# single-step kernel over hundreds of steps, alone and beside a second process
run61() {
P=gpurun_out/r06z_soak.txt
: > $P
echo "== automatic choices, alone" >> $P
timeout 600 python3 tools/soak_bitwise.py --more >> $P 2>&1
echo "== k_deep2 forced (LB_SOAK_VARIANT=119137)" >> $P
LB_SOAK_VARIANT=119137 timeout 600 python3 tools/soak_bitwise.py --more >> $P 2>&1
grep -c "bitwise equal" $P; grep -v "bitwise equal" $P | head -20
}

# round 6, GPU run 62: the driver's bench command on the last commit (traffic priced per kernel family), twice
run62() {
for rep in 1 2; do
  timeout 600 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r06zz_bench_steps20_$rep.json 2> gpurun_out/r06zz_bench_steps20_$rep.err
  python3 - $rep <<'PY'
import json, sys
d=json.loads(open("gpurun_out/r06zz_bench_steps20_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print(d["value"], r["frac"], r.get("frac_plain_launch"), r["kernel"][:12], r["traffic"], r["traffic_frac"] if "traffic_frac" in r else None, r["traffic_source"][:120])
print([(o["config"], o.get("value"), (o.get("kernel") or "")[:9], o.get("roofline_frac")) for o in d.get("other_configs", [])])
PY
done
}

# round 6, GPU run 63: the full GPU suite and smoke on the round's last commit
run63() {
timeout 1800 python3 -m pytest tests -m gpu -q > gpurun_out/r06_final_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_final_pytest_gpu.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/r06_final_pytest_gpu.log 2>&1
grep -n "passed\|failed\|smoke ok\|pytest rc" gpurun_out/r06_final_pytest_gpu.log | tail -4
}

if [ $# -ne 1 ] || ! declare -F "run$1" > /dev/null; then
  echo "usage: bash tools/r06/runs.sh <1 ... 63>" >&2; exit 2
fi
"run$1"
