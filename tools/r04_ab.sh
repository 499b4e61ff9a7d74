#!/bin/bash
# A/B of two builds of the library on ONE box (boxes of the pool differ by +-7 %): bench.py lines for configurations 4, 3, 5
# (and 2 when asked), the libraries alternating.  Usage (on the GPU box, through gpurun):
#   tools/r04_ab.sh <tag> <libA.so> <libB.so> [reps] [configs]
set -u
TAG=$1; A=$2; B=$3; REPS=${4:-3}; CFGS=${5:-"4 3 5"}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/ab_$TAG
mkdir -p $OUT
cd $REPO
for rep in $(seq 1 $REPS); do
  for c in $CFGS; do
    for lib in $A $B; do
      name=$(basename $lib .so)
      LB_LIB=$REPO/$lib timeout 300 python3 bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs \
        > $OUT/${name}_c${c}_r${rep}.json 2> $OUT/${name}_c${c}_r${rep}.err
    done
  done
done
python3 - <<PY
import glob, json, os, statistics as st
rows = {}
for f in sorted(glob.glob("$OUT/*.json")):
    lines = [l for l in open(f) if l.startswith("{")]
    if not lines:
        print("no line:", f); continue
    d = json.loads(lines[-1])
    name, c, r = os.path.basename(f)[:-5].rsplit("_", 2)
    rows.setdefault((c, name), []).append((d["value"], d["roofline"]["frac"], d["roofline"]["launch_ms"]))
for (c, name), v in sorted(rows.items()):
    print("%s %-16s MLUPS %s  frac %s  launch_ms %s" % (c, name, " ".join("%.0f" % x[0] for x in v),
          " ".join("%.4f" % x[1] for x in v), " ".join("%.4f" % x[2] for x in v)))
PY
