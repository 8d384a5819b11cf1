#!/bin/bash
# round 5: `python bench.py --gpus 6` rehearsed with gloo, six ranks sharing the one GPU of the box (the box allows at most 6
# GPU processes: the 8-rank launch itself stays the driver's), batch 8 per rank -> profiles/r05_bench_6rank_gloo_rehearsal.json
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_03; mkdir -p $R
CRIMAC_DIST_BACKEND=gloo timeout -k 10 900 python bench.py --gpus 6 --batch 8 --steps 4 --warmup 2 --roofline-steps 3 --roofline-warmup 1 --tiled-ordered \
  > $R/bench_6rank_gloo.json 2> $R/bench_6rank_gloo.err || { echo bench failed; tail -40 $R/bench_6rank_gloo.err; exit 1; }
tail -5 $R/bench_6rank_gloo.err
python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r5_03/bench_6rank_gloo.json")))
print({k: d[k] for k in ("value", "n_gpus", "ms_per_step")}, d["config"]["rank_census"], d.get("exchange", {}).get("exchange_ms_exposed"))
print(d["tiled"]["patches_per_s"], d["tiled"]["ordered_handoff"])
PY
echo r5_03 done
