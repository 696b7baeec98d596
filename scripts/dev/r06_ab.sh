#!/bin/bash
# Run ON THE GPU BOX: the default bench command's headline under a list of environment settings (A/B of a tuning knob).
#   usage: scripts/dev/r06_ab.sh <tag> "ENV1=a ENV2=b" "ENV1=c" ...      ("-" = no setting)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p "$ROOT/gpurun_out/$TAG"
cd "$ROOT"
for setting in "$@"; do
    name=$(echo "$setting" | tr ' =' '__')
    if [ "$setting" = "-" ]; then setting=""; name=default; fi
    for rep in 1 2; do
        env $setting timeout 150 python bench.py --no-cpu-baseline --no-other-configs --no-in-flight --steps 20 --warmup 5 \
            > "gpurun_out/$TAG/${name}_$rep.json" 2> "gpurun_out/$TAG/${name}_$rep.err"
        python - "gpurun_out/$TAG/${name}_$rep.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    ks = {k["name"].split("::")[-1]: round(k["avg_launch_us"], 2) for k in d["roofline"]["kernels"]}
    print("%-40s %.1f reg/s  %.3f ms  h2d %.1f  err %.2e  %s" % (sys.argv[2], d["value"], d["ms_per_step"], d.get("value_h2d_inclusive") or 0,
                                                            d["config"]["final_translation_error_m"], ks))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
    done
done
