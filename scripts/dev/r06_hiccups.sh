#!/bin/bash
# Run ON THE GPU BOX: how often a timed step of bench.py is a lone slow one (> 1.15 x the median), under environment settings
#   usage: scripts/dev/r06_hiccups.sh "ENV=a" "-" ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
for setting in "$@"; do
    name=$setting; [ "$setting" = "-" ] && { setting=""; name=default; }
    for rep in 1 2 3; do
        env $setting timeout 200 python bench.py --no-cpu-baseline --no-other-configs --no-in-flight --no-host-clouds --steps 400 --warmup 5 2>/dev/null | tail -1 | \
            python3 -c "
import json,sys,statistics
d=json.loads(sys.stdin.read()); s=d['config']['ms_each_step_rank0']; m=statistics.median(s)
slow=[round(x,2) for k,x in enumerate(s) if x>1.15*m and (k%20)!=19 and k!=len(s)-1]
print('%-22s' % '$name', 'value %.1f  median %.3f ms  slow steps %d of %d: %s' % (d['value'], m, len(slow), len(s), slow[:12]))"
    done
done
