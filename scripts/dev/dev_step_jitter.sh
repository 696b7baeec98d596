#!/bin/bash
# step-time outliers of the headline bench (a lone +0.35 ms step every ~10): launch-count or wall-clock correlated,
# and what the runtime's kernarg placement does to a step.  Output: gpurun_out/<tag>/jitter.txt
tag=${1:-jitter}
out=gpurun_out/$tag
mkdir -p $out
run() {  # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 60 --warmup 5 --no-in-flight --no-cpu-baseline --no-other-configs > $out/$name.json 2> $out/$name.err
  python - "$name" "$out/$name.json" <<'PY' >> $out/jitter.txt
import json, sys, statistics as st
d = json.load(open(sys.argv[2]))
s = d["config"]["ms_each_step_rank0"]
print(sys.argv[1], "mean %.3f median %.3f min %.3f max %.3f" % (d["ms_per_step"], st.median(s), min(s), max(s)))
print("   ", " ".join("%.2f" % x for x in s))
PY
}
: > $out/jitter.txt
run default A=1
run default_again A=1
run dev_kernarg_1 HIP_FORCE_DEV_KERNARG=1
run dev_kernarg_0 HIP_FORCE_DEV_KERNARG=0
python bench.py --steps 60 --warmup 5 --iters 25 --no-in-flight --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d[\"config\"][\"ms_each_step_rank0\"]; print(\"iters25\", \" \".join(\"%.2f\" % x for x in s))" >> $out/jitter.txt
cat $out/jitter.txt
