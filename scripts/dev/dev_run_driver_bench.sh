mkdir -p gpurun_out/r05u
t0=$(date +%s.%N)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05u/bench_line.json 2> gpurun_out/r05u/bench.err
t1=$(date +%s.%N)
echo "bench wall $(echo "$t1 - $t0" | bc) s"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r05u/bench_line.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"])
print({k:round(v,1) for k,v in d["config"].items() if "in_flight" in k or "h2d" in k})
for e in d.get("other_configs", []): print(str(e.get("workload"))[:70], e.get("ms_per_registration"), e.get("in_flight"), e.get("pairs_per_s"))
PY
