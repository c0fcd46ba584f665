python -m pytest tests/test_gpu_assemble.py tests/test_walkers.py tests/test_gpu_workspace.py -m gpu -x -q 2>&1 | tail -2
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.log
python - <<P
import json
r=json.loads([l for l in open("gpurun_out/bench_final.json") if l.startswith("{")][-1])
print(r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["avg_launch_ms"])
for v in r["variants"]: print(v["workload"], round(v["avg_launch_ms"]*1e3,2), round(v["frac"],3))
print(r["extra"]["ms_per_step"], r["extra"]["frac"])
P
bash tools/profile_variants.sh 2>&1 | tail -4
