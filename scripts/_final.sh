set -u
export FA_HEAD=$(cat .fa_head 2>/dev/null || echo unknown)
bash scripts/collect_profiles.sh r02 2>&1 | tail -1
python3 bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
python3 bench.py --batch 16 --no-cpu-baseline --clients 0 > gpurun_out/r02_bench_batch16.json 2>/dev/null
python3 bench.py --strong --steps 3 --warmup 1 > gpurun_out/r02_bench_strong_n1.json 2>/dev/null
python3 scripts/run_config3.py > gpurun_out/r02_config3.json 2>/dev/null
python3 scripts/run_config45.py 4 > gpurun_out/r02_config4.json 2>/dev/null
python3 scripts/bench_k1.py > gpurun_out/r02_k1.txt 2>/dev/null
python3 -c "
import json
d=json.loads(open('gpurun_out/r02_bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'], d['boundary_call']['ms_per_call'], d['concurrent_clients']['value'], d['cpu_baseline']['value'], d['parity_checked'])
"
