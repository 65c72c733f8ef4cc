for cfg in "--steps 20 --warmup 5 --merge 20 --inflight 1" "--steps 20 --warmup 5 --merge 10 --inflight 2" "--steps 20 --warmup 5 --merge 7 --inflight 3" "--steps 96 --warmup 6 --merge 32 --inflight 3" "--steps 96 --warmup 6 --merge 16 --inflight 3" "--steps 96 --warmup 6 --merge 48 --inflight 2" "--steps 96 --warmup 6 --merge 24 --inflight 2"; do
timeout 600 python bench.py $cfg --no-cpu-baseline --no-roofline > gpurun_out/b.json 2>gpurun_out/b.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/b.json").read().strip().splitlines()[-1])
    print("$cfg ->", d["value"], "img/s", d["ms_per_step"], "ms/step host", d["host_ms_per_step"])
except Exception as e:
    print("$cfg FAILED", e); print(open("gpurun_out/b.err").read()[-1500:])
PY
done
