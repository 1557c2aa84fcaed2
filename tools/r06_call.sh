#!/bin/bash
# round 6: the launcher forms of bench.py at HEAD -- torchrun with one rank over RCCL (the driver's N > 1 command shape), two ranks on one device over gloo
mkdir -p gpurun_out/r06c47
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fit --no-fp32 > gpurun_out/r06c47/bench_torchrun_n1.json 2> gpurun_out/r06c47/bench_torchrun_n1.err; echo "rc $?"
timeout 600 python bench.py --gpus 2 --single-device --backend gloo --steps 20 --warmup 5 --no-cpu-baseline --no-fit --no-fp32 > gpurun_out/r06c47/bench_n2.json 2> gpurun_out/r06c47/bench_n2.err; echo "rc $?"
python - <<'PY'
import json
for f in ("bench_torchrun_n1", "bench_n2"):
    for line in open(f"gpurun_out/r06c47/{f}.json"):
        if line.startswith("{"):
            d = json.loads(line)
            print(f, {k: d.get(k) for k in ("value", "n_gpus", "ms_per_step", "scaling", "collective_us", "exposed_us", "ranks_seen", "exchange", "dtype")})
PY
tail -3 gpurun_out/r06c47/bench_torchrun_n1.err
