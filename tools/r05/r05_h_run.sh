#!/bin/bash
# round 5, step h: the measurement table on the final sources, the driver's command line, the control loop's ticks
cd "$GRAFT_REPO_ROOT" || exit 1
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05_h_bench_driver_args.json 2>/dev/null; python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r05_h_bench_driver_args.json').read().splitlines() if l.startswith('{')][0])
print('driver args: value %.2f M ms_per_step %.4f first %.4f cold %.4f | roofline frac %.4f kernel_ms %.4f traffic %s | cpu %s' % (d['value']/1e6, d['ms_per_step'], d['first_block_ms_per_step'], d['cold']['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['traffic'], {k: d['cpu_baseline'][k] for k in ('value','cores','cpu_model','physical_cores_allowed','cpu_quota','bit_identical_to_portable_build')}))"
python3 bench.py > gpurun_out/r05_h_bench_default.json 2>/dev/null
bash tools/table.sh 2>&1 | grep -v amdgpu.ids
bash tools/loop_time.sh 2>&1 | cut -c1-330
