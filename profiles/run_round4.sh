#!/bin/bash
# Round-4 measurement batch (run on the GPU box through gpurun, from the repo root): bench lines of every workload, the
# profile passes (rocprofv3 kernel trace + PMC traffic / SQ counters: profiles/run_profiles.sh) of the workloads whose dominant
# kernel changed this round, the drop-in facade's end-to-end throughput.  part = 1 | 2 | 3 keeps a call inside gpurun's limit.
set -e
part=${1:-1}
out=$PWD/gpurun_out
mkdir -p "$out"
if [ "$part" = 1 ]; then
  python bench.py > "$out/r04_bench_f16x3.json" 2> "$out/r04_bench_f16x3.err"
  python bench.py --batch 4096 --steps 4 --warmup 1 --no-cpu-baseline --no-throughput-line > "$out/r04_bench_f16x3_b4096.json" 2> "$out/r04_b4096.err"
  python bench.py --workload ant-round > "$out/r04_bench_ant_round.json" 2> "$out/r04_ant_round.err"
  python bench.py --workload rollout > "$out/r04_bench_rollout.json" 2> "$out/r04_rollout.err"
  python bench.py --workload rollout --model ant > "$out/r04_bench_rollout_ant.json" 2> "$out/r04_rollout_ant.err"
  python bench.py --workload mppi > "$out/r04_bench_mppi.json" 2> "$out/r04_mppi.err"
  python bench.py --workload mppi --model ant > "$out/r04_bench_mppi_ant.json" 2> "$out/r04_mppi_ant.err"
  python profiles/facade_throughput.py 512 1024 8192 > "$out/r04_facade_throughput.txt" 2> "$out/r04_facade.err"
  echo part 1 done
elif [ "$part" = 2 ]; then
  bash profiles/run_profiles.sh r04_f16x3 --precision f16x3
  bash profiles/run_profiles.sh r04_rollout --workload rollout
  bash profiles/run_profiles.sh r04_rollout_ant --workload rollout --model ant
  echo part 2 done
else
  bash profiles/run_profiles.sh r04_ant_round_f16x3 --workload ant-round --no-cpu-baseline --no-early-exit-line --steps 3
  bash profiles/run_profiles.sh r04_mppi_ant --workload mppi --model ant
  bash profiles/run_profiles.sh r04_mppi --workload mppi
  echo part 3 done
fi
