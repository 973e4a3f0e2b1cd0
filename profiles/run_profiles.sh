#!/bin/bash
# Profiling recipe of a round (run on the GPU box from the repo root through gpurun):
#   bash profiles/run_profiles.sh r01
# 1. plain bench (the judged line), 2. rocprofv3 kernel trace + stats of the same command,
# 3./4. FETCH_SIZE / WRITE_SIZE counter passes (own runs, kernel-trace only).
set -e
tag=${1:-r01}
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
python bench.py --steps 10 --warmup 2 > "$out/bench.json" 2> "$out/bench.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o bench -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-early-exit-line > "$out/bench_under_rocprof.json" 2> "$out/rocprof.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o p -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-early-exit-line --no-profile > "$out/pmc_fetch.json" 2> "$out/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o p -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-early-exit-line --no-profile > "$out/pmc_write.json" 2> "$out/pmc_write.err"
echo profiles done
