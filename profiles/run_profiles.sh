#!/bin/bash
# Profiling recipe of a round (run on the GPU box from the repo root through gpurun):
#   bash profiles/run_profiles.sh r02_f16x3 --precision f16x3
#   bash profiles/run_profiles.sh r02_rollout --workload rollout
# 1. plain bench (the judged line), 2. rocprofv3 kernel trace + stats of the same command,
# 3./4. FETCH_SIZE / WRITE_SIZE counter passes, 5. SQ counter pass (MFMA busy, LDS conflicts, wave / wait cycles) --
# every counter pass is its own run with --kernel-trace only (never combined with the sys / hip trace domains).
set -e
tag=${1:-r02}
shift || true
extra="$@"
quiet="--no-cpu-baseline --no-early-exit-line --no-throughput-line"
case "$extra" in *--workload*) quiet="";; esac
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
python bench.py --steps 10 --warmup 2 $extra > "$out/bench.json" 2> "$out/bench.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o bench -- python /root/repo/bench.py --steps 5 --warmup 2 $quiet $extra > "$out/bench_under_rocprof.json" 2> "$out/rocprof.err"
noprof="--no-profile"
case "$extra" in *--workload*) noprof="";; esac
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o p -- python /root/repo/bench.py --steps 2 --warmup 1 $quiet $noprof $extra > "$out/pmc_fetch.json" 2> "$out/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o p -- python /root/repo/bench.py --steps 2 --warmup 1 $quiet $noprof $extra > "$out/pmc_write.json" 2> "$out/pmc_write.err"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$out/pmc_sq" -o p -- python /root/repo/bench.py --steps 2 --warmup 1 $quiet $noprof $extra > "$out/pmc_sq.json" 2> "$out/pmc_sq.err" || echo "SQ pass failed (see pmc_sq.err)"
cd /root/repo && rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --output-format csv -d "$out/pmc_sq2" -o p -- python /root/repo/bench.py --steps 2 --warmup 1 $quiet $noprof $extra > "$out/pmc_sq2.json" 2> "$out/pmc_sq2.err" || echo "SQ pass 2 failed (see pmc_sq2.err)"
cd /root/repo && python profiles/summarize.py "$out" "$tag" > "$out/summary.txt" 2>&1 || true
echo profiles done
