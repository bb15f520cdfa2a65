#!/bin/bash
# tools/prof_run.sh <tag> <bench.py arguments...>
# One workload under rocprofv3 on the GPU box: a --kernel-trace --stats pass, then the counter passes
# (each in its own run with --kernel-trace only: SQ set, FETCH_SIZE, WRITE_SIZE, TCC set), as
# MI355X_MICROARCH.md prescribes.  Raw output under gpurun_out/prof/<tag>/, summaries under
# gpurun_out/prof/<tag>/summary/ -- copy those into profiles/.
set -o pipefail
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/prof/$tag
rm -rf "$out" && mkdir -p "$out/summary"
B="python3 bench.py --no-cpu-baseline --repeats 1 --latency-solves 0 --sustained-s 0 --event-solves 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- $B --steps 100 --warmup 10 > "$out/bench_trace.json" 2> "$out/trace.err" || { tail -5 "$out/trace.err"; exit 1; }
python3 tools/prof_summary.py stats "$out/trace" "$out/summary/kernel_stats.csv"
PM="--steps 20 --warmup 5"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d "$out/sq" -- $B $PM > /dev/null 2> "$out/sq.err" || { tail -5 "$out/sq.err"; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- $B $PM > /dev/null 2> "$out/fetch.err" || { tail -5 "$out/fetch.err"; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write" -- $B $PM > /dev/null 2> "$out/write.err" || { tail -5 "$out/write.err"; exit 1; }
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/tcc" -- $B $PM > /dev/null 2> "$out/tcc.err" || { tail -5 "$out/tcc.err"; exit 1; }
# optional fifth pass: matrix-pipe counters (names differ between ROCm releases: skipped when the profiler refuses them)
extra=""
if rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_TRANS_F32 SQ_BUSY_CU_CYCLES --output-format csv -d "$out/mfma" -- $B $PM > /dev/null 2> "$out/mfma.err"; then extra="$out/mfma"; else echo "matrix-pipe counter pass skipped: $(tail -1 "$out/mfma.err")"; rm -rf "$out/mfma"; fi
PROF_BENCH_JSON="$out/bench_trace.json" python3 tools/prof_summary.py pmc "$out/summary/pmc.json" "rocprofv3 --kernel-trace --pmc (4 separate passes: SQ set | FETCH_SIZE | WRITE_SIZE | TCC set) -- $B $PM" "$out/sq" "$out/fetch" "$out/write" "$out/tcc" $extra
