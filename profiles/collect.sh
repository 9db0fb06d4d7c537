#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: kernel-trace stats of the default bench command,
# then two separate PMC passes (FETCH_SIZE, WRITE_SIZE) as MI355X_MICROARCH.md prescribes.
# Only the text summaries are kept (the rocpd databases exceed the gpurun_out quota).
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$1
SCR=/tmp/prof_$1
mkdir -p $OUT $SCR
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $SCR/trace -o bench -- python $ROOT/bench.py --steps 1000 --warmup 100 > $OUT/bench_traced.log 2>&1
python $ROOT/profiles/prof_query.py $SCR/trace/bench_results.db > $OUT/kernel_stats.md
python $ROOT/profiles/prof_query.py $SCR/trace/bench_results.db k_step_fd 24 | tail -24 > $OUT/timeline.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $SCR/pmc_fetch -o bench -- python $ROOT/bench.py --steps 200 --warmup 20 > $OUT/pmc_fetch.log 2>&1
python $ROOT/profiles/pmc_query.py $SCR/pmc_fetch/bench_results.db > $OUT/pmc_fetch.md
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $SCR/pmc_write -o bench -- python $ROOT/bench.py --steps 200 --warmup 20 > $OUT/pmc_write.log 2>&1
python $ROOT/profiles/pmc_query.py $SCR/pmc_write/bench_results.db > $OUT/pmc_write.md
cd $ROOT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench_traced.log | tail -2 | cut -c1-200
head -30 $OUT/pmc_fetch.md
