ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05e; SCR=/tmp/prof_r05e; mkdir -p $OUT $SCR
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  L=$(echo $C | tr A-Z a-z | sed 's/_size//')
  rocprofv3 --pmc $C --kernel-trace -d $SCR/pmc_$L -o bench -- python $ROOT/bench.py --steps 200 --warmup 20 > $OUT/pmc_$L.log 2>&1
  echo "$C rc=$?"
  if [ ! -f $SCR/pmc_$L/bench_results.db ]; then
    DPGO_NO_POOL=1 rocprofv3 --pmc $C --kernel-trace -d $SCR/pmc_$L -o bench -- python $ROOT/bench.py --steps 200 --warmup 20 > $OUT/pmc_$L.log 2>&1
    echo "$C (no pool) rc=$?"
  fi
  python $ROOT/profiles/pmc_query.py $SCR/pmc_$L/bench_results.db > $OUT/pmc_$L.md
done
head -20 $OUT/pmc_fetch.md
