# FETCH_SIZE of k_step_fe with the next-slab prefetch (24 of 32 chunks per lane) against the default build
cd /tmp && export TMPDIR=/tmp
for v in base pf24; do
  rm -rf /tmp/pmc
  if [ $v = base ]; then LIB=""; else LIB="DPGO_HIP_LIB=$GRAFT_REPO_ROOT/profiles/experiments/build/$v/libdpgo_hip.so"; fi
  env $LIB rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc -o p -- python $GRAFT_REPO_ROOT/profiles/experiments/rgd_run.py 600 > /tmp/pmc.log 2>&1
  echo "== $v"; python $GRAFT_REPO_ROOT/profiles/pmc_query.py /tmp/pmc/p_results.db | grep "k_step_fe"
done
