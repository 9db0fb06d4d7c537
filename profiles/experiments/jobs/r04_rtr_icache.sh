# instruction-cache and issue counters of the one-launch RTR solve (sphere2500 / 5, RTR + Nesterov)
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST_ANY\|SQ_INSTS_VALU\b\|SQ_INST_CYCLES_VMEM\|SQ_ACTIVE_INST_VALU\|SQ_WAVE_CYCLES\|SQ_WAIT_ANY\|SQ_INSTS_LDS\|SQ_ACTIVE_INST_LDS\|SQ_INSTS_SALU\|SQ_BUSY_CYCLES\|SQ_ACTIVE_INST_ANY" | sort | uniq | tr '\n' ' '
echo
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_IFETCH"; do
  rm -rf /tmp/pmc
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc -o p -- python $GRAFT_REPO_ROOT/profiles/experiments/rtr_run.py 60 > /tmp/pmc.log 2>&1
  tail -2 /tmp/pmc.log | cut -c1-150
  python $GRAFT_REPO_ROOT/profiles/pmc_query.py /tmp/pmc/p_results.db | grep "k_rtr_solve\|k_eval<\|kernel"
done
