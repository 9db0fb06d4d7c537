# LDS and issue counters of the main kernels over the default bench command
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_LDS[A-Z_]*\|SQ_INSTS_LDS\|SQ_ACTIVE_INST_LDS\|SQ_WAIT_INST_LDS\|SQ_LDS_UNALIGNED[A-Z_]*\|SQ_LDS_MEM_VIOLATIONS\|SQ_LDS_ADDR_CONFLICT\|SQ_LDS_DATA_FIFO_FULL" | sort | uniq | tr '\n' ' '; echo
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
  rm -rf /tmp/pmc
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 > /tmp/pmc.log 2>&1
  tail -1 /tmp/pmc.log | cut -c1-120
  python $GRAFT_REPO_ROOT/profiles/pmc_query.py /tmp/pmc/p_results.db | grep "k_step_fe\|k_step_fd\|k_rtr_solve<5, false\|k_precond<5, 3, 2048, false, true, false>\|k_precond<5, 3, 1024\|k_eval<5>\|k_eval_staged\|k_nest_pre\|kernel |"
done
