# lockstep ASAPP ticks on tunnels: kernel stats and the dispatch timeline of a few ticks inside a graph
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/asapp -o a -- python $GRAFT_REPO_ROOT/profiles/experiments/asapp_profile.py > /tmp/asapp.log 2>&1
tail -1 /tmp/asapp.log
python $GRAFT_REPO_ROOT/profiles/prof_query.py /tmp/asapp/a_results.db 3000 12 | tail -26
