ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pb -o x -- python -X faulthandler $ROOT/bench.py --steps 200 --warmup 20 > /tmp/pb.log 2>&1
echo rc=$?
grep -n "Fatal Python error" -A 25 /tmp/pb.log | head -60
