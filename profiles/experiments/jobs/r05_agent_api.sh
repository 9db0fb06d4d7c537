python - <<'P'
import bench, subprocess, os
exe = bench.build_agent_api_bench()
data = os.path.join(bench.ROOT, "data", "sphere2500.g2o")
for args in (["5","1","1","400","0.2","20","0.5"], ["5","0","1","100","0.2","50","0.01"]):
    for rep in range(2):
        print(subprocess.check_output([exe, data]+args, text=True).strip())
P
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_b20.json 2>gpurun_out/r05_b20.err
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r05_b20.json') if l.startswith('{')][-1])
print(d['value'], d['timing'])
P
