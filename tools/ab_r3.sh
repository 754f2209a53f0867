P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"].get("median_ms_per_step"))'
for rep in 1 2; do
for v in "A=1" "SCAT_WG_XCD=0" "SCAT_LIBPATH=$GRAFT_REPO_ROOT/tools/_bin/libscat_hip_slp.so"; do
  echo "== $v"; env $v timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --steps 60 --warmup 15 2>/dev/null | python -c "$P"
done; done
for v in "A=1" "SCAT_WG_XCD=0" "SCAT_LIBPATH=$GRAFT_REPO_ROOT/tools/_bin/libscat_hip_slp.so"; do
  echo "== $v"; env $v timeout -k 10 300 python tools/conv_bench.py --reps 10 --only wgrad 2>/dev/null | grep -v amdgpu > gpurun_out/r03_ab_$(echo $v | tr -c 'A-Za-z0-9' '_').txt
done
