for i in 1 2; do
for v in 0 on; do
  if [ $v = 0 ]; then export MV2_GATHER_DMA=0; else unset MV2_GATHER_DMA; fi
  echo "gather=$v B1: $(timeout 300 python tools/bench_exp.py --no-cpu-baseline --no-exact --no-configs34 --no-batch4 --no-two-streams --steps 300 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
  echo "gather=$v B4: $(timeout 300 python tools/bench_exp.py --config 2 --steps 100 --no-cpu-baseline --no-exact --no-configs34 --no-batch4 --no-two-streams 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
done; done
unset MV2_GATHER_DMA
for v in 0 on; do
  if [ $v = 0 ]; then export MV2_GATHER_DMA=0; else unset MV2_GATHER_DMA; fi
  echo "gather=$v cfg3: $(timeout 300 python tools/bench_exp.py --config 3 --steps 10 --no-cpu-baseline --no-exact --no-configs34 --no-batch4 --no-two-streams 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
done
