B="python bench.py --no-cpu-baseline --no-exact --no-configs34 --no-batch4 --no-two-streams --steps 300"
run() { echo "$1: $(env $2 timeout 300 $B 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"; }
for i in 1 2; do
run "default" "X=1"
run "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=1"
run "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=0"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
run "GPU_MAX_HW_QUEUES=1" "GPU_MAX_HW_QUEUES=1"
run "HSA_ENABLE_INTERRUPT=0" "HSA_ENABLE_INTERRUPT=0"
done
