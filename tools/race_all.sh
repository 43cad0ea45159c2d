# every leg of the engine under a steady background load (two other processes running forwards on the same GPU):
# repeated runs of one process must agree with each other (RACE_DISTINCT) -- inference forwards, training forwards, single ops
python tools/race_probe.py fwd DEFAULT3 2 fp32 1000000 > /dev/null 2>&1 & L1=$!
python tools/race_probe.py fwd DEFAULT3 4 bf16 1000000 > /dev/null 2>&1 & L2=$!
sleep 25
export RACE_DISTINCT=1
python tools/race_ops.py 200 2>&1 | grep -v amdgpu.ids
for args in "tfwd CFG1 4 fp32 200" "tfwd CFG1 4 bf16 200" "tfwd DEFAULT3 2 fp32 30" "fwd CFG1 4 fp32 200" "fwd DEFAULT3 2 fp32 30" "fwd DEFAULT3 2 bf16 30" "fwd CFG4_SMALL 2 fp32 100"; do
  python tools/race_probe.py $args 2>&1 | grep -v amdgpu.ids
done
kill $L1 $L2; wait $L1 $L2 2>/dev/null; true
