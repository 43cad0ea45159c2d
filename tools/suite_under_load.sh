# the whole GPU suite next to two other processes looping forwards on the same GPU (a kernel with a latent LDS / ordering hazard
# passes alone and fails here: DESIGN section 10, profiles/r04_race_under_load.txt)
python tools/race_probe.py fwd DEFAULT3 2 fp32 1000000 > /dev/null 2>&1 & L1=$!
python tools/race_probe.py fwd DEFAULT3 4 bf16 1000000 > /dev/null 2>&1 & L2=$!
sleep 25
python -m pytest tests -m gpu -q -p no:cacheprovider "$@" 2>&1 | tail -25
kill $L1 $L2; wait $L1 $L2 2>/dev/null; true
