# whole training steps (forward, backward, clip, AdamW) under a steady background load against the same run alone on the GPU:
# per step loss / gradient checksum / clipped-gradient checksum / parameter checksum must be the solo run's, bit for bit
cases=("train CFG1 4 fp32 8" "train CFG1 4 bf16 8" "train CFG1 4 fp16 8" "train CFG4_SMALL 2 fp32 6" "train CFG4_SMALL 2 bf16 6" "train DEFAULT3 2 fp32 4" "train DEFAULT3 2 bf16 4")
declare -a refs
for i in "${!cases[@]}"; do refs[$i]=$(python tools/race_probe.py ${cases[$i]} 2>/dev/null); done
python tools/race_probe.py fwd DEFAULT3 2 fp32 1000000 > /dev/null 2>&1 & L1=$!
python tools/race_probe.py fwd DEFAULT3 4 bf16 1000000 > /dev/null 2>&1 & L2=$!
sleep 25
for i in "${!cases[@]}"; do
  bad=0
  for rep in 1 2 3 4 5; do
    got=$(python tools/race_probe.py ${cases[$i]} 2>/dev/null)
    if [ "$got" != "${refs[$i]}" ]; then
      bad=$((bad+1))
      [ $bad -eq 1 ] && paste <(echo "${refs[$i]}" | tr ' ' '\n') <(echo "$got" | tr ' ' '\n') | awk '$1 != $2' | head -2
    fi
  done
  echo "${cases[$i]}: $bad of 5 loaded runs differ from the solo run"
done
kill $L1 $L2; wait $L1 $L2 2>/dev/null; true
