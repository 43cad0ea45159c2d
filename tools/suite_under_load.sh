# the whole GPU suite next to other processes on the same GPU (a kernel with a latent LDS / ordering hazard passes alone and fails
# here: DESIGN section 10, profiles/r04_race_under_load.txt).  LOAD=fwd (default): two processes looping U-Net forwards
# (fp32-equivalent, bf16); LOAD=train: two processes looping whole training steps (fp32 [N,C,H,W] tape, bf16 blocked tape);
# LOAD=mix: one of each plus a batch-1 sampler.
case "${LOAD:-fwd}" in
  fwd)   L=("fwd DEFAULT3 2 fp32" "fwd DEFAULT3 4 bf16");;
  train) L=("train DEFAULT3 2 fp32" "train DEFAULT3 4 bf16");;
  mix)   L=("fwd DEFAULT3 2 fp32" "train DEFAULT3 2 bf16" "fwd DEFAULT3 1 fp32");;
esac
pids=()
for l in "${L[@]}"; do python tools/race_probe.py $l 1000000 > /dev/null 2>&1 & pids+=($!); done
sleep 30
python -m pytest tests -m gpu -q -p no:cacheprovider "$@" 2>&1 | tail -12
kill "${pids[@]}"; wait "${pids[@]}" 2>/dev/null; true
