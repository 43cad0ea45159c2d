# tools/probes/probe_lds_mix.hip / probe_lds_read2.hip alone and next to two loader processes (binaries prebuilt into tools/probes/bin/)
for v in base VOL; do echo "alone  $v: $(tools/probes/bin/probe_lds_mix_$v 200)"; done
echo "alone:"; tools/probes/bin/probe_lds_read2 100
python tools/race_probe.py fwd DEFAULT3 2 fp32 1000000 > /dev/null 2>&1 & L1=$!
python tools/race_probe.py fwd DEFAULT3 4 bf16 1000000 > /dev/null 2>&1 & L2=$!
sleep 25
for v in base VOL FZ; do echo "loaded $v: $(tools/probes/bin/probe_lds_mix_$v 300)"; done
echo "loaded:"; tools/probes/bin/probe_lds_read2 300
kill $L1 $L2; wait $L1 $L2 2>/dev/null; true
