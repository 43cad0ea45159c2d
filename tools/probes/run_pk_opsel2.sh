# synthetic neighbours (tools/probes/probe_neighbour.hip) next to the packed-op probes
# (tools/probes/probe_lds_read2: 13 = v_pk_fma_f32 op_sel:[0,1,0]; 16 = v_pk_fma_f16 op_sel:[0,1,0]; 8, 17 = the unaffected op_sel_hi forms)
P=tools/probes/bin/probe_lds_read2
show() { grep -E "mode (8|13|16|17) " | sed -e 's/launches with wrong sums/wrong/' -e 's/; by lane quarter/ | lanes/' | cut -c1-130; }
echo "== alone"; $P 100 | show
for m in ${NEIGHBOURS:-0 1 2 3 4 5}; do
  echo "== next to probe_neighbour mode $m"
  tools/probes/bin/probe_neighbour $m > /dev/null 2>&1 & Q=$!; sleep 4
  $P 120 | show
  kill $Q 2>/dev/null; wait $Q 2>/dev/null
done
true
