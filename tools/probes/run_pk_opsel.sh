# what kind of neighbour makes v_pk_*_f32 with op_sel:[0,1] go wrong on lanes 48-63?  (tools/probes/probe_lds_read2 modes 13, 15; 8 = control)
P=tools/probes/bin/probe_lds_read2
show() { grep -E "mode (8|13) " | sed -e 's/launches with wrong sums/wrong/' -e 's/; by lane quarter/ | lanes/' | cut -c1-130; }
for nb in "$@"; do
  echo "== next to: $nb"
  python tools/probes/neighbour.py $nb > /dev/null 2>&1 & Q=$!; sleep 22
  if kill -0 $Q 2>/dev/null; then $P 120 | show; else echo "   (neighbour exited)"; fi
  kill $Q 2>/dev/null; wait $Q 2>/dev/null
done
true
