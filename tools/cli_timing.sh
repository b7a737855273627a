cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys; sys.path.insert(0,'.')
from paramugsy_amd import synth
synth.make_workload('/tmp/wbig', 20261003, n_left=4, n_right=4, genome_len=1000000, n_blocks=2500, n_deltas=16, entries_per_delta=6000, mean_len=1500)
PY
for i in 1 2 3 4; do time env PM_TIMING=1 ./bin/m_translate /tmp/wbig/profiles-l /tmp/wbig/profiles-r /tmp/wbig/nucmer.list /tmp/wbig/out.delta; done
time ./oracle/_ref/m_translate /tmp/wbig/profiles-l /tmp/wbig/profiles-r /tmp/wbig/nucmer.list /tmp/wbig/ref.delta
cmp /tmp/wbig/out.delta /tmp/wbig/ref.delta && echo IDENTICAL
echo "== dynamic loader statistics"
LD_DEBUG=statistics ./bin/m_translate /tmp/wbig/profiles-l /tmp/wbig/profiles-r /tmp/wbig/nucmer.list /tmp/wbig/out.delta 2>&1 | grep -i "total startup\|relocation\|load" | head
echo "== the same job over a device list {0,0}"
for i in 1 2; do time env PARAMUGSY_DEVICES=0,0 ./bin/m_translate /tmp/wbig/profiles-l /tmp/wbig/profiles-r /tmp/wbig/nucmer.list /tmp/wbig/out2.delta; done
cmp /tmp/wbig/out2.delta /tmp/wbig/ref.delta && echo IDENTICAL
