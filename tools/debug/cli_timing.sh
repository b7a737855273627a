cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sys, os, tempfile, subprocess, time
sys.path.insert(0, os.getcwd())
from paramugsy_amd import synth
tmp = tempfile.mkdtemp(prefix="cli")
w = synth.make_workload(tmp, 20261003, n_left=4, n_right=4, genome_len=1000000, n_blocks=2500, n_deltas=16, entries_per_delta=6000, mean_len=1500)
env = dict(os.environ, PM_TIMING="1", PARAMUGSY_SERVE_SOCKET="none")
for i in range(3):
    t = time.perf_counter()
    r = subprocess.run([os.path.join(os.getcwd(), "bin", "m_translate"), w.left_dir, w.right_dir, w.list_path, os.path.join(tmp, "out.delta")], env=env, capture_output=True, text=True)
    print("run", i, "rc", r.returncode, "wall %.3f s" % (time.perf_counter() - t))
    print(r.stderr[-3000:])
PY
