#!/usr/bin/env python3
"""Wall time of one Mugsy_profile node's make + make + translate prefix (lib/base/mugsy_profiles_task.ml:40-58) on the bench
job: three processes (as the task script runs them), `mugsy_profiles stage` (one process), and a resident `mugsy_profiles serve`
worker (one HIP context for many nodes); upstream's m_translate alone for scale.  Outputs are compared byte for byte."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from paramugsy_amd import synth  # noqa: E402

EXE = os.path.join(ROOT, "bin", "mugsy_profiles")


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    tmp = tempfile.mkdtemp(prefix="pm_stage_")
    rng = np.random.default_rng(20261003)
    lg = ["L%d.chr" % k for k in range(4)]
    rg = ["R%d.chr" % k for k in range(4)]
    glen, blocks = int(1000000 * scale), int(2500 * scale)
    for side, names in (("l", lg), ("r", rg)):
        open(os.path.join(tmp, side + ".maf"), "w").write(synth.side_to_maf_text(synth.gen_side(rng, names, glen, blocks)))
    paths = []
    for d in range(16):
        p = os.path.join(tmp, "n%d.delta" % d)
        open(p, "w").write(synth.gen_delta_text(rng, lg, rg, glen, glen, int(6000 * scale), mean_len=1500))
        paths.append(p)
    open(os.path.join(tmp, "nucmer.list"), "w").write("".join(p + "\n" for p in paths))
    out = {"maf_MB": [round(os.path.getsize(os.path.join(tmp, s + ".maf")) / 1e6, 1) for s in "lr"],
           "delta_MB": round(sum(os.path.getsize(p) for p in paths) / 1e6, 1)}

    def three(tag):
        t0 = time.perf_counter()
        for side in "lr":
            subprocess.run([EXE, "make", "-in_maf", side + ".maf", "-out_dir", "%s-%s" % (tag, side), "-basename", side], cwd=tmp, check=True)
        subprocess.run([EXE, "translate", "-profiles_left", tag + "-l", "-profiles_right", tag + "-r", "-nucmer_list", "nucmer.list",
                        "-out_delta", tag + ".delta"], cwd=tmp, check=True)
        return time.perf_counter() - t0

    def stage_args(tag):
        return ["stage", "-left_maf", "l.maf", "-left_dir", tag + "-l", "-left_basename", "l", "-right_maf", "r.maf", "-right_dir", tag + "-r",
                "-right_basename", "r", "-nucmer_list", "nucmer.list", "-out_delta", tag + ".delta"]

    def stage(tag):
        t0 = time.perf_counter()
        subprocess.run([EXE] + stage_args(tag), cwd=tmp, check=True)
        return time.perf_counter() - t0

    out["three_processes_s"] = min(three("p3"), three("p3"))
    out["stage_one_process_s"] = min(stage("p1"), stage("p1"))
    body = lambda f: open(os.path.join(tmp, f), "rb").read().split(b"\n", 1)[1]
    out["stage_bytes_equal_three"] = body("p1.delta") == body("p3.delta")
    # resident worker: time per node once the worker is up (5 nodes, the first one excluded as warm-up of buffers)
    w = subprocess.Popen([EXE, "serve"], cwd=tmp, stdin=subprocess.PIPE, stdout=subprocess.PIPE)
    times = []
    for k in range(6):
        t0 = time.perf_counter()
        w.stdin.write(("\t".join(stage_args("srv")) + "\n").encode())
        w.stdin.flush()
        ans = w.stdout.readline().decode().strip()
        times.append(time.perf_counter() - t0)
        assert ans == "done 0", ans
    w.stdin.write(b"quit\n")
    w.stdin.flush()
    w.wait()
    out["resident_worker_s_per_node"] = min(times[1:])
    out["resident_first_node_s"] = times[0]
    out["resident_bytes_equal_three"] = body("srv.delta") == body("p3.delta")
    ref = os.path.join(ROOT, "oracle", "_ref", "m_translate")
    if os.path.exists(ref):
        t0 = time.perf_counter()
        subprocess.run([ref, "p3-l", "p3-r", "nucmer.list", "ref.delta"], cwd=tmp, check=True)
        out["upstream_m_translate_alone_s"] = time.perf_counter() - t0
        os.rename(os.path.join(tmp, "p3-l"), os.path.join(tmp, "x-l"))
        out["upstream_bytes_equal"] = body("ref.delta") == body("p3.delta")
    print(json.dumps(out))
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
