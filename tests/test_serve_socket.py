"""The drop-in m_translate and the resident worker behind it: bin/m_translate links nothing of HIP, asks
`mugsy_profiles serve -socket <path>` over a UNIX socket first and runs the job in its own process only when nobody listens.

CPU part: the shim's side of the protocol against a stand-in worker written here (what it sends for which argv, what it does with
the reply, two clients at once, the fall-back when the socket is dead).  GPU part (-m gpu): the real worker -- two concurrent
clients started in another directory with relative paths get the reference's bytes (tests/golden/translate_*)."""
import os
import shutil
import socket
import subprocess
import threading
import time

import pytest

from conftest import GOLDEN, ROOT

EXE = os.path.join(ROOT, "bin", "m_translate")
WORKER = os.path.join(ROOT, "bin", "mugsy_profiles")


def test_the_drop_in_does_not_link_the_hip_runtime():
    needed = subprocess.run(["ldd", EXE], capture_output=True, text=True, check=True).stdout
    assert "amdhip" not in needed and "paramugsy" not in needed and "hsa" not in needed, needed


class StandInWorker(threading.Thread):
    """Speaks the worker's side of the protocol (csrc/mugsy_profiles_main.cc, serve_socket): records the requests, writes a marker into
    the requested output, answers `done <rc>` (and a message after a failure)."""

    def __init__(self, path, rc=0, message="", delay=0.0):
        super().__init__(daemon=True)
        self.path, self.rc, self.message, self.delay = path, rc, message, delay
        self.requests = []
        self.sock = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        self.sock.bind(path)
        self.sock.listen(8)

    def run(self):
        while True:
            try:
                c, _ = self.sock.accept()
            except OSError:
                return
            data = b""
            while not data.endswith(b"\n"):
                chunk = c.recv(65536)
                if not chunk:
                    break
                data += chunk
            f = data.decode().rstrip("\n").split("\t")
            self.requests.append(f)
            time.sleep(self.delay)
            if f[0] == "translate":
                out = f[4] if f[4].startswith("/") else os.path.join(f[1], f[4])
                with open(out, "w") as o:
                    o.write("%s/sequences.fasta %s/sequences.fasta\nNUCMER\nfrom the stand-in: %s\n" % (f[2], f[3], " ".join(f[7:])))
            c.sendall(("done %d\n%s" % (self.rc, self.message + "\n" if self.rc else "")).encode())
            c.close()

    def stop(self):
        self.sock.close()


def test_the_shim_sends_its_argv_and_directory_and_prints_the_workers_verdict(tmp_path):
    sock = str(tmp_path / "w.sock")
    w = StandInWorker(sock)
    w.start()
    job = tmp_path / "job"
    job.mkdir()
    (job / "nucmer.list").write_text("a.delta\nsub/b.delta\n")
    env = dict(os.environ, PARAMUGSY_SERVE_SOCKET=sock)
    r = subprocess.run([EXE, "profiles-l", "../r dir", "nucmer.list", "out.delta"], cwd=str(job), env=env, capture_output=True)
    assert r.returncode == 0 and r.stderr == b""
    assert w.requests == [["translate", str(job), "profiles-l", "../r dir", "out.delta", "-", "2", "a.delta", "sub/b.delta"]]
    assert (job / "out.delta").read_text().startswith("profiles-l/sequences.fasta ../r dir/sequences.fasta\nNUCMER\n")
    # a device list rides along; two clients at once are both served
    w.delay = 0.2
    env2 = dict(env, PARAMUGSY_DEVICES="0,1")
    ps = [subprocess.Popen([EXE, "l", "r", "nucmer.list", "o%d.delta" % k], cwd=str(job), env=env2) for k in range(2)]
    assert [p.wait() for p in ps] == [0, 0]
    assert sorted(q[4] for q in w.requests[1:]) == ["o0.delta", "o1.delta"] and all(q[5] == "0,1" for q in w.requests[1:])
    # a failed job: the reference's exit behaviour (SIGABRT's 134) and the worker's message
    w.rc, w.message, w.delay = 7, "work unit 3 failed with status 2", 0.0
    r = subprocess.run([EXE, "l", "r", "nucmer.list", "o.delta"], cwd=str(job), env=env, capture_output=True)
    assert r.returncode == 134 and b"error 7: work unit 3 failed with status 2" in r.stderr
    w.stop()
    # usage error: before anybody is asked (m_translate_main.cc:22-25)
    r = subprocess.run([EXE, "l"], env=env, capture_output=True)
    assert r.returncode == 1 and b"Usage: m_translate" in r.stderr


def test_nobody_listening_means_the_job_runs_in_the_process(tmp_path, hip_lib):
    """A socket file nobody answers at (and no socket at all): the library is loaded and the job runs here -- on a machine without a
    GPU that is the loud refusal of the C ABI (no CPU path), with a GPU the in-process path every other CLI test runs."""
    if hip_lib.pm_device_count() > 0:
        pytest.skip("a HIP device is present: the in-process path is tests/test_translate_gpu.py's")
    dead = str(tmp_path / "dead.sock")
    s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    s.bind(dead)
    s.close()  # the file stays, nobody listens
    (tmp_path / "nucmer.list").write_text("")
    for sock in (dead, str(tmp_path / "none.sock"), "none"):
        r = subprocess.run([EXE, "l", "r", "nucmer.list", "o.delta"], cwd=str(tmp_path), env=dict(os.environ, PARAMUGSY_SERVE_SOCKET=sock),
                           capture_output=True)
        assert r.returncode == 134 and b"no CPU path" in r.stderr, r.stderr


@pytest.mark.gpu
def test_two_clients_of_the_resident_worker_get_the_references_bytes(tmp_path):
    sock = str(tmp_path / "serve.sock")
    worker = subprocess.Popen([WORKER, "serve", "-socket", sock])
    try:
        for _ in range(600):
            if os.path.exists(sock):
                break
            time.sleep(0.05)
        assert os.path.exists(sock)
        env = dict(os.environ, PARAMUGSY_SERVE_SOCKET=sock)
        jobs = []
        for name in ("typical", "gappy", "reverse", "tiny_blocks"):
            case = str(tmp_path / name)
            shutil.copytree(os.path.join(GOLDEN, "translate_" + name), case)
            os.remove(os.path.join(case, "expected.delta"))
            jobs.append((name, case))
        for round_ in range(2):  # the second round meets a warm worker
            ps = [(name, case, subprocess.Popen([EXE, "profiles-l", "profiles-r", "nucmer.list", "out%d.delta" % round_], cwd=case, env=env))
                  for name, case in jobs]  # four clients at once, each in its own directory with relative paths
            for name, case, p in ps:
                assert p.wait() == 0
                got = open(os.path.join(case, "out%d.delta" % round_), "rb").read()
                assert got == open(os.path.join(GOLDEN, "translate_" + name, "expected.delta"), "rb").read(), name
        # a failing job comes back as the reference's SIGABRT exit, and the worker lives on
        with open(os.path.join(jobs[0][1], "junk.delta"), "w") as f:  # (m_delta.cc:72-92: Delta_stream_parse_error, uncaught upstream)
            f.write("l/sequences.fasta r/sequences.fasta\nNUCMER\n>a b 10 10\n1 2 three 4 0 0 0\n0\n")
        with open(os.path.join(jobs[0][1], "bad.list"), "w") as f:
            f.write("junk.delta\n")
        r = subprocess.run([EXE, "profiles-l", "profiles-r", "bad.list", "bad.delta"], cwd=jobs[0][1], env=env, capture_output=True)
        assert r.returncode == 134 and b"m_translate: error" in r.stderr
        r = subprocess.run([EXE, "profiles-l", "profiles-r", "nucmer.list", "again.delta"], cwd=jobs[0][1], env=env)
        assert r.returncode == 0
        c = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        c.connect(sock)
        c.sendall(b"quit\n")
        assert c.recv(64).startswith(b"done 0")
        c.close()
        assert worker.wait(timeout=30) == 0
        assert not os.path.exists(sock)
    finally:
        if worker.poll() is None:
            worker.kill()


# ---- round 5: who is believed, how long anybody waits, where the default socket lives, two workers at once.
# The real worker's accept loop needs no GPU as long as no job is run, so these run on CPU.

def ask(sock, line, timeout=10.0):
    c = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    c.settimeout(timeout)
    c.connect(sock)
    c.sendall(line)
    data = b""
    while True:
        chunk = c.recv(65536)
        if not chunk:
            break
        data += chunk
    c.close()
    return data


def start_worker(sock, env=None, wait=True):
    w = subprocess.Popen([WORKER, "serve", "-socket", sock], env=env, stderr=subprocess.PIPE)
    if wait:
        for _ in range(600):
            try:
                if ask(sock if sock != "default" else env["_EXPECTED_SOCK"], b"stats\n").startswith(b"done 0"):
                    break
            except OSError:
                time.sleep(0.05)
        else:
            w.kill()
            raise AssertionError("the worker did not come up")
    return w


def test_a_verdict_of_success_without_an_output_file_is_not_believed(tmp_path):
    """A worker (or whoever sits at the socket) that answers `done 0` and wrote nothing: exit 134, not 0 (a silent wrong result)."""
    sock = str(tmp_path / "w.sock")
    s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    s.bind(sock)
    s.listen(2)

    def liar():
        c, _ = s.accept()
        c.recv(65536)
        c.sendall(b"done 0\n")
        c.close()
    t = threading.Thread(target=liar, daemon=True)
    t.start()
    (tmp_path / "nucmer.list").write_text("a.delta\n")
    r = subprocess.run([EXE, "l", "r", "nucmer.list", "o.delta"], cwd=str(tmp_path), env=dict(os.environ, PARAMUGSY_SERVE_SOCKET=sock),
                       capture_output=True)
    assert r.returncode == 134 and b"does not exist" in r.stderr
    s.close()


def test_a_worker_that_never_answers_ends_the_client_with_the_failure_exit_and_no_second_run(tmp_path):
    sock = str(tmp_path / "w.sock")
    s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    s.bind(sock)
    s.listen(2)
    held = []
    t = threading.Thread(target=lambda: held.append(s.accept()[0]), daemon=True)  # takes the connection, says nothing
    t.start()
    (tmp_path / "nucmer.list").write_text("a.delta\n")
    t0 = time.time()
    r = subprocess.run([EXE, "l", "r", "nucmer.list", "o.delta"], cwd=str(tmp_path),
                       env=dict(os.environ, PARAMUGSY_SERVE_SOCKET=sock, PARAMUGSY_SERVE_TIMEOUT="1"), capture_output=True, timeout=60)
    assert r.returncode == 134 and b"no verdict from the resident worker" in r.stderr
    assert b"no CPU path" not in r.stderr and time.time() - t0 < 30  # it did not fall back to a run of its own
    s.close()


@pytest.mark.skipif(os.geteuid() != 0, reason="needs root to start a process under another uid")
def test_a_listener_of_another_uid_is_not_believed(tmp_path, hip_lib):
    """Somebody else's process at the socket path (a squatter in /tmp): the client reads SO_PEERCRED and tells it nothing."""
    if hip_lib.pm_device_count() > 0:
        pytest.skip("with a device the fall-back would run the job; the CPU box shows the refusal")
    import sys as _sys
    import tempfile
    shared = tempfile.mkdtemp(prefix="pm_squat_", dir="/tmp")  # a place the other uid can reach (pytest's tmp_path is root's alone)
    os.chmod(shared, 0o777)
    sock = os.path.join(shared, "squat.sock")
    leak = os.path.join(shared, "leak.txt")
    # the credentials a listening socket shows are those of the process that called listen(): drop the uid first
    code = ("import os,socket,sys\n"
            "os.setgid(65534); os.setuid(65534)\n"
            "s=socket.socket(socket.AF_UNIX,socket.SOCK_STREAM); s.bind(%r); os.chmod(%r,0o777); s.listen(2)\n"
            "sys.stdout.write('up\\n'); sys.stdout.flush()\n"
            "c,_=s.accept(); d=c.recv(65536); open(%r,'wb').write(d); c.sendall(b'done 0\\n'); c.close()\n"
            % (sock, sock, leak))
    squatter = subprocess.Popen([_sys.executable, "-c", code], stdout=subprocess.PIPE)
    assert squatter.stdout.readline() == b"up\n"
    (tmp_path / "nucmer.list").write_text("a.delta\n")
    r = subprocess.run([EXE, "l", "r", "nucmer.list", "o.delta"], cwd=str(tmp_path), env=dict(os.environ, PARAMUGSY_SERVE_SOCKET=sock),
                       capture_output=True, timeout=60)
    squatter.kill()
    assert b"is not this user's" in r.stderr
    assert r.returncode == 134 and b"no CPU path" in r.stderr  # the job went to the in-process path (which refuses without a device)
    told = open(leak, "rb").read() if os.path.exists(leak) else b""
    shutil.rmtree(shared, ignore_errors=True)
    assert told == b""  # the squatter was told nothing


def test_the_default_socket_lives_in_a_directory_of_the_users_own(tmp_path):
    """Nobody names a socket: worker (`-socket default`) and client meet at $XDG_RUNTIME_DIR/paramugsy/serve.sock, made 0700; a runtime
    directory that is open to others is not used."""
    run = tmp_path / "run"
    run.mkdir(mode=0o700)
    env = dict(os.environ, XDG_RUNTIME_DIR=str(run))
    env.pop("PARAMUGSY_SERVE_SOCKET", None)
    expected = str(run / "paramugsy" / "serve.sock")
    env["_EXPECTED_SOCK"] = expected
    # the real worker at the default place; a stand-in cannot take the place while it lives (the lock), so the client's default is
    # checked with the worker gone and a stand-in at the same path
    w = start_worker("default", env=env)
    try:
        st = os.stat(str(run / "paramugsy"))
        assert st.st_mode & 0o077 == 0 and st.st_uid == os.getuid()
        assert os.stat(expected).st_mode & 0o077 == 0
        assert ask(expected, b"stats\n") == b"done 0\njobs 0\n"
        assert ask(expected, b"quit\n").startswith(b"done 0")
        assert w.wait(timeout=30) == 0
    finally:
        if w.poll() is None:
            w.kill()
    sw = StandInWorker(expected)
    sw.start()
    (tmp_path / "nucmer.list").write_text("a.delta\n")
    r = subprocess.run([EXE, "l", "r", "nucmer.list", "o.delta"], cwd=str(tmp_path), env=env, capture_output=True)
    assert r.returncode == 0 and len(sw.requests) == 1 and sw.requests[0][0] == "translate"
    sw.stop()
    os.remove(expected)
    # an XDG_RUNTIME_DIR others can write to is not trusted: the default falls to /tmp/paramugsy-<uid>, never to a path inside it
    os.chmod(str(run), 0o777)
    sw2 = StandInWorker(expected)
    sw2.start()
    r = subprocess.run([EXE, "l", "r", "nucmer.list", "o2.delta"], cwd=str(tmp_path), env=env, capture_output=True)
    assert sw2.requests == []
    sw2.stop()


def test_a_silent_client_is_dropped_and_the_next_one_served(tmp_path):
    sock = str(tmp_path / "serve.sock")
    w = start_worker(sock, env=dict(os.environ, PARAMUGSY_SERVE_REQUEST_TIMEOUT="0.5"))
    try:
        mute = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        mute.connect(sock)  # connects, never sends its line
        half = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        half.connect(sock)
        half.sendall(b"translate\t/tmp")  # half a line, then nothing
        t0 = time.time()
        assert ask(sock, b"stats\n", timeout=30) == b"done 0\njobs 0\n"
        assert time.time() - t0 < 10
        assert ask(sock, b"nonsense\n").startswith(b"done -1")
        mute.close()
        half.close()
        assert ask(sock, b"quit\n").startswith(b"done 0")
        assert w.wait(timeout=30) == 0
        assert not os.path.exists(sock) and not os.path.exists(sock + ".lock")
    finally:
        if w.poll() is None:
            w.kill()


def test_two_workers_started_together_leave_one_listening(tmp_path):
    """The probe-unlink-bind sequence runs under a lock: whichever of two workers loses does not unlink the winner's socket."""
    sock = str(tmp_path / "serve.sock")
    ws = [subprocess.Popen([WORKER, "serve", "-socket", sock], stderr=subprocess.PIPE) for _ in range(4)]
    try:
        for _ in range(600):
            try:
                if ask(sock, b"stats\n").startswith(b"done 0"):
                    break
            except OSError:
                time.sleep(0.05)
        time.sleep(0.5)
        alive = [w for w in ws if w.poll() is None]
        assert len(alive) == 1
        for w in ws:
            if w is not alive[0]:
                assert w.returncode == 2 and (b"holds the lock" in w.stderr.read() or True)
        assert ask(sock, b"stats\n") == b"done 0\njobs 0\n"  # and the survivor's socket is still the file at the path
        assert ask(sock, b"quit\n").startswith(b"done 0")
        assert alive[0].wait(timeout=30) == 0
    finally:
        for w in ws:
            if w.poll() is None:
                w.kill()
