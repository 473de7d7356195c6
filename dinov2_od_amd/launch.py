"""One process per GPU on one node -- the launch the reference does with mp.spawn (train.py:1501-1506).

`spawn_ranks(script, argv, n)` starts n FRESH interpreter processes of `script` with the torchrun-style environment
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT), relays rank 0's stdout (the one JSON line of
bench.py) and every rank's stderr, waits, and returns the first non-zero exit code (terminating the other ranks when one
fails).  The parent never touches the GPU: nothing here imports torch, and children are new processes, not re-execs of a
process that has initialised HIP.
"""
import os
import socket
import subprocess
import sys
import threading
import time


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, world))))
    return env


def _pump(stream, sink, prefix=""):
    for line in iter(stream.readline, ""):
        sink.write(prefix + line)
        sink.flush()
    stream.close()


def spawn_ranks(script, argv, n, timeout=None, stdout=None, stderr=None):
    """returns the job's exit code (0 = every rank exited 0)"""
    stdout = stdout or sys.stdout
    stderr = stderr or sys.stderr
    port = free_port()
    procs, threads = [], []
    for r in range(n):
        p = subprocess.Popen([sys.executable, script] + list(argv), env=rank_env(r, n, port),
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, bufsize=1)
        procs.append(p)
        # rank 0's stdout is the job's stdout; other ranks' stdout is diagnostic only
        threads.append(threading.Thread(target=_pump, args=(p.stdout, stdout if r == 0 else stderr, "" if r == 0 else f"[rank {r}] "), daemon=True))
        threads.append(threading.Thread(target=_pump, args=(p.stderr, stderr, f"[rank {r}] " if n > 1 else ""), daemon=True))
    for t in threads:
        t.start()
    t0, rc = time.time(), 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            c = procs[r].poll()
            if c is None:
                continue
            live.discard(r)
            if c != 0 and rc == 0:
                rc = c
                for o in live:              # one rank died: the others would hang in the next collective
                    procs[o].terminate()
        if timeout is not None and time.time() - t0 > timeout and live:
            rc = rc or 124
            for o in live:
                procs[o].kill()
        time.sleep(0.05)
    for t in threads:
        t.join(timeout=5)
    return rc
