"""Build libsatt_hip.so for gfx950 with hipcc (in-tree; the .so travels with the repo snapshot to the GPU box)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["gemm.hip", "gemm_tile.hip", "flash.hip", "small_attn.hip", "elementwise.hip", "highway.hip", "lstm.hip", "lstm_cluster.hip", "attn_rnn.hip", "attn_cluster.hip", "decode.hip", "decode_mega2.hip", "api.hip"]
OUT = os.path.join(os.path.dirname(HERE), "libsatt_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc"] + os.environ.get("SATT_EXTRA_FLAGS", "").split()


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


IO_OUT = os.path.join(os.path.dirname(HERE), "libsatt_io.so")


def io_stale():
    """the library is missing or older than its source / header.  A source or header that is NOT THERE (an installed package that
    ships the prebuilt library without csrc/ or include/) cannot make an existing library stale."""
    src = os.path.join(HERE, "host_io.c")
    hdr = os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "satt_io.h")
    if not os.path.exists(IO_OUT):
        return True
    return any(os.path.exists(f) and _newer(f, IO_OUT) for f in (src, hdr))


def build_io(force=False):
    """the host-side input-pipeline library (include/satt_io.h): plain C, gcc, no ROCm dependency.  Several processes (DP ranks,
    test workers) may find it missing at the same time: the build runs under an exclusive file lock, gcc writes a temporary file
    and os.replace() publishes it - nobody can CDLL a half-written library."""
    import fcntl
    src = os.path.join(HERE, "host_io.c")
    if not (force or io_stale()):
        return IO_OUT
    with open(IO_OUT + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or io_stale():                  # (another process may have built it while this one waited for the lock)
                tmp = "%s.tmp.%d" % (IO_OUT, os.getpid())
                cmd = [os.environ.get("CC", "gcc"), "-O3", "-std=gnu11", "-fPIC", "-shared", "-pthread", "-Wall", "-Wextra", src, "-o", tmp]
                r = subprocess.run(cmd, capture_output=True, text=True)
                if r.returncode != 0:
                    if os.path.exists(tmp):
                        os.remove(tmp)
                    raise RuntimeError("gcc failed: %s\n%s" % (" ".join(cmd), r.stderr))
                os.replace(tmp, IO_OUT)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return IO_OUT


def build(force=False, verbose=False):
    build_io(force)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(HERE, h) for h in ("common.h", "matvec.h", "attn_common.h", "mfma_rec.h", "cluster_xchg.h", "gemm_tile.h")] + \
              [os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "satt_hip.h")]
    jobs = []
    for s in SOURCES:
        src = os.path.join(HERE, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _newer(src, obj) or any(_newer(h, obj) for h in headers):
            jobs.append((src, obj))

    def cc(job):
        cmd = [hipcc] + FLAGS + ["-c", job[0], "-o", job[1]]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stderr))
        if verbose and r.stderr:
            print(r.stderr, file=sys.stderr)
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if jobs or not os.path.exists(OUT):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", OUT]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stderr))
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
