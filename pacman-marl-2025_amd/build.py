"""Builds libpmx_hip.so (the C ABI of include/pmx.h + the gfx950 kernels) in-tree with hipcc."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpmx_hip.so")
SOURCES = ["pmx_step.hip", "pmx_api.hip", "pmx_train.hip"]
HEADERS = ["pmx_device.h", os.path.join("..", "..", "include", "pmx.h")]
# -ffp-contract=off: the float64 reward sums and the float32 GAE scan must round like the reference's Python/torch ops
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the pmx HIP library cannot be built")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile in-tree.  Serialised by a lock file and written through a temporary name, so that several ranks of one
    job importing the package at once (torch.distributed.run) cannot corrupt the library."""
    import fcntl
    if not force and not stale():
        return LIB
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not stale():          # another process built it while we waited
                return LIB
            tmp = LIB + ".tmp.%d" % os.getpid()
            cmd = [hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", tmp]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd, cwd=CSRC)
            os.replace(tmp, LIB)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
