"""Build libxlxmert_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so ships with the repo snapshot."""
import os
import shutil
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# Default library: what the training step, the samplers and the task rows call.  XL_EXPERIMENTAL=1 builds libxlxmert_hip_exp.so instead:
# the same plus the kernel / schedule variants that were measured slower inside the step (256x192 tiles, persistent tiles, paired
# launches, eight-wave q tiles, the role-trading relay kernel, K split with an epilogue) and their entry points (-DXL_EXPERIMENTAL).
EXPERIMENTAL = os.environ.get("XL_EXPERIMENTAL", "0") not in ("", "0")
LIB = os.path.join(HERE, "libxlxmert_hip_exp.so" if EXPERIMENTAL else "libxlxmert_hip.so")
SOURCES = ["gemm_pp.hip", "gemm_pp_nn.hip", "gemm_pp_duo.hip", "gemm.hip", "rowops.hip", "sdpa.hip", "optim.hip", "plan.hip", "comm.hip"]
SOURCES_EXPERIMENTAL = ["gemm_pp_192.hip", "gemm_pp_persist.hip", "gemm_pp_pair.hip", "gemm_q.hip", "gemm_relay.hip"]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]
if EXPERIMENTAL:
    SOURCES = SOURCES + SOURCES_EXPERIMENTAL
    FLAGS = FLAGS + ["-DXL_EXPERIMENTAL"]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: cannot build libxlxmert_hip.so")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "xlxmert_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True):
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build", "exp" if EXPERIMENTAL else "default")
    os.makedirs(objdir, exist_ok=True)

    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
              [os.path.join(HERE, "..", "include", "xlxmert_hip.h")]

    def cc(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        deps = [os.path.join(CSRC, src)] + headers
        if not force and os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps):
            return obj                                  # object newer than its source and every header: reuse
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        t_start = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        os.utime(obj, (t_start, t_start))               # a header edited DURING a minutes-long compile must still count as newer
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(cc, SOURCES))
    if not force and os.path.exists(LIB) and all(os.path.getmtime(o) <= os.path.getmtime(LIB) for o in objs):
        return LIB                                      # every object up to date and already linked
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", LIB], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
