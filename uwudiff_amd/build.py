"""Build libuwu_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m uwudiff_amd.build [--force]

Object files are cached under uwudiff_amd/csrc/_obj and rebuilt when a source or header is newer.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libuwu_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [*os.environ.get("UWU_EXTRA_FLAGS", "").split(), "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "uwu_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def compile_one(src, force, hm):
    obj = os.path.join(OBJ, src + ".o")
    sp = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(sp), hm):
        return obj, False
    cmd = [HIPCC, *FLAGS, "-c", sp, "-o", obj]
    if src.endswith(".cpp"):
        cmd = [HIPCC, "-O2", "-fPIC", "-std=c++17", "-c", sp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hm = headers_mtime()
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        res = list(ex.map(lambda s: compile_one(s, force, hm), srcs))
    objs = [o for o, _ in res]
    rebuilt = any(r for _, r in res)
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[uwudiff_amd.build] {LIB} ({'rebuilt' if rebuilt else 'up to date'}; {len(srcs)} sources)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
