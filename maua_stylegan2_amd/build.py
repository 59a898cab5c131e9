"""Build recipe for csrc/libmaua_hip.so (gfx950 only, plain hipcc, in-tree so it travels with gpurun snapshots).

    python -m maua_stylegan2_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects are rebuilt only when their source (or a header) is newer.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libmaua_hip.so")
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "maua_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = _sources()
    objs = [s[:-4] + ".o" for s in srcs]

    def compile_one(pair):
        src, obj = pair
        if not force and not _stale(obj, [src] + HEADERS):
            return None
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip() and verbose:
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        rebuilt = [o for o in ex.map(compile_one, zip(srcs, objs)) if o]
    if rebuilt or force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


def build_sanitized(device=True, verbose=True):
    """AddressSanitizer (+ UBSan on the host side) build of the same sources -> csrc/san/libmaua_hip_{asan,hostasan}.so, next to — never
    instead of — the product library (run with tools/asan_run.sh, which pre-loads the clang ASAN runtime into python
    and points MAUA_TEST_LIB at the build).  ``device=True``: host AND device code instrumented (gfx950:xnack+, -shared-libasan; the
    device checks need HSA_XNACK=1 and, for complete reports, the ASAN build of the ROCm runtime, which this image does not ship);
    ``device=False``: launchers / host tables only (-fno-gpu-sanitize), runs against the stock runtime."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out_dir = os.path.join(CSRC, "san")
    os.makedirs(out_dir, exist_ok=True)
    tag = "asan" if device else "hostasan"
    arch = "--offload-arch=gfx950:xnack+" if device else "--offload-arch=gfx950"
    flags = [arch, "-O2", "-g", "-std=c++17", "-fPIC", "-Wno-unused-function", "-fsanitize=address", "-shared-libasan"]
    if not device:
        # (-fno-sanitize=function: with the function sanitizer a kernel launched through a host function-pointer VARIABLE — `auto kern =
        # modconv_mfma_kernel<...>; hipLaunchKernelGGL(kern, ...)` — silently does not run; found in round 6 when the torch-free driver first
        # exercised those launchers, tools/asan_driver.cpp)
        flags += ["-fno-gpu-sanitize", "-fsanitize=undefined", "-fno-sanitize=vptr,function"]
    srcs = _sources()
    objs = [os.path.join(out_dir, os.path.basename(s)[:-4] + f".{tag}.o") for s in srcs]

    def compile_one(pair):
        src, obj = pair
        if not _stale(obj, [src] + HEADERS):
            return None
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr[-4000:]}")
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        list(ex.map(compile_one, zip(srcs, objs)))
    lib = os.path.join(out_dir, f"libmaua_hip_{tag}.so")
    cmd = [hipcc, arch, "-shared", "-fPIC", "-fsanitize=address", "-shared-libasan"] + ([] if device else ["-fno-gpu-sanitize", "-fsanitize=undefined"]) + ["-o", lib] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr[-4000:]}")
    return lib


if __name__ == "__main__":
    if "--asan" in sys.argv:
        print(build_sanitized(device=True))
    elif "--asan-host" in sys.argv:
        print(build_sanitized(device=False))
    else:
        print(build(force="--force" in sys.argv))
