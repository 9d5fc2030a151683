"""Build the gfx950 engine library in-tree (correlation_amd/liblk_engine.so).

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liblk_engine.so")
SOURCES = ["lk_engine.cpp", "lk_tracker.cpp", "lk_group.cpp", "lk_image_io.cpp", "lk_kernels.hip"]
HEADERS = ["lk_device.hpp", "lk_roi.hpp", os.path.join("..", "..", "include", "lk_engine.h"),
           os.path.join("..", "..", "include", "lk_tracker.h"), os.path.join("..", "..", "include", "lk_group.h")]


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into liblk_engine.so (one object per source under build/obj, compiled side by
    side and only when the source or a header is newer; then one link)."""
    if not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(os.path.dirname(HERE), "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    flags = [
        "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17",
        # the reference's x86-64 builds have no FMA: keep mul and add separate unless the
        # source asks for an fma explicitly (see lk_kernels.hip header)
        "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
        # v_pk_*_f32 run at the scalar-op rate on gfx950; SLP packing only adds register moves
        # to the sample loop (measured: 6 % slower with it)
        "-fno-slp-vectorize",
        "-Wall", "-Wextra",
    ] + os.environ.get("LK_EXTRA_HIPCC_FLAGS", "").split()   # (tuning experiments)
    if verbose:
        flags.insert(0, "-Rpass-analysis=kernel-resource-usage")
    tag = os.path.join(objdir, "flags.txt")
    same_flags = os.path.exists(tag) and open(tag).read() == " ".join(flags)
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS + ["lk_internal.hpp"] if os.path.exists(os.path.join(CSRC, h)))

    def compile_one(src):
        obj = os.path.join(objdir, src + ".o")
        src_path = os.path.join(CSRC, src)
        if (not force and same_flags and os.path.exists(obj)
                and os.path.getmtime(obj) > max(os.path.getmtime(src_path), hdr_time)):
            return obj, None
        r = subprocess.run([hipcc_path()] + flags + ["-c", src, "-o", obj], cwd=CSRC, capture_output=True, text=True)
        return obj, r

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:
        results = list(pool.map(compile_one, SOURCES))
    for obj, r in results:
        if r is not None and r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("hipcc failed building " + os.path.basename(obj))
        if r is not None and verbose:
            sys.stderr.write(r.stderr)
    with open(tag, "w") as f:
        f.write(" ".join(flags))
    link = [hipcc_path(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + [o for o, _ in results] + [
        "-lrocprofiler-sdk-roctx",   # roctx ranges (lk_engine.cpp: struct Range)
        "-lrccl",                    # lk_group.cpp: ncclBroadcast / ncclAllGather over xGMI
        "-lz",                       # lk_image_io.cpp: inflate + crc32 for PNG
    ]
    r = subprocess.run(link, cwd=CSRC, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed linking liblk_engine.so")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
