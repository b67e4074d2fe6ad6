"""Build libccc_amd.so (hand-written HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the built library lands in centroidalcontrolcollection_amd/lib/ (git-ignored,
but shipped to the GPU box with the working tree).
"""
import glob
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libccc_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# the ROCm installation the compiler belongs to ($ROCM_PATH, else two levels above hipcc)
ROCM_ROOT = os.environ.get("ROCM_PATH") or os.path.dirname(os.path.dirname(os.path.realpath(HIPCC)))


def host_fma_flags():
    """["-mfma"] on an x86 host whose CPU has FMA3 (faster std::fma in the host emulation of the tile kernel), else []."""
    import platform

    if platform.machine() not in ("x86_64", "AMD64"):
        return []
    try:
        with open("/proc/cpuinfo") as f:
            return ["-mfma"] if " fma " in f.read().replace("\n", " ") else []
    except OSError:
        return []
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


HASH_PATH = LIB_PATH + ".srchash"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def source_hash():
    """sha256 over everything the library is compiled from (file names + contents) and the compiler flags."""
    import hashlib

    h = hashlib.sha256(" ".join(FLAGS).encode())
    # (csrc/ includes ../../include/ccc_amd.h; a package installed without the include/ directory hashes csrc/ alone)
    deps = sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(CSRC, "*.inc"))) + sorted(
        glob.glob(os.path.join(_HERE, "..", "include", "*.h")))
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


# what each benched workload's kernels are compiled from (csrc/ file names): the hash of these files + the flags goes into
# the profile summaries the bench lines replay counters from, and a bench line refuses counters of another build
KERNEL_UNITS = dict(
    # (common.hip holds the order_by_count kernels of the ordered schedules, which run inside the timed and profiled launches)
    zmp=["zmp.hip", "zmp_k1.inc", "zmp_k2r.inc", "zmp_stage.inc", "sym_tableau.h", "wave_group.h", "common.h", "common.hip"],
    xy=["xy.hip", "wave_group.h", "common.h", "common.hip"],
    ddp=["ddp.hip", "ddp_tile.hip", "ddp_tile_body.inc", "ddp_tile.h", "ddp_batch.h", "w64.h", "common.h"],
)
for _alias in ("srb", "walk", "multi"):
    KERNEL_UNITS[_alias] = KERNEL_UNITS["ddp"]
KERNEL_UNITS["xywalk"] = KERNEL_UNITS["xy"]
KERNEL_UNITS["zmp100"] = KERNEL_UNITS["zmp"]  # (LinearMpcZmp at N = 100: zmp_stage.inc + zmp_k2r.inc)


def kernel_hash(workload):
    """sha256 (first 16 hex digits) over the csrc/ files the workload's kernels are built from and the compiler flags;
    None for a workload without an entry in KERNEL_UNITS."""
    import hashlib

    files = KERNEL_UNITS.get(workload)
    if files is None:
        return None
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for name in files:
        h.update(name.encode())
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def is_stale():
    """True when libccc_amd.so is missing or was not built from the sources in the tree (content hash recorded by
    build_lib next to the library; mtimes do not survive the copy to the GPU box)."""
    if not os.path.exists(LIB_PATH) or not os.path.exists(HASH_PATH):
        return True
    with open(HASH_PATH) as f:
        return f.read().strip() != source_hash()


_stale = is_stale


def build_lib(force=False, verbose=False):
    """Compile every HIP source into one shared library. Returns the library path.

    Safe under concurrent callers (every rank of a torchrun job imports the package): the build is serialised by a file
    lock, objects go to a per-process directory, and the library and its source hash are moved into place atomically --
    a rank that waited for the lock finds the library fresh and returns without compiling."""
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    import fcntl
    import shutil

    want = source_hash()
    with open(os.path.join(LIB_DIR, ".buildlock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():
                return LIB_PATH  # another process built it while this one waited
            # one hipcc per translation unit, side by side (no device code crosses a unit: every kernel is launched from
            # the unit that defines it), then one link
            from concurrent.futures import ThreadPoolExecutor

            obj_dir = os.path.join(LIB_DIR, "obj.%d" % os.getpid())
            os.makedirs(obj_dir, exist_ok=True)
            cflags = [f for f in FLAGS if f != "-shared"]

            def compile_one(src):
                obj = os.path.join(obj_dir, os.path.splitext(os.path.basename(src))[0] + ".o")
                cmd = [HIPCC] + cflags + ["-c", src, "-o", obj]
                if verbose:
                    print(" ".join(cmd))
                subprocess.check_call(cmd)
                return obj

            try:
                with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
                    objs = list(pool.map(compile_one, sources()))
                tmp_lib = os.path.join(obj_dir, "libccc_amd.so")
                cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp_lib]
                if verbose:
                    print(" ".join(cmd))
                subprocess.check_call(cmd)
                tmp_hash = os.path.join(obj_dir, "srchash")
                with open(tmp_hash, "w") as f:
                    f.write(want + "\n")
                if os.path.exists(HASH_PATH):
                    os.remove(HASH_PATH)  # (never a fresh hash beside a stale library)
                os.replace(tmp_lib, LIB_PATH)
                os.replace(tmp_hash, HASH_PATH)
            finally:
                shutil.rmtree(obj_dir, ignore_errors=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


if __name__ == "__main__":
    import sys

    if len(sys.argv) > 1 and sys.argv[1] == "--kernel-hashes":  # scripts/prof_*.sh: what the profiled build was made from
        import json

        print(json.dumps({k: kernel_hash(k) for k in sorted(KERNEL_UNITS)}))
    else:
        print(build_lib(force=True, verbose=True))
