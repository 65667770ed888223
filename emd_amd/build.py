"""In-tree build of the HIP extension (libemd_raster.so) for gfx950.  hipcc cross-compiles without a GPU."""
import os
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def build_native(force=False, verbose=False):
    so = os.path.join(_CSRC, "libemd_raster.so")
    if force:
        subprocess.check_call(["make", "-s", "-C", _CSRC, "clean"])
    cmd = ["make", "-C", _CSRC, "-j4", "libemd_raster.so"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    if not os.path.exists(so):
        raise RuntimeError(f"build did not produce {so}")
    return so
