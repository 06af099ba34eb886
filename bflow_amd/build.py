"""`python -m bflow_amd.build` -- compile libbflow_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def build(verbose: bool = True, force: bool = False) -> str:
    """force=True recompiles every .hip (clean build); the default only rebuilds stale objects."""
    script = os.path.join(HERE, "csrc", "build.sh")
    res = subprocess.run(["bash", script] + (["--force"] if force else []), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        sys.stdout.write(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libbflow_hip.so failed")
    return os.path.join(HERE, "lib", "libbflow_hip.so")


if __name__ == "__main__":
    build(force="--force" in sys.argv[1:])
