"""Build libgfy.so (HIP kernels + C ABI) in-tree for gfx950.

    python -m ginfinity_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU; the .so lands next to the sources
(ginfinity_amd/csrc/libgfy.so) so it travels with the repository snapshot.
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIBRARY = CSRC / "libgfy.so"
SOURCES = ("gfy_api.hip", "graph_build.hip", "csr_build.hip", "gine_f16.hip",
           "gine_f32.hip", "pairwise.hip", "gine_host.cpp", "gfy_base.cpp")
# device="cpu" (the reference's default device) without any ROCm runtime on the box: the host
# implementation and what it needs, built with the HOST compiler alone
HOST_LIBRARY = CSRC / "libgfy_host.so"
HOST_SOURCES = ("gine_host.cpp", "gfy_base.cpp")
ARCH = "gfx950"
# -ffp-contract=off: the rounding-point contract needs mul and add to round
# separately unless the source says fma.
FLAGS = ("-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}",
         "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function")


def _hipcc() -> str:
    for candidate in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if candidate and (Path(candidate).exists() or candidate == "hipcc"):
            return candidate
    raise RuntimeError("hipcc not found")


def _stale(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return True
    stamp = target.stat().st_mtime
    return any(dep.stat().st_mtime > stamp for dep in deps)


def build(force: bool = False, verbose: bool = False, stamps: bool = False) -> Path:
    """``stamps=True`` builds the diagnostic variant libgfy_stamps.so (per-phase
    s_memtime totals in the layer kernel); the product library never has them."""
    hipcc = _hipcc()
    headers = (list(CSRC.glob("*.h")) + list(CSRC.glob("*.inc"))
               + [CSRC.parents[1] / "include" / "gfy.h"])
    suffix = ".stamps.o" if stamps else ".o"
    library = CSRC / "libgfy_stamps.so" if stamps else LIBRARY
    extra = ("-DGFY_STAMPS",) if stamps else ()
    extra += tuple(os.environ.get("GFY_EXTRA_FLAGS", "").split())   # experiments (-DGFY_...)
    jobs = []
    for name in SOURCES:
        source, obj = CSRC / name, CSRC / (Path(name).stem + suffix)
        if force or _stale(obj, [source, *headers]):
            jobs.append([hipcc, *FLAGS, *extra, "-c", str(source), "-o", str(obj)])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        done = subprocess.run(cmd, capture_output=True, text=True)
        if done.returncode != 0:
            raise RuntimeError(f"{' '.join(cmd)}\n{done.stdout}\n{done.stderr}")
        if verbose and done.stderr.strip():
            print(done.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as pool:
        list(pool.map(run, jobs))
    objects = [CSRC / (Path(name).stem + suffix) for name in SOURCES]
    if force or jobs or _stale(library, objects):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}",
             "-o", str(library), *map(str, objects)])
    if not stamps:
        host_sources = [CSRC / name for name in HOST_SOURCES]
        if force or _stale(HOST_LIBRARY, [*host_sources, *headers]):
            run([os.environ.get("CXX", "g++"), "-O3", "-std=c++17", "-fPIC", "-shared",
                 "-ffp-contract=off", "-fno-fast-math", "-pthread", "-Wall",
                 "-o", str(HOST_LIBRARY), *map(str, host_sources)])
    return library


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--force", action="store_true")
    parser.add_argument("--verbose", action="store_true")
    parser.add_argument("--stamps", action="store_true",
                        help="diagnostic build with in-kernel phase stamps")
    args = parser.parse_args()
    print(build(args.force, args.verbose, args.stamps))
