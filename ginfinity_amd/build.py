"""Build libgfy.so (HIP kernels + C ABI) in-tree for gfx950.

    python -m ginfinity_amd.build [--force] [--verbose] [--host-only]

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


def _hipcc() -> str | None:
    """hipcc, or None where there is no ROCm toolchain (the host library builds without it)."""
    import shutil
    for candidate in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if candidate and Path(candidate).exists():
            return candidate
    return shutil.which("hipcc")


def _stale(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return True
    stamp = target.stat().st_mtime
    return any(dep.stat().st_mtime > stamp for dep in deps)


def _run(cmd: list[str], verbose: bool) -> None:
    if verbose:
        print(" ".join(cmd), flush=True)
    done = subprocess.run(cmd, capture_output=True, text=True)
    if done.returncode != 0:
        raise RuntimeError(f"{' '.join(cmd)}\n{done.stdout}\n{done.stderr}")
    if verbose and done.stderr.strip():
        print(done.stderr, file=sys.stderr)


def _headers() -> list[Path]:
    return (list(CSRC.glob("*.h")) + list(CSRC.glob("*.inc"))
            + [CSRC.parents[1] / "include" / "gfy.h"])


def build_host(force: bool = False, verbose: bool = False) -> Path:
    """libgfy_host.so (device="cpu") with the HOST compiler alone: no hipcc, no ROCm runtime.
    References inside the library bind inside it (-Bsymbolic): when libgfy.so — which defines
    the same helpers — is already loaded RTLD_GLOBAL, the host library still calls its own
    gfy::set_error and its error text is not lost."""
    host_sources = [CSRC / name for name in HOST_SOURCES]
    if force or _stale(HOST_LIBRARY, [*host_sources, *_headers()]):
        _run([os.environ.get("CXX", "g++"), "-O3", "-std=c++17", "-fPIC", "-shared",
              "-ffp-contract=off", "-fno-fast-math", "-pthread", "-Wall",
              "-Wl,-Bsymbolic-functions", "-Wl,-Bsymbolic",
              "-o", str(HOST_LIBRARY), *map(str, host_sources)], verbose)
    return HOST_LIBRARY


def build(force: bool = False, verbose: bool = False, stamps: bool = False,
          host_only: bool = False) -> Path:
    """``stamps=True`` builds the diagnostic variant libgfy_stamps.so (per-phase
    s_memtime totals in the layer kernel); the product library never has them.
    ``host_only=True`` builds libgfy_host.so and nothing else (no hipcc needed).  The two
    libraries are built independently: the host library first, and a missing hipcc only fails
    the GPU library (with a message that says so), never the host one."""
    if host_only:
        return build_host(force, verbose)
    host_error = None
    if not stamps:
        try:
            build_host(force, verbose)
        except (RuntimeError, OSError) as error:   # no host compiler: the GPU library still builds
            host_error = error
    hipcc = _hipcc()
    if hipcc is None:
        raise RuntimeError(
            "hipcc not found (set HIPCC): libgfy.so (device='cuda') cannot be built here; "
            "`python -m ginfinity_amd.build --host-only` builds libgfy_host.so (device='cpu') "
            "with the host compiler alone")
    headers = _headers()
    suffix = ".stamps.o" if stamps else ".o"
    library = CSRC / "libgfy_stamps.so" if stamps else LIBRARY
    extra = ("-DGFY_STAMPS",) if stamps else ()
    extra += tuple(os.environ.get("GFY_EXTRA_FLAGS", "").split())   # experiments (-DGFY_...)
    jobs = []
    for name in SOURCES:
        source, obj = CSRC / name, CSRC / (Path(name).stem + suffix)
        if force or _stale(obj, [source, *headers]):
            jobs.append([hipcc, *FLAGS, *extra, "-c", str(source), "-o", str(obj)])

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as pool:
        list(pool.map(lambda cmd: _run(cmd, verbose), jobs))
    objects = [CSRC / (Path(name).stem + suffix) for name in SOURCES]
    if force or jobs or _stale(library, objects):
        _run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}",
              "-o", str(library), *map(str, objects)], verbose)
    if host_error is not None:
        print(f"warning: libgfy_host.so (device='cpu') was not built: {host_error}",
              file=sys.stderr)
    return library


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--force", action="store_true")
    parser.add_argument("--verbose", action="store_true")
    parser.add_argument("--stamps", action="store_true",
                        help="diagnostic build with in-kernel phase stamps")
    parser.add_argument("--host-only", action="store_true",
                        help="libgfy_host.so only (device='cpu'): host compiler, no hipcc")
    args = parser.parse_args()
    print(build(args.force, args.verbose, args.stamps, args.host_only))
