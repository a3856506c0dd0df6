#!/bin/bash
# Drop-in check (this container only; /root/reference does not travel): run the REFERENCE'S OWN
# test files against ginfinity_amd.  A scratch directory gets (a) a shim package `ginfinity` that
# re-exports ginfinity_amd and maps ginfinity.{api,graph,cli,...} onto its modules, (b) a run-time
# copy of the reference's tests/ (never committed) and (c) the data directory where
# tests/test_api.py looks for it (<tests>/../src/ginfinity/data).  The reference implementation
# itself is not imported.  test_release.py (release tooling, out of scope) is skipped.
#   bash tools/run_reference_tests.sh [/root/reference]
set -e
REF=${1:-/root/reference}
REPO=$(cd "$(dirname "$0")/.." && pwd)
[ -d "$REF/tests" ] || { echo "no reference tree at $REF"; exit 2; }
S=$(mktemp -d)
trap 'rm -rf "$S"' EXIT
mkdir -p "$S/shim/ginfinity" "$S/tree/tests" "$S/tree/src/ginfinity"
cat > "$S/shim/ginfinity/__init__.py" <<'PY'
import importlib, sys
import ginfinity_amd as _pkg
for _name in dir(_pkg):
    if not _name.startswith("__"):
        globals()[_name] = getattr(_pkg, _name)
for _sub in ("api", "graph", "cli", "spec", "records", "table", "weights", "shard_io", "npz"):
    sys.modules["ginfinity." + _sub] = importlib.import_module("ginfinity_amd." + _sub)
PY
cp "$REF"/tests/*.py "$REF"/tests/*.tsv "$S/tree/tests/"
ln -s "$REPO/ginfinity_amd/data" "$S/tree/src/ginfinity/data"
cd "$S/tree/tests"
PYTHONPATH="$S/shim:$REPO" python -m pytest -q -p no:cacheprovider --ignore=test_release.py "${@:2}"
