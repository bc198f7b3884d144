#!/bin/bash
# Repeat GPU-tier tests N times in ONE pytest process (tests/conftest.py --repeat), every failure named (-rf, pytest.ini).
#   tools/loop_tests.sh <N> <log> [-x] <node id> [<node id> ...]      (environment passes through: PGGAN_DERIVED_EVENT=0, AMD_SERIALIZE_KERNEL=3 ...)
n=$1; log=$2; shift 2
extra=()
if [ "$1" == "-x" ]; then extra=(-x); shift; fi
mkdir -p "$(dirname "$log")"
python -m pytest -q -m gpu --repeat "$n" -p no:cacheprovider "${extra[@]}" "$@" > "$log" 2>&1
rc=$?
grep -c "^FAILED" "$log" | sed 's/^/failed: /'
tail -n 6 "$log"
exit $rc
