#!/bin/bash
# the GPU suite with its log kept: bash tools/gpu_suite.sh <tag> [pytest args]   (default: tests -m gpu)
out=gpurun_out/$1; shift
mkdir -p $out
if [ $# -eq 0 ]; then set -- tests -m gpu; fi
timeout 3000 python3 -m pytest "$@" -x -q > $out/pytest.log 2>&1
echo "pytest rc=$?"
grep -E "passed|failed|error" $out/pytest.log | tail -3
grep -E "^(FAILED|ERROR)" $out/pytest.log | head
