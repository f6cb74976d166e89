#!/bin/bash
# A/B the two exchange kernels in separate processes (AFQ_EXX_V1=1 selects exx_kernel)
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
unset AFQ_EXX_V1
python bench.py --steps 20 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('exx2', d['roofline']['kernel_ms'], d['roofline']['achieved'], d['ms_per_step'])"
export AFQ_EXX_V1=1
python -m pytest tests/test_gpu_ops.py tests/test_gpu_sizes.py -m gpu -x -q 2>&1 | tail -2
python bench.py --steps 20 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('exx1', d['roofline']['kernel_ms'], d['roofline']['achieved'], d['ms_per_step'])"
