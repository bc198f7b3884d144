cd /root/repo
python -m pytest tests/test_smallm.py -x -q -m gpu 2>&1 | tail -4
python tools/exp/smallm_bench.py 2>&1 | grep -v amdgpu
