PG_WINO_STRIP_MAXCIN=32 BW_ONLY=unpool python tools/bench_wino_strip.py 2>&1 | grep -v amdgpu.ids
