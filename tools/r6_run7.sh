timeout 600 python -m pytest tests/test_gpu_round6.py -q -k "shadow_only" 2>&1 | tail -3
for v in quad0 "" quad0 ""; do LC_DEV_LIB=$v timeout 300 python tools/r6_quad_ab.py 2>/dev/null | tail -1; done
