mkdir -p gpurun_out/r5f
# c2 / c3 product shapes on both kernels (NT: zx, dX, projection, dh)
X3_SHAPES="32000,1280,640;32000,640,1280;32000,1280,40;32000,320,320;32000,2048,1024;32000,1024,2048;32000,512,512;32000,5184,1024" X3_TN_SHAPES="640,1280,32000;320,1280,31968;320,320,32000" python tools/gemm_x3_probe.py > gpurun_out/r5f/gemm_x3_probe_c2c3.txt 2>&1
for f in 90 75 55 35; do
  for w in c2x3 c3x3; do
    LC_X3_MIN_FILL=$f python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5f/bench_${w}_fill$f.json 2>gpurun_out/r5f/bench_${w}_fill$f.err
  done
done
python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5f/bench_c2.json 2>&1
python bench.py --workload c3 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5f/bench_c3.json 2>&1
python - <<'PY'
import json,glob
for p in sorted(glob.glob("gpurun_out/r5f/bench_*.json")):
    try:
        l=[x for x in open(p).read().splitlines() if x.startswith("{")][-1]; d=json.loads(l)
        print(p, d["ms_per_step"], d["value"], d.get("breakdown_ms_per_step"), d["config"].get("product_kernels"))
    except Exception as e: print(p, "ERR", e)
PY
