mkdir -p gpurun_out/r04h/ab
L=chromosome3d_amd
for rep in 1 2; do
for v in a b; do
  if [ $v = b ]; then mv $L/_lib $L/_lib_a && mv $L/_lib_b $L/_lib; fi
  python bench.py --no-cpu-baseline --no-side-figures > gpurun_out/r04h/ab/default_${v}_$rep.json 2> gpurun_out/r04h/ab/err.txt
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-figures > gpurun_out/r04h/ab/s20_${v}_$rep.json 2>> gpurun_out/r04h/ab/err.txt
  if [ $v = b ]; then mv $L/_lib $L/_lib_b && mv $L/_lib_a $L/_lib; fi
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04h/ab/*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"])
PY
