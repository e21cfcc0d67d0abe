# config 4's block under two builds of the library, alternating: A = chromosome3d_amd/_lib (this tree), B = chromosome3d_amd/_lib_b
# (e.g. an older commit: git archive <rev> chromosome3d_amd/csrc include | tar -x -C /tmp/old && make -C /tmp/old/chromosome3d_amd/csrc OUT=$PWD/chromosome3d_amd/_lib_b all).
#   bash tools/ab_config4.sh [out dir]        then, for the prefetch / pairing grid on each build: python tools/config4_prefetch_ab.py <prefetch 0|1> <pair 0|1>
O=${1:-gpurun_out/ab_config4}
mkdir -p $O
L=chromosome3d_amd
for rep in 1 2 3; do
for v in a b; do
  if [ $v = b ]; then mv $L/_lib $L/_lib_a && mv $L/_lib_b $L/_lib; fi
  python -m chromosome3d_amd.batch --bench-block > $O/c4_${v}_$rep.json 2>> $O/err.txt
  if [ $v = b ]; then mv $L/_lib $L/_lib_b && mv $L/_lib_a $L/_lib; fi
done; done
python - $O <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/c4_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["wall_s"], d["per_rank"])
PY
