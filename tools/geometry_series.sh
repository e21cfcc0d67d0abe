for r in 24 20 16 10 8 5 3 2; do python tools/geometry_compare.py chr1_500kb $r; done
for c in chr4_1mb chr21_1mb chr13_1mb chr19_500kb chr22_1mb chr20_1mb; do python tools/geometry_compare.py $c 20; done
