"""BASELINE.json configs[3] — kept as an alias of the product entry:  python -m chromosome3d_amd.batch  (see its docstring)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chromosome3d_amd.batch import main
main()
