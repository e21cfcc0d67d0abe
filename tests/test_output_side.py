"""Output side of the boundary, pinned by the reference's own code.

tests/golden/output_side/*_solver_out.pdb are solver-output PDBs written by OUR writer (c3d_write_pdb) from the
coordinates of bundled reference models; *_final.pdb / *_model_info.log / output_side_golden.json are what the
REFERENCE's assess_dgsa subs (get_cns_energy :602-618, count_satisfied_tbl_rows :447-485, sum_noe_dev :581-600,
filter_nonCA :864-880, reindex_chain :831-862, sed :818, add_connect_rows :208-215) made of them in the build
container (tests/golden/make_output_golden.py / .pl eval the subs from /root/reference at run time).  No GPU needed."""
import filecmp
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from chromosome3d_amd import pipeline
from tests.util import GOLD, load_pdb_xyz, model_pdb

OUT = os.path.join(GOLD, "output_side")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(OUT, "output_side_golden.json")) as fh:
    G = json.load(fh)
ENERGIES = {"chr21_1mb": (54337.7461, 1203.25, 3.0625), "chr22_1mb": (39387.6953, 877.5, 0.125)}


@pytest.mark.parametrize("cid", sorted(G))
def test_writer_reproduces_the_file_the_reference_accepted(cid, tmp_path, built):
    """c3d_write_pdb today writes the bytes the reference's subs were fed (the goldens stay valid)."""
    xyz = load_pdb_xyz(model_pdb(cid)).astype(np.float32)
    p = tmp_path / f"{cid}_1.pdb"
    pipeline.write_pdb(str(p), xyz, *ENERGIES[cid], title=f"{cid}_1.pdb")
    assert filecmp.cmp(p, os.path.join(OUT, f"{cid}_solver_out.pdb"), shallow=False)


@pytest.mark.parametrize("cid", sorted(G))
def test_reference_reads_our_energy_and_our_assessment_equals_its(cid, built):
    """get_cns_energy parsed `REMARK noe` of our file (int, :617); count_satisfied_tbl_rows / sum_noe_dev of the
    reference on our file equal c3d_assess on the same coordinates."""
    assert G[cid]["noe_int"] == int(ENERGIES[cid][0])
    xyz = pipeline.read_pdb_ca(os.path.join(OUT, f"{cid}_solver_out.pdb"))
    rows = pipeline.read_tbl(os.path.join(GOLD, f"{cid}.contact.tbl"))
    sat, dev = pipeline.assess(xyz, rows)
    assert f"{sat}/{len(rows[0])}" == G[cid]["satisfied"]
    assert "%.2f" % dev == "%.2f" % G[cid]["sum_dev"]
    assert "%-9s             %-9s                %-25s" % (f"{sat}/{len(rows[0])}", "%.2f" % dev, f"{cid}_1") == G[cid]["table_row"]


@pytest.mark.parametrize("cid", sorted(G))
def test_c_abi_shaping_equals_reference_post_processing(cid, tmp_path, built):
    """c3d_shape_pdb == filter_nonCA + reindex_chain + sed + add_connect_rows, byte for byte, log included."""
    os.chdir(tmp_path)
    shutil.copy(os.path.join(OUT, f"{cid}_solver_out.pdb"), f"{cid}_1.pdb")
    pipeline.shape_pdb(f"./{cid}_1.pdb", None, "model_info.log")
    assert filecmp.cmp(f"{cid}_1.pdb", os.path.join(OUT, f"{cid}_final.pdb"), shallow=False)
    assert filecmp.cmp("model_info.log", os.path.join(OUT, f"{cid}_model_info.log"), shallow=False)


@pytest.mark.parametrize("cid", sorted(G))
def test_perl_driver_shaping_equals_reference_post_processing(cid, tmp_path):
    """The in-tree driver's own shaping (bin/chromosome3D_amd.pl, what it does to every model before ranking)."""
    if shutil.which("perl") is None:
        pytest.skip("perl not installed")
    shutil.copy(os.path.join(OUT, f"{cid}_solver_out.pdb"), tmp_path / f"{cid}_1.pdb")
    subprocess.check_call(["perl", os.path.join(ROOT, "bin", "chromosome3D_amd.pl"), "--shape", f"./{cid}_1.pdb", "-o", "."], cwd=tmp_path)
    assert filecmp.cmp(tmp_path / f"{cid}_1.pdb", os.path.join(OUT, f"{cid}_final.pdb"), shallow=False)
    assert filecmp.cmp(tmp_path / "model_info.log", os.path.join(OUT, f"{cid}_model_info.log"), shallow=False)


def test_shaping_drops_what_the_reference_drops(tmp_path, built):
    """Rows the reference's filters remove: non-CA atoms, alternative locations other than A, unknown residue names;
    residues are renumbered by change of the residue field, atoms consecutively."""
    rows = ["REMARK noe = 12.5\n",
            "ATOM      7  N   MET     3       1.000   2.000   3.000  1.00  0.00\n",
            "ATOM      8  CA  MET     3       1.500   2.000   3.000  1.00  0.00\n",
            "ATOM      9  CA BMET     3       1.600   2.000   3.000  1.00  0.00\n",
            "ATOM     10  CA  XYZ     4       2.500   2.000   3.000  1.00  0.00\n",
            "ATOM     11  CA AGLY     9       3.500   2.000   3.000  1.00  0.00\n",
            "HETATM   12  CA  MET    10       4.500   2.000   3.000  1.00  0.00\n",
            "END\n"]
    p = tmp_path / "m.pdb"
    p.write_text("".join(rows))
    pipeline.shape_pdb(str(p))
    got = p.read_text().split("\n")
    assert got[0] == "ATOM      1  CA  MET     1       1.500   2.000   3.000  1.00  0.00"
    assert got[1] == "ATOM      2  CA  GLY     2       3.500   2.000   3.000  1.00  0.00"      # altloc column blanked by the row formula
    assert got[2:] == ["", "CONECT    1    2", "END", ""]
