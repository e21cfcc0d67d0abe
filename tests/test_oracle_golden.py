"""The oracle against the reference's own outputs (tests/golden, made by running the reference's
Perl subs: tests/golden/make_golden.pl).  Pins the oracle's front half, assessment and metric."""
import hashlib
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.util import GOLD, golden, load_if, load_pdb_xyz, model_pdb, REF_SPEARMAN

G = golden()


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


@pytest.mark.parametrize("cid", sorted(G))
def test_front_half_md5(cid, tmp_path):
    """chromosome3D.pl:110-206,340-362 — .dist, .rr, contact.tbl byte-identical (md5 + counts)."""
    IF = load_if(cid)
    assert IF.shape[0] == G[cid]["n"]
    d = O.if_to_dist10(IF)
    rr = O.dist_to_rr(d)
    assert len(rr[0]) == G[cid]["restraints"]
    O.write_front_half(str(tmp_path), cid, d, rr)
    assert md5(tmp_path / "contact.tbl") == G[cid]["md5_tbl"]
    assert md5(tmp_path / f"{cid}.dist") == G[cid]["md5_dist"]
    assert md5(tmp_path / f"{cid}.rr") == G[cid]["md5_rr"]


@pytest.mark.parametrize("cid", ["chr21_1mb", "chr22_1mb"])
def test_front_half_files_verbatim(cid, tmp_path):
    IF = load_if(cid)
    d = O.if_to_dist10(IF)
    O.write_front_half(str(tmp_path), cid, d, O.dist_to_rr(d))
    for ours, theirs in [(f"{cid}.dist", f"{cid}.dist"), (f"{cid}.rr", f"{cid}.rr"), ("contact.tbl", f"{cid}.contact.tbl")]:
        assert open(tmp_path / ours, "rb").read() == open(os.path.join(GOLD, theirs), "rb").read()


def test_parse_text_matches_python_float():
    raw = open(os.path.join(GOLD, "inputs", "chr21_1mb_matrix.txt"), "rb").read()
    assert raw.endswith(b" \r\n")                  # the bundled line ends
    m = O.parse_if_text(raw)
    assert np.array_equal(m, load_if("chr21_1mb"))
    assert m.shape == (37, 37)


@pytest.mark.parametrize("cid", sorted(G))
def test_assessment_known_answers(cid):
    """chromosome3D.pl:447-485, 581-600 on the bundled model of the chromosome."""
    d = O.if_to_dist10(load_if(cid))
    rr = O.dist_to_rr(d)
    X = load_pdb_xyz(model_pdb(cid))
    sat, dev = O.assess(X, rr)
    assert f"{sat}/{len(rr[0])}" == G[cid]["satisfied"]
    assert "%.2f" % dev == "%.2f" % G[cid]["sum_dev"]


@pytest.mark.parametrize("cid", sorted(REF_SPEARMAN))
def test_spearman_of_bundled_models(cid):
    """spearman_IF_pdb.pl:42-70 restated; BASELINE.md table + scipy as an independent check."""
    from scipy.stats import spearmanr
    IF = load_if(cid)
    X = load_pdb_xyz(model_pdb(cid))
    rho = O.spearman_if_dist(IF, X, 3)
    assert abs(rho - REF_SPEARMAN[cid]) < 5e-5
    n = len(IF)
    i, j = np.where(np.abs(np.subtract.outer(np.arange(n), np.arange(n))) >= 3)
    dd = np.round(np.linalg.norm(X[i] - X[j], axis=1), 3)
    assert abs(rho - spearmanr(IF[i, j], dd)[0]) < 1e-12


def test_zero_if_gives_minus_one_and_no_restraint():
    IF = load_if("chr1_500kb")
    d = O.if_to_dist10(IF)
    assert ((IF == 0) == (d == -10)).all()
    ri, rj, rt = O.dist_to_rr(d)
    assert (rt > 0).all() and (rj - ri >= 5).all()
    assert len(ri) == 101426
