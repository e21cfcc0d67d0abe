"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/c3d.h declares,
refuses to run without a GPU, and its host-format helpers reproduce the reference's files."""
import ctypes as C
import hashlib
import os
import re

import numpy as np
import pytest

from tests.util import GOLD, golden, load_if, load_pdb_xyz, model_pdb, write_if_text, REF_SPEARMAN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = golden()


def test_library_exports_every_declared_symbol(built):
    from chromosome3d_amd import lib
    L = lib.load()
    hdr = open(os.path.join(ROOT, "include", "c3d.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(c3d_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/c3d.h but not exported"
    assert declared == set(lib.SIGNATURES), "python binding table out of sync with include/c3d.h"


def test_no_gpu_fails_loudly(built):
    """Without a device the product path must raise, never fall back to a CPU solver."""
    from chromosome3d_amd import lib
    L = lib.load()
    if L.c3d_device_count() > 0:
        pytest.skip("a GPU is visible")
    h = C.c_void_p()
    rc = L.c3d_create(0, C.byref(h))
    assert rc == -2 and b"no HIP device" in L.c3d_last_error()
    from chromosome3d_amd import Solver, C3DError
    with pytest.raises(C3DError):
        Solver(0)


def test_product_never_touches_oracle():
    """The oracle is test infrastructure: nothing under chromosome3d_amd/ may reference it."""
    pkg = os.path.join(ROOT, "chromosome3d_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".pl", "Makefile")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "oracle" not in txt.lower(), f"{f} mentions the oracle"


@pytest.mark.parametrize("cid", ["chr21_1mb", "chr22_1mb", "chr13_1mb", "chr1_500kb"])
def test_host_writers_reproduce_reference_files(built, cid, tmp_path):
    """c3d_write_front_half (dist2rr + carr2tbl formats, chromosome3D.pl:156-161,203-205,360)."""
    from chromosome3d_amd import pipeline
    from oracle import oracle as O
    d10 = O.if_to_dist10(load_if(cid))          # the K1 result is checked on the GPU; formats here
    n = pipeline.write_front_half(d10, str(tmp_path), cid)
    assert n == G[cid]["restraints"]
    md5 = lambda p: hashlib.md5(open(p, "rb").read()).hexdigest()
    assert md5(tmp_path / "contact.tbl") == G[cid]["md5_tbl"]
    assert md5(tmp_path / f"{cid}.dist") == G[cid]["md5_dist"]
    assert md5(tmp_path / f"{cid}.rr") == G[cid]["md5_rr"]
    ri, rj, rt = pipeline.read_tbl(str(tmp_path / "contact.tbl"))
    oi, oj, ot = O.dist_to_rr(d10)
    assert np.array_equal(ri, oi) and np.array_equal(rj, oj) and np.array_equal(rt, ot)


def test_host_parser_accepts_reference_text_format(built, tmp_path):
    from chromosome3d_amd import pipeline
    p = os.path.join(GOLD, "inputs", "chr21_1mb_matrix.txt")
    m = pipeline.parse_if_file(p)
    assert np.array_equal(m, load_if("chr21_1mb")) and pipeline.calc_len_IF(p) == 37
    # tabs / multiple blanks / LF-only / leading blanks are all accepted (split /\s+/, :118-126)
    IF = load_if("chr22_1mb")
    q = tmp_path / "m.txt"
    with open(q, "w") as fh:
        for row in IF:
            fh.write("  " + "\t ".join(repr(float(v)) for v in row) + "\n")
    assert np.array_equal(pipeline.parse_if_file(str(q)), IF)
    # ragged / empty input -> error, not a crash
    from chromosome3d_amd import C3DError
    (tmp_path / "bad.txt").write_text("1 2 3\n4 5\n")
    with pytest.raises(C3DError):
        pipeline.parse_if_file(str(tmp_path / "bad.txt"))
    (tmp_path / "empty.txt").write_text("")
    with pytest.raises(C3DError):
        pipeline.parse_if_file(str(tmp_path / "empty.txt"))
    with pytest.raises(C3DError):
        pipeline.parse_if_file(str(tmp_path / "missing.txt"))


@pytest.mark.parametrize("cid", sorted(G))
def test_host_assessment_and_spearman(built, cid):
    from chromosome3d_amd import pipeline
    from oracle import oracle as O
    IF = load_if(cid)
    rows = O.dist_to_rr(O.if_to_dist10(IF))
    X = load_pdb_xyz(model_pdb(cid))
    sat, dev = pipeline.assess(X, rows)
    assert f"{sat}/{len(rows[0])}" == G[cid]["satisfied"] and "%.2f" % dev == "%.2f" % G[cid]["sum_dev"]
    assert abs(pipeline.spearman_IF_pdb(IF, X) - REF_SPEARMAN[cid]) < 5e-5
    assert pipeline.spearman_IF_pdb(IF, X) == pytest.approx(O.spearman_if_dist(IF, X, 3), abs=1e-12)


def test_violation_table_equals_the_references_own(built, tmp_path):
    """A15's file: count_satisfied_tbl_rows (chromosome3D.pl:447-485) leaves contact_violation.txt behind — two '#' lines, then one row
    per restraint, violated rows first and, inside a flag group, in Perl's hash order (the rows are keys of a hash, :475-483: no two runs
    of the reference agree on it).  The golden is the reference's own sub run on the bundled chr21_1mb model (tests/golden/make_golden.pl):
    c3d_write_violations must write the same header, the same MULTISET of rows, the violated ones first, and return the known answers;
    the command-line twin (c3d_score --assess, what the Perl driver uses without the XS binding) the same."""
    import subprocess
    from chromosome3d_amd import pipeline
    from chromosome3d_amd.lib import LIB_PATH
    from oracle import oracle as O
    cid = "chr21_1mb"
    gold = open(os.path.join(GOLD, f"{cid}.contact_violation.txt")).read().splitlines()
    rows = O.dist_to_rr(O.if_to_dist10(load_if(cid)))
    X = load_pdb_xyz(model_pdb(cid))
    out = tmp_path / "viol.txt"
    sat, dev = pipeline.write_violations(X, rows, str(out))
    ours = open(out).read().splitlines()
    assert f"{sat}/{len(rows[0])}" == G[cid]["satisfied"] and "%.2f" % dev == "%.2f" % G[cid]["sum_dev"]
    assert ours[:2] == gold[:2] and len(ours) == len(gold) == 2 + len(rows[0])
    assert sorted(ours[2:]) == sorted(gold[2:])
    flags = [r[:3] for r in ours[2:]]
    assert flags == sorted(flags, reverse=True) and [r[:3] for r in gold[2:]] == flags          # violated ("  1") first, as many as the reference has
    # appended, not overwritten; the CLI twin writes the same table from the files
    pipeline.write_violations(X, rows, str(out), pdb_label="second.pdb")
    assert len(open(out).read().splitlines()) == 2 * len(gold)
    tbl = tmp_path / "contact.tbl"
    tbl.write_bytes(open(os.path.join(GOLD, f"{cid}.contact.tbl"), "rb").read())
    v2 = tmp_path / "viol2.txt"
    r = subprocess.run([os.path.join(os.path.dirname(LIB_PATH), "c3d_score"), "--assess", str(tbl), "0.5", str(v2), model_pdb(cid)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split("\t")[:2] == [G[cid]["satisfied"], "%.2f" % G[cid]["sum_dev"]]
    assert open(v2).read().splitlines()[2:] == ours[2:]


def test_violation_rows_equal_printf_formatting(built, tmp_path):
    """The rows are formatted without printf (c3d_host.cpp put_fixed2 / put_int3).  Against Python's % operator — C's printf — on the
    arithmetic of chromosome3D.pl:447-485 restated here: random coordinates, coordinates on the 0.0005 grid (distances and deviations on exact
    .xx5 ties), three-digit and four-digit residue numbers, large distances."""
    from chromosome3d_amd import pipeline
    rng = np.random.default_rng(21)
    n = 1100
    total = 0
    for trial, scale in enumerate((2.0, 30.0, 900.0, 5.0)):
        x = (rng.normal(size=(n, 3)) * scale).astype(np.float32)
        if scale > 100:
            x = np.abs(x)            # four-digit coordinates as a PDB can hold them: -999.999 .. 9999.999 (anything else is refused, below)
        if trial == 3:
            x = (np.round(x * 2000) / 2000).astype(np.float32)
        R = 4000
        ri = rng.integers(1, n, size=R).astype(np.int32)
        rj = np.minimum(ri + rng.integers(1, 400, size=R), n).astype(np.int32)
        rt10 = rng.integers(1, 30000 if scale > 100 else 900, size=R).astype(np.int32)
        if trial == 3:      # targets that put d - t within a hair of +-0.5 and +-0.2
            xr0 = np.array([[float("%.3f" % v) for v in row] for row in x.astype(np.float64)])
            d0 = np.sqrt(((xr0[ri - 1] - xr0[rj - 1]) ** 2).sum(1))
            rt10 = np.maximum(1, np.round((d0 + rng.choice([-0.5, -0.2, 0.0, 0.2, 0.5], size=R)) * 10)).astype(np.int32)
        out = tmp_path / f"v{trial}.txt"
        sat, dev = pipeline.write_violations(x, (ri, rj, rt10), str(out))
        got = open(out).read().splitlines()[2:]
        xr = np.array([[float("%.3f" % v) for v in row] for row in x.astype(np.float64)])
        want, count = [], 0
        for i, j, t10 in zip(ri, rj, rt10):
            dx = xr[i - 1] - xr[j - 1]
            d = float("%.3f" % np.sqrt(dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2]))
            t = t10 / 10.0
            flag, deviation = 1, d - t
            if d < t + 0.5:
                count += 1; flag = 0; deviation = 0.0
            if d < t - 0.5:
                count -= 1; flag = 1; deviation = -(t - d)
            want.append("%3d\t%.2f\t%.2f # assign45  resid %3d and name ca   resid %3d and name ca  %.2f 0.00 0.00" % (flag, deviation, d, i, j, t))
        assert sat == count
        assert got == [r for r in want if r[2] == "1"] + [r for r in want if r[2] == "0"]
        total += len(got)
    assert total == 16000


def test_violation_writer_error_behaviour(built, tmp_path):
    """Status codes, not crashes: a restraint row outside the model, an unwritable path, null arguments."""
    import ctypes as C
    from chromosome3d_amd import pipeline
    from chromosome3d_amd.lib import C3DError, load
    x = np.zeros((5, 3), dtype=np.float32)
    rows = (np.array([1], dtype=np.int32), np.array([9], dtype=np.int32), np.array([100], dtype=np.int32))
    with pytest.raises(C3DError, match="out of range"):
        pipeline.write_violations(x, rows, str(tmp_path / "v.txt"))
    assert not (tmp_path / "v.txt").exists()                                  # nothing is written before the rows have been checked
    ok = (np.array([1], dtype=np.int32), np.array([5], dtype=np.int32), np.array([100], dtype=np.int32))
    with pytest.raises(C3DError, match="cannot open"):
        pipeline.write_violations(x, ok, str(tmp_path / "no_such_dir" / "v.txt"))
    assert load().c3d_write_violations(None, 5, 0, None, None, None, 0.5, None, None, None, None, None) != 0
    sat, dev = pipeline.write_violations(x, ok, str(tmp_path / "v.txt"))       # d = 0 against a 10 A target: below the lower bound
    assert (sat, dev) == (0, 10.0)
    assert open(tmp_path / "v.txt").read().splitlines()[2] == "  1\t-10.00\t0.00 # assign45  resid   1 and name ca   resid   5 and name ca  10.00 0.00 0.00"


def test_pdb_roundtrip_and_layout(built, tmp_path):
    """Output layout the reference's Perl post-processing expects (parse_pdb_row :674-691,
    get_cns_energy :602-618, add_connect_rows :208-215)."""
    from chromosome3d_amd import pipeline
    x = (np.random.default_rng(1).normal(size=(41, 3)) * 30).astype(np.float32)
    p = tmp_path / "m.pdb"
    pipeline.write_pdb(str(p), x, 12345.678, 10.0, 2.5, title="t_1.pdb")
    lines = open(p).read().splitlines()
    noe = [l for l in lines if l.startswith("REMARK noe")]
    assert len(noe) == 1 and int(float(noe[0].replace(" ", "").split("=")[1])) == 12345
    atoms = [l for l in lines if l.startswith("ATOM")]
    assert len(atoms) == 41
    a = atoms[6]
    assert a[12:16].strip() == "CA" and a[17:20] == "MET" and a[21] == " " and int(a[22:27]) == 7 and int(a[6:11]) == 7
    assert abs(float(a[30:38]) - round(float(x[6, 0]), 3)) < 1e-6
    assert [l for l in lines if l.startswith("CONECT")][0] == "CONECT    1    2"
    assert len([l for l in lines if l.startswith("CONECT")]) == 40 and lines[-1] == "END"
    y = pipeline.read_pdb_ca(str(p))
    assert np.allclose(y, np.round(x.astype(np.float64), 3), atol=1e-3)


def test_fast_rounding_equals_printf_semantics(built):
    """c3d_assess / c3d_spearman round distances like sprintf("%.3f") without going through text
    (fma-residual tie handling).  Check against the oracle, which does go through snprintf, on
    many random and adversarial (exact-tie) coordinates."""
    from chromosome3d_amd import pipeline
    from oracle import oracle as O
    rng = np.random.default_rng(11)
    n = 60
    IF = rng.lognormal(2.0, 1.5, size=(n, n))
    IF = IF + IF.T
    d10 = O.if_to_dist10(IF)
    rows = O.dist_to_rr(d10)
    for trial in range(30):
        x = (rng.normal(size=(n, 3)) * rng.choice([0.5, 5.0, 40.0])).astype(np.float32)
        if trial % 3 == 0:      # coordinates on the 0.0005 grid: distances and roundings hit exact ties
            x = (np.round(x * 2000) / 2000).astype(np.float32)
        xr = np.array([[float("%.3f" % v) for v in row] for row in x.astype(np.float64)])   # what a PDB holds
        sat, dev = pipeline.assess(x, rows)
        sat_o, dev_o = O.assess(xr, rows)
        assert sat == sat_o and abs(dev - dev_o) < 1e-9 * max(1.0, abs(dev_o))
        assert pipeline.spearman_IF_pdb(IF, x) == pytest.approx(O.spearman_if_dist(IF, xr, 3), abs=1e-13)
    xs = (rng.normal(size=(5, n, 3)) * 10).astype(np.float32)
    batch = pipeline.spearman_IF_models(IF, xs)
    assert np.allclose(batch, [pipeline.spearman_IF_pdb(IF, xs[k]) for k in range(5)], atol=0, rtol=0)


def test_if_ranks_radix_and_symmetric_half_equal_the_plain_definition(built):
    """The IF side of the Spearman is ranked by a radix sort, and from the upper half when the matrix is symmetric (c3d_host.cpp
    if_pair_ranks).  Against the oracle and scipy's average ranks: heavy ties, zeros of both signs, negative values, values that
    differ in the last bit, an asymmetric matrix (falls back to all ordered pairs), and range 0/1/3."""
    from chromosome3d_amd import pipeline
    from oracle import oracle as O
    from scipy.stats import spearmanr
    rng = np.random.default_rng(5)
    n = 48
    x = (rng.normal(size=(n, 3)) * 6).astype(np.float32)
    xr = np.array([[float("%.3f" % v) for v in row] for row in x.astype(np.float64)])
    cases = {}
    a = rng.integers(0, 6, size=(n, n)).astype(np.float64)
    cases["ties"] = a + a.T
    b = rng.normal(size=(n, n))
    b = b + b.T
    b[rng.random((n, n)) < 0.2] = 0.0
    b = np.triu(b) + np.triu(b, 1).T
    b[3, 9] = -0.0; b[9, 3] = 0.0
    cases["signed_zeros"] = b
    c = np.full((n, n), 1.0)
    iu = np.triu_indices(n, 1)
    c[iu] = 1.0 + rng.integers(0, 4, size=iu[0].size) * np.finfo(np.float64).eps
    cases["last_bit"] = np.triu(c) + np.triu(c, 1).T
    cases["asymmetric"] = rng.lognormal(1.0, 1.0, size=(n, n))
    for name, IF in cases.items():
        for r in (1, 3):
            got = pipeline.spearman_IF_pdb(IF, x, r)
            assert got == pytest.approx(O.spearman_if_dist(IF, xr, r), abs=1e-13), (name, r)
            i, j = np.nonzero(np.abs(np.subtract.outer(np.arange(n), np.arange(n))) >= r)
            dd = np.round(np.sqrt(((xr[i] - xr[j]) ** 2).sum(1)), 3)
            assert abs(got - spearmanr(IF[i, j], dd)[0]) < 1e-10, (name, r)


def test_cross_resolution_similarity_reproduces_reference_table(built):
    """output_models/similarity.txt (reference data): the 500 kb model reduced to 1 Mb resolution against
    the 1 Mb model of the same chromosome.  Both bundled chr21 models are fixtures; the reduced model must
    also equal the reference's `_reduced` convention (pair means), checked on its defining property."""
    import json
    from chromosome3d_amd import pipeline
    ref = json.load(open(os.path.join(GOLD, "similarity_reference.json")))["chr21_500kb_rank04_a11"]
    a = load_pdb_xyz(model_pdb("chr21_500kb"))
    b = load_pdb_xyz(model_pdb("chr21_1mb"))
    r = pipeline.reduce_model(a)
    assert r.shape == ((len(a) + 1) // 2, 3)
    assert np.allclose(r[: len(a) // 2], 0.5 * (a[0:len(a) // 2 * 2:2] + a[1::2]), rtol=0, atol=1e-12)
    if len(a) % 2:
        assert np.array_equal(r[-1], a[-1])
    rho, rmsd = pipeline.model_similarity(r, b)
    assert abs(rho - ref["spearman"]) < 1e-12
    assert abs(rmsd - ref["rmsd"]) < 1e-10
    # a model is identical to itself; scale invariance of both measures
    rho2, rmsd2 = pipeline.model_similarity(b, 3.0 * b)
    assert abs(rho2 - 1.0) < 1e-12 and rmsd2 < 1e-9


def test_threaded_parser_is_exact_and_reports_bad_tokens(built, tmp_path):
    """Files above 8 MB are parsed by one chunk per host thread (two passes).  Same doubles as Python's
    float() for every entry; a malformed token anywhere, or a missing number, is an error."""
    from chromosome3d_amd import C3DError, pipeline
    n = 720
    rng = np.random.default_rng(5)
    a = rng.random((n, n)) * 10.0 ** rng.integers(-3, 6, size=(n, n))
    p = tmp_path / "big.txt"
    text = "".join(" ".join(repr(float(v)) for v in row) + " \r\n" for row in a)
    assert len(text) > (8 << 20)
    p.write_text(text, newline="")
    assert np.array_equal(pipeline.parse_if_file(str(p)), a)
    bad = text[: len(text) * 3 // 4] + "x" + text[len(text) * 3 // 4 + 1:]
    (tmp_path / "bad.txt").write_text(bad, newline="")
    with pytest.raises(C3DError):
        pipeline.parse_if_file(str(tmp_path / "bad.txt"))
    (tmp_path / "short.txt").write_text(text[: text.rstrip().rfind(" ")], newline="")
    with pytest.raises(C3DError):
        pipeline.parse_if_file(str(tmp_path / "short.txt"))


def test_parser_mid_size_split_and_tokens_outside_from_chars(built, tmp_path):
    """A 500 kb chromosome's ~1.5 MB goes to four threads; numbers are converted by std::from_chars with strtod behind it for what
    from_chars does not take whole (a leading '+', hex, out-of-range exponents).  Same doubles as Python's float() for every entry,
    integers, exponents, signed and subnormal values included."""
    from chromosome3d_amd import C3DError, pipeline
    n = 300
    rng = np.random.default_rng(8)
    a = rng.normal(size=(n, n)) * 10.0 ** rng.integers(-12, 12, size=(n, n))
    a[rng.random((n, n)) < 0.3] = 0.0
    k = rng.random((n, n)) < 0.2
    a[k] = np.round(a[k] % 5000)
    a[0, :6] = [5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, -0.0, 123456789012345678.0, 0.1]
    rows = []
    for row in a:
        rows.append("\t".join(("%d" % v) if (v == np.round(v) and abs(v) < 1e6 and not np.signbit(v)) else repr(float(v)) for v in row))
    text = "\n".join(rows) + "\n"
    assert (256 << 10) < len(text) < (8 << 20)
    p = tmp_path / "mid.txt"
    p.write_text(text)
    got = pipeline.parse_if_file(str(p))
    assert np.array_equal(got, a) and np.signbit(got[0, 3])
    small = tmp_path / "odd.txt"
    small.write_text("+1.5 0x10 1e400\n-1e-400 1E2 .5\n5. inf 7\n")
    got = pipeline.parse_if_file(str(small))
    assert np.array_equal(got, np.array([[1.5, 16.0, np.inf], [-0.0, 100.0, 0.5], [5.0, np.inf, 7.0]]))
    (tmp_path / "bad.txt").write_text(text[: len(text) // 2] + "1.2.3 " + text[len(text) // 2:].split(None, 1)[1])
    with pytest.raises(C3DError):
        pipeline.parse_if_file(str(tmp_path / "bad.txt"))


def test_residue_names_follow_an_installed_sequence(built, tmp_path):
    """A reference run names residue i after letter i of its fixed pseudo-protein (chromosome3D.pl:93-98; first residues ARG SER
    GLU ASP TRP GLN CYS); its bundled models carry MET everywhere, which stays the default.  With the
    sequence installed the writer names the beads accordingly, the shaping of :813-820 keeps the names, and beads beyond the
    sequence's end are MET."""
    import numpy as np
    from chromosome3d_amd import pipeline
    seq = pipeline.read_fasta(pipeline.REFSEQUENCE_FASTA)
    assert len(seq) == 663 and seq.startswith("RSEDWQC")
    xyz = np.cumsum(np.full((670, 3), 1.5, dtype=np.float32), axis=0)
    p = str(tmp_path / "m.pdb")
    try:
        pipeline.set_residue_sequence(seq)
        pipeline.write_pdb(p, xyz, 1.0, 2.0, 3.0, title="m.pdb")
        names = [l[17:20] for l in open(p) if l.startswith("ATOM")]
        assert names[:7] == ["ARG", "SER", "GLU", "ASP", "TRP", "GLN", "CYS"] and len(names) == 670
        assert names[662] != "" and names[663:] == ["MET"] * 7
        pipeline.shape_pdb(p)
        shaped = [l[17:20] for l in open(p) if l.startswith("ATOM")]
        assert shaped == names
    finally:
        pipeline.set_residue_sequence(None)
    pipeline.write_pdb(p, xyz[:5])
    assert [l[17:20] for l in open(p) if l.startswith("ATOM")] == ["MET"] * 5


def test_host_helpers_refuse_coordinates_a_pdb_cannot_hold(tmp_path):
    """Round 4's review: non-finite or huge coordinates went through round_dec3 and an out-of-range float -> integer conversion and came
    back as C3D_OK with rows like `-9223372036854776.32`.  Every public host helper that takes coordinates now returns C3D_ERR_INVALID
    with a message naming the bead — and writes nothing (tools/sanitize/run.sh drives the same under ASan + UBSan float-cast-overflow)."""
    from chromosome3d_amd import lib, pipeline
    rng = np.random.default_rng(5)
    n = 40
    good = (rng.normal(size=(n, 3)) * 10).astype(np.float32)
    ri, rj, rt = np.array([1, 2, 3], dtype=np.int32), np.array([10, 20, 30], dtype=np.int32), np.array([50, 60, 70], dtype=np.int32)
    IF = np.abs(rng.normal(size=(n, n))) + 1.0
    IF = IF + IF.T
    for bad in (np.inf, -np.inf, np.nan, 1e30, -1e30, 1e7, 10000.0, -1000.0):
        x = good.copy()
        x[6, 2] = bad
        out = tmp_path / "v.txt"
        for call in (lambda: pipeline.write_violations(x, (ri, rj, rt), str(out)), lambda: pipeline.assess(x, (ri, rj, rt)),
                     lambda: pipeline.write_pdb(str(tmp_path / "m.pdb"), x), lambda: pipeline.spearman_IF_pdb(IF, x),
                     lambda: pipeline.spearman_IF_models(IF, np.stack([good, x]))):
            with pytest.raises(lib.C3DError) as ei:
                call()
            assert "error -1" in str(ei.value) and ("bead 7" in str(ei.value) or "bead 47" in str(ei.value)), str(ei.value)
        assert not out.exists() and not (tmp_path / "m.pdb").exists()
    with pytest.raises(lib.C3DError):
        pipeline.model_similarity(np.where(np.arange(3 * n).reshape(n, 3) == 5, np.nan, good.astype(np.float64)), good)
    # the edge of the range is still taken, and fits its columns
    x = good.copy()
    x[0] = (9999.999, -999.999, 0.0)
    pipeline.write_pdb(str(tmp_path / "edge.pdb"), x)
    row = [l for l in open(tmp_path / "edge.pdb") if l.startswith("ATOM")][0]
    assert row[30:54] == "9999.999-999.999   0.000"
    assert np.allclose(pipeline.read_pdb_ca(str(tmp_path / "edge.pdb"))[0], x[0])


def test_executor_and_loader_under_thread_sanitizer(tmp_path):
    """csrc/c3d_api.cpp (context, code-object loader, executor of c3d_run) and csrc/c3d_batch_main.cpp (device lists, lanes, XCD broker)
    as they are, built with -fsanitize=thread against a fake HIP layer (tools/sanitize/hip_stub.cpp: device memory is host memory, kernels
    compute nothing) and driven through the start that met a device exception in round 5 — eight contexts of one process on one device —,
    the production shape (8 devices x 3 lanes) and an API storm through every code object.  Passes when ThreadSanitizer reports nothing
    and the stub saw no code-object load overlap a launch (the contract of c3d_api.cpp "code objects")."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    probe = subprocess.run(["g++", "-fsanitize=thread", "-x", "c++", "-", "-o", str(tmp_path / "probe")], input="int main(){return 0;}", capture_output=True, text=True)
    if probe.returncode != 0:
        pytest.skip("g++ has no ThreadSanitizer runtime here")
    ran = subprocess.run([str(tmp_path / "probe")], capture_output=True, text=True)
    if ran.returncode != 0:                                  # e.g. "unexpected memory mapping" under an ASLR setting TSan cannot live with
        pytest.skip("ThreadSanitizer binaries do not start on this box: " + ran.stderr[-200:])
    out = subprocess.run(["bash", os.path.join(root, "tools", "sanitize", "run.sh"), "executor"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    last = out.stdout.strip().splitlines()[-1]
    assert last.startswith("executor under TSan:") and last.endswith(" 0 load/launch overlaps"), last
