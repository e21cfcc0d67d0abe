"""Host-side mirror of the reference driver's interface for the hot path.

Same names, argument meaning and artefacts as chromosome3D.pl so that tests read like the
reference's flow (file:line under the reference tree):

    calc_len_IF      :164-179      IF2dist_new :110-162     dist2rr :181-206
    carr2tbl         :340-362      build_models :254-289    assess_dgsa :769-829
    spearman_IF_pdb  spearman_IF_pdb.pl:15-76

All arithmetic happens in libc3d.so (HIP kernels + host C++); this file only moves buffers and
file names around.
"""
import ctypes as C
import os

import numpy as np

from . import lib as _l
from .solver import Solver, default_fire, default_model, default_schedule

KSCALING = 11       # chromosome3D.pl:18
ALPHA = 0.5         # :19
SEPARATION = 5      # :20
MODELCOUNT = 20     # :21
MD_SEED = 82364     # :980
DISTRELAX = 0.5     # :74


def parse_if_file(path):
    """N x N float64 matrix from the whitespace-separated text format (:116-129)."""
    L = _l.load()
    p = C.POINTER(C.c_double)()
    n = C.c_int()
    _l.check(L.c3d_parse_if_file(os.fsencode(path), C.byref(p), C.byref(n)))
    m = np.ctypeslib.as_array(p, shape=(n.value, n.value)).copy()
    L.c3d_free(p)
    return m


def calc_len_IF(path):
    return parse_if_file(path).shape[0]


def IF2dist_new(solver, IF, K=KSCALING, alpha=ALPHA):
    """K1 on the GPU.  Returns the quantised distances in tenths of an Angstrom (int32 N x N),
    i.e. the numbers the reference prints into <ID>.dist."""
    solver.set_if_matrix(IF, alpha=float(alpha), K=float(K))
    return solver.dist10()


def write_front_half(dist10, out_dir, ID, min_sep=SEPARATION):
    """<ID>.dist, <ID>.rr (dist2rr) and contact.tbl (carr2tbl) byte-for-byte as the reference."""
    L = _l.load()
    d = np.ascontiguousarray(dist10, dtype=np.int32)
    nres = C.c_int()
    _l.check(L.c3d_write_front_half(_l.i32ptr(d), d.shape[0], min_sep,
                                    os.fsencode(os.path.join(out_dir, f"{ID}.dist")),
                                    os.fsencode(os.path.join(out_dir, f"{ID}.rr")),
                                    os.fsencode(os.path.join(out_dir, "contact.tbl")), C.byref(nres)))
    return nres.value


def read_tbl(path):
    L = _l.load()
    pi, pj, pt = (C.POINTER(C.c_int32)() for _ in range(3))
    R = C.c_int()
    _l.check(L.c3d_read_tbl(os.fsencode(path), C.byref(pi), C.byref(pj), C.byref(pt), C.byref(R)))
    r = max(R.value, 1)
    out = tuple(np.ctypeslib.as_array(p, shape=(r,))[:R.value].copy() for p in (pi, pj, pt))
    for p in (pi, pj, pt):
        L.c3d_free(p)
    return out


def restraints_from_dist10(dist10, min_sep=SEPARATION):
    """(i, j, t10) rows i<j, 1-based, row-major order (order does not matter to the solver)."""
    n = dist10.shape[0]
    i, j = np.triu_indices(n, min_sep)
    m = dist10[i, j] > 0
    return (i[m] + 1).astype(np.int32), (j[m] + 1).astype(np.int32), dist10[i, j][m].astype(np.int32)


REFSEQUENCE_FASTA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "refsequence.fasta")


def read_fasta(path):
    """letters of a FASTA / plain sequence file (header lines skipped) — seq_fasta, chromosome3D.pl:731-744"""
    return "".join(l.strip() for l in open(path) if not l.startswith(">"))


def set_residue_sequence(seq1):
    """Residue names of the models written from now on: None -> all MET (the bundled output_models), a string of one-letter
    codes -> residue i named after letter i as a reference run does (chromosome3D.pl:93-98); read_fasta(REFSEQUENCE_FASTA) is
    the reference's pseudo-protein."""
    _l.check(_l.load().c3d_set_residue_sequence(seq1.encode() if seq1 else None))


def write_pdb(path, xyz, e_noe=0.0, e_bond=0.0, e_rep=0.0, title=None):
    x = _l.as_f32(xyz)
    _l.check(_l.load().c3d_write_pdb(os.fsencode(path), _l.fptr(x), x.shape[0], e_noe, e_bond, e_rep,
                                     title.encode() if title else None))


def shape_pdb(in_path, out_path=None, log_path=None):
    """A16: filter_nonCA + reindex_chain + sed + add_connect_rows (:813-820) — the file assess_dgsa leaves behind."""
    _l.check(_l.load().c3d_shape_pdb(os.fsencode(in_path), os.fsencode(out_path or in_path),
                                     os.fsencode(log_path) if log_path else None))


def read_pdb_ca(path):
    L = _l.load()
    p = C.POINTER(C.c_float)()
    n = C.c_int()
    _l.check(L.c3d_read_pdb_ca(os.fsencode(path), C.byref(p), C.byref(n)))
    x = np.ctypeslib.as_array(p, shape=(n.value, 3)).copy()
    L.c3d_free(p)
    return x


def assess(xyz, rows, relax=DISTRELAX):
    """count_satisfied_tbl_rows + sum_noe_dev (:447-485, :581-600) -> (satisfied, sum_dev)."""
    ri, rj, rt = (np.ascontiguousarray(a, dtype=np.int32) for a in rows)
    x = _l.as_f32(xyz)
    sat, dev = C.c_int(), C.c_double()
    _l.check(_l.load().c3d_assess(_l.fptr(x), x.shape[0], len(ri), _l.i32ptr(ri), _l.i32ptr(rj), _l.i32ptr(rt),
                                  relax, C.byref(sat), C.byref(dev)))
    return sat.value, dev.value


def write_violations(xyz, rows, path, pdb_label="model.pdb", tbl_label="contact.tbl", relax=DISTRELAX):
    """assess() plus the violation table count_satisfied_tbl_rows leaves behind (:475-483), appended to `path`."""
    ri, rj, rt = (np.ascontiguousarray(a, dtype=np.int32) for a in rows)
    x = _l.as_f32(xyz)
    sat, dev = C.c_int(), C.c_double()
    _l.check(_l.load().c3d_write_violations(_l.fptr(x), x.shape[0], len(ri), _l.i32ptr(ri), _l.i32ptr(rj), _l.i32ptr(rt), relax,
                                            pdb_label.encode(), tbl_label.encode(), path.encode(), C.byref(sat), C.byref(dev)))
    return sat.value, dev.value


def spearman_IF_pdb(IF, xyz, rng=3):
    """Spearman(IF_ij, d_ij) over ordered pairs |i-j| >= rng; negative for good models."""
    IF = np.ascontiguousarray(IF, dtype=np.float64)
    x = _l.as_f32(xyz)
    rho = C.c_double()
    _l.check(_l.load().c3d_spearman_if_dist(_l.dptr(IF), _l.fptr(x), IF.shape[0], rng, C.byref(rho)))
    return rho.value


def spearman_IF_models(IF, xyz, rng=3):
    """spearman_IF_pdb for a stack of models [M, N, 3] of one matrix (IF ranked once)."""
    IF = np.ascontiguousarray(IF, dtype=np.float64)
    x = _l.as_f32(xyz)
    rho = np.empty(x.shape[0], dtype=np.float64)
    _l.check(_l.load().c3d_spearman_if_dist_batch(_l.dptr(IF), _l.fptr(x), IF.shape[0], x.shape[0], rng, _l.dptr(rho)))
    return rho


def reduce_model(xyz):
    """500 kb model -> 1 Mb resolution as the reference's `*_reduced.pdb` files: mean of consecutive bead pairs."""
    x = np.ascontiguousarray(xyz, dtype=np.float64)
    out = np.empty(((x.shape[0] + 1) // 2, 3), dtype=np.float64)
    _l.check(_l.load().c3d_reduce_model(_l.dptr(x), x.shape[0], _l.dptr(out)))
    return out


def model_similarity(xa, xb):
    """(Spearman, "RMSD") of two models' pairwise distances as in output_models/similarity.txt; the longer
    model is cut to the shorter one's length (a reduced 500 kb model can have one bead more or less)."""
    n = min(len(xa), len(xb))
    a = np.ascontiguousarray(np.asarray(xa, dtype=np.float64)[:n])
    b = np.ascontiguousarray(np.asarray(xb, dtype=np.float64)[:n])
    rho, rmsd = C.c_double(), C.c_double()
    _l.check(_l.load().c3d_model_similarity(_l.dptr(a), _l.dptr(b), n, C.byref(rho), C.byref(rmsd)))
    return rho.value, rmsd.value


def build_models(solver, model_count=MODELCOUNT, seed=MD_SEED, first_replica=0, model=None, stages=None, fire=None,
                 gtol=1e-2, check_every=250):
    """The replacement of `cns_solve < dgsa.inp` (:254-289): runs the whole annealing schedule
    for model_count replicas on the GPU and returns (xyz [M,N,3], energies [M,3])."""
    solver.set_model(model if model is not None else default_model())
    solver.set_schedule(stages if stages is not None else default_schedule(), fire or default_fire(), gtol, check_every)
    solver.init_replicas(model_count, seed, first_replica)
    solver.run()
    return solver.coords(), solver.energies()


def assess_dgsa(out_dir, ID, xyz, energies, rows, top=5, quiet=False):
    """Rank by int(E_noe) ascending (:796-802), print the satisfaction table (:804-810), write every
    model as <ID>_<k>.pdb, shape it as :813-820 does (CA rows renumbered, CONECT, REMARKs to model_info.log)
    and rename the best `top` to <ID>_model<i>.pdb (:822-828)."""
    M = xyz.shape[0]
    order = sorted(range(M), key=lambda r: (int(energies[r, 0]), r))
    report = []
    names = {}
    for r in range(M):
        names[r] = os.path.join(out_dir, f"{ID}_{r + 1}.pdb")
        write_pdb(names[r], xyz[r], *energies[r], title=os.path.basename(names[r]))
    say = (lambda *a: None) if quiet else print
    say(f"NOE_SATISFIED(+-{DISTRELAX}A)  SUM_OF_DEVIATIONS>= 0.2  PDB")
    for r in reversed(order):
        sat, dev = assess(xyz[r], rows)
        report.append((r, sat, dev))
        say("%-9s             %-9s                %-25s" % (f"{sat}/{len(rows[0])}", "%.2f" % dev,
                                                            os.path.basename(names[r])[:-4]))
    for r in reversed(order):
        shape_pdb(names[r], None, os.path.join(out_dir, "model_info.log"))
    for k, r in enumerate(order[:top]):
        dst = os.path.join(out_dir, f"{ID}_model{k + 1}.pdb")
        os.replace(names[r], dst)
        say(f"model{k + 1}.pdb <= {os.path.basename(names[r])}")
    return order, report


def reconstruct(matrix_path, out_dir, K=KSCALING, alpha=ALPHA, model_count=MODELCOUNT, device=0, **kw):
    """chromosome3D.pl top level (:86-106) for one matrix: front half, models, assessment."""
    os.makedirs(out_dir, exist_ok=True)
    ID = os.path.basename(matrix_path)
    ID = ID[:-4] if ID.endswith(".txt") else ID
    IF = parse_if_file(matrix_path)
    s = Solver(device)
    try:
        d10 = IF2dist_new(s, IF, K, alpha)
        nres = write_front_half(d10, out_dir, ID)
        print(f"Restraints : {nres} lines in tbl file")
        xyz, en = build_models(s, model_count, **kw)
        rows = restraints_from_dist10(d10)
        order, _ = assess_dgsa(out_dir, ID, xyz, en, rows)
        rho = list(spearman_IF_models(IF, xyz))
        return dict(ID=ID, n=IF.shape[0], restraints=nres, order=order, spearman=rho, energies=en, xyz=xyz)
    finally:
        s.close()
