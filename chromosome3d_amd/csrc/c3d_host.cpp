// c3d_host.cpp — host-side formats of the reference (no device code): IF matrix reader,
// <ID>.dist / <ID>.rr / contact.tbl writers, contact.tbl reader, PDB writer/reader,
// restraint-satisfaction report and the Spearman(IF, d) metric.  Part of libc3d.so.
//
// Reference (file:line): chromosome3D.pl:116-129,164-179 (matrix parse), :156-161 (.dist),
// :181-206 (.rr, Perl string sort), :340-362 (contact.tbl), :497-520 (tbl parse rules),
// :447-485,581-600,716-729 (assessment), :853-857,208-215 (final PDB layout),
// :93-94 (pseudo-sequence), spearman_IF_pdb.pl:42-70 (metric).
#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/c3d.h"
#include "c3d_host.h"

namespace c3d {
thread_local std::string g_err;
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
}  // namespace c3d

using c3d::fail;

extern "C" const char* c3d_last_error(void) { return c3d::g_err.c_str(); }
extern "C" const char* c3d_version(void) { return "chromosome3d_amd 0.1 (gfx950)"; }
extern "C" void c3d_free(void* p) { free(p); }

static bool read_file(const char* path, std::string& out) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, k);
    fclose(f);
    return true;
}
static inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\f' || c == '\v'; }

extern "C" int c3d_parse_if_file(const char* path, double** IF, int* n_out) {
    if (!path || !IF || !n_out) return fail(C3D_ERR_INVALID, "c3d_parse_if_file: null argument");
    std::string txt;
    if (!read_file(path, txt)) return fail(C3D_ERR_IO, std::string("cannot read IF matrix ") + path);
    // N = number of fields on the first non-empty line (calc_len_IF)
    size_t p = 0;
    int n = 0;
    bool in_tok = false;
    while (p < txt.size() && txt[p] != '\n') {
        const bool ws = is_ws(txt[p]);
        if (!ws && !in_tok) { ++n; in_tok = true; }
        if (ws) in_tok = false;
        ++p;
    }
    if (n < 1) return fail(C3D_ERR_IO, std::string("IF matrix has an empty first line: ") + path);
    const size_t nn = (size_t)n * n;
    double* m = (double*)malloc(sizeof(double) * nn);
    if (!m) return fail(C3D_ERR_NOMEM, "out of memory");
    // Large matrices (110 MB of text at N = 2500) are split at whitespace into one chunk per host thread:
    // pass 1 counts the tokens of every chunk, pass 2 converts them (correctly rounded, so the result does not
    // depend on the split); a 500 kb chromosome's 1.5 MB takes four threads.
    const char* base = txt.c_str();
    const size_t len = txt.size();
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    unsigned T = len > (size_t)(8u << 20) ? std::min(16u, hw) : len > (size_t)(256u << 10) ? std::min(4u, hw) : 1u;
    std::vector<size_t> cut(T + 1, len);
    cut[0] = 0;
    for (unsigned t = 1; t < T; ++t) {
        size_t q = std::max(cut[t - 1], len / T * t);
        while (q < len && !is_ws(base[q])) ++q;       // never inside a token
        cut[t] = q;
    }
    std::vector<size_t> count(T, 0), first(T + 1, 0);
    std::vector<int> bad(T, 0);
    auto each_chunk = [&](auto&& fn) {
        if (T == 1) { fn(0u); return; }
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; ++t) th.emplace_back(fn, t);
        for (auto& x : th) x.join();
    };
    each_chunk([&](unsigned t) {
        size_t c = 0;
        bool in = false;
        for (size_t q = cut[t]; q < cut[t + 1]; ++q) {
            const bool ws = is_ws(base[q]);
            if (!ws && !in) ++c;
            in = !ws;
        }
        count[t] = c;
    });
    for (unsigned t = 0; t < T; ++t) first[t + 1] = first[t] + count[t];
    const size_t cnt = first[T];
    if (cnt == nn) {
        each_chunk([&](unsigned t) {
            const char* s = base + cut[t];
            const char* end = base + cut[t + 1];
            size_t k = first[t];
            while (s < end) {
                while (s < end && is_ws(*s)) ++s;
                if (s >= end) break;
                // std::from_chars (correctly rounded like strtod, a third of its time); what it does not take whole — a leading '+', hex,
                // a value out of range — goes to strtod, which stops at the token's end: the text is NUL-terminated
                double v;
                const std::from_chars_result fc = std::from_chars(s, end, v);
                const char* e = fc.ptr;
                if (fc.ec != std::errc() || (e < base + len && !is_ws(*e))) {
                    char* se;
                    v = strtod(s, &se);
                    e = se;
                    if (e == s || (e < base + len && !is_ws(*e))) { bad[t] = 1; return; }
                }
                m[k++] = v;
                s = e;
            }
        });
        for (unsigned t = 0; t < T; ++t)
            if (bad[t]) { free(m); return fail(C3D_ERR_IO, std::string("non-numeric token in IF matrix ") + path); }
    }
    if (cnt != nn) {
        free(m);
        char b[256];
        snprintf(b, sizeof b, "IF matrix %s: expected %d x %d = %zu numbers, found %zu", path, n, n, nn, cnt);
        return fail(C3D_ERR_IO, b);
    }
    *IF = m;
    *n_out = n;
    return C3D_OK;
}

namespace {
// buffered text output with hand-rolled integer formatting (fprintf costs ~1 us per row: 3 M rows at N = 2500)
struct TextOut {
    FILE* f = nullptr;
    std::string buf;
    bool open(const char* path) { f = fopen(path, "w"); buf.reserve(1 << 20); return f != nullptr; }
    void flush() { if (!buf.empty()) { fwrite(buf.data(), 1, buf.size(), f); buf.clear(); } }
    void close() { flush(); fclose(f); f = nullptr; }
    void lit(const char* t) { buf.append(t); }
    void ch(char c) { buf.push_back(c); }
    void num(int v, int width = 0) {     // "%<width>d" for v >= 0
        char t[12];
        int k = 0;
        do { t[k++] = (char)('0' + v % 10); v /= 10; } while (v);
        for (int q = k; q < width; ++q) buf.push_back(' ');
        while (k) buf.push_back(t[--k]);
    }
    void tenths(int32_t t10, bool two_decimals) {   // "%.1f" / "%.2f" of t10 / 10 (exact)
        if (t10 < 0) { buf.push_back('-'); t10 = -t10; }
        num(t10 / 10);
        buf.push_back('.');
        buf.push_back((char)('0' + t10 % 10));
        if (two_decimals) buf.push_back('0');
    }
    void maybe_flush() { if (buf.size() > (1u << 20) - 256) flush(); }
};
}  // namespace

extern "C" int c3d_write_front_half(const int32_t* dist10, int n, int min_sep, const char* dist_path, const char* rr_path,
                                    const char* tbl_path, int* n_restraints) {
    if (!dist10 || n < 1) return fail(C3D_ERR_INVALID, "c3d_write_front_half: bad arguments");
    if (dist_path) {
        TextOut o;
        if (!o.open(dist_path)) return fail(C3D_ERR_IO, std::string("cannot write ") + dist_path);
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < n; ++j) {
                o.tenths(dist10[(size_t)i * n + j], false);
                o.ch(' ');
                o.maybe_flush();
            }
            o.ch('\n');
        }
        o.close();
    }
    // Perl `sort keys` on "i j": byte-wise string order.  A space sorts before every digit, so that is the
    // order of the decimal strings of i, then of j: enumerate both in that order instead of sorting R rows.
    std::vector<int> lex(n);
    {
        std::vector<std::string> name(n);
        for (int k = 0; k < n; ++k) { lex[k] = k + 1; name[k] = std::to_string(k + 1); }
        std::sort(lex.begin(), lex.end(), [&](int a, int b) { return name[a - 1] < name[b - 1]; });
    }
    TextOut rr, tbl;
    if (rr_path && !rr.open(rr_path)) return fail(C3D_ERR_IO, std::string("cannot write ") + rr_path);
    if (tbl_path && !tbl.open(tbl_path)) { if (rr.f) rr.close(); return fail(C3D_ERR_IO, std::string("cannot write ") + tbl_path); }
    int R = 0;
    for (int a = 0; a < n; ++a) {
        const int i = lex[a];
        for (int b = 0; b < n; ++b) {
            const int j = lex[b];
            if (j - i < min_sep) continue;
            const int32_t t = dist10[(size_t)(i - 1) * n + (j - 1)];
            if (t <= 0) continue;
            ++R;
            if (rr.f) {
                rr.num(i); rr.ch(' '); rr.num(j); rr.ch(' '); rr.tenths(t, true); rr.ch(' '); rr.tenths(t, true); rr.lit(" 1.0\n");
                rr.maybe_flush();
            }
            if (tbl.f) {
                tbl.lit("assign45 (resid "); tbl.num(i, 3); tbl.lit(" and name ca) (resid "); tbl.num(j, 3);
                tbl.lit(" and name ca) "); tbl.tenths(t, true); tbl.lit(" 0.00 0.00\n");
                tbl.maybe_flush();
            }
        }
    }
    if (rr.f) rr.close();
    if (tbl.f) tbl.close();
    if (n_restraints) *n_restraints = R;
    return C3D_OK;
}

extern "C" int c3d_read_tbl(const char* path, int32_t** ri, int32_t** rj, int32_t** rt10, int* R) {
    if (!path || !ri || !rj || !rt10 || !R) return fail(C3D_ERR_INVALID, "c3d_read_tbl: null argument");
    std::string txt;
    if (!read_file(path, txt)) return fail(C3D_ERR_IO, std::string("cannot read ") + path);
    std::vector<int32_t> vi, vj, vt;
    // rows "assign (resid I and name ca) (resid J and name ca) D DMINUS DPLUS": parentheses count as blanks; the tokens are scanned in place
    // (a std::string per token made this 101 426-row file the slowest part of an assessment)
    for (char& c : txt) if (c == '(' || c == ')') c = ' ';
    const char* base = txt.c_str();
    size_t p = 0;
    while (p < txt.size()) {
        size_t e = txt.find('\n', p);
        if (e == std::string::npos) e = txt.size();
        const char* tok[14];
        int nt = 0;
        size_t q = p;
        while (q < e) {
            while (q < e && is_ws(base[q])) ++q;
            const size_t b0 = q;
            while (q < e && !is_ws(base[q])) ++q;
            if (q > b0) { if (nt < 14) tok[nt] = base + b0; ++nt; }
        }
        const std::string line = nt == 0 || (nt >= 14 && strncmp(tok[0], "assign", 6) == 0) ? std::string() : txt.substr(p, e - p);
        const size_t p0 = p;
        p = e + 1;
        if (nt == 0) continue;
        if (strncmp(tok[0], "assign", 6) != 0) return fail(C3D_ERR_IO, std::string("contact.tbl: unexpected row: ") + line);
        if (nt < 14) return fail(C3D_ERR_IO, std::string("contact.tbl: short row: ") + line);
        vi.push_back(atoi(tok[2]));                  // (atoi / strtod stop at the token's end: a blank, or the text's final NUL)
        vj.push_back(atoi(tok[7]));
        double tv;
        const std::from_chars_result fc = std::from_chars(tok[11], base + e, tv);
        if (fc.ec != std::errc() || (fc.ptr < base + e && !is_ws(*fc.ptr))) tv = strtod(tok[11], nullptr);
        if (!(std::fabs(tv) < 1e7) || vi.back() < 1 || vj.back() < 1)       // NaN fails the comparison; atoi of a non-number gives 0
            return fail(C3D_ERR_IO, std::string("contact.tbl: bad residue number or distance: ") + txt.substr(p0, e - p0));
        vt.push_back((int32_t)llround(tv * 10.0));
    }
    const size_t r = vi.size();
    *ri = (int32_t*)malloc(sizeof(int32_t) * (r ? r : 1));
    *rj = (int32_t*)malloc(sizeof(int32_t) * (r ? r : 1));
    *rt10 = (int32_t*)malloc(sizeof(int32_t) * (r ? r : 1));
    if (!*ri || !*rj || !*rt10) return fail(C3D_ERR_NOMEM, "out of memory");
    for (size_t k = 0; k < r; ++k) { (*ri)[k] = vi[k]; (*rj)[k] = vj[k]; (*rt10)[k] = vt[k]; }
    *R = (int)r;
    return C3D_OK;
}

// Residue names carry no physics in the bead model; any of the 20 standard names keeps the
// reference's reindex_chain / seq_chain (chromosome3D.pl:847, :225) happy.  The bundled
// output_models use MET for every bead, so do we.
// residue names: one-letter codes of the pseudo-protein a caller may install (chromosome3D.pl:93-98), else MET
namespace {
std::mutex g_seq_mu;
std::string g_seq1;
const char* three_letter(char c) {     // %AA1TO3 (chromosome3D.pl:77-78)
    switch (c) {
        case 'A': return "ALA"; case 'N': return "ASN"; case 'C': return "CYS"; case 'Q': return "GLN"; case 'H': return "HIS";
        case 'L': return "LEU"; case 'M': return "MET"; case 'P': return "PRO"; case 'T': return "THR"; case 'Y': return "TYR";
        case 'R': return "ARG"; case 'D': return "ASP"; case 'E': return "GLU"; case 'G': return "GLY"; case 'I': return "ILE";
        case 'K': return "LYS"; case 'F': return "PHE"; case 'S': return "SER"; case 'W': return "TRP"; case 'V': return "VAL";
        default: return "MET";
    }
}
}  // namespace
extern "C" int c3d_set_residue_sequence(const char* seq1) {
    std::lock_guard<std::mutex> lk(g_seq_mu);
    g_seq1.clear();
    if (seq1)
        for (const char* p = seq1; *p; ++p)
            if (!is_ws(*p)) g_seq1.push_back((char)toupper((unsigned char)*p));
    return C3D_OK;
}

// Coordinates every host helper below accepts: finite and inside what a PDB's %8.3f columns hold (chromosome3D.pl:678-687 reads x, y, z
// from columns 31-54; a value that does not fit its eight columns shifts every later field).  Anything else is C3D_ERR_INVALID with a
// message — never rows of garbage, never an out-of-range float -> integer conversion (the device path reports a trajectory that blew up
// as C3D_ERR_DIVERGED; these helpers take coordinates from any caller).
static int check_coords(const float* xyz, size_t count, const char* who) {
    for (size_t k = 0; k < count; ++k) {
        const double v = (double)xyz[k];
        if (!(v > -999.9995 && v < 9999.9995)) {       // NaN fails both comparisons
            char msg[160];
            snprintf(msg, sizeof msg, "%s: coordinate %zu of bead %zu is %s (a PDB holds -999.999 .. 9999.999)", who, k % 3, k / 3 + 1,
                     std::isfinite(v) ? "out of range" : "not finite");
            return fail(C3D_ERR_INVALID, msg);
        }
    }
    return C3D_OK;
}
static int check_coords_d(const double* xyz, size_t count, const char* who) {
    for (size_t k = 0; k < count; ++k)
        if (!(std::fabs(xyz[k]) < 1e6)) return fail(C3D_ERR_INVALID, std::string(who) + ": coordinates not finite or out of range");
    return C3D_OK;
}

extern "C" int c3d_write_pdb(const char* path, const float* xyz, int n, double e_noe, double e_bond, double e_rep,
                             const char* title) {
    if (!path || !xyz || n < 1) return fail(C3D_ERR_INVALID, "c3d_write_pdb: bad arguments");
    if (!std::isfinite(e_noe) || !std::isfinite(e_bond) || !std::isfinite(e_rep)) return fail(C3D_ERR_INVALID, "c3d_write_pdb: energy not finite");
    if (int rc = check_coords(xyz, (size_t)3 * n, "c3d_write_pdb")) return rc;
    FILE* f = fopen(path, "w");
    if (!f) return fail(C3D_ERR_IO, std::string("cannot write ") + path);
    fprintf(f, "REMARK FILENAME=\"%s\"\n", title ? title : path);
    fprintf(f, "REMARK ===============================================================\n");
    fprintf(f, "REMARK overall = %.4f\n", e_noe + e_bond + e_rep);
    fprintf(f, "REMARK bon = %.4f\n", e_bond);
    fprintf(f, "REMARK vdw = %.4f\n", e_rep);
    fprintf(f, "REMARK noe = %.4f\n", e_noe);
    fprintf(f, "REMARK ===============================================================\n");
    std::string seq;
    { std::lock_guard<std::mutex> lk(g_seq_mu); seq = g_seq1; }
    for (int i = 0; i < n; ++i) {
        const char* rn = (size_t)i < seq.size() ? three_letter(seq[i]) : "MET";
        // cols: 1-6 ATOM, 7-11 serial, 13-16 name, 18-20 resName, 22 chain(blank), 23-26 resSeq, 31-54 xyz
        fprintf(f, "ATOM  %5d  CA  %3s  %4d    %8.3f%8.3f%8.3f  1.00  0.00\n", (i + 1) % 100000, rn, (i + 1) % 10000,
                (double)xyz[3 * i], (double)xyz[3 * i + 1], (double)xyz[3 * i + 2]);
    }
    for (int i = 1; i < n; ++i) fprintf(f, "CONECT%5d%5d\n", i, i + 1);
    fprintf(f, "END\n");
    fclose(f);
    return C3D_OK;
}

// A16 (chromosome3D.pl:813-820): filter_nonCA :864-880, reindex_chain :831-862, sed :818, add_connect_rows :208-215
extern "C" int c3d_shape_pdb(const char* in_path, const char* out_path, const char* log_path) {
    if (!in_path || !out_path) return fail(C3D_ERR_INVALID, "c3d_shape_pdb: null argument");
    std::string txt;
    if (!read_file(in_path, txt)) return fail(C3D_ERR_IO, std::string("cannot read ") + in_path);
    static const char* kResidues[] = {"ALA", "ASN", "CYS", "GLN", "HIS", "LEU", "MET", "PRO", "THR", "TYR",
                                      "ARG", "ASP", "GLU", "GLY", "ILE", "LYS", "PHE", "SER", "TRP", "VAL"};   // %AA3TO1, :77
    auto field = [](const std::string& row, size_t pos, size_t len) {   // Perl substr + s/\s+//g
        std::string f = pos < row.size() ? row.substr(pos, len) : std::string();
        f.erase(std::remove_if(f.begin(), f.end(), [](char ch) { return is_ws(ch); }), f.end());
        return f;
    };
    std::string log = std::string(in_path), body;   // print2line :870 writes the name without a line end
    std::string prev_rnum = "XX";
    int res_counter = 0, atom_counter = 0, n_ca = 0;
    size_t p = 0;
    while (p < txt.size()) {
        size_t e = txt.find('\n', p);
        const bool had_nl = e != std::string::npos;
        if (!had_nl) e = txt.size();
        const std::string row = txt.substr(p, e - p);
        p = e + 1;
        if (row.compare(0, 6, "REMARK") == 0) log += row + "\n";                 // filter_nonCA :873
        if (row.compare(0, 4, "ATOM") != 0) continue;                              // :874
        if (row.find("CA") == std::string::npos) continue;                         // :875 (the whole row is searched)
        // reindex_chain: alternative locations other than "" / "A" and unknown residue names are dropped (:846-847)
        const std::string alt = field(row, 16, 1);
        if (!(alt.empty() || alt == "A")) continue;
        const std::string rname = field(row, 17, 3);
        bool known = false;
        for (const char* r : kResidues) known = known || rname == r;
        if (!known) continue;
        const std::string rnum = field(row, 22, 5);
        if (rnum != prev_rnum) { prev_rnum = rnum; ++res_counter; }
        ++atom_counter;
        if (field(row, 12, 4) == "CA") ++n_ca;                                     // seq_chain counts CA atoms (:222)
        char num[32];
        std::string out = row.substr(0, 6);
        snprintf(num, sizeof num, "%5d", atom_counter);
        out += num;
        out += row.size() > 11 ? row.substr(11, 5) : std::string();
        out += " ";
        out += row.size() > 17 ? row.substr(17, 3) : std::string();
        out += "  ";
        snprintf(num, sizeof num, "%4d", res_counter);
        out += num;
        out += " ";
        if (row.size() > 27) out += row.substr(27);
        body += out + "\n";
    }
    if (n_ca < 1) return fail(C3D_ERR_IO, std::string(in_path) + " has less than 1 residue");   // seq_chain :231
    body += "\n";                                                                  // "END" written by :860, emptied by the sed of :818
    for (int i = 1; i < n_ca; ++i) {
        char row[32];
        snprintf(row, sizeof row, "CONECT%5d%5d\n", i, i + 1);
        body += row;
    }
    body += "END\n";
    // the sed of :818 removes the three letters wherever they stand
    if (log_path) {
        FILE* lf = fopen(log_path, "a");
        if (!lf) return fail(C3D_ERR_IO, std::string("cannot append to ") + log_path);
        fwrite(log.data(), 1, log.size(), lf);   // :879 appends an empty string
        fclose(lf);
    }
    FILE* f = fopen(out_path, "w");
    if (!f) return fail(C3D_ERR_IO, std::string("cannot write ") + out_path);
    fwrite(body.data(), 1, body.size(), f);
    fclose(f);
    return C3D_OK;
}

extern "C" int c3d_read_pdb_ca(const char* path, float** xyz, int* n_out) {
    if (!path || !xyz || !n_out) return fail(C3D_ERR_INVALID, "c3d_read_pdb_ca: null argument");
    std::string txt;
    if (!read_file(path, txt)) return fail(C3D_ERR_IO, std::string("cannot read ") + path);
    std::vector<float> v;
    size_t p = 0;
    while (p < txt.size()) {
        size_t e = txt.find('\n', p);
        if (e == std::string::npos) e = txt.size();
        const std::string line = txt.substr(p, e - p);
        p = e + 1;
        if (line.compare(0, 4, "ATOM") != 0 || line.size() < 54) continue;
        std::string an = line.substr(12, 4);
        an.erase(std::remove(an.begin(), an.end(), ' '), an.end());
        if (an != "CA") continue;
        v.push_back((float)atof(line.substr(30, 8).c_str()));
        v.push_back((float)atof(line.substr(38, 8).c_str()));
        v.push_back((float)atof(line.substr(46, 8).c_str()));
    }
    if (v.empty()) return fail(C3D_ERR_IO, std::string("no CA atoms in ") + path);
    *xyz = (float*)malloc(sizeof(float) * v.size());
    if (!*xyz) return fail(C3D_ERR_NOMEM, "out of memory");
    memcpy(*xyz, v.data(), sizeof(float) * v.size());
    *n_out = (int)(v.size() / 3);
    return C3D_OK;
}

// Value of sprintf("%.3f", d) read back as a number, without going through text: the exact
// product d*1000 is rounded to the nearest integer, ties to even (what glibc's printf does in
// the default rounding mode); the fma residual settles the cases where fl(d*1000) sits on a tie.
static inline long long round_milli(double d) {
    const double p = d * 1000.0;
    const double err = std::fma(d, 1000.0, -p);
    double r = std::nearbyint(p);
    const double diff = p - r;
    if (diff == 0.5 || diff == -0.5) {
        if (err > 0) r = std::floor(p) + 1.0;
        else if (err < 0) r = std::floor(p);
    }
    return (long long)r;
}
static inline double round_dec3(double d) { return (double)round_milli(d) / 1000.0; }

// sprintf("%.2f", v) without printf (|v| < 9e15): the magnitude's hundredths by the same exact-product rounding as round_milli, the sign as
// printf prints it (a negative value that rounds to zero keeps its '-').  ~10 ns against ~100.
static inline char* put_fixed2(char* o, double v) {
    if (std::signbit(v)) { *o++ = '-'; v = -v; }
    const double p = v * 100.0;
    const double err = std::fma(v, 100.0, -p);
    double r = std::nearbyint(p);
    const double diff = p - r;
    if (diff == 0.5 || diff == -0.5) {
        if (err > 0) r = std::floor(p) + 1.0;
        else if (err < 0) r = std::floor(p);
    }
    unsigned long long h = (unsigned long long)r, ip = h / 100;
    const unsigned frac = (unsigned)(h % 100);
    char tmp[24];
    int nd = 0;
    do { tmp[nd++] = (char)('0' + ip % 10); ip /= 10; } while (ip);
    while (nd) *o++ = tmp[--nd];
    *o++ = '.';
    *o++ = (char)('0' + frac / 10);
    *o++ = (char)('0' + frac % 10);
    return o;
}
// sprintf("%3d", v), v >= 0
static inline char* put_int3(char* o, int v) {
    char tmp[16];
    int nd = 0;
    do { tmp[nd++] = (char)('0' + v % 10); v /= 10; } while (v);
    for (int k = nd; k < 3; ++k) *o++ = ' ';
    while (nd) *o++ = tmp[--nd];
    return o;
}
static inline char* put_lit(char* o, const char* t, size_t len) { memcpy(o, t, len); return o + len; }

extern "C" int c3d_assess(const float* xyz, int n, int R, const int32_t* ri, const int32_t* rj, const int32_t* rt10,
                          double relax, int* satisfied, double* sum_dev) {
    if (!xyz || n < 1 || R < 0 || (R > 0 && (!ri || !rj || !rt10)) || !std::isfinite(relax)) return fail(C3D_ERR_INVALID, "c3d_assess: bad arguments");
    if (int rc = check_coords(xyz, (size_t)3 * n, "c3d_assess")) return rc;
    // the Perl reads coordinates back from the %8.3f PDB text
    std::vector<double> x((size_t)3 * n);
    for (size_t k = 0; k < x.size(); ++k) x[k] = round_dec3((double)xyz[k]);
    int count = 0;
    double sdev = 0;
    for (int k = 0; k < R; ++k) {
        const int i = ri[k] - 1, j = rj[k] - 1;
        if (i < 0 || j < 0 || i >= n || j >= n) return fail(C3D_ERR_INVALID, "c3d_assess: restraint index out of range");
        const double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
        const double d = round_dec3(sqrt(dx * dx + dy * dy + dz * dz));
        const double t = rt10[k] / 10.0;
        if (d < t + 0.0 + relax) ++count;
        if (d < t - 0.0 - relax) --count;
        if (d > t + 0.0 + 0.2) sdev += d - (t + 0.0);
        if (d < t - 0.0 - 0.2) sdev += (t - 0.0) - d;
    }
    if (satisfied) *satisfied = count;
    if (sum_dev) *sum_dev = sdev;
    return C3D_OK;
}

// count_satisfied_tbl_rows (chromosome3D.pl:447-485) with its violation table, and sum_noe_dev (:581-600), for one model: the numbers of
// c3d_assess plus, appended to `path`, the two '#' header lines and one row per restraint in the reference's row format
//   sprintf "%3s\t%.2f\t%.2f # assign45  resid %3d and name ca   resid %3d and name ca  %.2f 0.00 0.00", flag, deviation, distance, i, j, target
// violated rows first (the reference sorts its rows by the flag, descending; inside a flag group it leaves them in Perl's hash order —
// here: the order of the restraint rows).  20 models x 101 426 rows took the Perl driver 4 s, this loop 0.3 s with snprintf and 0.05 s without.
extern "C" int c3d_write_violations(const float* xyz, int n, int R, const int32_t* ri, const int32_t* rj, const int32_t* rt10, double relax,
                                    const char* pdb_label, const char* tbl_label, const char* path, int* satisfied, double* sum_dev) {
    if (!xyz || !path || n < 1 || R < 0 || (R > 0 && (!ri || !rj || !rt10)) || !std::isfinite(relax)) return fail(C3D_ERR_INVALID, "c3d_write_violations: bad arguments");
    if (int rc = check_coords(xyz, (size_t)3 * n, "c3d_write_violations")) return rc;
    for (int k = 0; k < R; ++k)      // the row buffer below is sized for targets a "%.1f" distance file can hold
        if (rt10[k] < -100000000 || rt10[k] > 100000000) return fail(C3D_ERR_INVALID, "c3d_write_violations: target out of range");
    std::vector<double> x((size_t)3 * n);
    for (size_t k = 0; k < x.size(); ++k) x[k] = round_dec3((double)xyz[k]);
    int count = 0;
    double sdev = 0;
    std::string viol, ok;
    viol.reserve((size_t)R * 96);
    char row[160];
    for (int k = 0; k < R; ++k) {
        const int i = ri[k] - 1, j = rj[k] - 1;
        if (i < 0 || j < 0 || i >= n || j >= n) return fail(C3D_ERR_INVALID, "c3d_write_violations: restraint index out of range");
        const double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
        const double d = round_dec3(sqrt(dx * dx + dy * dy + dz * dz));
        const double t = rt10[k] / 10.0;
        int flag = 1;
        double deviation = d - t;
        if (d < t + relax) { ++count; flag = 0; deviation = 0.0; }
        if (d < t - relax) { --count; flag = 1; deviation = -(t - d); }
        if (d > t + 0.2) sdev += d - t;
        if (d < t - 0.2) sdev += t - d;
        // "%3d\t%.2f\t%.2f # assign45  resid %3d and name ca   resid %3d and name ca  %.2f 0.00 0.00\n" (snprintf: 0.3 s for 2 M rows; this: 0.08 s)
        char* o = row;
        *o++ = ' '; *o++ = ' '; *o++ = (char)('0' + flag); *o++ = '\t';
        o = put_fixed2(o, deviation); *o++ = '\t';
        o = put_fixed2(o, d);
        o = put_lit(o, " # assign45  resid ", 19);
        o = put_int3(o, ri[k]);
        o = put_lit(o, " and name ca   resid ", 21);
        o = put_int3(o, rj[k]);
        o = put_lit(o, " and name ca  ", 14);
        o = put_fixed2(o, t);
        o = put_lit(o, " 0.00 0.00\n", 11);
        (flag ? viol : ok).append(row, (size_t)(o - row));
    }
    FILE* f = fopen(path, "a");
    if (!f) return fail(C3D_ERR_IO, std::string("c3d_write_violations: cannot open ") + path);
    fprintf(f, "#NOE violation check; %s against %s\n#violation-flag, deviation, actual-measurement, Input-NOE-restraint\n", pdb_label ? pdb_label : "model.pdb",
            tbl_label ? tbl_label : "contact.tbl");
    const bool good = fwrite(viol.data(), 1, viol.size(), f) == viol.size() && fwrite(ok.data(), 1, ok.size(), f) == ok.size();
    if (fclose(f) != 0 || !good) return fail(C3D_ERR_IO, std::string("c3d_write_violations: write failed: ") + path);
    if (satisfied) *satisfied = count;
    if (sum_dev) *sum_dev = sdev;
    return C3D_OK;
}

// average ranks (1-based, ties share the mean of their positions).  The values are sorted as (order-preserving 64-bit key, index) records by
// an LSD radix sort, 11 bits a pass, passes whose digit is the same for every value skipped — an indirect std::sort of 2 x 10^5 indices
// misses the cache at every comparison (18 -> 7 ms at N = 455 where both were timed).  The ranks do not depend on how ties are ordered.
// `copies` = 2 ranks the multiset in which every value appears twice (a symmetric matrix given by its upper half): the tie group at
// half-list positions k..e sits at 2k..2e+1 of the full list, so its rank is k + e + 1.5 — exact, and the same double the full sort gives.
static inline uint64_t order_key(double d) {
    uint64_t u;
    memcpy(&u, &d, 8);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

static void avg_ranks(const std::vector<double>& v, std::vector<double>& r, int copies = 1) {
    struct KV { uint64_t k; uint32_t i; };
    const size_t m = v.size();
    std::vector<KV> kv(m), tmp(m);
    for (size_t k = 0; k < m; ++k) kv[k] = KV{order_key(v[k]), (uint32_t)k};
    constexpr int B = 11, P = 6, R = 1 << B;
    std::vector<uint32_t> hist((size_t)P * R, 0);
    for (size_t k = 0; k < m; ++k)
        for (int p = 0; p < P; ++p) ++hist[(size_t)p * R + ((kv[k].k >> (p * B)) & (R - 1))];
    KV *src = kv.data(), *dst = tmp.data();
    for (int p = 0; p < P; ++p) {
        uint32_t* h = &hist[(size_t)p * R];
        bool one_digit = false;
        for (int d = 0; d < R && !one_digit; ++d) one_digit = h[d] == m;
        if (one_digit) continue;
        uint32_t at = 0;
        for (int d = 0; d < R; ++d) { const uint32_t c = h[d]; h[d] = at; at += c; }
        for (size_t k = 0; k < m; ++k) { const KV e = src[k]; dst[h[(e.k >> (p * B)) & (R - 1)]++] = e; }
        std::swap(src, dst);
    }
    r.resize(m);
    size_t k = 0;
    while (k < m) {
        size_t e = k;
        const double vk = v[src[k].i];
        while (e + 1 < m && v[src[e + 1].i] == vk) ++e;      // -0.0 and +0.0 have adjacent keys and compare equal here
        const double rank = copies == 2 ? ((double)k + (double)e) + 1.5 : 0.5 * ((double)k + (double)e) + 1.0;
        for (size_t q = k; q <= e; ++q) r[src[q].i] = rank;
        k = e + 1;
    }
}

// ranks of the ordered pairs |i-j| >= range of IF (spearman_IF_pdb.pl:30-44 ranks both (i,j) and (j,i)).  A symmetric matrix — every matrix
// the pipeline makes — is ranked from its upper half; the sums run over the ordered pairs in the reference's order either way.
void c3d::if_pair_ranks(const double* IF, int n, int range, std::vector<double>& rank_matrix, size_t& m, double& mean_rank, double& saa) {
    std::vector<double> a, ra;
    std::vector<size_t> pos;
    bool symmetric = true;
    for (int i = 0; i < n && symmetric; ++i)
        for (int j = i + (range > 0 ? range : 0); j < n; ++j)
            if (IF[(size_t)i * n + j] != IF[(size_t)j * n + i]) { symmetric = false; break; }
    if (range <= 0) symmetric = false;                         // the diagonal would be counted once, not twice
    for (int i = 0; i < n; ++i)
        for (int j = symmetric ? i + range : 0; j < n; ++j) {
            if (std::abs(i - j) < range) continue;
            a.push_back(IF[(size_t)i * n + j]);
            pos.push_back((size_t)i * n + j);
        }
    rank_matrix.assign((size_t)n * n, 0.0);
    mean_rank = 0; saa = 0; m = 0;
    if (a.empty()) return;
    avg_ranks(a, ra, symmetric ? 2 : 1);
    for (size_t k = 0; k < a.size(); ++k) {
        rank_matrix[pos[k]] = ra[k];
        if (symmetric) rank_matrix[(pos[k] % n) * n + pos[k] / n] = ra[k];
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            if (std::abs(i - j) >= range) { mean_rank += rank_matrix[(size_t)i * n + j]; ++m; }
    mean_rank /= (double)m;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            if (std::abs(i - j) >= range) { const double d = rank_matrix[(size_t)i * n + j] - mean_rank; saa += d * d; }
}

// Spearman for M models of one matrix: the IF ranks are computed once; distances are integers in
// thousandths of an Angstrom (the "%.3f" rounding of spearman_IF_pdb.pl:47), so their average
// ranks come from a counting pass instead of a sort.
extern "C" int c3d_spearman_if_dist_batch(const double* IF, const float* xyz, int n, int n_models, int range, double* rho) {
    if (!IF || !xyz || !rho || n < 2 || n_models < 1) return fail(C3D_ERR_INVALID, "c3d_spearman_if_dist_batch: bad arguments");
    std::vector<double> rank_matrix, ra;
    size_t m = 0;
    double ma = 0, saa = 0;
    for (size_t k = 0; k < (size_t)n * n; ++k)
        if (std::isnan(IF[k])) return fail(C3D_ERR_INVALID, "c3d_spearman_if_dist_batch: the matrix holds a NaN");
    c3d::if_pair_ranks(IF, n, range, rank_matrix, m, ma, saa);
    if (m < 2) return fail(C3D_ERR_INVALID, "c3d_spearman_if_dist_batch: range leaves no pairs");
    std::vector<uint32_t> pi, pj;
    pi.reserve(m); pj.reserve(m); ra.reserve(m);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            if (std::abs(i - j) < range) continue;
            pi.push_back((uint32_t)i); pj.push_back((uint32_t)j);
            ra.push_back(rank_matrix[(size_t)i * n + j]);
        }
    std::vector<double> x((size_t)3 * n), rb(m);
    std::vector<long long> dq(m);
    if (int rc = check_coords(xyz, (size_t)3 * n * n_models, "c3d_spearman_if_dist_batch")) return rc;
    for (int mdl = 0; mdl < n_models; ++mdl) {
        const float* xm = xyz + (size_t)mdl * n * 3;
        for (size_t k = 0; k < x.size(); ++k) x[k] = round_dec3((double)xm[k]);
        long long dmax = 0;
        for (size_t k = 0; k < m; ++k) {
            const size_t i = pi[k], j = pj[k];
            const double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
            dq[k] = round_milli(sqrt(dx * dx + dy * dy + dz * dz));
            if (dq[k] > dmax) dmax = dq[k];
        }
        if (dmax > 50000000LL) return fail(C3D_ERR_INVALID, "c3d_spearman_if_dist_batch: coordinates out of range");
        std::vector<uint32_t> cnt((size_t)dmax + 2, 0);
        for (size_t k = 0; k < m; ++k) ++cnt[(size_t)dq[k]];
        std::vector<double> rank_of((size_t)dmax + 1);
        size_t below = 0;
        for (size_t v = 0; v <= (size_t)dmax; ++v) {
            if (cnt[v]) rank_of[v] = 0.5 * ((double)below + (double)(below + cnt[v] - 1)) + 1.0;
            below += cnt[v];
        }
        double mb = 0;
        for (size_t k = 0; k < m; ++k) { rb[k] = rank_of[(size_t)dq[k]]; mb += rb[k]; }
        mb /= m;
        double sab = 0, sbb = 0;
        for (size_t k = 0; k < m; ++k) { sab += (ra[k] - ma) * (rb[k] - mb); sbb += (rb[k] - mb) * (rb[k] - mb); }
        rho[mdl] = sab / sqrt(saa * sbb);
    }
    return C3D_OK;
}

extern "C" int c3d_spearman_if_dist(const double* IF, const float* xyz, int n, int range, double* rho) {
    return c3d_spearman_if_dist_batch(IF, xyz, n, 1, range, rho);
}

// ---- cross-resolution similarity (output_models/similarity.txt of the reference: data only, no code) ----
// The bundled "<ID>_reduced.pdb" of a 500 kb model is the mean of consecutive bead pairs (an odd last bead
// is kept); its similarity to the 1 Mb model of the same chromosome is reported as the Spearman
// correlation of the i<j distances and as an "RMSD" of those distances after scaling the first model's
// by mean(d_b)/mean(d_a).  Both definitions were recovered from the bundled files and reproduce every
// number of similarity.txt to 1e-12.
extern "C" int c3d_reduce_model(const double* xyz, int n, double* out) {
    if (!xyz || !out || n < 1) return fail(C3D_ERR_INVALID, "c3d_reduce_model: bad arguments");
    if (int rc = check_coords_d(xyz, (size_t)3 * n, "c3d_reduce_model")) return rc;
    const int m = (n + 1) / 2;
    for (int k = 0; k < m; ++k)
        for (int c = 0; c < 3; ++c) {
            const double a = xyz[(size_t)(2 * k) * 3 + c];
            out[(size_t)k * 3 + c] = 2 * k + 1 < n ? 0.5 * (a + xyz[(size_t)(2 * k + 1) * 3 + c]) : a;
        }
    return C3D_OK;
}

extern "C" int c3d_model_similarity(const double* a, const double* b, int n, double* spearman, double* rmsd) {
    if (!a || !b || n < 3) return fail(C3D_ERR_INVALID, "c3d_model_similarity: bad arguments");
    if (int rc = check_coords_d(a, (size_t)3 * n, "c3d_model_similarity")) return rc;
    if (int rc = check_coords_d(b, (size_t)3 * n, "c3d_model_similarity")) return rc;
    const size_t m = (size_t)n * (n - 1) / 2;
    std::vector<double> da(m), db(m), ra, rb;
    size_t k = 0;
    double sa = 0, sb = 0;
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j, ++k) {
            double qa = 0, qb = 0;
            for (int c = 0; c < 3; ++c) {
                const double ua = a[(size_t)i * 3 + c] - a[(size_t)j * 3 + c], ub = b[(size_t)i * 3 + c] - b[(size_t)j * 3 + c];
                qa += ua * ua; qb += ub * ub;
            }
            da[k] = sqrt(qa); db[k] = sqrt(qb);
            sa += da[k]; sb += db[k];
        }
    if (spearman) {
        avg_ranks(da, ra);
        avg_ranks(db, rb);
        const double mean = 0.5 * ((double)m + 1.0);
        double sab = 0, saa = 0, sbb = 0;
        for (size_t q = 0; q < m; ++q) {
            const double x = ra[q] - mean, y = rb[q] - mean;
            sab += x * y; saa += x * x; sbb += y * y;
        }
        *spearman = sab / sqrt(saa * sbb);
    }
    if (rmsd) {
        const double scale = sa > 0 ? sb / sa : 1.0;     // ratio of the mean distances
        double acc = 0;
        for (size_t q = 0; q < m; ++q) { const double e = scale * da[q] - db[q]; acc += e * e; }
        *rmsd = sqrt(acc / (double)m);
    }
    return C3D_OK;
}
