// c3d_solve — command-line stand-in for the `cns_solve < dgsa.inp` line of the reference's
// job.sh (chromosome3D.pl:258-284).  It is a thin shell over the C ABI (include/c3d.h):
//
//   c3d_solve --if <matrix.txt> --out <dir> [--id ID] [-k 11] [-a 0.5] [-m 20] ...
//       front half on the GPU (K1) + <ID>.dist/.rr/contact.tbl + M models <ID>_<k>.pdb
//   c3d_solve --tbl contact.tbl --n N --out <dir> --id ID [-m 20] ...
//       exactly the cns_solve role: restraints in, <ID>_<k>.pdb out
//
// Success convention of the reference: <ID>_<M>.pdb exists, iam.running removed; on failure
// iam.running is renamed iam.failed and the exit code is non-zero (:266-283).
#include <sys/stat.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/c3d.h"

static void usage() {
    fprintf(stderr,
            "usage: c3d_solve (--if <IF matrix> | --tbl <contact.tbl> --n <beads>) --out <dir> [--id <ID>]\n"
            "                 [-k <K=11>] [-a <alpha=0.5>] [-m <models=20>] [--seed <82364>] [--first-replica <0>]\n"
            "                 [--device <0>] [--min-steps <3000>] [--gtol <1e-2>] [--final-minimiser <1>] [--embed] [--no-graph] [--quiet]\n"
            "                 [--seq <one-letter residue codes | @fasta file>   residue names of the models (default: all MET)]\n"
            "                 [--accepted   also write <ID>a_<k>.pdb beside every <ID>_<k>.pdb, as CNS does for structures it accepts]\n");
}

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc__ = (call);                                                       \
        if (rc__ != C3D_OK) {                                                    \
            fprintf(stderr, "c3d_solve: %s failed (%d): %s\n", #call, rc__, c3d_last_error()); \
            return fail_exit(out_dir);                                           \
        }                                                                        \
    } while (0)

static int fail_exit(const std::string& out_dir) {
    if (!out_dir.empty()) {
        const std::string a = out_dir + "/iam.running", b = out_dir + "/iam.failed";
        if (rename(a.c_str(), b.c_str()) != 0) { FILE* f = fopen(b.c_str(), "w"); if (f) fclose(f); }
    }
    fprintf(stderr, "ERROR! Final structures not found!\nC3D FAILED!\n");
    return 1;
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    // A device exception (memory fault, queue error) must reach stderr in the runtime's own words: by default ROCr first pipes a GPU core
    // dump to the helper named in /proc/sys/kernel/core_pattern, and where that helper does not exist the process dies on the broken pipe
    // (rc -13, "GPU coredump: execvp failed") before the runtime has said WHICH exception — round 5 lost the only evidence of one that way.
    setenv("HSA_DISABLE_COREDUMP_ON_EXCEPTION", "1", 0);          // (0: a user's own setting wins)
    const double t_start = now_s();
    std::string if_path, tbl_path, out_dir, id, seq_arg;
    double K = 11, alpha = 0.5, gtol = 1e-2;
    int final_min = 1;
    int models = 20, device = 0, n_beads = 0, min_steps = 3000, use_graph = 1, quiet = 0, embed = 0, accepted = 0;
    unsigned long long seed = 82364ULL;
    unsigned first_rep = 0;
    for (int a = 1; a < argc; ++a) {
        const std::string s = argv[a];
        auto next = [&](const char* what) -> const char* {
            if (a + 1 >= argc) { fprintf(stderr, "c3d_solve: %s needs a value\n", what); exit(2); }
            return argv[++a];
        };
        if (s == "--if" || s == "-i" || s == "-if") if_path = next("--if");
        else if (s == "--tbl") tbl_path = next("--tbl");
        else if (s == "--n") n_beads = atoi(next("--n"));
        else if (s == "--out" || s == "-o") out_dir = next("--out");
        else if (s == "--id") id = next("--id");
        else if (s == "-k") K = atof(next("-k"));
        else if (s == "-a") alpha = atof(next("-a"));
        else if (s == "-m") models = atoi(next("-m"));
        else if (s == "--seed") seed = strtoull(next("--seed"), nullptr, 10);
        else if (s == "--first-replica") first_rep = (unsigned)strtoul(next("--first-replica"), nullptr, 10);
        else if (s == "--device") device = atoi(next("--device"));
        else if (s == "--min-steps") min_steps = atoi(next("--min-steps"));
        else if (s == "--gtol") gtol = atof(next("--gtol"));
        else if (s == "--final-minimiser") final_min = atoi(next("--final-minimiser"));   // 0 = FIRE throughout (rounds 1-4), 1 = two-point steps then FIRE (default)
        else if (s == "--embed") embed = 1;   // distance-geometry start (deck :1471-1525) instead of the random coil
        else if (s == "--no-graph") use_graph = 0;
        else if (s == "--quiet") quiet = 1;
        else if (s == "--accepted") accepted = 1;   // the deck's printaccept writes <ID>a_<k>.pdb for structures CNS accepts, beside the trial file (:1818-1828)
        else if (s == "--seq") seq_arg = next("--seq");
        else if (s == "-h" || s == "--help") { usage(); return 0; }
        else { fprintf(stderr, "c3d_solve: unknown option %s\n", s.c_str()); usage(); return 2; }
    }
    if (out_dir.empty() || (if_path.empty() == tbl_path.empty()) || models < 1) { usage(); return 2; }
    if (id.empty()) {
        std::string base = if_path.empty() ? std::string("model") : if_path.substr(if_path.find_last_of('/') + 1);
        if (base.size() > 4 && base.substr(base.size() - 4) == ".txt") base.resize(base.size() - 4);
        id = base;
    }
    if (!seq_arg.empty()) {           // residue names: letters, or @file in FASTA form (header lines skipped)
        std::string letters = seq_arg;
        if (seq_arg[0] == '@') {
            FILE* f = fopen(seq_arg.c_str() + 1, "r");
            if (!f) { fprintf(stderr, "c3d_solve: cannot read %s\n", seq_arg.c_str() + 1); return 2; }
            letters.clear();
            char line[4096];
            while (fgets(line, sizeof line, f)) if (line[0] != '>') letters += line;
            fclose(f);
        }
        c3d_set_residue_sequence(letters.c_str());
    }
    mkdir(out_dir.c_str(), 0755);   // like the reference (:45): create the output directory if missing
    { FILE* f = fopen((out_dir + "/iam.running").c_str(), "w"); if (!f) { fprintf(stderr, "c3d_solve: cannot write into %s\n", out_dir.c_str()); return 1; } fclose(f); }
    remove((out_dir + "/iam.failed").c_str());

    c3d_ctx* ctx = nullptr;
    CHECK(c3d_create(device, &ctx));
    const double t_ctx = now_s();
    c3d_model model;
    c3d_default_model(&model);
    CHECK(c3d_set_model(ctx, &model));

    int n = 0, R = 0;
    std::vector<int32_t> ri, rj, rt;
    double* IF = nullptr;
    if (!if_path.empty()) {
        CHECK(c3d_parse_if_file(if_path.c_str(), &IF, &n));
        CHECK(c3d_set_if_matrix(ctx, IF, n, alpha, K));
        std::vector<int32_t> d10((size_t)n * n);
        CHECK(c3d_get_dist10(ctx, d10.data()));
        CHECK(c3d_write_front_half(d10.data(), n, model.min_sep, (out_dir + "/" + id + ".dist").c_str(),
                                   (out_dir + "/" + id + ".rr").c_str(), (out_dir + "/contact.tbl").c_str(), &R));
        if (!quiet) printf("L          : %d\nRestraints : %d lines in tbl file\n", n, R);
    } else {
        int32_t *pi = nullptr, *pj = nullptr, *pt = nullptr;
        CHECK(c3d_read_tbl(tbl_path.c_str(), &pi, &pj, &pt, &R));
        ri.assign(pi, pi + R); rj.assign(pj, pj + R); rt.assign(pt, pt + R);
        c3d_free(pi); c3d_free(pj); c3d_free(pt);
        n = n_beads;
        for (int k = 0; k < R; ++k) { if (ri[k] > n) n = ri[k]; if (rj[k] > n) n = rj[k]; }
        CHECK(c3d_set_restraints(ctx, n, R, ri.data(), rj.data(), rt.data()));
    }

    const double t_front = now_s();
    std::vector<c3d_stage> stages(c3d_default_schedule(nullptr, 0, min_steps));
    c3d_default_schedule(stages.data(), (int)stages.size(), min_steps);
    c3d_fire_params fire;
    c3d_default_fire(&fire);
    CHECK(c3d_set_option(ctx, "final_minimiser", final_min));
    CHECK(c3d_set_schedule(ctx, stages.data(), (int)stages.size(), &fire, (float)gtol, 250));
    CHECK(c3d_set_option(ctx, "use_graph", use_graph));
    CHECK(c3d_init_replicas(ctx, models, seed, first_rep));
    if (embed) CHECK(c3d_embed_replicas(ctx, 50));
    CHECK(c3d_run(ctx));
    const double t_run = now_s();

    std::vector<float> xyz((size_t)models * n * 3);
    std::vector<double> en((size_t)models * 3);
    CHECK(c3d_get_coords(ctx, xyz.data()));
    CHECK(c3d_get_energies(ctx, en.data()));
    for (int r = 0; r < models; ++r) {
        char name[64];
        snprintf(name, sizeof name, "%s_%u.pdb", id.c_str(), first_rep + (unsigned)r + 1u);
        CHECK(c3d_write_pdb((out_dir + "/" + name).c_str(), xyz.data() + (size_t)r * n * 3, n, en[3 * r], en[3 * r + 1],
                            en[3 * r + 2], name));
        if (accepted) {
            // the accepted twin: same coordinates, same REMARK rows (the reference's assess_dgsa then drops the trial file, :791-795).  CNS's
            // acceptance thresholds are defined on covalent geometry a bead model does not have: with --accepted every model is "accepted"
            snprintf(name, sizeof name, "%sa_%u.pdb", id.c_str(), first_rep + (unsigned)r + 1u);
            CHECK(c3d_write_pdb((out_dir + "/" + name).c_str(), xyz.data() + (size_t)r * n * 3, n, en[3 * r], en[3 * r + 1],
                                en[3 * r + 2], name));
        }
    }
    double ms = 0;
    long steps = 0, launches = 0;
    c3d_last_timing(ctx, &ms, &steps, &launches);
    if (!quiet) {
        printf(accepted ? "trial and accepted structures written.\n" : "trial structures written.\n");
        printf("c3d_solve: %d beads, %d restraints, %d models, %ld SA steps/model in %.1f ms on device %d (%.3g replica-steps/s)\n",
               n, R, models, steps, ms, device, ms > 0 ? 1e3 * (double)steps * models / ms : 0.0);
        if (IF) {
            std::vector<double> rho(models);
            if (c3d_score_replicas(ctx, IF, 3, nullptr, nullptr, rho.data()) == C3D_OK)      // on the device, from the resident coordinates (K6)
                for (int r = 0; r < models; ++r)
                    printf("  model %2u  E_noe %14.2f  Spearman(IF,d) %.4f\n", first_rep + r + 1, en[3 * r], rho[r]);
        }
    }
    if (!quiet)
        printf("c3d_solve wall: device init %.3f s, parse + K1 + front-half files %.3f s, anneal (incl. graph build) %.3f s, "
               "read-back + PDB + scoring %.3f s, total %.3f s\n", t_ctx - t_start, t_front - t_ctx, t_run - t_front, now_s() - t_run,
               now_s() - t_start);
    if (IF) c3d_free(IF);
    c3d_destroy(ctx);
    remove((out_dir + "/iam.running").c_str());
    return 0;
}
