// c3d_batch — every chromosome of a directory in ONE process, one host thread and one libc3d context per GPU.
//
// The reference's only concurrency is test.sh:4-12: one `chromosome3D.pl -if <matrix> -o <dir> &` per matrix, each a
// Perl process that shells out to CNS.  Here the whole per-matrix flow of chromosome3D.pl:86-106 runs natively behind
// the C ABI (include/c3d.h) for every matrix:
//   calc_len_IF / IF2dist_new / dist2rr / carr2tbl (:87-89)   c3d_parse_if_file, c3d_set_if_matrix (K1), c3d_write_front_half
//   build_models (:104)                                       c3d_init_replicas, c3d_run
//   assess_dgsa (:106, :769-829)                              c3d_rank, c3d_assess, c3d_write_pdb, c3d_shape_pdb, top 5 renamed
// Matrices go to GPUs by longest-processing-time-first on N^2 (the restraint count), largest first on every GPU.
//
//   c3d_batch <dir with *_matrix.txt | matrix files...> --out <root> [--devices <all>] [--lanes 3] [--order warm|lpt] [-m 20] [-k 11] [-a 0.5]
//             [--seed 82364] [--min-steps 3000] [--gtol 1e-2] [--pattern _500kb_]
//             [--pair 1]             1 (default): two small chromosomes anneal side by side on disjoint halves of a device's XCDs; 0: one at a time
//             [--map-devices-to P]   rehearsal hook: every logical device 0 .. N-1 of --devices N is physical device P (the per-device
//                                    LPT lists, threads and contexts of an N-GPU node on a box with fewer GPUs; anneals are serialised
//                                    per PHYSICAL device, so the models are what --devices 1 gives; its timings mean nothing)
// Output: <root>/<chromosome>/ with the files a reference run leaves (<ID>.dist, .rr, contact.tbl, model_info.log,
// <ID>_model1..5.pdb, <ID>_<k>.pdb) and <root>/<chromosome>.log with the satisfaction table; one summary line per matrix.
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/c3d.h"

namespace {

struct Job {
    std::string path, id, chrom;
    double cost = 0;
    int device = -1, half = -1;    // half: which four XCDs it annealed on (-1: the whole device)
    bool ok = false;
    std::string summary;
};
struct Options {
    std::string out;
    double K = 11, alpha = 0.5, gtol = 1e-2;
    int final_min = 1;          // --final-minimiser 0: FIRE throughout (rounds 1-4)
    int models = 20, min_steps = 3000;
    unsigned long long seed = 82364ULL;
    bool violations = false;       // --violations: also leave contact_violation.txt (:475-483; 180 MB of text per chromosome at N = 455)
};
std::mutex g_print;
bool g_timeline = false;           // env C3D_BATCH_TIMELINE
double g_t0 = 0;
bool g_warm = true;                // --order lpt: a GPU's lanes start on its largest jobs (round 4); warm (round 5): see main
bool g_pair = true;                // --pair 0: every anneal has the whole device to itself (round 4)

// Who anneals where on one physical device.  The multi-step kernel places replica r on XCD base + r % count (c3d.h, option
// cluster_xcd_count / cluster_xcd_base): a chromosome small enough for a four-XCD geometry takes ONE half of the device (XCDs 0-3 or
// 4-7) and a second small chromosome anneals beside it on the other half — no CU is shared, the models are the same bits
// (profiles/r05_config4_paired_anneals.txt: 1.4-1.8 x for the pairs of config 4 with N <= 287); a large one takes both halves.
struct XcdBroker {
    std::mutex mu;
    std::condition_variable cv;
    bool busy[2] = {false, false};
    int waiting_full = 0;
    // returns the half taken (0 / 1), or -1 = both
    int acquire(bool half) {
        std::unique_lock<std::mutex> lk(mu);
        if (!half) {
            ++waiting_full;
            cv.wait(lk, [&] { return !busy[0] && !busy[1]; });
            --waiting_full;
            busy[0] = busy[1] = true;
            return -1;
        }
        cv.wait(lk, [&] { return waiting_full == 0 && (!busy[0] || !busy[1]); });      // a waiting whole-device job goes first
        const int h = busy[0] ? 1 : 0;
        busy[h] = true;
        return h;
    }
    void release(int h) {
        { std::lock_guard<std::mutex> lk(mu); if (h < 0) busy[0] = busy[1] = false; else busy[h] = false; }
        cv.notify_all();
    }
};
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

#define TRY(call)                                                                                  \
    do {                                                                                           \
        if ((call) != C3D_OK) { job.summary = std::string(#call) + ": " + c3d_last_error(); return false; } \
    } while (0)

bool solve_one(c3d_ctx* ctx, const Options& o, Job& job, XcdBroker& gpu) {
    const double t0 = now_s();
    const std::string dir = o.out + "/" + job.chrom;
    mkdir(dir.c_str(), 0755);
    c3d_model model;
    c3d_default_model(&model);
    TRY(c3d_set_model(ctx, &model));
    double* IF = nullptr;
    int n = 0, R = 0;
    TRY(c3d_parse_if_file(job.path.c_str(), &IF, &n));
    std::vector<double> if_copy(IF, IF + (size_t)n * n);
    c3d_free(IF);
    TRY(c3d_set_if_matrix(ctx, if_copy.data(), n, o.alpha, o.K));
    const double t_k1 = now_s();
    std::vector<int32_t> d10((size_t)n * n);
    TRY(c3d_get_dist10(ctx, d10.data()));
    const std::string tbl = dir + "/contact.tbl";
    TRY(c3d_write_front_half(d10.data(), n, model.min_sep, (dir + "/" + job.id + ".dist").c_str(), (dir + "/" + job.id + ".rr").c_str(),
                             tbl.c_str(), &R));
    {   // <ID>.fasta (:92-98): one residue per bead, every bead MET as in the bundled output_models
        FILE* fa = fopen((dir + "/" + job.id + ".fasta").c_str(), "w");
        if (!fa) { job.summary = "cannot write " + job.id + ".fasta"; return false; }
        fprintf(fa, ">%s\n%s\n", job.id.c_str(), std::string((size_t)n, 'M').c_str());
        if (fclose(fa) != 0) { job.summary = "cannot write " + job.id + ".fasta"; return false; }
    }
    std::vector<c3d_stage> stages(c3d_default_schedule(nullptr, 0, o.min_steps));
    c3d_default_schedule(stages.data(), (int)stages.size(), o.min_steps);
    c3d_fire_params fire;
    c3d_default_fire(&fire);
    TRY(c3d_set_option(ctx, "final_minimiser", o.final_min));
    TRY(c3d_set_schedule(ctx, stages.data(), (int)stages.size(), &fire, (float)o.gtol, 250));
    // a chromosome for which a four-XCD geometry exists anneals on half of the device (chosen when its turn comes)
    bool half = false;
    if (g_pair && n <= 270) {      // (a four-XCD geometry exists up to 288 beads; above ~270 it costs more than pairing returns: profiles/r05_config4_paired_anneals.txt)
        TRY(c3d_set_option(ctx, "cluster_xcd_base", 0));
        TRY(c3d_set_option(ctx, "cluster_xcd_count", 4));
        TRY(c3d_init_replicas(ctx, o.models, o.seed, 0));
        double ok = 0;
        c3d_get_stat(ctx, "cluster_ok", &ok);
        half = ok != 0;
    }
    if (!half) {
        TRY(c3d_set_option(ctx, "cluster_xcd_base", 0));
        TRY(c3d_set_option(ctx, "cluster_xcd_count", 8));
        TRY(c3d_init_replicas(ctx, o.models, o.seed, 0));
    }
    const double t_front = now_s();
    double t_anneal0 = t_front;
    {   // one anneal at a time per GPU: the multi-step kernel wants every CU; the host phases of the other lanes run meanwhile.
        // Their short device phases (K1: two kernels of ~20 us, coordinate copies) are NOT serialised: a cluster launch whose
        // workgroups find a CU busy with one of them becomes resident as soon as it drains, microseconds later, far inside
        // the 0.3 s after which a launch gives up (c3d_cluster.hip); the fallback counter stays 0 in profiles/r02_config4_*.
        struct Hold {                               // released on every exit path (TRY returns early)
            XcdBroker& b; int h;
            ~Hold() { b.release(h); }
        } hold{gpu, gpu.acquire(half)};
        if (half) TRY(c3d_set_option(ctx, "cluster_xcd_base", 4 * hold.h));
        t_anneal0 = now_s();                        // (the wait for the GPU, when another lane anneals, is not this job's anneal)
        TRY(c3d_run(ctx));
        job.half = half ? hold.h : -1;
    }
    const double t_run = now_s();
    double ms = 0;
    long steps = 0, launches = 0;
    c3d_last_timing(ctx, &ms, &steps, &launches);
    const int M = o.models;
    std::vector<float> xyz((size_t)M * n * 3);
    std::vector<double> en((size_t)M * 3), rho(M);
    std::vector<int32_t> rank(M);
    TRY(c3d_get_coords(ctx, xyz.data()));
    TRY(c3d_get_energies(ctx, en.data()));
    TRY(c3d_rank(ctx, rank.data()));
    // Spearman(IF, d) of every model and the satisfaction table's two numbers (:447-485, :581-600) on the device, from the coordinates
    // resident there (K6, c3d_score_replicas: equal to the host functions c3d_spearman_if_dist_batch / c3d_assess — integers exact, sums
    // to rounding, a -m gpu test — which took 33 + ~15 ms per chromosome at N = 455, and the 7 MB contact.tbl need not be read back)
    std::vector<int32_t> sats(M);
    std::vector<double> devs(M);
    TRY(c3d_score_replicas(ctx, if_copy.data(), 3, sats.data(), devs.data(), rho.data()));
    const double t_score = now_s();
    const int Rt = R;                                // rows of contact.tbl = the restraints the context holds
    struct FileCloser {                              // closes the log on every exit path (the TRY macros return early)
        FILE* f;
        ~FileCloser() { if (f) fclose(f); }
    } lgc{fopen((o.out + "/" + job.chrom + ".log").c_str(), "w")};
    FILE* const lg = lgc.f;
    if (!lg) { job.summary = "cannot write the log"; return false; }
    fprintf(lg, "L          : %d\nRestraints : %d lines in tbl file\n\nNOE_SATISFIED(+-0.5A)  SUM_OF_DEVIATIONS>= 0.2  PDB\n", n, R);
    std::vector<std::string> names(M);
    std::vector<int32_t> vri, vrj, vrt;
    remove((dir + "/model_info.log").c_str());
    for (int k = M - 1; k >= 0; --k) {               // the reference lists (and shapes) from the highest energy down (:805, :813)
        const int r = rank[k];
        char name[96];
        snprintf(name, sizeof name, "%s_%d.pdb", job.id.c_str(), r + 1);
        names[r] = dir + "/" + name;
        TRY(c3d_write_pdb(names[r].c_str(), xyz.data() + (size_t)r * n * 3, n, en[3 * r], en[3 * r + 1], en[3 * r + 2], name));
        int sat = sats[r];
        double dev = devs[r];
        if (o.violations) {                          // the reference's violation table (and, with it, the host's numbers: the same)
            if (vri.empty()) {
                int32_t *pi = nullptr, *pj = nullptr, *pt = nullptr;
                int Rr = 0;
                TRY(c3d_read_tbl(tbl.c_str(), &pi, &pj, &pt, &Rr));
                vri.assign(pi, pi + Rr); vrj.assign(pj, pj + Rr); vrt.assign(pt, pt + Rr);
                c3d_free(pi); c3d_free(pj); c3d_free(pt);
                remove((dir + "/contact_violation.txt").c_str());
            }
            TRY(c3d_write_violations(xyz.data() + (size_t)r * n * 3, n, (int)vri.size(), vri.data(), vrj.data(), vrt.data(), 0.5, ("./" + std::string(name)).c_str(),
                                     "contact.tbl", (dir + "/contact_violation.txt").c_str(), &sat, &dev));
        }
        char cnt[48], sd[48];
        snprintf(cnt, sizeof cnt, "%d/%d", sat, Rt);
        snprintf(sd, sizeof sd, "%.2f", dev);
        std::string base(name);
        base.resize(base.size() - 4);
        fprintf(lg, "%-9s             %-9s                %-25s\n", cnt, sd, base.c_str());
    }
    for (int k = M - 1; k >= 0; --k) TRY(c3d_shape_pdb(names[rank[k]].c_str(), names[rank[k]].c_str(), (dir + "/model_info.log").c_str()));
    fprintf(lg, "\n");
    for (int k = 0; k < M && k < 5; ++k) {            // :822-828
        char dst[96];
        snprintf(dst, sizeof dst, "%s_model%d.pdb", job.id.c_str(), k + 1);
        fprintf(lg, "model%d.pdb <= %s\n", k + 1, names[rank[k]].c_str());
        if (rename(names[rank[k]].c_str(), (dir + "/" + dst).c_str()) != 0) { job.summary = "rename failed"; return false; }
    }
    lgc.f = nullptr;
    if (fclose(lg) != 0) { job.summary = "cannot write the log"; return false; }
    char buf[512];
    const double t_end = now_s();
    snprintf(buf, sizeof buf, "%-14s N=%4d R=%6d  %2d models  best: replica %2d  E_noe %12.1f  Spearman(IF,1/d) %.4f  anneal %6.1f ms (%ld steps)  "
                              "end-to-end %.2f s  GPU %d%s  [phases: parse+K1 %.3f, front-half files+start structures %.3f, anneal %.3f, read-back+rank+Spearman %.3f, "
                              "PDB+assessment+shaping %.3f s]",
             job.chrom.c_str(), n, R, M, rank[0] + 1, en[3 * rank[0]], -rho[rank[0]], ms, steps, t_end - t0, job.device, job.half < 0 ? "" : (job.half ? " XCDs 4-7" : " XCDs 0-3"), t_k1 - t0, t_front - t_k1,
             t_run - t_anneal0, t_score - t_run, t_end - t_score);
    job.summary = buf;
    if (g_timeline) {                                 // C3D_BATCH_TIMELINE=1: when each phase of each job ran, seconds since the executor's start
        std::lock_guard<std::mutex> lk(g_print);
        fprintf(stderr, "timeline %-12s N=%4d start %.4f parsed+K1 %.4f front %.4f got-GPU %.4f annealed %.4f scored %.4f end %.4f%s\n", job.chrom.c_str(), n,
                t0 - g_t0, t_k1 - g_t0, t_front - g_t0, t_anneal0 - g_t0, t_run - g_t0, t_score - g_t0, t_end - g_t0, job.half < 0 ? "" : (job.half ? " half 1" : " half 0"));
    }
    return true;
}

void collect(const std::string& arg, const std::string& pattern, std::vector<Job>& jobs) {
    struct stat st;
    if (stat(arg.c_str(), &st) != 0) { fprintf(stderr, "c3d_batch: %s not found\n", arg.c_str()); exit(2); }
    std::vector<std::string> files;
    if (S_ISDIR(st.st_mode)) {
        DIR* d = opendir(arg.c_str());
        while (dirent* e = d ? readdir(d) : nullptr) {
            const std::string f = e->d_name;
            if (f.size() > 11 && f.compare(f.size() - 11, 11, "_matrix.txt") == 0) files.push_back(arg + "/" + f);
        }
        if (d) closedir(d);
    } else {
        files.push_back(arg);
    }
    std::sort(files.begin(), files.end());
    for (const std::string& f : files) {
        if (!pattern.empty() && f.find(pattern) == std::string::npos) continue;
        Job j;
        j.path = f;
        std::string base = f.substr(f.find_last_of('/') + 1);
        j.id = base.size() > 4 && base.compare(base.size() - 4, 4, ".txt") == 0 ? base.substr(0, base.size() - 4) : base;
        j.chrom = j.id.size() > 7 && j.id.compare(j.id.size() - 7, 7, "_matrix") == 0 ? j.id.substr(0, j.id.size() - 7) : j.id;
        stat(f.c_str(), &st);
        j.cost = (double)st.st_size;                 // text bytes ~ N^2 ~ restraints
        jobs.push_back(j);
    }
}

}  // namespace

int main(int argc, char** argv) {
    // a device exception must reach stderr in the runtime's own words, not end in the GPU-core-dump helper's broken pipe (c3d_solve_main.cpp)
    setenv("HSA_DISABLE_COREDUMP_ON_EXCEPTION", "1", 0);
    Options o;
    std::vector<std::string> inputs;
    std::string pattern;
    int devices = -1, lanes = 3, map_to = -1;
    for (int a = 1; a < argc; ++a) {
        const std::string s = argv[a];
        auto next = [&](const char* what) -> const char* {
            if (a + 1 >= argc) { fprintf(stderr, "c3d_batch: %s needs a value\n", what); exit(2); }
            return argv[++a];
        };
        if (s == "--out" || s == "-o") o.out = next("--out");
        else if (s == "--devices") devices = atoi(next("--devices"));
        else if (s == "--pair") g_pair = atoi(next("--pair")) != 0;
        else if (s == "--order") g_warm = std::string(next("--order")) != "lpt";
        else if (s == "--map-devices-to") map_to = atoi(next("--map-devices-to"));
        else if (s == "--lanes") lanes = std::max(1, std::min(8, atoi(next("--lanes"))));
        else if (s == "-m") o.models = atoi(next("-m"));
        else if (s == "-k") o.K = atof(next("-k"));
        else if (s == "-a") o.alpha = atof(next("-a"));
        else if (s == "--seed") o.seed = strtoull(next("--seed"), nullptr, 10);
        else if (s == "--min-steps") o.min_steps = atoi(next("--min-steps"));
        else if (s == "--gtol") o.gtol = atof(next("--gtol"));
        else if (s == "--final-minimiser") o.final_min = atoi(next("--final-minimiser"));
        else if (s == "--pattern") pattern = next("--pattern");
        else if (s == "--violations") o.violations = true;
        else if (s == "-h" || s == "--help") { printf("usage: c3d_batch <dir | matrix files...> --out <root> [--devices N] [--lanes 3] [-m 20] [-k 11] [-a 0.5] [--seed S] [--min-steps 3000] [--gtol 1e-2] [--final-minimiser 1] [--pattern text] [--violations] [--pair 1] [--order warm|lpt] [--map-devices-to P (rehearsal)]\n"); return 0; }
        else inputs.push_back(s);
    }
    if (o.out.empty() || inputs.empty() || o.models < 1) { fprintf(stderr, "c3d_batch: need input matrices and --out <root> (see --help)\n"); return 2; }
    std::vector<Job> jobs;
    for (const std::string& in : inputs) collect(in, pattern, jobs);
    if (jobs.empty()) { fprintf(stderr, "c3d_batch: no *_matrix.txt found\n"); return 2; }
    const int ndev = c3d_device_count();
    if (ndev < 1) { fprintf(stderr, "c3d_batch: no HIP device visible: libc3d has no CPU fallback\n"); return 1; }
    if (map_to >= ndev) { fprintf(stderr, "c3d_batch: --map-devices-to %d: this machine has %d device(s)\n", map_to, ndev); return 2; }
    if (map_to >= 0) { if (devices < 1) devices = 1; if (devices > 64) devices = 64; }
    else if (devices < 1 || devices > ndev) devices = ndev;
    mkdir(o.out.c_str(), 0755);
    // longest-processing-time-first: biggest job to the least loaded GPU
    std::vector<size_t> order(jobs.size());
    for (size_t k = 0; k < order.size(); ++k) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return jobs[a].cost > jobs[b].cost; });
    std::vector<double> load(devices, 0.0);
    std::vector<std::vector<size_t>> mine(devices);
    for (size_t k : order) {
        const int g = (int)(std::min_element(load.begin(), load.end()) - load.begin());
        jobs[k].device = g;
        load[g] += jobs[k].cost;
        mine[g].push_back(k);
    }
    // Order WITHIN a GPU's queue (the assignment to GPUs above stays LPT).  The GPU has nothing to anneal until the first job has been
    // parsed and its front half written, and a large matrix takes 30-50 ms of that on a cold lane: with "warm" the first lane starts on the
    // GPU's largest job and the other lanes on its SMALLEST ones (1 ms of parsing), so the device is annealing a few milliseconds after the
    // contexts exist and the large parses run beside those anneals; the rest follows largest first.  Results do not depend on the order.
    if (g_warm && lanes > 1)
        for (int g = 0; g < devices; ++g) {
            std::vector<size_t>& q = mine[g];
            const size_t small = std::min((size_t)(lanes - 1), q.size() > 1 ? q.size() - 1 : 0);
            if (small == 0) continue;
            std::vector<size_t> head(1, q[0]);
            for (size_t k = 0; k < small; ++k) head.push_back(q[q.size() - 1 - k]);        // smallest first
            head.insert(head.end(), q.begin() + 1, q.end() - small);
            q.swap(head);
        }
    const double t0 = now_s();
    g_t0 = t0;
    g_timeline = getenv("C3D_BATCH_TIMELINE") != nullptr;
    std::atomic<int> failed{0};
    // per GPU: `lanes` host threads, each with its own context, take that GPU's jobs in LPT order; while one lane anneals
    // (the GPU phase, serialised per GPU) the other parses, writes the front-half files, scores and writes models
    std::vector<std::thread> workers;
    std::vector<XcdBroker> gpu_mu(devices);
    std::vector<std::atomic<size_t>> next_job(devices);
    std::vector<std::atomic<int>> lanes_up(devices);
    for (int g = 0; g < devices; ++g) { next_job[g] = 0; lanes_up[g] = 0; }
    for (int g = 0; g < devices; ++g)
      for (int lane = 0; lane < lanes; ++lane)
        workers.emplace_back([&, g]() {
            c3d_ctx* ctx = nullptr;
            const int phys = map_to >= 0 ? map_to : g;   // (rehearsal: all logical devices on one physical one)
            if (c3d_create(phys, &ctx) != C3D_OK) {     // this lane only: the GPU's other lanes take its jobs (see below if none came up)
                std::lock_guard<std::mutex> lk(g_print);
                fprintf(stderr, "c3d_batch: GPU %d: %s\n", g, c3d_last_error());
                return;
            }
            ++lanes_up[g];
            if (g_timeline) { std::lock_guard<std::mutex> lk(g_print); fprintf(stderr, "timeline lane of GPU %d has its context at %.4f\n", g, now_s() - g_t0); }
            for (;;) {
                const size_t at = next_job[g]++;
                if (at >= mine[g].size()) break;
                Job& job = jobs[mine[g][at]];
                job.ok = solve_one(ctx, o, job, gpu_mu[map_to >= 0 ? 0 : g]);   // one anneal at a time per PHYSICAL device
                std::lock_guard<std::mutex> lk(g_print);
                if (job.ok) printf("%s\n", job.summary.c_str());
                else { printf("FAILED: %s (%s)\n", job.chrom.c_str(), job.summary.c_str()); ++failed; }
                fflush(stdout);
            }
            c3d_destroy(ctx);
        });
    for (std::thread& w : workers) w.join();
    for (int g = 0; g < devices; ++g) {              // a GPU on which no lane got a context: its jobs were never taken
        if (lanes_up[g] > 0) continue;
        for (size_t at = 0; at < mine[g].size(); ++at) {
            printf("FAILED: %s (no context on GPU %d)\n", jobs[mine[g][at]].chrom.c_str(), g);
            ++failed;
        }
    }
    printf("c3d_batch: %zu matrices x %d models on %d GPU(s), %d lane(s) each, in %.2f s, %d failed; models under %s/<chromosome>/<ID>_model1..5.pdb\n", jobs.size(),
           o.models, devices, lanes, now_s() - t0, failed.load(), o.out.c_str());
    return failed.load() ? 1 : 0;
}
