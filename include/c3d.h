/*
 * c3d.h — C ABI of libc3d.so, the MI355X (gfx950) solver that replaces the
 * `cns_solve < dgsa.inp` process the reference shells out to.
 *
 * Reference boundary being replaced (file:line under the reference tree):
 *   chromosome3D.pl:87-89    IF2dist_new / dist2rr / carr2tbl  -> c3d_set_if_matrix,
 *                            c3d_get_dist10, c3d_write_front_half
 *   chromosome3D.pl:254-289  build_models: job.sh -> `cns_solve < dgsa.inp`, success iff
 *                            <ID>_<M>.pdb exists                -> c3d_run / c3d_write_models
 *   chromosome3D.pl:882-1846 the dgsa.inp deck (knobs :1093-1126, protocol :1574-1829)
 *                                                               -> c3d_model / c3d_stage
 *   chromosome3D.pl:769-829  assess_dgsa (rank by int(REMARK noe)) -> c3d_rank, c3d_assess
 *   spearman_IF_pdb.pl:26-70 scoring                            -> c3d_spearman_if_dist
 *
 * Conventions: plain C types only; every function returns C3D_OK (0) or a negative
 * c3d_status; c3d_last_error() gives a thread-local message.  The caller owns all host
 * buffers, the library owns all device memory.  Nothing here falls back to a CPU solver:
 * without a usable HIP device c3d_create fails with C3D_ERR_NO_DEVICE.
 */
#ifndef C3D_H_
#define C3D_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    C3D_OK = 0,
    C3D_ERR_INVALID = -1,    /* bad argument / call order */
    C3D_ERR_NO_DEVICE = -2,  /* no HIP device, or the gfx950 code object cannot load */
    C3D_ERR_HIP = -3,        /* a HIP runtime call failed */
    C3D_ERR_IO = -4,         /* file could not be read / written / parsed */
    C3D_ERR_NOMEM = -5,
    C3D_ERR_DIVERGED = -6    /* non-finite coordinates/forces after a run */
} c3d_status;

typedef struct c3d_ctx c3d_ctx;

/* Energy model of one bead chain (defaults: c3d_default_model).  Mirrors the deck:
 * con_wt chromosome3D.pl:66,1111,1120; mass/fbeta :1415-1416; SEPARATION :20. */
typedef struct {
    int32_t min_sep;   /* 5: restraints only for |i-j| >= min_sep                        */
    int32_t noe_pot;   /* 0 symmetric soft-square, 1 X-PLOR soft-square, 2 square,
                          3 CNS soft-square with a soft LOWER side too (mrswitch, masym, msoexp; default) */
    int32_t rep_sep;   /* repel acts on |i-j| >= rep_sep (1..3)                          */
    int32_t ang_mode;  /* (i,i+2) term: 0 lower bound only, 1 harmonic                   */
    float s_noe;       /* NOE scale = con_wt = 10                                        */
    float rswitch;     /* 0.5: the upper side is square up to d - t = rswitch            */
    float asym;        /* 2.0: slope of the upper tail in units of rswitch (tail slope = asym x rswitch x S = 10) */
    float k_bond, b0;  /* pseudo-bond (i,i+1)                                            */
    float k_ang, a0;   /* pseudo-angle (i,i+2)                                           */
    float r0_rep;      /* bead contact distance, scaled by stage `repel`                 */
    float k_rep;       /* bead-level repel multiplier                                    */
    float mass;        /* 100 amu                                                        */
    float fbeta;       /* 10 /ps                                                         */
    float masym;       /* noe_pot 3: asymptote slope of the lower side (CNS masymptote)  */
    float mrswitch;    /* noe_pot 3: the lower side is square up to t - d = mrswitch     */
    int32_t msoexp;    /* noe_pot 3: exponent of the lower side's soft form a + b / D^msoexp + masym D beyond mrswitch
                          (CNS msoexponent), 1 or 2; 0 = the default (2).  Shipped: mrswitch 10, masym 0, msoexp 2 — X-PLOR's own defaults
                          (rswitch 10, asymptote 0, soexponent 2), which the deck never overrides for the minus side:
                          the push on a pair far inside its target rises to 2 S mrswitch and then DECAYS as D^-3     */
} c3d_model;   /* ABI: no size / version member — every caller fills the struct through c3d_default_model of the library it links (the Perl
                  binding, c3d_solve, c3d_batch and the ctypes mirror do); `msoexp` was appended in round 4 ("ABI notes") */

/* One stage of the annealing schedule (defaults: c3d_default_schedule, which restates
 * chromosome3D.pl:1631-1700 hot stages, :1729-1782 slow cool, :1790-1803 minimisation). */
typedef struct {
    int32_t kind;      /* 0 MD + T-coupling, 1 MD + velocity rescale, 2 FIRE minimise,
                          5 minimise with two-point (Barzilai-Borwein) step sizes — one force evaluation a step, no energy, the length
                            of a move = the inverse of a one-number curvature estimate from the previous move and the change of the force
                            over it, taken one evaluation late (the replica sums of a step reach the next one) — for the first
                            `final_minimiser_steps` (1000) steps, then FIRE for what is left of nsteps: half the evaluations FIRE needs
                            to the same exit test, the same minima; the default schedule's final stage since round 5
                            (option final_minimiser = 0: a stage of kind 5 is a FIRE stage)                                        */
    int32_t nsteps;
    float dt;          /* ps (MD)                                                        */
    float w_all;       /* `weights * w`                                                  */
    float w_vdw;       /* vdw weight                                                     */
    float repel_s;     /* nbonds repel=                                                  */
    float t_bath;      /* K                                                              */
} c3d_stage;

typedef struct {
    float dt_start, dt_max, f_inc, f_dec, alpha_start, f_alpha, max_step;
    int32_t n_min;
} c3d_fire_params;

/* --- lifecycle ---------------------------------------------------------------------- */
const char* c3d_last_error(void);
const char* c3d_version(void);
int c3d_device_count(void);
int c3d_create(int device, c3d_ctx** out);
void c3d_destroy(c3d_ctx* ctx);

void c3d_default_model(c3d_model* m);
void c3d_default_fire(c3d_fire_params* f);
/* Fills up to `cap` stages; returns the number of stages of the default schedule
 * (1 pre-minimisation + 5 hot + 81 cool + 1 final minimisation of `min_steps`). */
int c3d_default_schedule(c3d_stage* stages, int cap, int min_steps);

/* --- problem set-up ----------------------------------------------------------------- */
/* K1 on device: D = K * mean(IF^alpha) / IF^alpha, quantised like "%.1f"; builds the n x n
 * target matrix (restraints for |i-j| >= min_sep, IF > 0).  IF is row-major n*n (host). */
int c3d_set_if_matrix(c3d_ctx* ctx, const double* IF, int n, double alpha, double K);
/* Alternative entry: restraint rows as in contact.tbl (1-based i, j; target in tenths of A). */
int c3d_set_restraints(c3d_ctx* ctx, int n, int R, const int32_t* ri, const int32_t* rj, const int32_t* rt10);
/* n*n int32, tenths of an Angstrom, -10 where IF == 0 (the <ID>.dist content). */
int c3d_get_dist10(c3d_ctx* ctx, int32_t* dist10);
int c3d_num_beads(const c3d_ctx* ctx);
int c3d_num_restraints(const c3d_ctx* ctx);

int c3d_set_model(c3d_ctx* ctx, const c3d_model* m);
int c3d_set_schedule(c3d_ctx* ctx, const c3d_stage* stages, int n_stages, const c3d_fire_params* fire,
                     float gtol, int check_every);
/* Execution knobs.  The three launch forms (many steps per launch, one launch per step, symmetric tiles) and every setting
 * of the knobs below except `precision`, `symmetric` and `start` end in the same bits.
 *   resident        -1 (default) / 0 / 1: run step ranges as ONE multi-step launch of the cluster kernel (c3d_cluster.hip)
 *                   wherever a geometry exists / never / as -1, without the back-off after an abandoned launch
 *   cluster         0: never use the cluster kernel (test knob)
 *   cluster_geometry  100 x compute waves + 10 x rows per wave + helper waves (e.g. 1244): force that geometry of the cluster
 *                   kernel instead of the planner's choice (measurement knob; 0 = planner); before c3d_init_replicas
 *   cluster_late_tiles  1 (default) / 0: how the per-tile sums of a step reach the wave that needs them in the cluster kernel —
 *                   1: where the planner finds the pair loop long enough, that wave fetches them after the next step has
 *                   started (off the critical path); 0: always gathered with the rows before it starts.  Same bits either way
 *                   (measurement knob); before c3d_init_replicas
 *   narrow_columns  1 (default) / 0: column layout of the pair loop — the lanes of the last 256-column block own 1..4 columns each and
 *                   up to 8 columns behind it are summed separately (N = 455: 7 column slots per row instead of 8); 0 = four
 *                   columns per lane everywhere (round 2's layout).  The two layouts sum in different orders: results agree to
 *                   rounding, not bit for bit; every launch form follows the layout in force.  Before c3d_init_replicas.
 *   cluster_static_placement  1 (default): a cluster launch numbers the workgroups of an XCD as blockIdx / 8 and every workgroup
 *                   checks its XCC id against blockIdx % 8; a mismatch abandons the launch and switches the context to 0 =
 *                   per-XCD atomic slot counters (2: test hook, the next launch's workgroup 0 reports a mismatch)
 *   cluster_xcd_count  8 (default) .. 1: the multi-step launches of this context live on that many XCDs (replica r on XCD cluster_xcd_base +
 *                   r % count; the planner fits the replicas of the fullest of THEM into its 32 CUs), workgroups elsewhere exit at once; before
 *                   c3d_init_replicas.  cluster_xcd_base (default 0) = the first of them, may change between launches.  Two contexts with
 *                   disjoint XCD sets anneal side by side without sharing a CU (c3d_batch pairs config 4's small chromosomes this way); the
 *                   same bits whatever the set (a replica's trajectory does not depend on where it runs)
 *   cluster_num_xcc, cluster_inject_incomplete, resident_inject_timeout   test hooks of the cluster kernel's safety net
 *                   (a device that does not expose 8 XCDs gets no cluster plan; a launch that ends without its completion
 *                   mark or with a time-out is re-run on the per-step path)
 *   precision       32 (default) or 64: the fp64 reference kernels (c3d_f64.hip); call before c3d_init_replicas.  The fp64 step stages a
 *                   replica's coordinates in LDS: at most 2560 beads (c3d_init_replicas returns C3D_ERR_INVALID beyond; fp32: 8192)
 *   symmetric       1: symmetric-tile kernels for large N (c3d_sym.hip; opt-in); call before c3d_init_replicas
 *   eval_rows_per_wave  4 (default) / 2 / -2: form of the forces hook (c3d_eval_forces) — four rows per wave with the scalar pair term; 2 = two
 *                   rows per wave, the step kernels' code (shipped potential: the packed pair term); -2 = two rows per wave, scalar pair
 *                   term.  2 and -2 return the same bits (a -m gpu test); test knob
 *   pair_targets    1 (default) / 0: beyond the multi-step kernel's reach (n > 1024) the per-step kernel of the shipped potential reads
 *                   resident pre-scaled targets of row pairs (built once per matrix and model) instead of forming the per-pair
 *                   constants from the target matrix in every step.  Same bits either way (measurement knob)
 *   wide_tiles      1 (default) / 0: beyond the multi-step kernel's reach (n > 1024) the per-step kernel of the shipped potential runs 16 rows
 *                   a workgroup and four a wave (two packed row pairs; needs pair_targets 1) instead of 8 and two: a wave's fixed work per
 *                   step is shared by twice the pair terms (N = 2500 x 8: 29.8 -> 26.2 us per step).  Through round 4 equal to the narrow form within rounding
 *                   only; since round 5 — a row's force is one explicit fma in every form — the same bits in every problem of tools/fuzz_wide.py
 *                   (measurement knob)
 *   prefetch_ranks  1 (default) / 0: c3d_set_if_matrix starts the IF side of the Spearman coefficient (average ranks of the matrix's ordered
 *                   pairs |i-j| >= 3, spearman_IF_pdb.pl:30-44: 5 ms of host time at N = 455) on a helper thread over a copy of the matrix;
 *                   c3d_score_replicas takes it when its IF argument holds the same numbers, else computes it as before.  Up to 2048 beads
 *                   only (the worker keeps 16 bytes per pair until the context goes: 67 MB there).  Same result either way
 *                   (measurement knob; stat "rank_prefetch_hits")
 *   final_minimiser 1 (default) / 0: what a stage of kind 5 (the default schedule's final stage) runs — two-point step sizes handing over to
 *                   FIRE after final_minimiser_steps, or FIRE throughout as in rounds 1-4.  Stages of kind 2 are FIRE whatever this says
 *   final_minimiser_steps   1000 (default): two-point steps of a kind-5 stage before FIRE takes it over (>= 2)
 *   start           0 (default) random coil, 1 extended strand (chromosome3D.pl:2413-2416)
 *   use_graph       != 0: per-step path replays hipGraphs (default 1)
 *   replica_groups  1..4 stream groups of the per-step path (default 2);  graph_chunk, rows_per_wave,
 *   stage_dma       tuning and test knobs of the per-step kernel
 *   event_timing    1 (default) / 0: HIP-event pair around c3d_run / c3d_run_steps (feeds c3d_last_timing)
 *   kernel_timing   1: start/stop events attached to every multi-step launch (stat "last_kernel_us")
 *   spin_wait_us    how long c3d_run_steps watches the completion mark a multi-step launch writes into host-mapped memory before it
 *                   falls back to hipStreamSynchronize (default 400; 0 = always synchronise).  Results are untouched by it. */
int c3d_set_option(c3d_ctx* ctx, const char* key, double value);

/* Process-wide switches, to be set before the first c3d_create (no environment variable is read by the library):
 *   preload         which code objects c3d_create loads before it returns (the library never leaves a load to the runtime's first-launch
 *                   path and never loads beside a launch: csrc/c3d_api.cpp "code objects"):
 *                   1 (default)  what a default job launches from — K1 + per-step unit, both multi-step units of the shipped potential,
 *                                scoring — about 9 ms, once per process and device;
 *                   2            all sixteen units (long-lived executors: nothing is ever loaded after the first c3d_create of a device);
 *                   0            none: each unit at the first entry that needs it (measurement knob).
 *                   Units beyond the default set (other potentials, fp64, symmetric tiles, embedding) load at the first entry that
 *                   needs them in every mode, while no other thread of the process is inside a launching entry.  Results are untouched. */
int c3d_set_process_option(const char* key, double value);

/* --- replicas ----------------------------------------------------------------------- */
/* n_replicas chains with ids first_replica .. first_replica+n_replicas-1; the RNG is
 * Philox4x32-10 keyed by (seed, replica id), so a replica's trajectory does not depend on
 * how replicas are spread over processes or GPUs (seed 82364: chromosome3D.pl:980). */
int c3d_init_replicas(c3d_ctx* ctx, int n_replicas, uint64_t seed, uint32_t first_replica);
/* A7 (deck chromosome3D.pl:1471-1525, bead-level restatement): replace the random-coil start of every
 * replica by a metric-matrix distance-geometry embedding — bounds from the restraints, triangle
 * smoothing, random trial distances (Philox, keyed by replica id), 3 leading eigenvectors found with
 * `iters` orthogonal iterations (50 is plenty).  Call between c3d_init_replicas and c3d_run. */
int c3d_embed_replicas(c3d_ctx* ctx, int iters);
/* overwrite coordinates (n_replicas*n*3, xyz interleaved) — tests and restarts */
int c3d_set_coords(c3d_ctx* ctx, const float* xyz);
int c3d_get_coords(c3d_ctx* ctx, float* xyz);
/* (after a range that ended inside the two-point part of a final stage — kind 5, its first final_minimiser_steps steps — the velocity
 *  slot holds the previous evaluation's FORCE per bead, kcal/mol/A: that minimiser has no velocities and keeps its history there;
 *  after MD and FIRE steps it is the velocity in A/ps) */
int c3d_get_velocities(c3d_ctx* ctx, float* v);

/* --- solve -------------------------------------------------------------------------- */
/* whole schedule, with the gtol exit of the final minimisation; centres the models. */
int c3d_run(c3d_ctx* ctx);
/* advance by at most nsteps SA steps of the schedule (no early exit); returns steps done
 * through *done.  Used by the benchmark and by tests that follow a trajectory. */
int c3d_run_steps(c3d_ctx* ctx, long nsteps, long* done);
long c3d_schedule_length(const c3d_ctx* ctx);   /* SA steps in the whole schedule */
long c3d_steps_done(const c3d_ctx* ctx);
int c3d_centre(c3d_ctx* ctx);
/* device time of the last c3d_run / c3d_run_steps, from HIP events on the solver's stream */
int c3d_last_timing(const c3d_ctx* ctx, double* ms_total, long* steps, long* launches);
/* Counters of the context since c3d_create, for benchmarks and tests (no reference counterpart: the reference's
 * only instrument is the wall clock around `./job.sh`, chromosome3D.pl:287).  Keys: "graph_captures" (hipGraphs
 * captured + instantiated), "graph_launches", "graphs_cached", "step_launches" (k_step dispatches), "resident_launches",
 * "cluster_launches", "resident_fallbacks" (multi-step launches abandoned for the per-step path), "cluster_incomplete"
 * (those of them that ended without every (replica, part) workgroup reporting), "cluster_static_placement",
 * "cluster_placement_mismatches", "cluster_xcd_count", "cluster_xcd_base", "cluster_ok" (1: a multi-step geometry exists for the replicas as initialised), "spin_completions" (multi-step launches whose end was seen on the completion mark), "num_xcc", "last_path"
 * (0 per-step, 2 k_cluster), "cluster_parts", "cluster_rows_per_wave", "cluster_late_tiles", "last_host_launch_us", "last_host_sync_us" (host time inside the launch / synchronise call of the last c3d_run_steps, cluster launches), "cluster_compute_waves", "replica_groups",
 * "k1_recomputed" (elements of the last c3d_set_if_matrix that sat within 1e-10 of a "%.1f" rounding tie and were redone
 * on the host in the reference's operation order), "k1_patched" (how many of those changed, since c3d_create),
 * "rms_force" (largest RMS force component over the replicas at the last minimiser step: the quantity c3d_run holds
 * against gtol, the stand-in for L-BFGS's convergence test of chromosome3D.pl:1800-1803). */
int c3d_get_stat(const c3d_ctx* ctx, const char* key, double* value);
/* Test hook, no reference counterpart: the multi-step kernel's hand-off trusts a 16-byte unit once its tag word matches — i.e. that a
 * 16-byte aligned store is never observed half-written by a 16-byte load on gfx950.  This runs that exact store / load pair (one producer
 * workgroup, a consumer workgroup on every other CU, the context's stream) for `iterations` rewrites of 1024 units and returns the number
 * of unit reads, of TORN units (must be 0) and of reads that saw a new value. */
int c3d_debug_tear16(c3d_ctx* ctx, int iterations, unsigned long long* unit_reads, unsigned long long* torn, unsigned long long* fresh);
/* Name of the step kernel the last c3d_run / c3d_run_steps ran on, as a profiler prints it (thread-local string). */
const char* c3d_step_kernel_name(const c3d_ctx* ctx);

/* One evaluation through the production pair kernel at the replicas' current coordinates:
 * F (n_replicas*n*3) = total weighted force; e (n_replicas*3) = unweighted (noe, bond+angle,
 * repel) energies in fp64. Either may be NULL. */
int c3d_eval(c3d_ctx* ctx, float w_all, float w_vdw, float repel_s, float* F, double* e);
/* per replica: e[3*r + {0,1,2}] = E_noe, E_bond(+angle), E_repel at the final weights */
int c3d_get_energies(c3d_ctx* ctx, double* e);
/* K6 on the device, for every replica at its current coordinates: the restraint-satisfaction count and
 * the sum of deviations of chromosome3D.pl:447-485 / :581-600 (relax 0.5 A, threshold 0.2 A) and, if IF
 * (the n*n matrix given to c3d_set_if_matrix) and rho are non-NULL, Spearman(IF, d) over |i-j| >= range
 * as spearman_IF_pdb.pl:42-70 defines it.  Any output pointer may be NULL. */
int c3d_score_replicas(c3d_ctx* ctx, const double* IF, int range, int32_t* satisfied, double* sum_dev, double* rho);
/* rank[k] = replica index with the k-th lowest int(E_noe) (chromosome3D.pl:796-802,822-828);
 * ties broken by replica id. */
int c3d_rank(c3d_ctx* ctx, int32_t* rank);

/* --- host helpers (formats of the reference; no device needed) ------------------------ */
/* chromosome3D.pl:116-129,164-179: whitespace-separated numbers, N = fields on line 1.
 * *IF is malloc'ed (free with c3d_free). */
int c3d_parse_if_file(const char* path, double** IF, int* n);
void c3d_free(void* p);
/* <ID>.dist, <ID>.rr, contact.tbl exactly as chromosome3D.pl:156-161, 203-205, 360 write them */
int c3d_write_front_half(const int32_t* dist10, int n, int min_sep, const char* dist_path,
                         const char* rr_path, const char* tbl_path, int* n_restraints);
/* contact.tbl reader (format chromosome3D.pl:360; parse rules :497-520) */
int c3d_read_tbl(const char* path, int32_t** ri, int32_t** rj, int32_t** rt10, int* R);
/* CA-only model in the layout assess_dgsa leaves (chromosome3D.pl:853-857, 208-215) preceded
 * by REMARK lines carrying the energies (`REMARK noe = ...`, :611-614). resname from the
 * bundled output_models (MET). */
int c3d_write_pdb(const char* path, const float* xyz, int n, double e_noe, double e_bond, double e_rep,
                  const char* title);
/* Residue names of the models c3d_write_pdb writes.  The reference names residue i after letter i of a fixed 663-letter
 * pseudo-protein (`$REFSEQUENCE`, chromosome3D.pl:93-98; 3-letter codes through %AA1TO3, :78) — the chemistry CNS needs and a
 * bead model does not; its bundled output_models were re-exported with every residue MET, which is the default here.  seq1 =
 * one-letter amino-acid codes, one per bead (chromosome3d_amd/data/refsequence.fasta holds the reference's); beads beyond its
 * end, and letters outside the 20 standard ones, are MET.  NULL or "" restores all-MET.  Process-wide; call before writing. */
int c3d_set_residue_sequence(const char* seq1);
int c3d_read_pdb_ca(const char* path, float** xyz, int* n);
/* A16, what assess_dgsa does to every solver-output PDB before it ranks them (chromosome3D.pl:813-820): filter_nonCA
 * :864-880 (REMARK rows go to `log_path`, appended after a line with the input path; ATOM rows containing "CA" stay),
 * reindex_chain :831-862 (atoms and residues renumbered from 1, chain id blanked), `sed -i "s/END//g"` :818 (the END row
 * becomes an empty line), add_connect_rows :208-215 (CONECT i i+1, END).  `out_path` may equal `in_path`; `log_path` may
 * be NULL.  The result is byte-identical to the file the reference leaves behind (tests/golden/output_side). */
int c3d_shape_pdb(const char* in_path, const char* out_path, const char* log_path);
/* chromosome3D.pl:447-485, 581-600 on coordinates rounded to 3 decimals as a PDB holds them */
int c3d_assess(const float* xyz, int n, int R, const int32_t* ri, const int32_t* rj, const int32_t* rt10,
               double relax, int* satisfied, double* sum_dev);
/* The same two numbers and, appended to `path`, the violation table count_satisfied_tbl_rows leaves behind (chromosome3D.pl:475-483): two
 * '#' lines naming pdb_label and tbl_label, then one row per restraint in the reference's format, violated rows first. */
int c3d_write_violations(const float* xyz, int n, int R, const int32_t* ri, const int32_t* rj, const int32_t* rt10, double relax,
                         const char* pdb_label, const char* tbl_label, const char* path, int* satisfied, double* sum_dev);
/* spearman_IF_pdb.pl:42-70 */
int c3d_spearman_if_dist(const double* IF, const float* xyz, int n, int range, double* rho);
/* the same for n_models models (n_models*n*3 coordinates) of one matrix: IF is ranked once */
int c3d_spearman_if_dist_batch(const double* IF, const float* xyz, int n, int n_models, int range, double* rho);

/* Cross-resolution check of the reference's output_models/similarity.txt (data only; the definitions
 * were recovered from the bundled models and reproduce its numbers to 1e-12):
 *   c3d_reduce_model      mean of consecutive bead pairs (an odd last bead is kept): out has (n+1)/2 beads
 *   c3d_model_similarity  two models of n beads: Spearman of the i<j distances, and the RMS difference of
 *                         those distances after scaling a's by mean(d_b)/mean(d_a) ("RMSD" in that file) */
int c3d_reduce_model(const double* xyz, int n, double* out);
int c3d_model_similarity(const double* a, const double* b, int n, double* spearman, double* rmsd);

#ifdef __cplusplus
}
#endif
#endif /* C3D_H_ */
