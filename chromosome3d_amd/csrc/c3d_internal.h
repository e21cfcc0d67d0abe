// c3d_internal.h — shared between the HIP kernels (c3d_device.hip) and the C-ABI host (c3d_api.cpp).
// Not part of the public boundary (that is include/c3d.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace c3d {

constexpr float kBoltz = 0.0019872f;  // kcal/mol/K (X-PLOR/CNS AKMA)
constexpr float kAccel = 418.4f;      // kcal/mol/A/amu -> A/ps^2

// rows of the pair matrix owned by one workgroup of the step kernel (waves per workgroup = kTileRows / rpw)
constexpr int kTileRows = 8;
// the forces-only test hook runs the same tile with 4 rows per wave
constexpr int kEvalRowsPerWave = 4;
constexpr int kEvalBlock = 64 * kTileRows / kEvalRowsPerWave;

// far-away coordinates for the padding beads j in [n, npad): no NOE (target 0), repel term vanishes
constexpr float kPadCoord = 1.0e4f;

struct DevModel {
    int n, npad, ntiles, nrep;
    int rep_base, nrep_g;          // replica group of this launch: [rep_base, rep_base + nrep_g)
    int stage_dma;                 // 1: coordinates staged with global_load_lds (async), 0: through registers
    int rpw;                       // rows per wave of the step kernel (1, 2 or 4); waves/WG = kTileRows/rpw
    int noe_pot, ang_mode, rep_sep; // noe_pot: the c3d_model's 0..3, or 4 = 3 with the lower side's fast soft form (msoexp 2, masym 0)
    int mexp;                      // noe_pot 3 / 4, lower side: exponent of the soft form (c3d_model::msoexp, 1 or 2)
    float rs, tail_c, tail_b;      // soft tail: dE/dD = tail_c - tail_b / D^2  (D > rs)
    float mrs, mtail_c, mtail_b;   // noe_pot 3, lower side: dE/dD = mtail_c - mtail_b / D^(mexp + 1)  (D = t - d > mrs)
    float nmrs;                    // -mrs
    float inv_rs, nm_rs;           // 1 / rs, -mrs / rs: the clamp form works on (d - t) / (rs d), see pair_term (device potential 4: 1 / mrs, rs / mrs)
    // Column layout of the pair loop (c3d_step_core.h): blocks of 256 columns, lane l owns 4 consecutive ones — except in the LAST
    // block, where it owns wl (1..4) consecutive ones: column 256 (nb - 1) + wl l + c.  Up to 8 columns beyond the last block
    // (jl0 .. jl0 + nleft - 1 = n - 1) are "left over": their pair terms are evaluated eight to a row in a separate short pass
    // and summed in their own fixed tree.  n = 455: 256 + 64 x 3 + 7 -> 7 column slots per row instead of 8.
    int wl, nleft, jl0;
    float k_bond2, b0;             // 2*k_bond
    float k_ang2, a0;              // 2*k_ang
    float acc;                     // kAccel / mass
    float t_fac;                   // T = t_fac * sum(v^2):  mass / kAccel / (ndf * kBoltz)
    float fbeta;
    float inv_n;
    // Per-step kernel, device potential 4, no narrow column block (every n > 1024): the per-pair constant t / mrs of ROW PAIRS, resident
    // (Buffers::tgs2, built once per matrix and model by launch_pair_targets): [row pair q][column block jb][lane][column c][row & 1] —
    // the wave's two rows against its four columns of a block are 32 consecutive bytes, two register-pair-aligned float4 loads — with
    // "no restraint" encoded as 1e30 (such a pair feels exactly nothing under the decaying lower bound: pair_term2); nullptr = not in use
    const float* tgs2;
};

struct DevStep {
    int kind;        // 0 MD T-coupling, 1 MD velocity rescale, 2 FIRE step, 3 first FIRE step of a stage, 4 MD begin,
                     // 5 two-point step-size (Barzilai-Borwein) minimiser step, 6 its first step of a stage
    float dt;
    float w_all;     // weights * w
    float w_noe2n;   // -2 * w_all * s_noe
    float w_rep4;    // 4 * w_vdw * k_rep
    float rep_r2;    // (repel_s * r0_rep)^2
    float inv_rep_r2;// 1 / rep_r2
    float w_rep4r2;  // w_rep4 * rep_r2
    float w_rs;      // w_noe2n * rs (device potential 4: w_noe2n * mrs): the factor the clamp form leaves out of every pair term and applies once per row
    float kq;        // w_rep4r2 / w_rs: the repel weight relative to it (w_noe2n != 0; else the general kernels run)
    float t_bath;
};

struct DevFire {
    float dt_start, dt_max, f_inc, f_dec, alpha_start, f_alpha, max_step;
    int n_min;
};

struct FireState {   // per replica, double buffered
    float dt, alpha;
    int npos, pad;
};

// All device pointers of one context.  Layouts (npad = n rounded up to 256, one column block):
//   tgt   [n][npad]            encoded restraint target (see encode_target)
//   X,V   [2][nrep][3][npad]   SoA coordinates / velocities, double buffered by step parity
//   Vinit [nrep][3][npad]
//   P     [2][nrep][ntiles][4] per-tile partial sums
//   S     [2][nrep]            FIRE state
struct DevBuffers {
    float* tgt;
    float* tgs2;    // see DevModel::tgs2 (nullptr unless built)
    float* X[2];
    float* V[2];
    float* Vinit;
    float* P[2];
    FireState* S[2];
    double* E;      // [nrep][4]
};

// host-callable launchers (defined in c3d_device.hip)
// wide: 16 rows a workgroup and four a wave (the shipped potential, clamp forms, no narrow last block; the caller decides: n > 1024)
hipError_t launch_step(const DevModel& m, const DevStep& p, const DevFire& fp, const DevBuffers& b, int parity,
                       bool general_tail, bool wide, hipStream_t s);
hipError_t launch_eval_forces(const DevModel& m, const DevStep& p, const DevBuffers& b, int parity, float* Fout,
                              bool general_tail, int rows_per_wave, hipStream_t s);
hipError_t launch_energy(const DevModel& m, const DevStep& p, const DevBuffers& b, int parity, float s_noe,
                         float k_rep, double rep_r2, hipStream_t s);
hipError_t launch_centre(const DevModel& m, const DevBuffers& b, int parity, hipStream_t s);
size_t pair_targets_floats(int n, int npad);           // size of DevBuffers::tgs2
hipError_t launch_pair_targets(const DevModel& m, const float* tgt, float* tgs2, hipStream_t s);
struct StepRun {    // `count` consecutive steps with the same parameters
    DevStep p;
    int count;
};
struct AnnealIO {   // state buffers of one multi-step launch (io = anneal_io(buffers, parity)), passed by value in the kernel arguments
    const float *pin, *xin, *vin, *vinit;
    const FireState* sin;
    float *xout, *vout, *pout;
    FireState* sout;
};
AnnealIO anneal_io(const DevBuffers& b, int parity);
// cluster kernel (c3d_cluster.hip): many SA steps of the replica group [m.rep_base, m.rep_base + m.nrep_g) in ONE launch;
// reads parity `parity` (through io), writes parity^1 once at the end.  A replica runs on `parts` workgroups of `threads`
// threads (`cw` compute waves x `rpw` rows + `helpers` helper waves) of ONE XCD, `wgs_per_cu` workgroups per CU, grid =
// wgs_per_cu x number of CUs.
// `runs` is the run-length coded step list of the WHOLE program (uploaded once); the launch starts `skip0` steps into
// run `run0` and makes `nsteps` steps.  `tag_base` (launch sequence number << 20) keeps the tags of different launches
// apart, `claim` points at 16 zeroed words that no other launch has used ([0..7] slot counters per XCD, [8] completion
// counter), timeout[0] = 0 (set by a workgroup that gives up), timeout[1] = 0 (set to tag_base | 1 by the workgroup that
// completes the launch's pl.expected).
struct ClusterPlan {
    int rpw, cw, helpers, wgs_per_cu, parts, per_xcd, grid, threads, units, device;
    int static_place = 1;                         // slot = blockIdx / 8, checked against the XCC id (0: per-XCD atomic counters)
    int xcd_base = 0, xcd_count = 8;              // the launch lives on XCDs xcd_base .. xcd_base + xcd_count - 1 (replica r on XCD xcd_base + r % xcd_count): two
    bool two_point = false;        // the launch's range holds two-point minimiser steps (kinds 5 / 6): k_cluster_tp
                                                  // contexts with disjoint sets anneal side by side (c3d_set_option "cluster_xcd_count" / "cluster_xcd_base")
    int late_tiles = 0;                           // tile sums fetched by H0 after the step has started instead of gating it (c3d_cluster.hip)
    unsigned expected = 0;                        // workgroups that must report completion: replicas x parts (the host may raise it: test hook)
    size_t lds;
    hipEvent_t t0 = nullptr, t1 = nullptr;        // when set: the launch stamps them with the kernel's own start and end
};
bool cluster_plan(const DevModel& m, int num_cus, int num_xcc, int forced_geom, int forced_late, int xcd_count, ClusterPlan* plan);
size_t cluster_record_bytes(const DevModel& m, const ClusterPlan& pl);
hipError_t launch_cluster(const DevModel& m, const DevFire& fp, const ClusterPlan& pl, const AnnealIO& io, const float* tgt, void* rec,
                          const StepRun* runs, int run0, int skip0, int nsteps, unsigned tag_base, unsigned* timeout,
                          unsigned* claim, hipStream_t s);
// Code objects.  The runtime loads a code object at the first use of one of its kernels; libc3d does not leave that to chance: every
// translation unit that holds kernels exports a function that loads its code object on the current device (and, for the multi-step
// units, gives every instantiation its dynamic-LDS allowance), and the loader of c3d_api.cpp ("code objects") calls them one at a time,
// under a lock that every launching entry of the library holds shared — no code object is loaded while a thread of the process can launch.
hipError_t preload_device_unit();
hipError_t preload_cluster_base_unit();
hipError_t preload_cluster_unit(int pot, bool two_point);
hipError_t preload_score_unit();
hipError_t preload_embed_unit();
hipError_t preload_f64_unit();
hipError_t preload_sym_unit();
// the hand-off's 16-byte atomicity, watched: one producer workgroup, one consumer workgroup on every other CU (c3d_cluster.hip k_tear16)
hipError_t launch_tear16(int num_cus, void* buf, unsigned* stop, unsigned long long* stats, int iters, hipStream_t s);
// symmetric-tile step for large N (c3d_sym.hip): every pair once.  tiles = sym_tile_list() uploaded, scratch =
// sym_scratch_floats() floats of device memory (row-side and column-side partial forces of one step).
void sym_geometry(const DevModel& m, int* Q, int* G, int* ntiles_offdiag, int* ntiles_diag);
size_t sym_scratch_floats(const DevModel& m);
void sym_tile_list(const DevModel& m, int2* out);
hipError_t launch_step_sym(const DevModel& m, const DevStep& p, const DevFire& fp, const DevBuffers& b, int parity, const void* tiles,
                           float* scratch, hipStream_t s);
// fp64 step (c3d_f64.hip, option "precision" = 64): the CPU restatement's algorithm in its precision on the GPU, one launch per SA
// step of a replica group (k64_step), double buffered by step parity like the fp32 per-step path.
// model_host[15] = s_noe, rswitch, asym, masym, mrswitch, k_bond, b0, k_ang, a0, r0_rep, k_rep, mass, fbeta, min_sep, msoexp;
// step_host[6] = kind, dt, w_all, w_vdw, repel_s, t_bath; fire_host[7] = dt_start, dt_max, f_inc, f_dec, alpha_start, f_alpha, max_step.
// Layouts (np = cols64(n): n rounded up to 128): T [n][np] targets in Angstrom (0.1 * t10, 0 = none); X, V [2][nrep][3][np] SoA;
// Vinit [nrep][3][np]; P [2][nrep][ntiles][4] per-tile sums; S [2][nrep] x fire_state64_bytes().
struct Buffers64 {
    int32_t* t10 = nullptr;        // the integer tenths T is (re)built from whenever the model changes
    double* T = nullptr;
    double* X[2] = {nullptr, nullptr};
    double* V[2] = {nullptr, nullptr};
    double* Vinit = nullptr;
    double* P[2] = {nullptr, nullptr};
    void* S[2] = {nullptr, nullptr};
};
int cols64(int n);
constexpr int kMaxBeads64 = 2560;     // 3 * 8 * np bytes of LDS must stay below the 64 KB a launch gets without opt-in
hipError_t launch_step64(const DevModel& d, const double* model_host, const double* step_host, const double* fire_host, int fire_n_min,
                         const Buffers64& b, int parity, hipStream_t s);
hipError_t launch_targets64(const DevModel& d, const double* model_host, int min_sep, const int32_t* t10, double* T, hipStream_t s);
hipError_t launch_import64(const DevModel& d, const float* Xf, const Buffers64& b, hipStream_t s);
hipError_t launch_export64(const DevModel& d, const Buffers64& b, int parity, float* Xf, float* Vf, float* Pf, hipStream_t s);
size_t fire_state64_bytes();
// K1: IF (n*n fp64, device) -> dist10 (n*n int32, device) and encoded targets (n*npad, device)
hipError_t launch_if_to_target(const double* IF, int n, int npad, double alpha, double K, int min_sep, int rep_sep,
                               double* scratchP, double* partial, int npartial, int32_t* dist10, float* tgt,
                               unsigned char* flags, unsigned* nflag, hipStream_t s);

// A7 (c3d_embed.hip): bead-level metric-matrix distance geometry for every replica
hipError_t launch_dg_embed(const float* tgt, int n, int npad, int nrep, float b0, float lower, uint64_t seed,
                           uint32_t first_replica, int iters, const float* v0, float* U, float* L, float* D2, float* x0,
                           float* x1, hipStream_t s);

// K6 (c3d_score.hip): satisfied / sum-of-deviations / Spearman partial sums for every replica
hipError_t launch_score(const float* xin, const float* tgt, const double* rankA, int n, int npad, int nrep, int range, int min_sep,
                        unsigned nbins, double ma, double mb, double relax, double* xr, unsigned* hist, unsigned* below,
                        double* partial, int* overflow, hipStream_t s);

// Target matrix entry: NOE target in Angstrom, 0 = no restraint (host c3d_set_restraints and K1).
inline float encode_target_host(float t) { return t > 0 ? t : 0.0f; }

}  // namespace c3d
