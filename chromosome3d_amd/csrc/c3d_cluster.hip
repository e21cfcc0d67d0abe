// c3d_cluster.hip — K3/K4 as ONE launch for many SA steps: a replica on a few LARGE workgroups of ONE XCD (gfx950).
//
// The solver is bound by the vector ALU (a pair term is ~27 ns of VALU pipe per SIMD: tools/microbench/valu_forms, pair_loop_insitu, and the
// SQ counters in profiles/), so a step costs what its instruction count costs — and in a kernel where every wave also
// reduces the replica sums, evaluates chain terms, updates rows and handles records, that overhead is as large as the
// pair loop.  Here the waves of a workgroup are specialised:
//
//   compute waves (CW of them, RPW rows each)   pair loop with targets and NOE weights in registers, butterfly sum,
//                                               three LDS words per row — nothing else
//   helper H0   (lane k <-> row k of the workgroup)   replica sums -> step scalars; after the barrier: force = pair sum
//                                               + chain sum, row update, tile sums (DPP), publishes the record
//   helpers H1..H3   the chain terms of 16 rows each, one (row, neighbour) per lane, quad sum
//
// and the workgroups of a replica are few and share an XCD:
//
//   placement   (round 3) the dispatcher deals the workgroups of a launch to the XCDs round-robin with an offset that is constant for
//               the launch, so slot = blockIdx / 8 numbers the workgroups of every XCD: slot -> (replica, part), replica r on XCD r % 8.
//               Every workgroup checks HW_REG_XCC_ID against blockIdx % 8 and ORs the offset it sees into one word; the completion
//               mark is refused unless exactly one offset was seen (then: per-step re-run, and the context falls back to round 2's
//               per-XCD atomic slot counters).  A CU without work exits at once.
//   hand-off    16-byte units with ONE tag word, published with PLAIN stores (they write through the CU's L1 into the XCD's L2 and
//               stay there): {tag, x, y, z} per row the moment the row is integrated, then {tag, s0, s1, s2} {tag, s3, 0, 0} per 8-row
//               tile.  Every thread gathers its row units of the replica's P parts with L1-bypassing (sc1) loads until the tags
//               match; the tile units are fetched by H0 alone, after the next step has started where the pair loop is long enough
//               (template parameter LATE).  Producer and consumers share one L2; nothing crosses the fabric.  tag = (launch number
//               << 20) + step + 1; two record buffers alternate by step parity.  A unit is trusted as a whole once its tag matches:
//               that a 16-byte aligned store is never seen half-written by a 16-byte load is a property of this hardware, watched by
//               k_tear16 below (c3d_debug_tear16: the same store, the same load, the solver's stream) in the -m gpu suite.
//   P == 1      (N <= 64) a replica is one workgroup: no record at all, positions go LDS -> LDS.
//   completion  every workgroup that finishes its last step bumps a counter; the one that completes replicas x parts writes a mark
//               into host-mapped memory (no mark -> the host re-runs the range step by step); the host may end its wait on that
//               mark instead of hipStreamSynchronize (c3d_api.cpp run_cluster, option spin_wait_us).
//
// Arithmetic: c3d_step_core.h, every row's force and every sum formed in the order k_step forms it, so a range run
// here ends in the bits of the per-step path.  Every spin is bounded; a workgroup that gives up sets *timeout and the
// host re-runs the range step by step (the launch only writes its outputs in its last step).
// Deck: chromosome3D.pl:1646-1700 (hot MD), :1729-1782 (cooling), :1790-1803 (minimisation).
#include <hip/hip_ext.h>
#include <atomic>
#include <type_traits>
#include <cstdio>
#include <cstdlib>

#include "c3d_step_core.h"

namespace c3d {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kClMaxThreads = 1024;

// Diagnostic build only (-DC3D_STAMPS, tools/stamps): cycles per phase of helper H0 of workgroup (replica 0, part 0),
// summed over the steps of a launch, plus the number of gather sweeps.
#ifdef C3D_STAMPS
// raw s_memrealtime stamps (100 MHz) of the last 64 steps of a launch: [step & 63][point], written by one lane of the
// stamping wave (H0 of replica 0 part 0 for points 0..5, compute wave 0 of the same workgroup for points 6..7)
__device__ unsigned long long g_cstamps[64][8];
// launch phases of the stamping workgroup: [0] kernel entry, [1] prologue done (targets, coordinates, constants in place), [2] first
// step started (past the first B1), [3] last step finished (outputs stored, completion counted)
__device__ unsigned long long g_pstamps[8];
#define CSTAMP(k) do { if (stamper) g_cstamps[s & 63][k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define CSTAMP_C(k) do { if (cstamper) g_cstamps[s & 63][k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CSTAMP(k) do { } while (0)
#define CSTAMP_C(k) do { } while (0)
#endif

// The kernel's body.  TWO: the launch's range holds steps of the two-point minimiser (kinds 5 / 6).  Their row update is compiled into a
// kernel of its own, k_cluster_tp: with it in the loop the other kinds' steps measured 0.3-1 % longer (same box, A/B, round 5) although
// they never run it — the lone H0 wave's path is that sensitive to what else the loop holds — so the launches of the hot and cool stages
// keep the kernel they had.
template <int POT, int RPW, int NB, int WL, bool LATE, bool TWO>
__device__ __forceinline__ void cluster_body(
    // argument order: the first 16 dwords arrive in SGPRs with the wave (kernarg preload, Makefile), the rest after a ~0.7 us fetch —
    // what the placement and the first loads of the prologue need comes first
    // (14 dwords here: 16 user SGPRs less the kernarg pointer)
    const float* __restrict__ tgt, unsigned* __restrict__ claim, const StepRun* __restrict__ runs, const int P, const int CW, const int NH,
    const int static_place, const int nbeads, const int run0, const int skip0, const int nsteps,
    const unsigned tag_base, u32x4* __restrict__ rec, volatile unsigned* __restrict__ timeout, const unsigned expected,
    const AnnealIO& io, const DevModel& m, const DevFire& fp) {
    constexpr int NPAD = 256 * NB;
    constexpr int MAXT = NPAD / 8;
    constexpr int KUMAX = NB > 2 ? 3 : 2;         // gather loads per thread: P * 2 RW <= threads * KUMAX
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* ys = smem + NPAD;
    float* zs = smem + 2 * NPAD;
    float* ps = smem + 3 * NPAD;                  // [MAXT][4] per-tile sums of the previous step
    float* fbuf = ps + 4 * MAXT;                  // [3][64] pair sums of this workgroup's rows (compute waves -> H0)
    float* cbuf = fbuf + 3 * 64;                  // [3][64] chain sums of this workgroup's rows (H1..H3 -> H0)
    float* lbuf = cbuf + 3 * 64;                  // [3][64] sums over the left-over columns of this workgroup's rows (H1..H3 -> H0)
    float* lcon = lbuf + 3 * 64;                  // [3 helpers][4 passes][2][64] pair constants (b, a) of the left-over columns: registers are scarce
    float* dump = lcon + 3 * 4 * 2 * 64;          // [4] nobody reads
    int* s_slot = reinterpret_cast<int*>(dump + 4);
    // dump + 8 ..: [CW][RPW * NB][64] float4, the compute waves' pair_a constants
#ifdef C3D_STAMPS
    const unsigned long long t_entry = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthreads = (CW + NH) * 64;           // NH helper waves: H0 + (NH - 1) chain helpers
    const int RW = CW * RPW;                      // rows of one workgroup: a multiple of 8 (whole tiles), <= 64

    // ---- placement: (replica, part) from a slot number inside this workgroup's XCD -------------------------
    // static_place: the workgroups of a launch are dealt to the XCDs round-robin in launch order — XCD (blockIdx + o) % 8 with
    // an offset o that is constant for the launch (0 on the null stream, 7 on this library's stream: tools/microbench/
    // xcc_dispatch.hip) — so blockIdx / 8 numbers the workgroups of every XCD without collision, and without the atomic round
    // trip of a slot counter (~2 us of every launch).  Not trusted blindly: every workgroup ORs the offset IT sees into one
    // word, and the workgroup that completes the launch refuses the completion mark unless exactly one offset was seen
    // (timeout[2] tells the host why; it re-runs the range step by step and switches this context to the counters).
    // `static_place` also carries the launch's XCD set (bits 8-15: first XCD, bits 16-23: how many; the preloaded argument dwords are
    // all taken): replica r lives on XCD first + r % count, and a workgroup on any other XCD has no work.  Two contexts with disjoint
    // sets anneal side by side (config 4 pairs its small chromosomes: c3d_batch); the default 0 / 8 is the whole device.
    const int xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0x7);     // HW_REG_XCC_ID
    const int sp_mode = static_place & 0xff, xcd0 = (static_place >> 8) & 0xff, nxcd = (static_place >> 16) & 0xff;
    int slot;
    if (sp_mode) {
        const unsigned off = ((unsigned)xcc - blockIdx.x + (sp_mode == 2 && blockIdx.x == 0 ? 1u : 0u)) & 7u;      // 2: test hook
        if (tid == 0) atomicOr(&claim[9], 1u << off);
        slot = (int)(blockIdx.x >> 3);
    } else {
        if (tid == 0) *s_slot = (int)atomicAdd(&claim[xcc], 1u);
        __syncthreads();
        slot = *s_slot;
    }
    const int lxcd = xcc - xcd0;                                  // this XCD's index inside the launch's set
    const int lrep = lxcd + nxcd * (slot / P), part = slot % P;   // replica index inside this launch's group
#ifdef C3D_STAMPS
    const unsigned long long t_placed = __builtin_amdgcn_s_memrealtime();
#endif
    // waves 0 .. NH-1 are the helpers (the OLDEST waves of their SIMDs: instruction arbitration favours age, and the
    // helpers' chains are the serial part of a step), waves NH .. NH+CW-1 compute
    const bool is_compute = wave >= NH, is_h0 = wave == 0;
    const int cwave = wave - NH;                  // compute wave index
    const int wg_row0 = part * RW;
    const int row0 = wg_row0 + cwave * RPW;       // compute waves: first row of the wave
    // The long loads of the prologue leave first, from the arguments that came with the wave: the first run's parameters (a scalar
    // load nobody would otherwise ask for before the first barrier) and this wave's rows of the target matrix, 96 KB per workgroup
    // (the same for every replica: before it is known whether this CU has work at all)
    const StepRun first_run = runs[run0];
    float4 traw[RPW][NB];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
            float4 t = make_float4(0, 0, 0, 0);
            if (is_compute) {
                const float* trow = tgt + (size_t)min(row0 + r, nbeads - 1) * NPAD + 256 * jb;
                if (jb < NB - 1 || WL == 4) t = *reinterpret_cast<const float4*>(trow + 4 * lane);
                else if constexpr (WL == 3) {       // the last block's lanes own WL columns: one 12-byte load
                    struct F3 { float a, b, c; };
                    const F3 q = *reinterpret_cast<const F3*>(trow + 3 * lane);
                    t = make_float4(q.a, q.b, q.c, 0.0f);
                } else if constexpr (WL == 2) {
                    const float2 q = *reinterpret_cast<const float2*>(trow + 2 * lane);
                    t = make_float4(q.x, q.y, 0.0f, 0.0f);
                } else t = make_float4(trow[lane], 0.0f, 0.0f, 0.0f);
            }
            traw[r][jb] = t;
        }
    if (lxcd < 0 || lxcd >= nxcd || lrep >= m.nrep_g) return;     // this CU has nothing to do
    const int rep = m.rep_base + lrep;
    const bool solo = P == 1;
    if (!is_compute) __builtin_amdgcn_s_setprio(3);

    const size_t roff = (size_t)rep * 3 * NPAD;
    const int hrow = wg_row0 + lane;              // H0: the row of this lane
    const bool hfin = is_h0 && lane < RW && hrow < m.n;
#define C3D_HROW_INDEX const size_t ix = roff + hrow, iy = ix + NPAD, iz = iy + NPAD   /* formed where used: H0 only */
    // units of one (parity, replica): row r at r, tile t at NPAD + 2 t and + 1; only real rows and tiles travel.  The ROW units gate
    // the next step (gathered into LDS before B1).  The TILE units are wanted by H0 alone, for the step scalars it forms while the
    // compute waves already run, and travel in one of two ways (a template parameter: as a run-time switch the two forms cost the
    // short-loop problems 7 % — H0's chain is their critical path and every extra branch and SGPR reload is on it; cluster_plan chooses):
    //   LATE        H0 fetches them itself after B1 (tile lane + 64 k into lane's registers); H0 publishes them after B3, while the
    //               other waves already gather rows: the tile sums and their L2 round trip are off the critical path of a step, but
    //               H0's scalars are ready ~0.35 us later (0.76 us into the step) — wins where the pair loop is longer than that
    //   otherwise   they gate B1 with the rows (every thread gathers its share of rows + tiles; B3 follows H0's last store)
    constexpr int rec_stride = NPAD + NPAD / 4;
    constexpr int KT = (MAXT + 63) / 64;          // tiles per H0 lane
    constexpr bool late = LATE;
    const int gthreads = late ? nthreads - 64 : nthreads;         // gathering threads (late: all but H0)
    const int gtid = late ? tid - 64 : tid;
    const int units = late ? m.n : m.n + 2 * m.ntiles;
    const int ku = (units + gthreads - 1) / gthreads;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(rec, 0, (int)(sizeof(u32x4) * 2 * m.nrep_g * rec_stride), 0x00020000);

    // ---- prologue: everything that stays for the whole launch --------------------------------------
    // this replica's state: issued now (the pointers arrived with the second half of the arguments), consumed below while the
    // targets are still coming in
    float vcx = 0.0f, vcy = 0.0f, vcz = 0.0f;     // H0: velocity of this lane's row, carried from step to step
    FireState st;
    float4 cin[2], pin4 = make_float4(0, 0, 0, 0);
    {
        const float* xin = io.xin;
        const float* pin = io.pin;
        const float* vin = io.vin;
#pragma unroll
        for (int k = 0; k < 2; ++k) {             // 3 NPAD floats over the workgroup: at most two float4 per thread (3 NPAD <= 8 x 512)
            const int b = 4 * (tid + k * nthreads);
            cin[k] = b < 3 * NPAD ? *reinterpret_cast<const float4*>(xin + roff + b) : make_float4(0, 0, 0, 0);
        }
        if (tid < m.ntiles) pin4 = reinterpret_cast<const float4*>(pin)[(size_t)rep * m.ntiles + tid];
        if (hfin) { C3D_HROW_INDEX; vcx = vin[ix]; vcy = vin[iy]; vcz = vin[iz]; }
        st = io.sin[rep];
    }
    // chain helpers: pair constants of the left-over columns of "their" rows — pass q of helper `wave` covers rows 8 (wave - 1) +
    // 8 (NH - 1) q + (lane >> 3) of the workgroup, lane & 7 = the column; at most 4 passes (RW <= 64, two helpers)
    float* const lc = lcon + (size_t)((wave >= 1 && wave <= 3) ? wave - 1 : 0) * (4 * 2 * 64);
    if (!is_compute && !is_h0 && wave <= 3 && m.nleft > 0) {
        for (int q = 0; q < 4; ++q) {
            const int k = 8 * (wave - 1) + 8 * (NH - 1) * q + (lane >> 3), c = lane & 7;
            float b = 0.0f, a = 0.0f;
            if (k < RW && c < m.nleft && wg_row0 + k < m.n) {
                const float t = tgt[(size_t)(wg_row0 + k) * NPAD + m.jl0 + c];
                b = t * m.inv_rs;
                a = t > 0.0f ? m.inv_rs : 0.0f;
            }
            lc[(2 * q) * 64 + lane] = b; lc[(2 * q + 1) * 64 + lane] = a;
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int b = 4 * (tid + k * nthreads);
        if (b < 3 * NPAD) *reinterpret_cast<float4*>(smem + b) = cin[k];
    }
    if (tid < m.ntiles) reinterpret_cast<float4*>(ps)[tid] = pin4;
    if (tid < 3 * 64) lbuf[tid] = 0.0f;           // H0 reads it every step; written by the helpers only where left-over columns exist
#ifdef C3D_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t_coords = __builtin_amdgcn_s_memrealtime();
#endif
    // per-pair constants of the clamp form (c3d_step_core.h: pair_b = target / rswitch in registers, pair_a = 1 / rswitch where
    // a restraint exists in wave-private LDS), the same for every step of the launch
    // (device potential 4, the shipped model: row PAIRS per column, the layout of the packed pair term — c3d_step_core.h pair_term2 —, "no
    //  restraint" as a target of 1e30 A, so that the second constant is 1 / mrs for every pair and only an odd last row keeps it in LDS)
    constexpr bool PK = POT == 4;
    float4 tv[PK ? 1 : RPW][NB];
    PairConsts2<PK ? RPW : 1, NB> pc;
    float4* const mw = reinterpret_cast<float4*>(dump + 8) + (size_t)(is_compute ? cwave : 0) * (RPW * NB * 64);
    if constexpr (PK) {
        pair_consts2_build<RPW, NB>(m, traw, is_compute, lane, pc, mw);
    } else {
        DevStep p0{};
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int jb = 0; jb < NB; ++jb) {
                tv[r][jb] = pair_b<false>(m, traw[r][jb]);
                if (is_compute) mw[(r * NB + jb) * 64 + lane] = pair_a<false>(m, p0, traw[r][jb]);
            }
    }
#ifdef C3D_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t_targets = __builtin_amdgcn_s_memrealtime();
#endif
    // gather bookkeeping: gathering thread g takes units g, g + gthreads, ...: row u -> {x, y, z} of bead u; tile unit j = 2 t + h ->
    // three sums (h = 0) or the fourth (h = 1) of tile t; float offsets into smem, words nobody wants go to `dump`
    int gsrc[KUMAX], gda[KUMAX], gdb[KUMAX], gdc[KUMAX];
    const int dump_off = (int)(dump - smem);
#pragma unroll
    for (int k = 0; k < KUMAX; ++k) {
        const int u = min(max(gtid + gthreads * k, 0), units - 1);
        int src, da, db = dump_off, dc = dump_off;
        if (u < m.n) { src = u; da = u; db = NPAD + u; dc = 2 * NPAD + u; }
        else {
            const int j = u - m.n, t = j >> 1;
            src = NPAD + j;
            if ((j & 1) == 0) { da = 3 * NPAD + 4 * t; db = da + 1; dc = da + 2; }
            else da = 3 * NPAD + 4 * t + 3;
        }
        if (gtid < 0 || gtid + gthreads * k >= units) da = db = dc = dump_off;
        gsrc[k] = src; gda[k] = da; gdb[k] = db; gdc[k] = dc;
    }

#ifdef C3D_STAMPS
    const bool stamper = lrep == 0 && part == 0 && is_h0 && lane == 0;
    const bool cstamper = lrep == 0 && part == 0 && cwave == 0 && lane == 0;
    if (stamper) { g_pstamps[0] = t_entry; g_pstamps[1] = __builtin_amdgcn_s_memrealtime(); g_pstamps[6] = t_placed; }
    if (cstamper) { g_pstamps[4] = t_targets; g_pstamps[5] = t_coords; }
#endif
    // One step loop PER ROLE (compute wave, H0, chain helper): the waves of a workgroup keep their role for the whole launch, and
    // written as one loop with role branches inside, every role's registers are live in every other role's code (the compiler
    // cannot know the branches never mix) — the 32 target registers of the compute waves next to H0's row state and the helpers'
    // pair terms overflowed the 128 a wave has and were spilled INSIDE the pair loop.  Three loops, same barriers in the same
    // order: each role's code is allocated on its own.
    auto role_loop = [&](auto role) {
    constexpr int ROLE = decltype(role)::value;     // 0 compute wave, 1 H0, 2 chain helper
    int s = 0;
    StepRun cur = first_run;
    for (int run = run0;; ++run) {                  // left by the `return` of the last step
      if (run != run0) cur = runs[run];
      const DevStep p = cur.p;
      const int count = cur.count;
      const bool needs_partials = p.kind == 0 || p.kind == 1 || p.kind == 2 || (TWO && p.kind == 5);
      for (int it = run == run0 ? skip0 : 0; it < count; ++it, ++s) {
#ifdef C3D_STAMPS
        if (cstamper && s == 0) g_pstamps[7] = __builtin_amdgcn_s_memrealtime();
#endif
        __syncthreads();                            // B1: xs/ys/zs/ps of this step are in LDS
        CSTAMP(0);                                  // step start (H0 past B1)
#ifdef C3D_STAMPS
        if (stamper && s == 0) g_pstamps[2] = __builtin_amdgcn_s_memrealtime();
#endif
        CSTAMP_C(6);                                // compute wave 0 past B1

        float hx0 = 0.0f, hy0 = 0.0f, hz0 = 0.0f;  // H0: position of this lane's row
        StepScalars sc;
        sc.lam = 1.0f; sc.cmx = sc.cmy = sc.cmz = 0.0f; sc.keep = 0.0f; sc.mix = 0.0f;
        if constexpr (ROLE == 0) {
            // ---- K2: pair terms of RPW rows, butterfly sums, three words per row for H0 ------------------
            if (p.kind != 4) {
                float Fx, Fy, Fz;
                if constexpr (PK) tile_pair_sums_pk<RPW, NB, WL>(m, p, pc, mw, xs, ys, zs, row0, lane, Fx, Fy, Fz);
                else tile_pair_sums_reg<POT, RPW, NB, WL, 1>(m, p, tv, mw, xs, ys, zs, row0, lane, Fx, Fy, Fz);
                if (lane < RPW) {
                    const int k = cwave * RPW + lane;
                    fbuf[k] = Fx; fbuf[64 + k] = Fy; fbuf[128 + k] = Fz;
                }
                CSTAMP_C(7);                        // compute wave 0: pair loop + reduce done
            }
        } else if constexpr (ROLE == 1) {
            // ---- replica sums of the previous step -> scalars of this one (only H0 needs them) -----------
            if (p.kind != 2 && !(TWO && p.kind == 5)) { st.dt = fp.dt_start; st.alpha = fp.alpha_start; st.npos = 0; st.pad = 0; }
            float4 psum = make_float4(0, 0, 0, 0);
            if (needs_partials) {
                if (late && !solo && s > 0) {
                    // the tile sums the replica's parts published in the previous step's tail (tag and parity of THIS step's input);
                    // nobody else reads ps.  Same bound and same way out as the row gather below.
                    const unsigned tagp = tag_base + (unsigned)s;
                    const int basep = ((s & 1) * m.nrep_g + lrep) * rec_stride + NPAD;
                    u32x4 a[KT], b[KT];
                    unsigned spins = 0;
                    for (;;) {
                        bool ok = true;
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int k = 0; k < KT; ++k) {
                            const int t = min(lane + 64 * k, m.ntiles - 1);
                            a[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (basep + 2 * t) * 16, 0, 16);
                            b[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (basep + 2 * t + 1) * 16, 0, 16);
                        }
#pragma unroll
                        for (int k = 0; k < KT; ++k) ok &= a[k].x == tagp && b[k].x == tagp;
                        if (__all(ok)) break;
                        __builtin_amdgcn_s_sleep(1);
                        ++spins;
                        if (spins > (1u << 18) || ((spins & 1023u) == 0 && *timeout != 0u)) {
                            if (lane == 0) *timeout = 1u;
                            return;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < KT; ++k)
                        if (lane + 64 * k < m.ntiles)
                            reinterpret_cast<float4*>(ps)[lane + 64 * k] = make_float4(__uint_as_float(a[k].y), __uint_as_float(a[k].z), __uint_as_float(a[k].w), __uint_as_float(b[k].y));
                }
                for (int t = lane; t < m.ntiles; t += 64) {
                    const float4 q = reinterpret_cast<const float4*>(ps)[t];
                    psum.x += q.x; psum.y += q.y; psum.z += q.z; psum.w += q.w;
                }
                psum = wave_sum4(psum);
                // the butterfly's association differs from quad to quad: every row takes lane 0's sums (the per-step kernel's
                // finisher lanes sit in quad 0)
                psum.x = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(psum.x)));
                psum.y = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(psum.y)));
                psum.z = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(psum.z)));
                psum.w = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(psum.w)));
            }
            sc = step_scalars<TWO>(m, p, fp, psum, st);
            // this row's position: read now, the barrier below is long
            if (lane < RW && hrow < NPAD) { hx0 = xs[hrow]; hy0 = ys[hrow]; hz0 = zs[hrow]; }
            CSTAMP(1);                              // H0: sums + scalars done
        } else if (p.kind != 4) {                  // ROLE 2
            // ---- chain terms: lane = (row, neighbour), 16 rows per helper and pass ------------------------
            for (int cb = 16 * (wave - 1); cb < RW; cb += 16 * (NH - 1)) {
                const int k = cb + (lane >> 2);
                float cx, cy, cz;
                chain_term(m, p, xs, ys, zs, wg_row0 + k, lane & 3, k < RW, cx, cy, cz);
                cx = quad_chain_sum(cx); cy = quad_chain_sum(cy); cz = quad_chain_sum(cz);
                if ((lane & 3) == 0 && k < RW) { cbuf[k] = cx; cbuf[64 + k] = cy; cbuf[128 + k] = cz; }
            }
            // ---- left-over columns: lane = (row, column), 8 rows per helper and pass, the row's sum in the tree of leftover_terms ----
            if (m.nleft > 0) {
                const PairK kk = pair_k(m, p);
                for (int q = 0; q < 4; ++q) {
                    const int k = 8 * (wave - 1) + 8 * (NH - 1) * q + (lane >> 3), c = lane & 7;
                    if (8 * (wave - 1) + 8 * (NH - 1) * q < RW) {           // wave-uniform
                        const bool active = k < RW && c < m.nleft && wg_row0 + k < m.n;
                        float lx, ly, lz;
                        leftover_terms<POT, false>(m, p, kk, xs, ys, zs, min(wg_row0 + k, m.n - 1), c, active, lc[(2 * q) * 64 + lane], lc[(2 * q + 1) * 64 + lane], lx, ly, lz);
                        if (c == 0 && k < RW) { lbuf[k] = lx; lbuf[64 + k] = ly; lbuf[128 + k] = lz; }
                    }
                }
            }
        }
        __syncthreads();                            // B2: all LDS reads of this step are done; fbuf / cbuf complete
        CSTAMP(2);                                  // H0 past B2 (all compute waves done)
        const bool last = s + 1 == nsteps;
        const unsigned tag = tag_base + (unsigned)s + 1u;
        const int base = (((s + 1) & 1) * m.nrep_g + lrep) * rec_stride;

        if constexpr (ROLE == 1) {
            // ---- row update, lane k <-> row k of the workgroup ------------------------------------------
            // One copy of the update per step kind (a run has one kind; the switch is scalar): between B2 and the rows' store every
            // instruction of this lone wave costs ~5 ns of the step, and written once with the kind tested inside, the path held some
            // twenty scalar branches and three dependent LDS round trips (the reads sat in different blocks)
            float4 q = make_float4(0, 0, 0, 0);
            float xn = 0.0f, yn = 0.0f, zn = 0.0f;
            auto row_update = [&](auto kc) {
                constexpr int K = decltype(kc)::value;
                DevStep pk = p;
                pk.kind = K;
                if (hfin) {
                    float Fx = 0.0f, Fy = 0.0f, Fz = 0.0f;
                    if constexpr (K != 4) {
                        // nine words, one round trip (lbuf stays zero where there are no left-over columns: read, not used)
                        const float sx0 = fbuf[lane], sy0 = fbuf[64 + lane], sz0 = fbuf[128 + lane];
                        const float lx = lbuf[lane], ly = lbuf[64 + lane], lz = lbuf[128 + lane];
                        const float cx = cbuf[lane], cy = cbuf[64 + lane], cz = cbuf[128 + lane];
                        const bool lo = m.nleft > 0;
                        const float sx = lo ? sx0 + lx : sx0, sy = lo ? sy0 + ly : sy0, sz = lo ? sz0 + lz : sz0;
                        Fx = row_total<false>(pk, sx, cx);        // one explicit fma, as in reduce_and_chain (c3d_step_core.h)
                        Fy = row_total<false>(pk, sy, cy);
                        Fz = row_total<false>(pk, sz, cz);
                    }
                    float vx0 = vcx, vy0 = vcy, vz0 = vcz;
                    if constexpr (K == 3 || K == 6) { vx0 = vy0 = vz0 = 0.0f; }
                    else if constexpr (K == 4) { C3D_HROW_INDEX; const float* vinit = io.vinit; vx0 = vinit[ix]; vy0 = vinit[iy]; vz0 = vinit[iz]; }
                    finish_row(m, pk, fp, sc, st, Fx, Fy, Fz, hx0, hy0, hz0, vx0, vy0, vz0, xn, yn, zn, vcx, vcy, vcz, q);
                    if (!last && !solo) {               // the row's new position leaves at once; the tile sums follow below
                        u32x4 o;
                        o.x = tag; o.y = __float_as_uint(xn); o.z = __float_as_uint(yn); o.w = __float_as_uint(zn);
                        __builtin_amdgcn_raw_buffer_store_b128(o, rsrc, (base + hrow) * 16, 0, 0);     // plain: the line stays in this XCD's L2
                    }
                } else {
                    xn = hx0; yn = hy0; zn = hz0;       // padding row (one-workgroup replicas write it back as it is)
                }
            };
            bool two_point_step = false;
            if constexpr (TWO) {                            // (the two-point minimiser's kinds: k_cluster_tp only)
                two_point_step = p.kind >= 5;
                if (p.kind == 5) row_update(std::integral_constant<int, 5>{});
                else if (p.kind == 6) row_update(std::integral_constant<int, 6>{});
            }
            if (!two_point_step) switch (p.kind) {
                case 0: row_update(std::integral_constant<int, 0>{}); break;
                case 1: row_update(std::integral_constant<int, 1>{}); break;
                case 2: row_update(std::integral_constant<int, 2>{}); break;
                case 3: row_update(std::integral_constant<int, 3>{}); break;
                default: row_update(std::integral_constant<int, 4>{}); break;
            }
            // B3 (see the other roles below): the rows are out; what follows here — tile sums, their two units — is wanted by the
            // H0s of the replica only, well after the next step has started, and runs while the other waves gather rows
            if (late && !last && !solo) { CSTAMP(4); __builtin_amdgcn_s_barrier(); }      // bare: no wait for the stores' acknowledgement
            // tile sums, the tree of tile_sum8 over eight consecutive lanes: lane 8 t ends with tile t's four sums
            float4 t = q;
            t.x += dpp_mov<0xB1>(t.x); t.y += dpp_mov<0xB1>(t.y); t.z += dpp_mov<0xB1>(t.z); t.w += dpp_mov<0xB1>(t.w);
            t.x += dpp_mov<0x4E>(t.x); t.y += dpp_mov<0x4E>(t.y); t.z += dpp_mov<0x4E>(t.z); t.w += dpp_mov<0x4E>(t.w);
            t.x += dpp_mov<0x12C>(t.x); t.y += dpp_mov<0x12C>(t.y); t.z += dpp_mov<0x12C>(t.z); t.w += dpp_mov<0x12C>(t.w);   // row_ror:12 = lane + 4
            CSTAMP(3);                              // H0: tile sums done
            if (last) {                             // hand the state back to the ordinary buffers
                if (hfin) {
                    C3D_HROW_INDEX;
                    float* xout = io.xout;
                    float* vout = io.vout;
                    xout[ix] = xn; xout[iy] = yn; xout[iz] = zn;
                    vout[ix] = vcx; vout[iy] = vcy; vout[iz] = vcz;
                }
                const int tl = (wg_row0 + lane) >> 3;
                if ((lane & 7) == 0 && lane < RW && tl < m.ntiles) reinterpret_cast<float4*>(io.pout)[(size_t)rep * m.ntiles + tl] = t;
                if (lane == 0 && part == 0) io.sout[rep] = st;
                // completion: claim[8] counts the workgroups that reached their last step; the one that completes the launch's
                // `expected` = replicas x parts says so in the host-mapped word next to *timeout.  A launch in which some
                // (replica, part) was never claimed (fewer workgroups on an XCD than the plan assumes) or abandoned never
                // writes it, and the host re-runs the range on the per-step path instead of accepting stale state.
                if (lane == 0 && atomicAdd(&claim[8], 1u) + 1u == expected) {
                    const unsigned seen = sp_mode ? atomicOr(&claim[9], 0u) : 1u;       // XCD offsets the workgroups saw
                    if ((seen & (seen - 1u)) == 0u) timeout[1] = tag_base | 1u;
                    else timeout[2] = 1u;
                }
#ifdef C3D_STAMPS
                if (stamper) g_pstamps[3] = __builtin_amdgcn_s_memrealtime();
#endif
                return;
            }
            if (solo) {
                // one workgroup owns the replica: new positions and tile sums go straight back into LDS
                if (lane < RW && hrow < NPAD) { xs[hrow] = xn; ys[hrow] = yn; zs[hrow] = zn; }
                if ((lane & 7) == 0 && lane < RW) reinterpret_cast<float4*>(ps)[lane >> 3] = t;
            } else if ((lane & 7) == 0 && lane < RW && ((wg_row0 + lane) >> 3) < m.ntiles) {
                // publish the tile's four sums (lane 8 t holds them)
                u32x4 o0, o1;
                o0.x = tag; o0.y = __float_as_uint(t.x); o0.z = __float_as_uint(t.y); o0.w = __float_as_uint(t.z);
                o1.x = tag; o1.y = __float_as_uint(t.w); o1.z = 0u; o1.w = 0u;
                const int u0 = base + NPAD + 2 * ((wg_row0 + lane) >> 3);
                __builtin_amdgcn_raw_buffer_store_b128(o0, rsrc, u0 * 16, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(o1, rsrc, (u0 + 1) * 16, 0, 0);
            }
            CSTAMP(5);                              // H0: tile units stored
            if (solo || late) continue;             // late: H0 has passed B3 and gathers nothing
        }
        {
            if (last) return;
            if (solo) continue;
            // B3: nobody polls before this workgroup's own rows are out: no gather can finish earlier than that, and the
            // spinning waves cost H0 issue slots (measured: 4.43 us per step with it, 4.53 without).  A timing device only — no LDS or
            // global access is ordered by it — hence the bare instruction: __syncthreads() would hold H0 until its stores are
            // acknowledged (~0.2 us) before anybody may look for them
            __builtin_amdgcn_s_barrier();
            // ---- gather the replica's units of step s+1 into LDS: re-read until every tag matches ----------
            // (one sweep at a time, no pause before the first: two sweeps in flight did not help, and every 64-clock pause before the
            //  first sweep adds its own length to the step — the workgroups of a replica run in lock-step, the first sweep is the one)
            u32x4 v[KUMAX];
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
                asm volatile("" ::: "memory");     // the loads below must be re-issued on every sweep
#pragma unroll
                for (int k = 0; k < KUMAX; ++k)
                    if (k < ku) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (base + gsrc[k]) * 16, 0, 16);   // aux 16 = sc1: bypass L1
#pragma unroll
                for (int k = 0; k < KUMAX; ++k)
                    if (k < ku) ok &= v[k].x == tag;
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
                ++spins;
                // ~0.3 s: the workgroups of this replica are not all resident; or another workgroup has already given up
                if (spins > (1u << 18) || ((spins & 1023u) == 0 && *timeout != 0u)) {
                    if (lane == 0) *timeout = 1u;
                    return;
                }
            }
#pragma unroll
            for (int k = 0; k < KUMAX; ++k)
                if (k < ku) { smem[gda[k]] = __uint_as_float(v[k].y); smem[gdb[k]] = __uint_as_float(v[k].z); smem[gdc[k]] = __uint_as_float(v[k].w); }
        }
      }
    }
    };
    if (is_compute) role_loop(std::integral_constant<int, 0>{});
    else if (is_h0) role_loop(std::integral_constant<int, 1>{});
    else role_loop(std::integral_constant<int, 2>{});
}

#define C3D_CLUSTER_ARGS                                                                                                                  \
    const float* __restrict__ tgt, unsigned* __restrict__ claim, const StepRun* __restrict__ runs, const int P, const int CW, const int NH, \
    const int static_place, const int nbeads, const int run0, const int skip0, const int nsteps, const unsigned tag_base,                 \
    u32x4* __restrict__ rec, volatile unsigned* __restrict__ timeout, const unsigned expected, const AnnealIO io, const DevModel m, const DevFire fp
#define C3D_CLUSTER_PASS tgt, claim, runs, P, CW, NH, static_place, nbeads, run0, skip0, nsteps, tag_base, rec, timeout, expected, io, m, fp
// MD and FIRE steps (every launch of the hot and cool stages, the regularisation, FIRE's part of the final stage)
template <int POT, int RPW, int NB, int WL, bool LATE>
__global__ __launch_bounds__(kClMaxThreads) void k_cluster(C3D_CLUSTER_ARGS) { cluster_body<POT, RPW, NB, WL, LATE, false>(C3D_CLUSTER_PASS); }
// ranges that hold two-point minimiser steps (the first part of the final stage)
template <int POT, int RPW, int NB, int WL, bool LATE>
__global__ __launch_bounds__(kClMaxThreads) void k_cluster_tp(C3D_CLUSTER_ARGS) { cluster_body<POT, RPW, NB, WL, LATE, true>(C3D_CLUSTER_PASS); }
#undef C3D_CLUSTER_ARGS
#undef C3D_CLUSTER_PASS

// Built as six translation units (Makefile): this file once with -DC3D_CLUSTER_SPLIT — planner, dispatch, k_tear16, no instantiation of
// k_cluster — and once per device potential with -DC3D_CLUSTER_POT=0..4, each carrying that potential's geometries alone.  The 5 MB code
// object of all 200 instantiations took 9-13 ms to load at the first launch of a process and over two minutes to compile; a job loads
// the unit of its potential only (the shipped one: 86 kernels).  Without either macro the file is the single unit it used to be
// (tools/stamps/build.sh).
constexpr bool cluster_late_ok(int pot, int rpw, int nb, int wl) { return pot >= 3 && rpw * (4 * (nb - 1) + wl) >= 6; }
#ifndef C3D_CLUSTER_POT
#ifdef C3D_STAMPS
hipError_t read_cluster_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cstamps), sizeof(unsigned long long) * 64 * 8); }
hipError_t read_cluster_pstamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pstamps), sizeof(unsigned long long) * 8); }
#endif

// ---- the hand-off's hardware assumption, watched (c3d_debug_tear16) -----------------------------------------
// One producer workgroup rewrites 1024 units {i, i, i, i}, i = 1 .. iters, with the plain 16-byte store of the row hand-off; every other
// workgroup (grid = one per CU: consumers on the producer's XCD and on all others) re-reads them with the sc1 16-byte load of the gather
// and counts units whose four words differ.  stats: [0] unit reads, [1] torn units, [2] reads that saw a new value.
__global__ __launch_bounds__(kClMaxThreads) void k_tear16(u32x4* buf, unsigned* stop, unsigned long long* stats, int iters) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(buf, 0, kClMaxThreads * 16, 0x00020000);
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        for (int i = 1; i <= iters; ++i) {
            u32x4 v; v.x = v.y = v.z = v.w = (unsigned)i;
            __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, tid * 16, 0, 0);
        }
        __threadfence();
        if (tid == 0) atomicExch(stop, 1u);
        return;
    }
    const auto ssrc = __builtin_amdgcn_make_buffer_rsrc(stop, 0, 4, 0x00020000);
    unsigned long long reads = 0, torn = 0, fresh = 0;
    unsigned lastv = 0;
    for (unsigned guard = 0; guard < (1u << 26); ++guard) {                // bounded whatever happens to the producer
        asm volatile("" ::: "memory");
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, tid * 16, 0, 16);
        ++reads;
        if (!(v.x == v.y && v.y == v.z && v.z == v.w)) ++torn;
        if (v.x != lastv) { ++fresh; lastv = v.x; }
        if ((reads & 255) == 0 && __builtin_amdgcn_raw_buffer_load_b32(ssrc, 0, 0, 16) != 0u) break;
    }
    atomicAdd(&stats[0], reads); atomicAdd(&stats[1], torn); atomicAdd(&stats[2], fresh);
}
hipError_t launch_tear16(int num_cus, void* buf, unsigned* stop, unsigned long long* stats, int iters, hipStream_t s) {
    hipLaunchKernelGGL(k_tear16, dim3(num_cus), dim3(kClMaxThreads), 0, s, reinterpret_cast<u32x4*>(buf), stop, stats, iters);
    return hipGetLastError();
}

// ---- host side ---------------------------------------------------------------------------------------
// Geometry.  A workgroup = CW compute waves x RPW rows (RW = CW * RPW rows, a multiple of 8, at most 64) + 2 or 4 helpers;
// a replica = P = ceil(n / RW) workgroups; the replicas of the fullest XCD must fit its CUs, one or two workgroups per CU.
// Among the geometries that fit, the cheapest by an instruction-count model of one step (VALU issue is the limiter):
//   per SIMD: ceil(CW / 4) compute waves x (column slots x instructions per slot of RPW rows + 60) wave-instructions,
//   + H0's serial tail (~200) + the hand-off (140 + 0.92 n instruction-times: 0.8 us at n = 455) unless P == 1.
// Tile sums late (H0 fetches them after B1, LATE = true) where the pair loop of a compute wave — rows x column slots — outlasts
// H0's later scalars; measured on 13 problems, N = 76 .. 455 (profiles/r03_late_tiles_ab.txt).  Shipped potential only: every
// (geometry, LATE) pair is one more kernel to compile.

bool cluster_plan(const DevModel& m, int num_cus, int num_xcc, int forced_geom, int forced_late, int xcd_count, ClusterPlan* plan) {
    // the placement arithmetic (replica r on XCD r % 8, XCC_ID & 7) is written for the 8 XCDs of an unpartitioned MI355X:
    // on a partitioned or CU-masked device the cluster kernel is not used at all
    if (m.npad > 1024 || num_xcc != 8 || num_cus < 8 || num_cus % 8) return false;
    const int cus_per_xcd = num_cus / 8;
    if (xcd_count < 1 || xcd_count > 8) return false;
    const int per_xcd = (m.nrep_g + xcd_count - 1) / xcd_count;      // replicas on the fullest XCD of the launch's set
    const int nb = m.npad / 256;
    // {compute waves, rows per wave, helper waves, workgroups per CU}
    static const int geoms[][4] = {{8, 1, 4, 1}, {8, 2, 4, 1}, {12, 2, 4, 1}, {8, 4, 4, 1}, {10, 4, 4, 1}, {12, 4, 4, 1}, {8, 3, 4, 1},
                                   {4, 2, 4, 1}, {4, 4, 4, 1}, {6, 4, 4, 1}};
    // (two 512-thread workgroups per CU — {6, 4, 2, 2} and the like, meant to overlap one replica's serial phase with
    //  another's pair loop — measured slower at every size: twice the records per replica, twice the helper work per CU)
    double best = 1e30;
    bool found = false;
    // measurement knob (c3d_set_option "cluster_geometry" = 100 CW + 10 RPW + helpers, e.g. 1244): only that geometry
    const int fcw = forced_geom / 100, frpw = forced_geom / 10 % 10, fnh = forced_geom % 10;
    for (const auto& g : geoms) {
        const int cw = g[0], rpw = g[1], nh = g[2], wpc = g[3], rw = cw * rpw;
        if (fcw && (cw != fcw || rpw != frpw || nh != fnh)) continue;
        if (rw % 8 || rw > 64) continue;
        if (rpw * nb > 8) continue;               // targets in registers: rpw * nb float4 per lane
        if (m.noe_pot < 3 && (m.wl != 4 || m.nleft != 0)) continue;     // narrow last blocks and left-over columns: the CNS soft-square (3, 4) only
        if (m.nleft > 0 && (nh != 4 || (rw + 8 * (nh - 1) - 1) / (8 * (nh - 1)) > 4)) continue;       // left-over passes per chain helper (three of them)
        const int P = (m.n + rw - 1) / rw;
        if (per_xcd * P > cus_per_xcd * wpc) continue;
        const int threads = (cw + nh) * 64;
        const int kumax = nb > 2 ? 3 : 2;
        // tile sums late (H0 fetches them after B1) where the pair loop outlasts H0's later scalars: rows x column slots of a compute wave
        // (round 4: what has to outlast H0's later scalars is the pair loop of a SIMD, compute waves x rows x column slots — with the packed
        //  pair term a lone compute wave per SIMD is through 8 row-slots before them: profiles/r04_late_tiles_ab.txt)
        const int late = P > 1 && forced_late != 0 && cluster_late_ok(m.noe_pot, rpw, nb, m.wl) &&
                         (m.noe_pot != 4 || ((cw + 3) / 4) * rpw * (4 * (nb - 1) + m.wl) >= 12);
        if ((late ? m.n : m.n + 2 * ((m.n + 7) / 8)) > (late ? threads - 64 : threads) * kumax) continue;      // gather: units per thread
        // coordinates, sums, row buffers + the compute waves' NOE weights
        size_t lds = sizeof(float) * (3 * m.npad + 4 * (m.npad / 8) + 9 * 64 + 24 * 64 + 16) + (size_t)cw * rpw * nb * 64 * 16;
        if (wpc == 1) { if (lds < 84 * 1024) lds = 84 * 1024; }    // more than half of a CU's 160 KB: one workgroup per CU
        else if (lds > 78 * 1024 || threads > 512) continue;        // two per CU must fit
        const int wps = (cw * wpc + 3) / 4;       // compute waves per SIMD; a lone wave issues at ~0.6 of the multi-wave rate
        // instructions per column slot of a compute wave: 15 per row (clamp forms); device potential 4: 22 per row PAIR in the packed
        // form (c3d_step_core.h pair_term2), 19 for an odd last row
        const double per_slot = m.noe_pot == 4 ? (rpw / 2) * 22.0 + (rpw & 1) * 19.0 : rpw * 15.0;
        const double valu = (wps == 1 ? 1.6 : (wps == 2 ? 1.15 : 1.0)) * wps * ((4 * (nb - 1) + m.wl) * per_slot + 60.0);
        // the hand-off grows with the units gathered (n rows): ~560 instruction-times at n = 455, ~170 at n = 37 (where five small parts of a
        // replica beat one large workgroup: 1.33 against 1.40 us per step, profiles/r04_cluster_geometry_sweep.txt)
        const double serial = 200.0 + (P > 1 ? 140.0 + 0.92 * m.n : 0.0);
        // one workgroup per CU: the serial tail of a step follows its pair loop; two per CU: the tails hide behind the
        // other workgroup's loop where there is one
        const double cost = wpc == 1 ? valu + serial : (valu > serial ? valu + 0.35 * serial : 0.5 * valu + serial);
        if (cost < best) {
            best = cost; found = true;
            plan->rpw = rpw; plan->cw = cw; plan->helpers = nh; plan->wgs_per_cu = wpc; plan->parts = P; plan->per_xcd = per_xcd;
            plan->grid = num_cus * wpc; plan->threads = threads; plan->units = 2 * rw; plan->device = 0; plan->lds = lds;
            plan->expected = (unsigned)(m.nrep_g * P);
            plan->late_tiles = late;
            plan->xcd_count = xcd_count;
        }
    }
    return found;
}

// two parities x replicas x (one 16-byte unit per row + two per 8-row tile)
size_t cluster_record_bytes(const DevModel& m, const ClusterPlan& pl) { (void)pl; return (size_t)2 * m.nrep_g * (m.npad + m.npad / 4) * 16; }

#endif  // !C3D_CLUSTER_POT

template <int POT, int RPW, int NB, int WL, bool LATE>
static hipError_t cluster_go(const DevModel& m, const DevFire& fp, const ClusterPlan& pl, const AnnealIO& io, const float* tgt, void* rec,
                             const StepRun* runs, int run0, int skip0, int nsteps, unsigned tag_base, unsigned* timeout, unsigned* claim,
                             hipStream_t s) {
    // which of the two kernels this translation unit holds: a unit built with -DC3D_CLUSTER_TP carries k_cluster_tp (ranges that hold
    // two-point minimiser steps) and nothing else, the others k_cluster alone; the single-unit build (neither split macro) holds both.
    // Nothing is set up here: the unit was loaded, and every instantiation given its dynamic LDS size, by cluster_prepare_unit below —
    // before this process's first launch on the device, under the loader's exclusive lock (c3d_api.cpp "code objects")
#if defined(C3D_CLUSTER_TP)
    constexpr bool kHasLean = false, kHasTp = true;
#elif defined(C3D_CLUSTER_POT)
    constexpr bool kHasLean = true, kHasTp = false;
#else
    constexpr bool kHasLean = true, kHasTp = true;
#endif
    const int tp = pl.two_point ? 1 : 0;         // the range holds two-point minimiser steps: the kernel that carries their row update
    if ((tp && !kHasTp) || (!tp && !kHasLean)) return hipErrorInvalidValue;
    const int sp = (pl.static_place & 0xff) | ((pl.xcd_base & 0xff) << 8) | ((pl.xcd_count & 0xff) << 16);      // decoded at the kernel's top
#define C3D_CL_LAUNCH(K)                                                                                                                              \
    do {                                                                                                                                              \
        if (pl.t0 && pl.t1)                                                                                                                           \
            hipExtLaunchKernelGGL((K<POT, RPW, NB, WL, LATE>), dim3(pl.grid), dim3(pl.threads), pl.lds, s, pl.t0, pl.t1, 0, tgt, claim, runs,          \
                                  pl.parts, pl.cw, pl.helpers, sp, m.n, run0, skip0, nsteps, tag_base, reinterpret_cast<u32x4*>(rec), timeout, pl.expected, io, m, fp); \
        else                                                                                                                                          \
            hipLaunchKernelGGL((K<POT, RPW, NB, WL, LATE>), dim3(pl.grid), dim3(pl.threads), pl.lds, s, tgt, claim, runs, pl.parts, pl.cw, pl.helpers, \
                               sp, m.n, run0, skip0, nsteps, tag_base, reinterpret_cast<u32x4*>(rec), timeout, pl.expected, io, m, fp);                \
    } while (0)
    if constexpr (kHasTp) { if (tp) C3D_CL_LAUNCH(k_cluster_tp); }
    if constexpr (kHasLean) { if (!tp) C3D_CL_LAUNCH(k_cluster); }
#undef C3D_CL_LAUNCH
    return hipGetLastError();
}
template <int POT>
static hipError_t cluster_geom(const DevModel& m, const DevFire& fp, const ClusterPlan& pl, const AnnealIO& io, const float* tgt, void* rec,
                               const StepRun* runs, int run0, int skip0, int nsteps, unsigned tag_base, unsigned* timeout, unsigned* claim,
                               hipStream_t s) {
#define C3D_GO(R, B, W)                                                                                                 \
    do {                                                                                                                \
        if constexpr (cluster_late_ok(POT, R, B, W)) {                                                                  \
            if (pl.late_tiles) return cluster_go<POT, R, B, W, true>(m, fp, pl, io, tgt, rec, runs, run0, skip0, nsteps, tag_base, timeout, claim, s); \
        }                                                                                                               \
        if (pl.late_tiles) return hipErrorInvalidValue;                                                                 \
        return cluster_go<POT, R, B, W, false>(m, fp, pl, io, tgt, rec, runs, run0, skip0, nsteps, tag_base, timeout, claim, s); \
    } while (0)
#define C3D_CL(R, B)                                                                                                    \
    if (pl.rpw == R && m.npad == 256 * B) {                                                                             \
        if (m.wl == 4) C3D_GO(R, B, 4);                                                                                 \
        if constexpr (POT >= 3) {   /* narrower last blocks: the CNS soft-square only (cluster_plan refuses the others)    */      \
            if (m.wl == 3) C3D_GO(R, B, 3);                                                                             \
            if (m.wl == 2) C3D_GO(R, B, 2);                                                                             \
            if (m.wl == 1) C3D_GO(R, B, 1);                                                                             \
        }                                                                                                               \
        return hipErrorInvalidValue;                                                                                    \
    }
    C3D_CL(1, 1); C3D_CL(1, 2); C3D_CL(1, 3); C3D_CL(1, 4);
    C3D_CL(2, 1); C3D_CL(2, 2); C3D_CL(2, 3); C3D_CL(2, 4);
    C3D_CL(3, 1); C3D_CL(3, 2);
    C3D_CL(4, 1); C3D_CL(4, 2);
#undef C3D_CL
#undef C3D_GO
    return hipErrorInvalidValue;
}

// Loads the code object that holds this unit's kernels on the CURRENT device and allows every instantiation in it the dynamic LDS a
// launch may ask for (more than the 64 KB a kernel gets by default; hipFuncSetAttribute is per function and device).  Called once per
// (unit, device) by the loader of c3d_api.cpp while it holds its lock exclusively: after it, a launch from this unit changes no state of
// the runtime.  Walks the table cluster_geom dispatches over.
template <int POT, bool TP>
static hipError_t cluster_prepare_unit() {
    hipError_t e = hipSuccess;
#define C3D_PREP1(R, B, W, L)                                                                                            \
    do {                                                                                                                 \
        if (e == hipSuccess) {                                                                                           \
            if constexpr (TP) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cluster_tp<POT, R, B, W, L>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            else e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cluster<POT, R, B, W, L>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        }                                                                                                                \
    } while (0)
#define C3D_PREP_W(R, B, W)                                                                                              \
    do {                                                                                                                 \
        C3D_PREP1(R, B, W, false);                                                                                       \
        if constexpr (cluster_late_ok(POT, R, B, W)) C3D_PREP1(R, B, W, true);                                           \
    } while (0)
#define C3D_PREP(R, B)                                                                                                   \
    do {                                                                                                                 \
        C3D_PREP_W(R, B, 4);                                                                                             \
        if constexpr (POT >= 3) { C3D_PREP_W(R, B, 3); C3D_PREP_W(R, B, 2); C3D_PREP_W(R, B, 1); }                      \
    } while (0)
    C3D_PREP(1, 1); C3D_PREP(1, 2); C3D_PREP(1, 3); C3D_PREP(1, 4);
    C3D_PREP(2, 1); C3D_PREP(2, 2); C3D_PREP(2, 3); C3D_PREP(2, 4);
    C3D_PREP(3, 1); C3D_PREP(3, 2);
    C3D_PREP(4, 1); C3D_PREP(4, 2);
#undef C3D_PREP
#undef C3D_PREP_W
#undef C3D_PREP1
    return e;
}

#define C3D_CL_CAT2(a, b) a##b
#define C3D_CL_CAT(a, b) C3D_CL_CAT2(a, b)
#if defined(C3D_CLUSTER_POT) && defined(C3D_CLUSTER_TP)
// one potential's unit of k_cluster_tp
hipError_t C3D_CL_CAT(launch_cluster_tp_pot, C3D_CLUSTER_POT)(const DevModel& m, const DevFire& fp, const ClusterPlan& pl, const AnnealIO& io, const float* tgt, void* rec,
                          const StepRun* runs, int run0, int skip0, int nsteps, unsigned tag_base, unsigned* timeout, unsigned* claim,
                          hipStream_t s) {
    return cluster_geom<C3D_CLUSTER_POT>(m, fp, pl, io, tgt, rec, runs, run0, skip0, nsteps, tag_base, timeout, claim, s);
}
hipError_t C3D_CL_CAT(preload_cluster_tp_pot, C3D_CLUSTER_POT)() { return cluster_prepare_unit<C3D_CLUSTER_POT, true>(); }
#elif defined(C3D_CLUSTER_POT)
// one potential's unit: its dispatch and its preparation (load + LDS attribute of every kernel in it)
hipError_t C3D_CL_CAT(launch_cluster_pot, C3D_CLUSTER_POT)(const DevModel& m, const DevFire& fp, const ClusterPlan& pl, const AnnealIO& io, const float* tgt, void* rec,
                          const StepRun* runs, int run0, int skip0, int nsteps, unsigned tag_base, unsigned* timeout, unsigned* claim,
                          hipStream_t s) {
    return cluster_geom<C3D_CLUSTER_POT>(m, fp, pl, io, tgt, rec, runs, run0, skip0, nsteps, tag_base, timeout, claim, s);
}
hipError_t C3D_CL_CAT(preload_cluster_pot, C3D_CLUSTER_POT)() { return cluster_prepare_unit<C3D_CLUSTER_POT, false>(); }
#else
#if defined(C3D_CLUSTER_SPLIT)
#define C3D_CL_DECL(P) hipError_t launch_cluster_pot##P(const DevModel& m, const DevFire& fp, const ClusterPlan& pl, const AnnealIO& io, const float* tgt, void* rec, const StepRun* runs, int run0, int skip0, int nsteps, unsigned tag_base, unsigned* timeout, unsigned* claim, hipStream_t s); hipError_t preload_cluster_pot##P();
C3D_CL_DECL(0) C3D_CL_DECL(1) C3D_CL_DECL(2) C3D_CL_DECL(3) C3D_CL_DECL(4)
#undef C3D_CL_DECL
#define C3D_CL_DECL(P) hipError_t launch_cluster_tp_pot##P(const DevModel& m, const DevFire& fp, const ClusterPlan& pl, const AnnealIO& io, const float* tgt, void* rec, const StepRun* runs, int run0, int skip0, int nsteps, unsigned tag_base, unsigned* timeout, unsigned* claim, hipStream_t s); hipError_t preload_cluster_tp_pot##P();
C3D_CL_DECL(0) C3D_CL_DECL(1) C3D_CL_DECL(2) C3D_CL_DECL(3) C3D_CL_DECL(4)
#undef C3D_CL_DECL
#define C3D_CL_POT(P) (pl.two_point ? launch_cluster_tp_pot##P(m, fp, pl, io, tgt, rec, runs, run0, skip0, nsteps, tag_base, timeout, claim, s) : launch_cluster_pot##P(m, fp, pl, io, tgt, rec, runs, run0, skip0, nsteps, tag_base, timeout, claim, s))
#else
#define C3D_CL_POT(P) cluster_geom<P>(m, fp, pl, io, tgt, rec, runs, run0, skip0, nsteps, tag_base, timeout, claim, s)
#endif
hipError_t launch_cluster(const DevModel& m, const DevFire& fp, const ClusterPlan& pl, const AnnealIO& io, const float* tgt, void* rec,
                          const StepRun* runs, int run0, int skip0, int nsteps, unsigned tag_base, unsigned* timeout, unsigned* claim,
                          hipStream_t s) {
    switch (m.noe_pot) {
        case 0: return C3D_CL_POT(0);
        case 1: return C3D_CL_POT(1);
        case 3: return C3D_CL_POT(3);
        case 4: return C3D_CL_POT(4);
        default: return C3D_CL_POT(2);
    }
}
#undef C3D_CL_POT

AnnealIO anneal_io(const DevBuffers& b, int parity) {
    const int q = parity ^ 1;
    AnnealIO io;
    io.pin = b.P[parity]; io.xin = b.X[parity]; io.vin = b.V[parity]; io.vinit = b.Vinit; io.sin = b.S[parity];
    io.xout = b.X[q]; io.vout = b.V[q]; io.pout = b.P[q]; io.sout = b.S[q];
    return io;
}

// the unit that holds the multi-step kernels of device potential `pot` — k_cluster (two_point false) or k_cluster_tp —, loaded and
// prepared on the current device (cluster_prepare_unit); pot outside 0..4 is an error
hipError_t preload_cluster_unit(int pot, bool two_point) {
#if defined(C3D_CLUSTER_SPLIT)
    switch (pot) {
        case 0: return two_point ? preload_cluster_tp_pot0() : preload_cluster_pot0();
        case 1: return two_point ? preload_cluster_tp_pot1() : preload_cluster_pot1();
        case 2: return two_point ? preload_cluster_tp_pot2() : preload_cluster_pot2();
        case 3: return two_point ? preload_cluster_tp_pot3() : preload_cluster_pot3();
        case 4: return two_point ? preload_cluster_tp_pot4() : preload_cluster_pot4();
        default: return hipErrorInvalidValue;
    }
#else
    switch (pot) {
        case 0: return two_point ? cluster_prepare_unit<0, true>() : cluster_prepare_unit<0, false>();
        case 1: return two_point ? cluster_prepare_unit<1, true>() : cluster_prepare_unit<1, false>();
        case 2: return two_point ? cluster_prepare_unit<2, true>() : cluster_prepare_unit<2, false>();
        case 3: return two_point ? cluster_prepare_unit<3, true>() : cluster_prepare_unit<3, false>();
        case 4: return two_point ? cluster_prepare_unit<4, true>() : cluster_prepare_unit<4, false>();
        default: return hipErrorInvalidValue;
    }
#endif
}
// the planner's own unit: k_tear16 (c3d_debug_tear16) and nothing else in the split build
hipError_t preload_cluster_base_unit() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_tear16));
}
#endif  // C3D_CLUSTER_POT

}  // namespace c3d
