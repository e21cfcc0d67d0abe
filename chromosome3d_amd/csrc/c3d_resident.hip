// c3d_resident.hip — K3/K4 as ONE launch for many SA steps (gfx950).
//
// The per-step kernel (k_step, c3d_device.hip) pays a kernel boundary, a cold L2 and a dispatch ramp on
// every one of the ~5000 dependent steps of an annealing run; at N ~ 500 that is most of a step.  Here the
// workgroups stay resident for a whole range of steps: workgroup (replica, tile) keeps its 8 rows of the
// target matrix in REGISTERS (8 * npad / 256 = 16 VGPRs per lane at N <= 512), its rows' velocities and the
// FIRE state in registers, and exchanges only what the other tiles of the SAME replica need for the next
// step — 8 new positions and the tile's four partial sums, one 256-byte record — through HBM-side memory:
//
//   publish   wave 0 writes the record as sixteen 16-byte units {tag, v, tag, v} with ONE write-through
//             (sc1) store instruction; tag = step + 1, two buffers alternate by step parity
//   gather    every wave re-reads its share of the replica's ntiles records with sc1 loads (they bypass the
//             CU's L1) until every tag matches: the data is its own flag, no counter, no fence
//             (the {tag, value} granule hand-off of the CDNA4 guide; replicas never wait for each other)
//
// Safety of two buffers: a tile publishes step s+1 only after it has read every record of step s, and the
// record it overwrites (step s-1) was read by all tiles before they could publish step s.
// The arithmetic is c3d_step_core.h, the same functions in the same order as k_step: a resident range and the
// same range run step by step give bit-identical coordinates (tests/test_gpu_parity.py).
// Every spin is bounded: a tile that waits ~0.3 s sets *timeout and leaves; the host then runs the same steps on the per-step path.
#include "c3d_step_core.h"

namespace c3d {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kRecUnits = 16;                 // 16-byte units per tile record
constexpr int kRecValues = 2 * kRecUnits;     // x[8] y[8] z[8] partial[4] pad[4]
static_assert(kTileRows == 8, "record layout assumes 8 rows per tile");

// Diagnostic build only (-DC3D_STAMPS, tools/stamps): cycles per phase of workgroup 0, summed over the steps of
// a launch by its thread 0 (s_memtime), plus the number of gather sweeps.
#ifdef C3D_STAMPS
__device__ unsigned long long g_rstamps[16];
#define RSTAMP(k)                                                                                       \
    do {                                                                                                \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                                      \
            const unsigned long long t_ = __builtin_readcyclecounter();                                 \
            racc[k] += t_ - rlast[0];                                                                   \
            rlast[0] = t_;                                                                              \
        }                                                                                               \
    } while (0)
#else
#define RSTAMP(k) do { } while (0)
#endif

// Five waves per SIMD (20 replicas x 57 tiles x 4 waves on 1024 SIMDs) means <= 96 VGPRs and no SGPR to waste:
// the ten state pointers are read from `io` where they are needed (start / end of the launch), not held in
// registers across the step loop.
// the same pointer, but opaque to the optimiser: loads through it are issued where they are written
__device__ __forceinline__ const AnnealIO* late_pointers(const AnnealIO* io) {
    asm volatile("" : "+s"(io));
    return io;
}

template <int POT, bool GEN, int NB>
__global__ __launch_bounds__(256, 5) void k_anneal(
    const AnnealIO* __restrict__ io, const float* __restrict__ tgt, u32x4* __restrict__ rec,
    const StepRun* __restrict__ runs, const int nruns, const int nsteps, unsigned* __restrict__ timeout, const DevModel m,
    const DevFire fp) {
    constexpr int RPW = 2, WAVES = kTileRows / RPW, BLOCK = 64 * WAVES;
    constexpr int NPAD = 256 * NB;
    constexpr int MAXT = NPAD / kTileRows;    // most tiles a replica can have at this NB
    constexpr int KU = (MAXT * kRecUnits) / BLOCK;   // gather loads per thread (= 2 * NB)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* ys = smem + NPAD;
    float* zs = smem + 2 * NPAD;
    float* ps = smem + 3 * NPAD;              // [MAXT][4] per-tile sums of the previous step
    float* stage = ps + 4 * MAXT;             // [32] this tile's record under construction
    float* wpart = stage + kRecValues;        // [WAVES][4]
#ifdef C3D_STAMPS
    unsigned long long* racc = reinterpret_cast<unsigned long long*>(wpart + 4 * WAVES + 4);   // [10]
    unsigned long long* rlast = racc + 10;
    if (blockIdx.x == 0 && threadIdx.x == 0) { for (int k = 0; k < 10; ++k) racc[k] = 0; rlast[0] = __builtin_readcyclecounter(); }
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rep = blockIdx.x % m.nrep, tile = blockIdx.x / m.nrep;
    const size_t roff = (size_t)rep * 3 * NPAD;
    const int row0 = tile * kTileRows + wave * RPW;
    const int row = row0 + (lane & (RPW - 1));    // the row this lane finishes (lanes < RPW only)
    const bool finisher = lane < RPW && row < m.n;
    const size_t ix = roff + row, iy = ix + NPAD, iz = iy + NPAD;
    const int units = m.ntiles * kRecUnits;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(rec, 0, (int)(sizeof(u32x4) * 2 * m.nrep * units), 0x00020000);

    // ---- prologue: everything that stays for the whole launch ----------------------------------
    float4 tv[RPW][NB];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int jb = 0; jb < NB; ++jb)
            tv[r][jb] = *reinterpret_cast<const float4*>(tgt + (size_t)min(row0 + r, m.n - 1) * NPAD + 256 * jb + 4 * lane);
    float vcx = 0.0f, vcy = 0.0f, vcz = 0.0f;      // velocity of this lane's row, carried from step to step
    FireState st;
    {
        const float* xin = io->xin;
        const float* pin = io->pin;
        const float* vin = io->vin;
        lds_dma_copy<BLOCK>(xin + roff, smem, 3 * NPAD, tid);
        for (int t = tid; t < m.ntiles; t += BLOCK)
            reinterpret_cast<float4*>(ps)[t] = reinterpret_cast<const float4*>(pin)[(size_t)rep * m.ntiles + t];
        if (finisher) { vcx = vin[ix]; vcy = vin[iy]; vcz = vin[iz]; }
        st = io->sin[rep];
    }
    // gather bookkeeping of this thread: unit u = tid + 256 k holds values (2 qd, 2 qd + 1) of record u / 16
    const int ulast = units - 1;
    const int qd = tid & (kRecUnits - 1), cat = qd >> 2, within = (qd & 3) * 2;
    float* const dump = wpart + 4 * WAVES;                              // 2 floats nobody reads
    float* const dst0 = cat < 3 ? smem + cat * NPAD + (tid >> 4) * kTileRows + within
                                : (qd < 14 ? ps + (tid >> 4) * 4 + within : dump);
    const int dstride = cat < 3 ? (BLOCK / kRecUnits) * kTileRows : (qd < 14 ? (BLOCK / kRecUnits) * 4 : 0);

    int s = 0;
    for (int run = 0; run < nruns; ++run) {
      const DevStep p = runs[run].p;
      const int count = runs[run].count;
      const bool needs_partials = p.kind == 0 || p.kind == 1 || p.kind == 2;
      for (int it = 0; it < count; ++it, ++s) {
        if (p.kind != 2) { st.dt = fp.dt_start; st.alpha = fp.alpha_start; st.npos = 0; st.pad = 0; }
        RSTAMP(0);                                  // LDS writes of the gather / loop bookkeeping
        __syncthreads();                            // xs/ys/zs/ps of this step are in LDS
        RSTAMP(1);                                  // barrier wait

        // ---- scalars per wave -----------------------------------------------------------------
        float4 psum = make_float4(0, 0, 0, 0);
        if (needs_partials) {
            for (int t = lane; t < m.ntiles; t += 64) {
                const float4 q = reinterpret_cast<const float4*>(ps)[t];
                psum.x += q.x; psum.y += q.y; psum.z += q.z; psum.w += q.w;
            }
            psum = wave_sum4(psum);
        }
        const StepScalars sc = step_scalars(m, p, fp, psum, st);
        RSTAMP(2);                                  // sums + scalars

        // ---- K2 -------------------------------------------------------------------------------
        float Fx = 0.0f, Fy = 0.0f, Fz = 0.0f;
        if (p.kind != 4) tile_forces_reg<POT, GEN, RPW, NB>(m, p, tv, xs, ys, zs, row0, lane, Fx, Fy, Fz);

        RSTAMP(3);                                  // pair loop + reductions
        // ---- epilogue: lanes 0..RPW-1 finish one row each ---------------------------------------
        float4 q = make_float4(0, 0, 0, 0);
        float xn = 0.0f, yn = 0.0f, zn = 0.0f;
        if (finisher) {
            float vx0 = vcx, vy0 = vcy, vz0 = vcz;
            if (p.kind == 3) { vx0 = vy0 = vz0 = 0.0f; }
            else if (p.kind == 4) { const float* vinit = late_pointers(io)->vinit; vx0 = vinit[ix]; vy0 = vinit[iy]; vz0 = vinit[iz]; }
            finish_row(m, p, fp, sc, st, Fx, Fy, Fz, xs[row], ys[row], zs[row], vx0, vy0, vz0, xn, yn, zn, vcx, vcy, vcz, q);
        } else if (lane < RPW) {
            xn = xs[row]; yn = ys[row]; zn = zs[row];    // padding row of the last tile: republish as is
        }
        if (lane < RPW) {
            const int k = row - tile * kTileRows;
            stage[k] = xn; stage[kTileRows + k] = yn; stage[2 * kTileRows + k] = zn;
        }
        q.x = quad_sum<RPW>(q.x); q.y = quad_sum<RPW>(q.y); q.z = quad_sum<RPW>(q.z); q.w = quad_sum<RPW>(q.w);
        if (lane == 0) reinterpret_cast<float4*>(wpart)[wave] = q;
        RSTAMP(4);                                  // row update
        __syncthreads();                            // all LDS reads of this step are done
        RSTAMP(5);                                  // barrier wait
        const bool last = s + 1 == nsteps;
        const unsigned tag = (unsigned)s + 1u;
        const int base = (((s + 1) & 1) * m.nrep + rep) * units;

        // ---- wave 0: tile sums (fixed order over the waves), then publish the record or hand the state back
        if (wave == 0) {
            float4 tsum = make_float4(0, 0, 0, 0);
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const float4 u = reinterpret_cast<float4*>(wpart)[w];
                tsum.x += u.x; tsum.y += u.y; tsum.z += u.z; tsum.w += u.w;
            }
            if (last) {
                if (tid == 0) {
                    const AnnealIO* o = late_pointers(io);
                    reinterpret_cast<float4*>(o->pout)[(size_t)rep * m.ntiles + tile] = tsum;
                    if (tile == 0) o->sout[rep] = st;
                }
            } else if (lane < kRecUnits) {
                const float2 sv = *reinterpret_cast<const float2*>(stage + 2 * min(lane, 11));
                const float a = lane < 12 ? sv.x : (lane == 12 ? tsum.x : (lane == 13 ? tsum.z : 0.0f));
                const float b = lane < 12 ? sv.y : (lane == 12 ? tsum.y : (lane == 13 ? tsum.w : 0.0f));
                u32x4 o;
                o.x = tag; o.y = __float_as_uint(a); o.z = tag; o.w = __float_as_uint(b);
                __builtin_amdgcn_raw_buffer_store_b128(o, rsrc, (base + tile * kRecUnits + lane) * 16, 0, 16);   // aux 16 = sc1
            }
        }
#ifdef C3D_STAMPS
        if (last && blockIdx.x == 0 && threadIdx.x == 0) for (int k = 0; k < 10; ++k) g_rstamps[k] = racc[k];
#endif
        if (last) {                                 // hand the state back to the ordinary buffers
            if (finisher) {
                const AnnealIO* o = late_pointers(io);
                float* xout = o->xout;
                float* vout = o->vout;
                xout[ix] = xn; xout[iy] = yn; xout[iz] = zn;
                vout[ix] = vcx; vout[iy] = vcy; vout[iz] = vcz;
            }
            return;
        }
        RSTAMP(6);                                  // publish
        // ---- gather the replica's records of step s+1 into LDS: all loads in flight, one wait, re-read
        //      until every tag matches (a thread with fewer than KU units re-reads the replica's last one)
        {
            u32x4 v[KU];
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
                asm volatile("" ::: "memory");     // the loads below must be re-issued on every sweep
#pragma unroll
                for (int k = 0; k < KU; ++k) {
                    const int u = min(tid + BLOCK * k, ulast);
                    v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (base + u) * 16, 0, 16);
                }
#pragma unroll
                for (int k = 0; k < KU; ++k) ok &= v[k].x == tag && v[k].z == tag;
#ifdef C3D_STAMPS
                if (blockIdx.x == 0 && threadIdx.x == 0) racc[9] += 1;
#endif
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 18)) {         // ~0.3 s: the tiles of this replica are not all resident
                    if (lane == 0) atomicOr(timeout, 1u);
                    return;
                }
            }
            RSTAMP(7);                              // gather: sweeps until every tag matches
#pragma unroll
            for (int k = 0; k < KU; ++k) {
                float* const dst = tid + BLOCK * k <= ulast ? dst0 + k * dstride : dump;
                *reinterpret_cast<float2*>(dst) = make_float2(__uint_as_float(v[k].y), __uint_as_float(v[k].w));
            }
        }
      }
    }
}

#ifdef C3D_STAMPS
hipError_t read_resident_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rstamps), sizeof(unsigned long long) * 16); }
#endif

bool anneal_supported(const DevModel& m) { return m.npad <= 1024 && m.rpw == 2; }

static size_t anneal_lds_bytes(int nb) { return sizeof(float) * ((size_t)3 * 256 * nb + 4 * (256 * nb / kTileRows) + kRecValues + 4 * (kTileRows / 2) + 4) + 128; }

size_t anneal_record_bytes(const DevModel& m) { return (size_t)2 * m.nrep * m.ntiles * kRecUnits * 16; }

template <int POT, bool GEN, int NB>
static hipError_t anneal_go(bool query, int* blocks_per_cu, const DevModel& m, const DevFire& fp, const AnnealIO* io,
                            const float* tgt, void* rec, const StepRun* runs, int nruns, int nsteps, unsigned* timeout, hipStream_t s) {
    const size_t lds = anneal_lds_bytes(NB);
    if (query) return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, k_anneal<POT, GEN, NB>, 256, lds);
    hipLaunchKernelGGL((k_anneal<POT, GEN, NB>), dim3(m.nrep * m.ntiles), dim3(256), lds, s, io, tgt,
                       reinterpret_cast<u32x4*>(rec), runs, nruns, nsteps, timeout, m, fp);
    return hipGetLastError();
}
template <int POT, bool GEN>
static hipError_t anneal_nb(bool query, int* bpc, const DevModel& m, const DevFire& fp, const AnnealIO* io, const float* tgt, void* rec,
                            const StepRun* runs, int nruns, int nsteps, unsigned* timeout, hipStream_t s) {
    switch (m.npad / 256) {
        case 1: return anneal_go<POT, GEN, 1>(query, bpc, m, fp, io, tgt, rec, runs, nruns, nsteps, timeout, s);
        case 2: return anneal_go<POT, GEN, 2>(query, bpc, m, fp, io, tgt, rec, runs, nruns, nsteps, timeout, s);
        case 3: return anneal_go<POT, GEN, 3>(query, bpc, m, fp, io, tgt, rec, runs, nruns, nsteps, timeout, s);
        default: return anneal_go<POT, GEN, 4>(query, bpc, m, fp, io, tgt, rec, runs, nruns, nsteps, timeout, s);
    }
}
static hipError_t anneal_dispatch(bool query, int* bpc, const DevModel& m, const DevFire& fp, const AnnealIO* io, const float* tgt,
                                  bool general_tail, void* rec, const StepRun* runs, int nruns, int nsteps, unsigned* timeout, hipStream_t s) {
    if (!general_tail) {
        switch (m.noe_pot) {
            case 0: return anneal_nb<0, false>(query, bpc, m, fp, io, tgt, rec, runs, nruns, nsteps, timeout, s);
            case 1: return anneal_nb<1, false>(query, bpc, m, fp, io, tgt, rec, runs, nruns, nsteps, timeout, s);
            default: return anneal_nb<2, false>(query, bpc, m, fp, io, tgt, rec, runs, nruns, nsteps, timeout, s);
        }
    }
    switch (m.noe_pot) {
        case 0: return anneal_nb<0, true>(query, bpc, m, fp, io, tgt, rec, runs, nruns, nsteps, timeout, s);
        case 1: return anneal_nb<1, true>(query, bpc, m, fp, io, tgt, rec, runs, nruns, nsteps, timeout, s);
        default: return anneal_nb<2, true>(query, bpc, m, fp, io, tgt, rec, runs, nruns, nsteps, timeout, s);
    }
}

// workgroups of this kernel that can be resident per CU (occupancy API, capped as the hardware admits them)
hipError_t anneal_blocks_per_cu(const DevModel& m, bool general_tail, int* blocks_per_cu) {
    DevFire fp{};
    hipError_t e = anneal_dispatch(true, blocks_per_cu, m, fp, nullptr, nullptr, general_tail, nullptr, nullptr, 0, 0, nullptr, nullptr);
    if (e == hipSuccess && *blocks_per_cu > 7) *blocks_per_cu = 7;   // .sgpr_count <= 96 (Makefile prints it): 7 admitted
    return e;
}

hipError_t launch_anneal(const DevModel& m, const DevFire& fp, const AnnealIO* io, const float* tgt, bool general_tail, void* rec,
                         const StepRun* runs, int nruns, int nsteps, unsigned* timeout, hipStream_t s) {
    return anneal_dispatch(false, nullptr, m, fp, io, tgt, general_tail, rec, runs, nruns, nsteps, timeout, s);
}

AnnealIO anneal_io(const DevBuffers& b, int parity) {
    const int q = parity ^ 1;
    AnnealIO io;
    io.pin = b.P[parity]; io.xin = b.X[parity]; io.vin = b.V[parity]; io.vinit = b.Vinit; io.sin = b.S[parity];
    io.xout = b.X[q]; io.vout = b.V[q]; io.pout = b.P[q]; io.sout = b.S[q];
    return io;
}

}  // namespace c3d
