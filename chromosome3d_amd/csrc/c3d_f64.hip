// c3d_f64.hip — the SA step in fp64 (option "precision" = 64): the precision reference of the fp32 hot path.
//
// The reference's arithmetic is fp64 throughout (Perl, CNS); the product kernels run fp32.  This is the
// same algorithm — same energy model, same lagged sums, same leap-frog / FIRE update (deck :1646-1700, :1729-1782,
// :1790-1803 restated in DESIGN.md section 3) — in plain fp64, one launch pair per step, written for clarity, not speed:
//   k64_force    one wave per (replica, row): every term of the row's force, fp64 tree sum over the lanes
//   k64_update   one workgroup per replica: the replica sums, thermostat / FIRE state, new velocities and positions
// It puts a number on what fp32 costs (bench.py --dtype f64 prints the f64 line beside the f32 one) and, because the
// CPU restatement the tests hold is the same algorithm in the same precision, it ties the GPU to it over LONG trajectories (fp32
// trajectories leave any reference after a few hundred chaotic steps; tests/test_gpu_parity.py::test_fp64_path_*).
// Targets come from the integer tenths (0.1 * t10, as the CPU restatement forms them), not from the fp32 target matrix.
#include "c3d_internal.h"

namespace c3d {

struct Model64 {
    int n, min_sep, noe_pot, rep_sep, ang_mode;
    double s_noe, rs, tail_c, tail_b, mrs, mtail_c, mtail_b;
    double k_bond, b0, k_ang, a0, r0_rep, k_rep, mass, fbeta;
};
struct Step64 {
    int kind;
    double dt, w_all, w_vdw, repel_s, t_bath;
};
struct Fire64 {
    double dt_start, dt_max, f_inc, f_dec, alpha_start, f_alpha, max_step;
    int n_min;
};
struct FireState64 {
    double dt, alpha;
    int npos, pad;
};
constexpr double kBoltz64 = 0.0019872, kAccel64 = 418.4;

// dE/dDelta of the NOE term without S and w
__device__ __forceinline__ double noe_grad64(const Model64& m, double delta) {
    const double ad = fabs(delta);
    if (m.noe_pot == 0) { if (ad > m.rs) { const double g = m.tail_c - m.tail_b / (ad * ad); return delta > 0 ? g : -g; } return 2.0 * delta; }
    if (m.noe_pot == 1) return delta > m.rs ? m.tail_c - m.tail_b / (ad * ad) : 2.0 * delta;
    if (m.noe_pot == 3) {
        if (delta > m.rs) return m.tail_c - m.tail_b / (ad * ad);
        if (delta < -m.mrs) return -(m.mtail_c - m.mtail_b / (ad * ad));
        return 2.0 * delta;
    }
    return 2.0 * delta;
}

__global__ __launch_bounds__(64) void k64_force(const Model64 m, const Step64 p, const int32_t* __restrict__ t10,
                                               const double* __restrict__ X, double* __restrict__ F) {
    const int i = blockIdx.x, rep = blockIdx.y, lane = threadIdx.x, n = m.n;
    const double* x = X + (size_t)rep * 3 * n;
    const double xi = x[3 * i], yi = x[3 * i + 1], zi = x[3 * i + 2];
    const double R2 = (p.repel_s * m.r0_rep) * (p.repel_s * m.r0_rep);
    double fx = 0, fy = 0, fz = 0;
    for (int j = lane; j < n; j += 64) {
        if (j == i) continue;
        const double dx = xi - x[3 * j], dy = yi - x[3 * j + 1], dz = zi - x[3 * j + 2];
        double r2 = dx * dx + dy * dy + dz * dz;
        if (r2 < 1e-12) r2 = 1e-12;
        const int sep = j > i ? j - i : i - j;
        double coef = 0.0;
        const int32_t t = t10[(size_t)i * n + j];
        if (sep >= m.min_sep && t > 0) {
            const double d = sqrt(r2);
            coef -= p.w_all * m.s_noe * noe_grad64(m, d - 0.1 * t) / d;
        }
        if (sep == 1) {
            const double d = sqrt(r2);
            coef -= p.w_all * 2.0 * m.k_bond * (d - m.b0) / d;
        }
        if (sep >= m.rep_sep && r2 < R2) coef += p.w_vdw * m.k_rep * 4.0 * (R2 - r2);
        if (sep == 2 && m.k_ang > 0 && (m.ang_mode == 1 || r2 < m.a0 * m.a0)) {
            const double d = sqrt(r2);
            coef -= p.w_all * 2.0 * m.k_ang * (d - m.a0) / d;
        }
        fx += coef * dx; fy += coef * dy; fz += coef * dz;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { fx += __shfl_xor(fx, off, 64); fy += __shfl_xor(fy, off, 64); fz += __shfl_xor(fz, off, 64); }
    if (lane == 0) { double* f = F + ((size_t)rep * n + i) * 3; f[0] = fx; f[1] = fy; f[2] = fz; }
}

__device__ __forceinline__ double block_sum64(double v, double* scratch, int tid) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((tid & 63) == 0) scratch[tid >> 6] = v;
    __syncthreads();
    return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

// L: [nrep][4] sums of the previous FIRE evaluation (v.F, F.F, v.v); fs: FIRE state per replica
__global__ __launch_bounds__(256) void k64_update(const Model64 m, const Step64 p, const Fire64 fp, double* __restrict__ X,
                                                 double* __restrict__ V, const double* __restrict__ F,
                                                 const double* __restrict__ Vinit, double* __restrict__ L, FireState64* __restrict__ fs) {
    __shared__ double scratch[4];
    const int rep = blockIdx.x, tid = threadIdx.x, n = m.n;
    double* x = X + (size_t)rep * 3 * n;
    double* v = V + (size_t)rep * 3 * n;
    const double* f = F + (size_t)rep * 3 * n;
    if (p.kind == 4) {                              // MD begin: Maxwell velocities, no move
        for (int k = tid; k < 3 * n; k += 256) v[k] = Vinit[(size_t)rep * 3 * n + k];
        return;
    }
    if (p.kind == 0 || p.kind == 1) {
        double ke2 = 0, c0 = 0, c1 = 0, c2 = 0;
        for (int k = tid; k < 3 * n; k += 256) ke2 += v[k] * v[k];
        for (int i = tid; i < n; i += 256) { c0 += v[3 * i]; c1 += v[3 * i + 1]; c2 += v[3 * i + 2]; }
        ke2 = block_sum64(ke2, scratch, tid);
        c0 = block_sum64(c0, scratch, tid) / n; c1 = block_sum64(c1, scratch, tid) / n; c2 = block_sum64(c2, scratch, tid) / n;
        const int ndf = 3 * n - 3;
        double tprev = m.mass * ke2 / kAccel64 / ((ndf > 0 ? ndf : 1) * kBoltz64);
        if (tprev < 1e-2) tprev = 1e-2;
        double lam;
        if (p.kind == 0) { double l2 = 1.0 + p.dt * m.fbeta * (p.t_bath / tprev - 1.0); if (l2 < 0) l2 = 0; lam = sqrt(l2); }
        else lam = sqrt(p.t_bath / tprev);
        const double acc = p.dt * kAccel64 / m.mass;
        __syncthreads();
        for (int k = tid; k < 3 * n; k += 256) {
            const double cm = (k % 3) == 0 ? c0 : ((k % 3) == 1 ? c1 : c2);
            const double vn = lam * (v[k] - cm) + acc * f[k];
            v[k] = vn;
            x[k] += p.dt * vn;
        }
        return;
    }
    // FIRE with the power test on the previous evaluation's sums
    FireState64 st = fs[rep];
    double L0 = L[4 * rep], L1 = L[4 * rep + 1], L2 = L[4 * rep + 2];
    if (p.kind == 3) {                              // first step of a stage: v = 0, L = 0, fresh state
        for (int k = tid; k < 3 * n; k += 256) v[k] = 0.0;
        L0 = L1 = L2 = 0.0;
        st.dt = fp.dt_start; st.alpha = fp.alpha_start; st.npos = 0;
        __syncthreads();
    }
    double vf = 0, ff = 0, vv = 0;
    for (int k = tid; k < 3 * n; k += 256) { vf += v[k] * f[k]; ff += f[k] * f[k]; vv += v[k] * v[k]; }
    vf = block_sum64(vf, scratch, tid); ff = block_sum64(ff, scratch, tid); vv = block_sum64(vv, scratch, tid);
    __syncthreads();
    if (L0 > 0) {
        const double mix = st.alpha * sqrt(L2 / (L1 > 1e-30 ? L1 : 1e-30));
        for (int k = tid; k < 3 * n; k += 256) v[k] = (1.0 - st.alpha) * v[k] + mix * f[k];
        if (st.npos > fp.n_min) { st.dt = st.dt * fp.f_inc < fp.dt_max ? st.dt * fp.f_inc : fp.dt_max; st.alpha *= fp.f_alpha; }
        st.npos += 1;
    } else {
        for (int k = tid; k < 3 * n; k += 256) v[k] = 0.0;
        st.alpha = fp.alpha_start; st.dt *= fp.f_dec; st.npos = 0;
    }
    __syncthreads();
    const double acc = st.dt * kAccel64 / m.mass;
    for (int i = tid; i < n; i += 256) {
        double dr[3], d2 = 0;
        for (int c = 0; c < 3; ++c) {
            const int k = 3 * i + c;
            v[k] += acc * f[k];
            dr[c] = st.dt * v[k];
            d2 += dr[c] * dr[c];
        }
        const double sc = d2 > fp.max_step * fp.max_step ? fp.max_step / sqrt(d2) : 1.0;
        for (int c = 0; c < 3; ++c) x[3 * i + c] += sc * dr[c];
    }
    if (tid == 0) { L[4 * rep] = vf; L[4 * rep + 1] = ff; L[4 * rep + 2] = vv; L[4 * rep + 3] = 0; fs[rep] = st; }
}

// fp32 SoA buffers <-> fp64 AoS state (the solver's read-back, energies and scoring work on the fp32 copy)
__global__ __launch_bounds__(256) void k64_import(int n, int npad, const float* __restrict__ Xf, double* __restrict__ X, double* __restrict__ V) {
    const int rep = blockIdx.x;
    for (int k = threadIdx.x; k < 3 * n; k += 256) {
        const int i = k / 3, c = k % 3;
        X[(size_t)rep * 3 * n + k] = (double)Xf[((size_t)rep * 3 + c) * npad + i];
        V[(size_t)rep * 3 * n + k] = 0.0;
    }
}
__global__ __launch_bounds__(256) void k64_export(int n, int npad, int ntiles, const double* __restrict__ X, const double* __restrict__ V,
                                                 const double* __restrict__ L, float* __restrict__ Xf, float* __restrict__ Vf, float* __restrict__ Pf) {
    const int rep = blockIdx.x;
    for (int k = threadIdx.x; k < 3 * n; k += 256) {
        const int i = k / 3, c = k % 3;
        Xf[((size_t)rep * 3 + c) * npad + i] = (float)X[(size_t)rep * 3 * n + k];
        Vf[((size_t)rep * 3 + c) * npad + i] = (float)V[(size_t)rep * 3 * n + k];
    }
    // the minimiser's sums go where the host looks for them (max RMS force, finiteness): tile 0 carries the replica's totals
    for (int t = threadIdx.x; t < ntiles; t += 256) {
        float4 q = make_float4(0, 0, 0, 0);
        if (t == 0) q = make_float4((float)L[4 * rep], (float)L[4 * rep + 1], (float)L[4 * rep + 2], 0.0f);
        reinterpret_cast<float4*>(Pf)[(size_t)rep * ntiles + t] = q;
    }
}

// ---- host side ---------------------------------------------------------------------------------------
static Model64 model64(const DevModel& d, const double* host) {
    // host[]: s_noe, rswitch, asym, masym, mrswitch, k_bond, b0, k_ang, a0, r0_rep, k_rep, mass, fbeta, min_sep
    Model64 m;
    m.n = d.n; m.min_sep = (int)host[13]; m.noe_pot = d.noe_pot; m.rep_sep = d.rep_sep; m.ang_mode = d.ang_mode;
    m.s_noe = host[0]; m.rs = host[1];
    m.tail_c = host[2] * host[1]; m.tail_b = (m.tail_c - 2.0 * m.rs) * m.rs * m.rs;
    m.mrs = host[4]; m.mtail_c = host[3]; m.mtail_b = (m.mtail_c - 2.0 * m.mrs) * m.mrs * m.mrs;
    m.k_bond = host[5]; m.b0 = host[6]; m.k_ang = host[7]; m.a0 = host[8]; m.r0_rep = host[9]; m.k_rep = host[10]; m.mass = host[11]; m.fbeta = host[12];
    return m;
}

hipError_t launch_step64(const DevModel& d, const double* model_host, const double* step_host, const double* fire_host, int fire_n_min,
                         const int32_t* t10, double* X, double* V, double* F, const double* Vinit, double* L, void* fs, hipStream_t s) {
    const Model64 m = model64(d, model_host);
    Step64 p;   // step_host[]: kind, dt, w_all, w_vdw, repel_s, t_bath
    p.kind = (int)step_host[0]; p.dt = step_host[1]; p.w_all = step_host[2]; p.w_vdw = step_host[3]; p.repel_s = step_host[4]; p.t_bath = step_host[5];
    Fire64 fp;
    fp.dt_start = fire_host[0]; fp.dt_max = fire_host[1]; fp.f_inc = fire_host[2]; fp.f_dec = fire_host[3]; fp.alpha_start = fire_host[4];
    fp.f_alpha = fire_host[5]; fp.max_step = fire_host[6]; fp.n_min = fire_n_min;
    if (p.kind != 4) hipLaunchKernelGGL(k64_force, dim3(d.n, d.nrep), dim3(64), 0, s, m, p, t10, X, F);
    hipLaunchKernelGGL(k64_update, dim3(d.nrep), dim3(256), 0, s, m, p, fp, X, V, F, Vinit, L, reinterpret_cast<FireState64*>(fs));
    return hipGetLastError();
}
hipError_t launch_import64(const DevModel& d, const float* Xf, double* X, double* V, hipStream_t s) {
    hipLaunchKernelGGL(k64_import, dim3(d.nrep), dim3(256), 0, s, d.n, d.npad, Xf, X, V);
    return hipGetLastError();
}
hipError_t launch_export64(const DevModel& d, const double* X, const double* V, const double* L, float* Xf, float* Vf, float* Pf, hipStream_t s) {
    hipLaunchKernelGGL(k64_export, dim3(d.nrep), dim3(256), 0, s, d.n, d.npad, d.ntiles, X, V, L, Xf, Vf, Pf);
    return hipGetLastError();
}
size_t fire_state64_bytes() { return sizeof(FireState64); }

}  // namespace c3d
