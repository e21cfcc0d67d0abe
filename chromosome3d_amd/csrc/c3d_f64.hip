// c3d_f64.hip — the SA step in fp64 (option "precision" = 64): the reference's precision on the GPU.
//
// The reference's arithmetic is fp64 throughout (Perl, CNS); the product kernels run fp32.  This is the
// same algorithm — same energy model, same lagged sums, same leap-frog / FIRE update (deck :1646-1700, :1729-1782,
// :1790-1803 restated in DESIGN.md section 3) — in fp64, and since round 4 in the per-step kernel's SHAPE (round 3: one wave
// per row over array-of-structures coordinates, a force and an update launch per step, 29.5 us per 20-replica step):
//
//   k64_step   ONE launch per SA step of a replica group.  A workgroup (4 waves) owns kTileRows = 8 consecutive rows of one
//              replica's N x N pair matrix, 2 rows per wave; the replica's coordinates are staged once in LDS as
//              structure-of-arrays doubles (3 x 8 x npad bytes), lanes run along the columns (column 64 k + lane: every
//              target load is one 512-byte line per wave, every coordinate read a conflict-free ds_read_b64), two columns
//              in flight per lane.  1 / d comes from v_rsq_f64 (23 bits) and two coupled Newton steps that deliver d and
//              1 / d together (8 fused operations instead of the ~30 of a correctly rounded sqrt and a division); the two
//              rows of a wave are reduced in ONE transposing DPP butterfly on the 32-bit halves.  The row's owner lane
//              adds the chain terms, integrates the row and leaves its contribution to the replica sums; the sums a step
//              needs (kinetic energy, centre-of-mass velocity, FIRE power and norms) are the per-tile partial sums the
//              previous step left, added in a fixed order by every wave — the kernel boundary is the only synchronisation,
//              exactly as in k_step (c3d_device.hip), so the hipGraph replay and the replica groups on two streams of the
//              per-step path carry it unchanged.
//
// It puts a number on what fp32 buys (bench.py prints the f64 line beside the f32 one) and, because the CPU restatement the
// tests hold is the same algorithm in the same precision, it ties the GPU to it over LONG trajectories (fp32 trajectories leave
// any reference after a few hundred chaotic steps; tests/test_gpu_parity.py::test_fp64_path_*).  Targets are 0.1 * t10 formed
// in fp64 from the integer tenths (as the CPU restatement forms them), not the fp32 target matrix.
#include "c3d_internal.h"

namespace c3d {

struct Model64 {
    int n, np, ntiles, min_sep, noe_pot, rep_sep, ang_mode, mexp;      // noe_pot as DevModel's (4 = the fast soft lower side)
    double s_noe, rs, tail_c, tail_b, mrs, mtail_c, mtail_b;
    double k_bond, b0, k_ang, a0, r0_rep, k_rep, mass, fbeta;
    double nmrs4;                                  // -mrs^4 (the fast soft lower side's bound is nmrs4 / D^3)
    double t_fac, inv_n;                           // T = t_fac * sum v^2; 1 / n
};
struct Step64 {
    int kind;
    double dt, w_all, w_vdw, repel_s, t_bath;
    // uniform factors of a step, formed on the host (fp64 has no scalar ALU: formed in the kernel they are vector registers every wave
    // holds through its pair loop): R2 = (repel_s r0_rep)^2, wr4 = 4 w_vdw k_rep, nws4 = -4 w_all S, wq = wr4 / nws4 (0 where nws4 = 0)
    double R2, wr4, nws4, wq;
    double acc;        // MD: dt kAccel / mass (the kernel's own division was ~30 fp64 operations on every wave's path after its pair loop)
    double kb4, ka4;   // chain terms: -4 w_all k_bond, -4 w_all k_ang
    double kacc;       // kAccel / mass (FIRE: times the step's dt)
    double a0sq;       // a0^2
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
constexpr int kRows64 = 2;                       // rows per wave
constexpr int kWaves64 = kTileRows / kRows64;    // 4 waves: one tile of 8 rows per workgroup (the fp32 step's tile)
constexpr int kBlock64 = 64 * kWaves64;
constexpr int kColPad64 = 128;                   // columns padded to two per lane
constexpr double kNoTarget64 = 1.0e300;          // "no restraint" in the target matrix of the fast soft lower side (pair64)

// ---- 64-bit values through the 32-bit cross-lane paths (DPP, permlane swaps): VALU only, no LDS round trip ----------
template <int CTRL>
__device__ __forceinline__ double dpp_mov64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double xrow_sum64(double v) {    // + the other three 16-lane rows, every lane
    const long long b = __double_as_longlong(v);
    unsigned lo = (unsigned)(b & 0xffffffffll), hi = (unsigned)(b >> 32);
    auto l16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto h16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __longlong_as_double(((long long)h16[0] << 32) | l16[0]) + __longlong_as_double(((long long)h16[1] << 32) | l16[1]);
    const long long c = __double_as_longlong(v);
    lo = (unsigned)(c & 0xffffffffll); hi = (unsigned)(c >> 32);
    auto l32 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto h32 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __longlong_as_double(((long long)h32[0] << 32) | l32[0]) + __longlong_as_double(((long long)h32[1] << 32) | l32[1]);
}
__device__ __forceinline__ double wave_sum64(double v) {    // total in every lane, fixed tree
    v += dpp_mov64<0xB1>(v);
    v += dpp_mov64<0x4E>(v);
    v += dpp_mov64<0x124>(v);
    v += dpp_mov64<0x128>(v);
    return xrow_sum64(v);
}
// a0 / a1: this lane's partial sums for the wave's two rows; returns, in every lane l, the wave total of row (l & 1)
__device__ __forceinline__ double reduce_rows64(double a0, double a1, int lane) {
    const bool odd = lane & 1;
    double k = odd ? a1 : a0;
    const double s = odd ? a0 : a1;
    k += dpp_mov64<0xB1>(s);
    k += dpp_mov64<0x4E>(k);
    k += dpp_mov64<0x124>(k);
    k += dpp_mov64<0x128>(k);
    return xrow_sum64(k);
}

// d = sqrt(r2) and h = 1 / (2 d) together: v_rsq_f64 (2^29 ulp: 23 bits) + two coupled Newton steps (Goldschmidt form);
// the results are within an ulp or two of the correctly rounded values (r2 >= 1e-12: no denormal, no zero)
__device__ __forceinline__ void sqrt_hrsqrt64(double r2, double& d, double& h) {
    // 23-bit seeds from the fp32 unit (v_rsq_f64 is no better and no faster), the halving done there too; 1e-30 <= r2 <= 1e9
    const float yf = __builtin_amdgcn_rsqf((float)r2);
    double g = r2 * (double)yf;
    h = (double)(0.5f * yf);
    double r = fma(-g, h, 0.5);
#ifdef C3D_F64_TWO_NEWTON
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-g, h, 0.5);
#endif
    d = fma(g, r, g); h = fma(h, r, h);
}

// 1 / x and sqrt(x) to a rounding or two (fp32 seed, two Newton steps each): the per-workgroup scalars; x > 0 (sqrt64: 0 allowed)
__device__ __forceinline__ double rcp64(double x) {
    double y = (double)__builtin_amdgcn_rcpf((float)x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}
__device__ __forceinline__ double sqrt64(double x) {
    if (!(x > 0.0)) return 0.0;
    const double y = (double)__builtin_amdgcn_rsqf((float)x);
    double g = x * y, h = 0.5 * y;
    double r = fma(-g, h, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-g, h, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    return fma(fma(-g, g, x), h, g);                // one more correction of g alone
}

// HALF of dE/dDelta of the NOE term without S and w.  GEN = tails with a 1/D^2 part.
template <int POT, bool GEN>
__device__ __forceinline__ double half_noe_grad64(const Model64& m, double delta) {
    if constexpr (!GEN) {      // the force stays at its value at the switch distance: slope 2 rs above, 2 mrs below (POT 3)
        if constexpr (POT == 0) return fmin(fmax(delta, -m.rs), m.rs);
        else if constexpr (POT == 1) return fmin(delta, m.rs);
        else if constexpr (POT == 3) return fmin(fmax(delta, -m.mrs), m.rs);
        else if constexpr (POT == 4) {
            // lower side beyond mrs: dE/dD = 2 mrs^4 / D^3 (soft form, exponent 2, no asymptote) = the lower bound -mrs (mrs / D)^3 of
            // the same clamp; 1 / D from v_rcp_f64 and two Newton steps, D held at >= mrs (the bound is -mrs inside the square part)
            // D = |delta| (a free source modifier; round 4, second session: it was max(-delta, mrs)): inside the square part and above the
            // target the bound -mrs^4 / D^3 then lies BELOW delta and does not bind; delta = 0 (or below the fp32 seed's range): the seed is
            // inf, the Newton step NaN, and max(delta, NaN) = delta
            const double D = fabs(delta);
            double y = (double)__builtin_amdgcn_rcpf((float)D);
            double e = fma(-D, y, 1.0);
            y = fma(y, e, y);
#ifdef C3D_F64_TWO_NEWTON
            e = fma(-D, y, 1.0);
            y = fma(y, e, y);
#endif
            return fmin(fmax(delta, (y * y) * (y * m.nmrs4)), m.rs);       // nmrs4 = -mrs^4
        }
        else return delta;
    } else {
        const double ad = fabs(delta);
        const double a2 = ad * ad > 1e-300 ? ad * ad : 1e-300;
        const double up = 0.5 * (m.tail_c - m.tail_b / a2), lo = -0.5 * (m.mtail_c - m.mtail_b / (m.mexp == 2 ? a2 * ad : a2));
        if constexpr (POT == 0) return ad > m.rs ? (delta > 0 ? up : -up) : delta;
        else if constexpr (POT == 1) return delta > m.rs ? up : delta;
        else if constexpr (POT == 3 || POT == 4) return delta > m.rs ? up : (delta < -m.mrs ? lo : delta);
        else return delta;
    }
}

// One pair term of a row against column j (T = 0.1 * t10 where restrained, else 0; pad columns sit 1e4 A away with T = 0).
// Straight-line code, no branch around the square root (a branch serialises the four pair terms a lane has in flight):
//   NOE    -w S g(d - T) / d = nws4 * (g / 2) * h,  nws4 = -4 w S where restrained else 0, h = 1 / (2 d)
//   repel  on EVERY column: 4 w_vdw k_rep max(0, R2 - r2); the self term has dx = 0, the |i-j| < rep_sep neighbours are taken
//          back out with the chain terms (chain64)
// FOLD (fast soft lower side only): the row's sum is formed WITHOUT the NOE weight — nws4 = 1 here, wr4 = the repel weight divided by the NOE
// weight — and multiplied by it once per row (k64_step): one multiplication less per pair term.  Needs a non-zero NOE weight.
template <int POT, bool GEN, bool FOLD = false>
__device__ __forceinline__ void pair64(const Model64& m, double nws4, double wr4, double R2, double T, double xi, double yi, double zi,
                                       double xj, double yj, double zj, double& fx, double& fy, double& fz) {
    const double dx = xi - xj, dy = yi - yj, dz = zi - zj;
    // The guard against r2 = 0 (the self term; the CPU restatement clamps at 1e-12, which no pair of distinct beads ever reaches) rides in the
    // fma chain: 1e-30 is below half an ulp of any r2 > 1e-14, so every real pair keeps its bits, and the self term stays finite (x 0 = 0).
    const double r2 = fma(dx, dx, fma(dy, dy, fma(dz, dz, 1e-30)));
    double d, h;
    sqrt_hrsqrt64(r2, d, h);
    double wn = nws4;
    // no restraint: T = 0 and the weight is switched off — except for the fast soft lower side (POT 4), where such a pair carries
    // T = kNoTarget64 = 1e300 instead: a pair that far inside its "target" feels exactly nothing (D = 1e300: the fp32 seed of 1 / D is 0, the
    // Newton steps keep it, the bound is -0, the force +0), which saves the compare and the two selects of every pair term
    if constexpr (!(POT == 4 && !GEN)) wn = T > 0.0 ? nws4 : 0.0;
    double cn = half_noe_grad64<POT, GEN>(m, d - T) * h;
    if constexpr (!(FOLD && POT == 4 && !GEN)) cn = wn * cn;
    const double coef = fma(wr4, fmax(R2 - r2, 0.0), cn);
    fx = fma(coef, dx, fx); fy = fma(coef, dy, fy); fz = fma(coef, dz, fz);
}

// pseudo-bond (i,i+-1) and pseudo-angle (i,i+-2) terms of `row` against neighbour row + off
__device__ __forceinline__ void chain64(const Model64& m, const Step64& p, double wr4, double R2, const double* xs, const double* ys,
                                        const double* zs, int row, int off, double& cx, double& cy, double& cz) {
    cx = cy = cz = 0.0;
    const int jn = row + off, sep = off < 0 ? -off : off;
    if (row >= m.n || jn < 0 || jn >= m.n) return;
    const double dx = xs[row] - xs[jn], dy = ys[row] - ys[jn], dz = zs[row] - zs[jn];
    const double r2 = fmax(fma(dx, dx, fma(dy, dy, dz * dz)), 1e-12);
    double d, h;
    sqrt_hrsqrt64(r2, d, h);                          // h = 1 / (2 d)
    double coef = 0.0;
    if (sep == 1) coef = p.kb4 * (d - m.b0) * h;
    else if (m.k_ang > 0 && (m.ang_mode == 1 || r2 < p.a0sq)) coef = p.ka4 * (d - m.a0) * h;
    if (sep < m.rep_sep) coef -= wr4 * fmax(R2 - r2, 0.0);      // the pair loop applied the repel term to every column
    cx = coef * dx; cy = coef * dy; cz = coef * dz;
}

// P: [nrep][ntiles][4] per-tile sums of the previous step — MD kinds: (sum v^2, sum vx, vy, vz); FIRE: (v.F, F.F, v.v, 0)
template <int POT, bool GEN, bool FOLD = false>
__global__ __launch_bounds__(kBlock64) __attribute__((amdgpu_waves_per_eu(5))) void k64_step(const Model64 m, const Step64 p, const Fire64 fp, const int rep_base,
                                                    const double* __restrict__ T, const double* __restrict__ xin,
                                                    const double* __restrict__ vin, const double* __restrict__ vinit,
                                                    const double* __restrict__ pin, const FireState64* __restrict__ sin,
                                                    double* __restrict__ xout, double* __restrict__ vout, double* __restrict__ pout,
                                                    FireState64* __restrict__ sout) {
    extern __shared__ __attribute__((aligned(16))) double sm64[];
    const int tile = blockIdx.x, rep = rep_base + blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = m.n, np = m.np;
    double* xs = sm64;
    double* ys = sm64 + np;
    double* zs = sm64 + 2 * np;
    double* rowq = sm64 + 3 * np;                       // [kTileRows][4]
    const size_t roff = (size_t)rep * 3 * np;
    // ---- stage the replica's coordinates; the previous step's sums meanwhile ----
    for (int b = 2 * tid; b < 3 * np; b += 2 * kBlock64) *reinterpret_cast<double2*>(sm64 + b) = *reinterpret_cast<const double2*>(xin + roff + b);
    // ---- loads whose latency would otherwise be exposed later leave now: the first targets of this wave's rows, the velocities of the two
    //      rows it finishes ----
    const int row0 = tile * kTileRows + wave * kRows64;
    const int ra = min(row0, n - 1), rb = min(row0 + 1, n - 1);
    const double* Ta = T + (size_t)ra * np + lane;
    const double* Tb = T + (size_t)rb * np + lane;
    double ta0 = Ta[0], ta1 = Ta[64], tb0 = Tb[0], tb1 = Tb[64];        // np >= 128: in bounds whatever n is
    const int row = row0 + lane;
    double v0x = 0, v0y = 0, v0z = 0;
    if (lane < kRows64 && row < n && p.kind != 3 && p.kind != 6) {
        const double* vsrc = p.kind == 4 ? vinit : vin;
        const size_t ix = roff + row;
        v0x = vsrc[ix]; v0y = vsrc[ix + np]; v0z = vsrc[ix + 2 * np];
    }
    // ---- the replica's scalars of this step: ONE wave forms them (the sums of 57 tiles through four butterflies, six fp64 divisions and
    //      a square root are ~400 instruction slots — as much as two thirds of a wave's pair loop) and leaves them in LDS before the
    //      barrier everybody waits at anyway ----
    double* scal = rowq + 4 * kTileRows;                // [8] lam, cm0, cm1, cm2, keep, mix, dt, (unused)
    FireState64 st;
    st.dt = fp.dt_start; st.alpha = fp.alpha_start; st.npos = 0; st.pad = 0;
    if (wave == 0 && (p.kind == 2 || p.kind == 5)) st = sin[rep];        // (asked for here, used after the sums have arrived)
    if (wave == 0) {
        const bool needs = p.kind == 0 || p.kind == 1 || p.kind == 2 || p.kind == 5;
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        if (needs) {
            const double* pp = pin + (size_t)rep * m.ntiles * 4;
            for (int t = lane; t < m.ntiles; t += 64) { s0 += pp[4 * t]; s1 += pp[4 * t + 1]; s2 += pp[4 * t + 2]; s3 += pp[4 * t + 3]; }
            s0 = wave_sum64(s0); s1 = wave_sum64(s1); s2 = wave_sum64(s2); s3 = wave_sum64(s3);
        }
        // (reciprocals and square roots by seed + two Newton steps, the constant factors folded on the host: the correctly rounded
        //  divisions and sqrt of the straightforward form are ~230 dependent fp64 operations — a microsecond on every workgroup's
        //  critical path, more than the launch boundary hides; these are a rounding or two away from them)
        double lam = 1.0, cm0 = 0, cm1 = 0, cm2 = 0, keep = 0.0, mix = 0.0;
        if (p.kind == 0 || p.kind == 1) {
            double tprev = m.t_fac * s0;                // mass / kAccel / (ndf kBoltz) * sum v^2
            if (tprev < 1e-2) tprev = 1e-2;
            const double ratio = p.t_bath * rcp64(tprev);
            if (p.kind == 0) { double l2 = 1.0 + p.dt * m.fbeta * (ratio - 1.0); if (l2 < 0) l2 = 0; lam = sqrt64(l2); }
            else lam = sqrt64(ratio);
            cm0 = s1 * m.inv_n; cm1 = s2 * m.inv_n; cm2 = s3 * m.inv_n;
        } else if (p.kind == 2 || p.kind == 3) {
            if (s0 > 0) {                               // power of the previous evaluation positive (kind 3: sums are 0)
                keep = 1.0 - st.alpha;
                mix = st.alpha * sqrt64(s2 * rcp64(s1 > 1e-30 ? s1 : 1e-30));
                if (st.npos > fp.n_min) { st.dt = st.dt * fp.f_inc < fp.dt_max ? st.dt * fp.f_inc : fp.dt_max; st.alpha *= fp.f_alpha; }
                st.npos += 1;
            } else {
                st.alpha = fp.alpha_start; st.dt *= fp.f_dec; st.npos = 0;
            }
            if (tile == 0 && lane == 0) sout[rep] = st;
        } else if (p.kind == 5 || p.kind == 6) {        // two-point step size, the length one evaluation late (c3o_bb_step): lam = previous length, mix = this one
            const int k = p.kind == 6 ? 0 : st.npos;
            const double a_prev = st.dt;
            double a = a_prev;
            if (k == 0) a = fp.dt_start * fp.dt_start * p.kacc;
            else if (k >= 2) {
                if (s0 > 0) a = (k & 1) ? s0 * rcp64(s2) : s3 * rcp64(s0);
                else a = 2.0 * a_prev;
                if (!(a >= 1e-7)) a = 1e-7;
                if (a > 1e2) a = 1e2;
            }
            lam = a_prev; mix = a;
            st.dt = a; st.npos = k + 1;
            if (tile == 0 && lane == 0) sout[rep] = st;
        }
        if (lane == 0) { scal[0] = lam; scal[1] = cm0; scal[2] = cm1; scal[3] = cm2; scal[4] = keep; scal[5] = mix; scal[6] = st.dt; }
    }
    __syncthreads();

    // ---- pair forces of this wave's two rows ----
    double fxa = 0, fya = 0, fza = 0, fxb = 0, fyb = 0, fzb = 0;
    const double R2 = p.R2;
    const double wr4 = p.wr4;
    if (p.kind != 4) {
        const double xa = xs[ra], ya = ys[ra], za = zs[ra], xb = xs[rb], yb = ys[rb], zb = zs[rb];
        // (FOLD: the pair terms carry the repel weight relative to the NOE weight, the row sums get the NOE weight below)
        const double nws4p = FOLD ? 1.0 : p.nws4, wr4p = FOLD ? p.wq : wr4;
        // Columns: two per lane and pass (j, j + 64) over the first n & ~127 of them, the next two in flight while these two compute; then
        // ONE column per lane if 64 or more are left, then the last n % 64 columns — both rows of the wave in one pass where they fit
        // (lane = (row, column)).  No lane evaluates a padding column pair by pair any more (455 beads: 15 pair terms per lane, not 16);
        // a lane without a column takes the padding bead n (1e4 A away, no target: an exact zero).
        const int nmain = n & ~127;
        if (nmain > 0) {
            for (int j = lane; j < nmain; j += 128) {
                const int jn = j + 128 < nmain ? 128 : 0;       // the last pass re-reads itself (in bounds)
                Ta += jn; Tb += jn;
                const double na0 = Ta[0], na1 = Ta[64], nb0 = Tb[0], nb1 = Tb[64];
                const double x0 = xs[j], y0 = ys[j], z0 = zs[j], x1 = xs[j + 64], y1 = ys[j + 64], z1 = zs[j + 64];
                pair64<POT, GEN, FOLD>(m, nws4p, wr4p, R2, ta0, xa, ya, za, x0, y0, z0, fxa, fya, fza);
                pair64<POT, GEN, FOLD>(m, nws4p, wr4p, R2, tb0, xb, yb, zb, x0, y0, z0, fxb, fyb, fzb);
                pair64<POT, GEN, FOLD>(m, nws4p, wr4p, R2, ta1, xa, ya, za, x1, y1, z1, fxa, fya, fza);
                pair64<POT, GEN, FOLD>(m, nws4p, wr4p, R2, tb1, xb, yb, zb, x1, y1, z1, fxb, fyb, fzb);
                ta0 = na0; ta1 = na1; tb0 = nb0; tb1 = nb1;
            }
        }
        const double* Tra = T + (size_t)ra * np;
        const double* Trb = T + (size_t)rb * np;
        int c0 = nmain;
        if (n - c0 >= 64) {
            const int j = c0 + lane;
            const double x0 = xs[j], y0 = ys[j], z0 = zs[j];
            pair64<POT, GEN, FOLD>(m, nws4p, wr4p, R2, Tra[j], xa, ya, za, x0, y0, z0, fxa, fya, fza);
            pair64<POT, GEN, FOLD>(m, nws4p, wr4p, R2, Trb[j], xb, yb, zb, x0, y0, z0, fxb, fyb, fzb);
            c0 += 64;
        }
        const int left = n - c0;                             // 0 .. 63 columns
        if (left > 0 && 2 * left <= 64) {
            const bool second = lane >= left;                // lanes [0, left): row a; [left, 2 left): row b; beyond: the padding bead
            const int c = lane - (second ? left : 0);
            const int j = c < left ? c0 + c : n;
            const double xr = second ? xb : xa, yr = second ? yb : ya, zr = second ? zb : za;
            double tx = 0, ty = 0, tz = 0;
            pair64<POT, GEN, FOLD>(m, nws4p, wr4p, R2, (second ? Trb : Tra)[j], xr, yr, zr, xs[j], ys[j], zs[j], tx, ty, tz);
            if (second) { fxb += tx; fyb += ty; fzb += tz; } else { fxa += tx; fya += ty; fza += tz; }
        } else if (left > 0) {
            const int j = lane < left ? c0 + lane : n;
            const double x0 = xs[j], y0 = ys[j], z0 = zs[j];
            pair64<POT, GEN, FOLD>(m, nws4p, wr4p, R2, Tra[j], xa, ya, za, x0, y0, z0, fxa, fya, fza);
            pair64<POT, GEN, FOLD>(m, nws4p, wr4p, R2, Trb[j], xb, yb, zb, x0, y0, z0, fxb, fyb, fzb);
        }
    }
    double Fx = reduce_rows64(fxa, fxb, lane), Fy = reduce_rows64(fya, fyb, lane), Fz = reduce_rows64(fza, fzb, lane);
    if constexpr (FOLD) { Fx *= p.nws4; Fy *= p.nws4; Fz *= p.nws4; }
    // chain terms: lane 4 r + nb evaluates neighbour nb (offsets -2, -1, +1, +2) of row row0 + r; quad sum; to lane r
    {
        const int r = (lane >> 2) & 1, nb = lane & 3;
        double cx = 0, cy = 0, cz = 0;
        if (lane < 8 && p.kind != 4) chain64(m, p, wr4, R2, xs, ys, zs, row0 + r, nb < 2 ? nb - 2 : nb - 1, cx, cy, cz);
        cx += dpp_mov64<0xB1>(cx); cy += dpp_mov64<0xB1>(cy); cz += dpp_mov64<0xB1>(cz);
        cx += dpp_mov64<0x4E>(cx); cy += dpp_mov64<0x4E>(cy); cz += dpp_mov64<0x4E>(cz);
        const double ox = dpp_mov64<0x12C>(cx), oy = dpp_mov64<0x12C>(cy), oz = dpp_mov64<0x12C>(cz);   // row_ror:12 = lane + 4
        if (lane == 0) { Fx += cx; Fy += cy; Fz += cz; }
        if (lane == 1) { Fx += ox; Fy += oy; Fz += oz; }
    }
    // ---- lanes 0, 1 finish one row each (the CPU restatement's update, c3o_md_step / c3o_fire_step) ----
    const double lam = scal[0], cm0 = scal[1], cm1 = scal[2], cm2 = scal[3], keep = scal[4], mix = scal[5];
    st.dt = scal[6];
    double q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    if (lane < kRows64 && row < n) {
        const size_t ix = roff + row, iy = ix + np, iz = iy + np;
        const double x0 = xs[row], y0 = ys[row], z0 = zs[row];
        double vx, vy, vz, xn, yn, zn;
        if (p.kind == 4) {                              // MD begin: Maxwell velocities, no move
            vx = v0x; vy = v0y; vz = v0z; xn = x0; yn = y0; zn = z0;
            q0 = vx * vx + vy * vy + vz * vz; q1 = vx; q2 = vy; q3 = vz;
        } else if (p.kind == 0 || p.kind == 1) {
            const double acc = p.acc;
            vx = lam * (v0x - cm0) + acc * Fx; vy = lam * (v0y - cm1) + acc * Fy; vz = lam * (v0z - cm2) + acc * Fz;
            xn = x0 + p.dt * vx; yn = y0 + p.dt * vy; zn = z0 + p.dt * vz;
            q0 = vx * vx + vy * vy + vz * vz; q1 = vx; q2 = vy; q3 = vz;
        } else if (p.kind == 5 || p.kind == 6) {
            const double ms2 = fp.max_step * fp.max_step;
            auto clamp_scale = [&](double d2) {
                double scl = 1.0;
                if (d2 > ms2) { double dd, hh; sqrt_hrsqrt64(d2, dd, hh); hh = fma(fma(-dd, hh, 0.5), hh, hh); scl = fp.max_step * (hh + hh); }
                return scl;
            };
            q1 = Fx * Fx + Fy * Fy + Fz * Fz;
            if (p.kind == 5) {
                double sx = lam * v0x, sy = lam * v0y, sz = lam * v0z;
                const double scp = clamp_scale(sx * sx + sy * sy + sz * sz);
                sx *= scp; sy *= scp; sz *= scp;
                const double yx = v0x - Fx, yy = v0y - Fy, yz = v0z - Fz;
                q0 = sx * yx + sy * yy + sz * yz; q2 = yx * yx + yy * yy + yz * yz; q3 = sx * sx + sy * sy + sz * sz;
            }
            const double dxs = mix * Fx, dys = mix * Fy, dzs = mix * Fz;
            const double scl = clamp_scale(dxs * dxs + dys * dys + dzs * dzs);
            xn = x0 + scl * dxs; yn = y0 + scl * dys; zn = z0 + scl * dzs;
            vx = Fx; vy = Fy; vz = Fz;
        } else {
            q0 = v0x * Fx + v0y * Fy + v0z * Fz; q1 = Fx * Fx + Fy * Fy + Fz * Fz; q2 = v0x * v0x + v0y * v0y + v0z * v0z;
            const double acc = st.dt * p.kacc;
            vx = keep * v0x + mix * Fx; vy = keep * v0y + mix * Fy; vz = keep * v0z + mix * Fz;
            vx += acc * Fx; vy += acc * Fy; vz += acc * Fz;
            const double dxs = st.dt * vx, dys = st.dt * vy, dzs = st.dt * vz;
            const double d2 = dxs * dxs + dys * dys + dzs * dzs;
            double scl = 1.0;
            if (d2 > fp.max_step * fp.max_step) { double dd, hh; sqrt_hrsqrt64(d2, dd, hh); hh = fma(fma(-dd, hh, 0.5), hh, hh); scl = fp.max_step * (hh + hh); }
            xn = x0 + scl * dxs; yn = y0 + scl * dys; zn = z0 + scl * dzs;
        }
        xout[ix] = xn; xout[iy] = yn; xout[iz] = zn;
        vout[ix] = vx; vout[iy] = vy; vout[iz] = vz;
    }
    if (lane < kRows64) { double* q = rowq + 4 * (wave * kRows64 + lane); q[0] = q0; q[1] = q1; q[2] = q2; q[3] = q3; }
    __syncthreads();
    if (tid < 4) {                                      // tile sums, fixed tree ((q0+q1)+(q2+q3))+((q4+q5)+(q6+q7))
        const double* q = rowq + tid;
        pout[((size_t)rep * m.ntiles + tile) * 4 + tid] = ((q[0] + q[4]) + (q[8] + q[12])) + ((q[16] + q[20]) + (q[24] + q[28]));
    }
}

// T[i][j] = 0.1 * t10 where a restraint exists (|i-j| >= min_sep, t10 > 0), else 0; np columns per row
__global__ __launch_bounds__(256) void k64_targets(int n, int np, int min_sep, double none, const int32_t* __restrict__ t10, double* __restrict__ T) {
    const int i = blockIdx.x;
    for (int j = threadIdx.x; j < np; j += 256) {
        double v = none;
        if (j < n) {
            const int sep = j > i ? j - i : i - j;
            const int32_t t = t10[(size_t)i * n + j];
            if (sep >= min_sep && t > 0) v = 0.1 * t;
        }
        T[(size_t)i * np + j] = v;
    }
}

// fp32 SoA buffers [nrep][3][npad] <-> fp64 SoA state [nrep][3][np] (the solver's read-back, energies and scoring work on the fp32 copy)
__global__ __launch_bounds__(256) void k64_import(int n, int npad, int np, const float* __restrict__ Xf, double* __restrict__ X0,
                                                 double* __restrict__ X1, double* __restrict__ V0, double* __restrict__ V1) {
    const int rep = blockIdx.x;
    for (int k = threadIdx.x; k < 3 * np; k += 256) {
        const int c = k / np, i = k - c * np;
        // pad beads far away and apart from each other (repel and NOE terms vanish), as in the fp32 layout
        const double x = i < n ? (double)Xf[((size_t)rep * 3 + c) * npad + i] : (double)kPadCoord * (c + 1) + 16.0 * (i - n);
        X0[(size_t)rep * 3 * np + k] = x; X1[(size_t)rep * 3 * np + k] = x;
        V0[(size_t)rep * 3 * np + k] = 0.0; V1[(size_t)rep * 3 * np + k] = 0.0;
    }
}
__global__ __launch_bounds__(256) void k64_export(int n, int npad, int np, int ntiles, const double* __restrict__ X, const double* __restrict__ V,
                                                 const double* __restrict__ P, float* __restrict__ Xf, float* __restrict__ Vf, float* __restrict__ Pf) {
    const int rep = blockIdx.x;
    for (int k = threadIdx.x; k < 3 * n; k += 256) {
        const int c = k / n, i = k - c * n;
        Xf[((size_t)rep * 3 + c) * npad + i] = (float)X[((size_t)rep * 3 + c) * np + i];
        Vf[((size_t)rep * 3 + c) * npad + i] = (float)V[((size_t)rep * 3 + c) * np + i];
    }
    // the per-tile sums go where the host looks for them (max RMS force of the minimiser, finiteness)
    for (int t = threadIdx.x; t < 4 * ntiles; t += 256) Pf[(size_t)rep * ntiles * 4 + t] = (float)P[(size_t)rep * ntiles * 4 + t];
}

// ---- host side ---------------------------------------------------------------------------------------
int cols64(int n) { return (n + kColPad64 - 1) / kColPad64 * kColPad64; }
size_t fire_state64_bytes() { return sizeof(FireState64); }

static Model64 model64(const DevModel& d, const double* host) {
    // host[]: s_noe, rswitch, asym, masym, mrswitch, k_bond, b0, k_ang, a0, r0_rep, k_rep, mass, fbeta, min_sep, msoexp
    Model64 m;
    m.n = d.n; m.np = cols64(d.n); m.ntiles = d.ntiles;
    m.min_sep = (int)host[13]; m.noe_pot = d.noe_pot; m.rep_sep = d.rep_sep; m.ang_mode = d.ang_mode; m.mexp = (int)host[14] == 2 ? 2 : 1;
    m.s_noe = host[0]; m.rs = host[1];
    m.tail_c = host[2] * host[1]; m.tail_b = (m.tail_c - 2.0 * m.rs) * m.rs * m.rs;
    // lower side beyond mrs: dE/dD = mtail_c - mtail_b / D^(mexp + 1)
    m.mrs = host[4]; m.mtail_c = host[3]; m.mtail_b = (m.mtail_c - 2.0 * m.mrs) * m.mrs * m.mrs * (m.mexp == 2 ? m.mrs : 1.0);
    m.nmrs4 = -(m.mrs * m.mrs) * (m.mrs * m.mrs);
    if (m.noe_pot == 4 && !(m.mexp == 2 && m.mtail_c == 0.0 && m.tail_b == 0.0 && m.tail_c == 2.0 * m.rs)) m.noe_pot = 3;   // (cannot happen: same test in doubles)
    m.k_bond = host[5]; m.b0 = host[6]; m.k_ang = host[7]; m.a0 = host[8]; m.r0_rep = host[9]; m.k_rep = host[10]; m.mass = host[11]; m.fbeta = host[12];
    { const int ndf = 3 * d.n - 3; m.t_fac = m.mass / kAccel64 / ((ndf > 0 ? ndf : 1) * kBoltz64); m.inv_n = 1.0 / d.n; }
    return m;
}
static bool general64(const Model64& m) {
    if (!(m.tail_b == 0.0 && m.tail_c == 2.0 * m.rs)) return true;
    return m.noe_pot == 3 && !(m.mtail_b == 0.0 && m.mtail_c == 2.0 * m.mrs);     // potential 4 has a fast form of its own
}

hipError_t launch_step64(const DevModel& d, const double* model_host, const double* step_host, const double* fire_host, int fire_n_min,
                         const Buffers64& b, int parity, hipStream_t s) {
    const Model64 m = model64(d, model_host);
    Step64 p;   // step_host[]: kind, dt, w_all, w_vdw, repel_s, t_bath
    p.kind = (int)step_host[0]; p.dt = step_host[1]; p.w_all = step_host[2]; p.w_vdw = step_host[3]; p.repel_s = step_host[4]; p.t_bath = step_host[5];
    p.R2 = (p.repel_s * m.r0_rep) * (p.repel_s * m.r0_rep);
    p.wr4 = p.w_vdw * m.k_rep * 4.0;
    p.nws4 = -4.0 * p.w_all * m.s_noe;
    p.wq = p.nws4 != 0.0 ? p.wr4 / p.nws4 : 0.0;
    p.acc = p.dt * kAccel64 / m.mass;
    p.kb4 = -p.w_all * 4.0 * m.k_bond; p.ka4 = -p.w_all * 4.0 * m.k_ang;
    p.kacc = kAccel64 / m.mass;
    p.a0sq = m.a0 * m.a0;
    Fire64 fp;
    fp.dt_start = fire_host[0]; fp.dt_max = fire_host[1]; fp.f_inc = fire_host[2]; fp.f_dec = fire_host[3]; fp.alpha_start = fire_host[4];
    fp.f_alpha = fire_host[5]; fp.max_step = fire_host[6]; fp.n_min = fire_n_min;
    const int q = parity ^ 1;
    const dim3 grid(d.ntiles, d.nrep_g), blk(kBlock64);
    const size_t lds = sizeof(double) * ((size_t)3 * m.np + 4 * kTileRows + 8);
    FireState64* sin = reinterpret_cast<FireState64*>(b.S[parity]);
    FireState64* sout = reinterpret_cast<FireState64*>(b.S[q]);
#define C3D_STEP64(POT, GEN) hipLaunchKernelGGL((k64_step<POT, GEN>), grid, blk, lds, s, m, p, fp, d.rep_base, b.T, b.X[parity], b.V[parity], b.Vinit, \
                                                b.P[parity], sin, b.X[q], b.V[q], b.P[q], sout)
    if (!general64(m)) {
        if (m.noe_pot == 0) C3D_STEP64(0, false); else if (m.noe_pot == 1) C3D_STEP64(1, false); else if (m.noe_pot == 3) C3D_STEP64(3, false);
        else if (m.noe_pot == 4) {
            if (p.w_all != 0.0) hipLaunchKernelGGL((k64_step<4, false, true>), grid, blk, lds, s, m, p, fp, d.rep_base, b.T, b.X[parity], b.V[parity], b.Vinit, b.P[parity],
                                                   sin, b.X[q], b.V[q], b.P[q], sout);
            else C3D_STEP64(4, false);
        } else C3D_STEP64(2, false);
    } else {
        if (m.noe_pot == 0) C3D_STEP64(0, true); else if (m.noe_pot == 1) C3D_STEP64(1, true); else if (m.noe_pot == 3) C3D_STEP64(3, true); else C3D_STEP64(2, true);
    }
#undef C3D_STEP64
    return hipGetLastError();
}
hipError_t launch_targets64(const DevModel& d, const double* model_host, int min_sep, const int32_t* t10, double* T, hipStream_t s) {
    const Model64 m = model64(d, model_host);
    const double none = (m.noe_pot == 4 && !general64(m)) ? kNoTarget64 : 0.0;        // what pair64 of the kernel that will run expects
    hipLaunchKernelGGL(k64_targets, dim3(d.n), dim3(256), 0, s, d.n, cols64(d.n), min_sep, none, t10, T);
    return hipGetLastError();
}
hipError_t launch_import64(const DevModel& d, const float* Xf, const Buffers64& b, hipStream_t s) {
    hipLaunchKernelGGL(k64_import, dim3(d.nrep), dim3(256), 0, s, d.n, d.npad, cols64(d.n), Xf, b.X[0], b.X[1], b.V[0], b.V[1]);
    return hipGetLastError();
}
hipError_t launch_export64(const DevModel& d, const Buffers64& b, int parity, float* Xf, float* Vf, float* Pf, hipStream_t s) {
    hipLaunchKernelGGL(k64_export, dim3(d.nrep), dim3(256), 0, s, d.n, d.npad, cols64(d.n), d.ntiles, b.X[parity], b.V[parity], b.P[parity], Xf, Vf, Pf);
    return hipGetLastError();
}

hipError_t preload_f64_unit() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k64_import));
}

}  // namespace c3d
