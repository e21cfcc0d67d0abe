// c3d_sym.hip — K2/K3/K4 for LARGE bead counts (N > ~1000, BASELINE config 5): every pair evaluated ONCE (gfx950).
// OPT-IN (option "symmetric" = 1), not the default: measured at N = 2500 x 8 replicas it takes 41 us per step against
// 26 us for k_step.  The pair count drops to 0.6 x (tile granularity on the diagonal), but a pair costs three more
// fused multiply-adds for the column side, the three launches and the slab traffic add ~6 us, and 4-wave workgroups
// with a 1600-instruction stream per wave fill the SIMDs worse than k_step's short ones.
//
// The step kernels are bound by VALU issue, i.e. by how many pair terms they evaluate; k_step walks the full row of
// every bead, so each pair (i, j) is computed twice.  Here the pair matrix is cut into 128-row x 256-column tiles and
// only the tiles on or above the diagonal are visited; a pair term c * (x_i - x_j) goes to BOTH beads:
//
//   k_pairs_sym   one workgroup (4 waves) per tile and replica; wave w owns 16 rows of the tile, lanes run along the
//                 columns (4 per lane, as in k_step).  Row side: butterfly sum per 4 rows -> rowpart[q][row].  Column
//                 side: the lane keeps -c * d for its 4 columns in registers over all 32 rows; the 4 waves' values are
//                 summed through LDS in wave order -> colpart[g][column].  Tiles crossing the diagonal keep j > i only.
//   k_update_sym  one wave per 64 rows (lane = row): force = sum_q rowpart + sum_g colpart (fixed order) + chain terms,
//                 then the same step scalars / row update / 8-row tile sums as k_step (c3d_step_core.h).
//
// No atomics, no inter-workgroup hand-off inside a launch: deterministic, independent of placement and GPU count.  The
// summation order differs from k_step's (so do the last bits); parity is checked in fp64 by tests/test_gpu_large.py.
// Slabs: rowpart [replica][Q][3][npad], colpart [replica][G][3][npad], Q = npad / 256, G = ceil(n / 64): 1.5 MB per
// replica at N = 2500, written and read once per step.
#include "c3d_step_core.h"

#pragma clang fp contract(off)

namespace c3d {

constexpr int kSymRows = 64, kSymCols = 256, kSymRPW = 4, kSymRowsPerWave = 16;

template <int POT, bool RS1, bool DIAG>
__global__ __launch_bounds__(256) void k_pairs_sym(const float* __restrict__ xin, const float* __restrict__ tgt,
                                                   const int2* __restrict__ tiles, float* __restrict__ rowpart,
                                                   float* __restrict__ colpart, const int Q, const int G, const DevModel m,
                                                   const DevStep p) {
    __shared__ __attribute__((aligned(16))) float xr[3][kSymRows];      // coordinates of the tile's rows
    __shared__ __attribute__((aligned(16))) float xc[3][kSymCols];      // ... and of its columns
    __shared__ __attribute__((aligned(16))) float cw[4][3][kSymCols];   // column sums of the four waves
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int2 tile = tiles[blockIdx.x];
    const int g = tile.x, q = tile.y;
    const int rep = m.rep_base + blockIdx.y;
    const int npad = m.npad;
    const float* X = xin + (size_t)rep * 3 * npad;
    const int r_base = g * kSymRows, c_base = q * kSymCols;
    for (int k = tid; k < 3 * kSymRows; k += 256) {
        const int comp = k / kSymRows, r = k - comp * kSymRows;
        xr[comp][r] = X[comp * npad + min(r_base + r, npad - 1)];
    }
    for (int k = tid; k < 3 * kSymCols; k += 256) {
        const int comp = k / kSymCols, c = k - comp * kSymCols;
        xc[comp][c] = X[comp * npad + c_base + c];
    }
    __syncthreads();
    const float4 xj = *reinterpret_cast<const float4*>(&xc[0][4 * lane]);
    const float4 yj = *reinterpret_cast<const float4*>(&xc[1][4 * lane]);
    const float4 zj = *reinterpret_cast<const float4*>(&xc[2][4 * lane]);
    float ccx[4] = {0, 0, 0, 0}, ccy[4] = {0, 0, 0, 0}, ccz[4] = {0, 0, 0, 0};   // column-side sums of this lane's four columns
    const int j0 = c_base + 4 * lane;
    float* rp = rowpart + (((size_t)rep * Q + q) * 3) * npad;
    for (int ch = 0; ch < kSymRowsPerWave / kSymRPW; ++ch) {
        const int lr0 = wave * kSymRowsPerWave + ch * kSymRPW;           // first row of the chunk inside the tile
        float fx[kSymRPW], fy[kSymRPW], fz[kSymRPW];
#pragma unroll
        for (int r = 0; r < kSymRPW; ++r) {
            const int i = r_base + lr0 + r;
            const float xi = xr[0][lr0 + r], yi = xr[1][lr0 + r], zi = xr[2][lr0 + r];
            const float4 tv = *reinterpret_cast<const float4*>(tgt + (size_t)min(i, m.n - 1) * npad + j0);
            float4 mw = noe_weights(p, tv);
            if (i >= m.n) mw = make_float4(0, 0, 0, 0);
            fx[r] = fy[r] = fz[r] = 0.0f;
            const float tt[4] = {tv.x, tv.y, tv.z, tv.w}, ww[4] = {mw.x, mw.y, mw.z, mw.w};
            const float xx[4] = {xj.x, xj.y, xj.z, xj.w}, yy[4] = {yj.x, yj.y, yj.z, yj.w}, zz[4] = {zj.x, zj.y, zj.z, zj.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // one pair term (the arithmetic of pair_term, c3d_step_core.h), applied to both beads
                const float dx = xi - xx[k], dy = yi - yy[k], dz = zi - zz[k];
                const float r2 = fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, 1e-12f)));
                const float rinv = __builtin_amdgcn_rsqf(r2);
                const float u = fmaf(-tt[k], rinv, 1.0f);
                float lim;
                if constexpr (RS1) lim = rinv; else lim = m.rs * rinv;
                float s;
                if constexpr (POT == 1) s = fminf(u, lim);
                else if constexpr (POT == 0) s = fminf(fmaxf(u, -lim), lim);
                else if constexpr (POT == 3) s = __builtin_amdgcn_fmed3f(u, m.nmrs * rinv, lim);
                else if constexpr (POT == 4) {      // lower side soft beyond mrs (c3d_step_core.h pair_term, in this kernel's variable u = (d - t) / d)
                    const float kr = m.nmrs * rinv, z = kr * __builtin_amdgcn_rcpf(fabsf(u));
                    s = __builtin_amdgcn_fmed3f(u, ((kr * z) * z) * fabsf(z), lim);
                }
                else s = u;
                float c = ww[k] * s;
                float q01;
                asm("v_fma_f32 %0, -%1, %2, 1.0 clamp" : "=v"(q01) : "v"(r2), "v"(p.inv_rep_r2));
                c = fmaf(p.w_rep4r2, q01, c);
                // padding rows / columns are 1e4 A away (no repel, no target); rows beyond n are switched off entirely
                if (i >= m.n) c = 0.0f;
                if constexpr (DIAG) { if (j0 + k <= i) c = 0.0f; }       // the tile crosses the diagonal: each pair once
                fx[r] = fmaf(c, dx, fx[r]); fy[r] = fmaf(c, dy, fy[r]); fz[r] = fmaf(c, dz, fz[r]);
                ccx[k] = fmaf(-c, dx, ccx[k]); ccy[k] = fmaf(-c, dy, ccy[k]); ccz[k] = fmaf(-c, dz, ccz[k]);
            }
        }
        const float Fx = reduce_rows<kSymRPW>(fx, lane), Fy = reduce_rows<kSymRPW>(fy, lane), Fz = reduce_rows<kSymRPW>(fz, lane);
        const int i = r_base + lr0 + lane;
        if (lane < kSymRPW && i < npad) { rp[i] = Fx; rp[npad + i] = Fy; rp[2 * npad + i] = Fz; }
    }
    // column side: the four waves' sums, in wave order
    *reinterpret_cast<float4*>(&cw[wave][0][4 * lane]) = make_float4(ccx[0], ccx[1], ccx[2], ccx[3]);
    *reinterpret_cast<float4*>(&cw[wave][1][4 * lane]) = make_float4(ccy[0], ccy[1], ccy[2], ccy[3]);
    *reinterpret_cast<float4*>(&cw[wave][2][4 * lane]) = make_float4(ccz[0], ccz[1], ccz[2], ccz[3]);
    __syncthreads();
    float* cp = colpart + (((size_t)rep * G + g) * 3) * npad + c_base;
    for (int k = tid; k < 3 * kSymCols; k += 256) {
        const int comp = k / kSymCols, c = k - comp * kSymCols;
        cp[comp * npad + c] = (cw[0][comp][c] + cw[1][comp][c]) + (cw[2][comp][c] + cw[3][comp][c]);
    }
}

// one wave per 64 rows of one replica, lane = row
template <int DUMMY>
__global__ __launch_bounds__(64) void k_update_sym(const float* __restrict__ pin, const float* __restrict__ xin, const float* __restrict__ vin,
                                                  const float* __restrict__ vinit, const FireState* __restrict__ sin,
                                                  const float* __restrict__ rowpart, const float* __restrict__ colpart,
                                                  float* __restrict__ xout, float* __restrict__ vout, float* __restrict__ pout,
                                                  FireState* __restrict__ sout, const int Q, const int G, const DevModel m,
                                                  const DevStep p, const DevFire fp) {
    const int lane = threadIdx.x;
    const int rep = m.rep_base + blockIdx.y;
    const int npad = m.npad;
    const int row = blockIdx.x * 64 + lane;
    const bool live = row < m.n;
    const size_t roff = (size_t)rep * 3 * npad;
    const float* X = xin + roff;
    const bool needs_partials = p.kind == 0 || p.kind == 1 || p.kind == 2 || p.kind == 5;
    FireState st;
    st.dt = fp.dt_start; st.alpha = fp.alpha_start; st.npos = 0; st.pad = 0;
    if (p.kind == 2 || p.kind == 5) st = sin[rep];
    float4 psum = make_float4(0, 0, 0, 0);
    if (needs_partials) {
        const float4* pp = reinterpret_cast<const float4*>(pin) + (size_t)rep * m.ntiles;
        for (int t = lane; t < m.ntiles; t += 64) {
            const float4 qq = pp[t];
            psum.x += qq.x; psum.y += qq.y; psum.z += qq.z; psum.w += qq.w;
        }
        psum = wave_sum4(psum);
        psum.x = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(psum.x)));
        psum.y = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(psum.y)));
        psum.z = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(psum.z)));
        psum.w = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(psum.w)));
    }
    const StepScalars sc = step_scalars(m, p, fp, psum, st);
    if ((p.kind == 2 || p.kind == 3 || p.kind == 5 || p.kind == 6) && blockIdx.x == 0 && lane == 0) sout[rep] = st;

    float Fx = 0.0f, Fy = 0.0f, Fz = 0.0f;
    if (p.kind != 4 && live) {
        // row side: the tiles (g, q) of this row's group; column side: every row group whose tiles reach this column block
        const int g = row / kSymRows, qc = row / kSymCols;
        for (int q = (g * kSymRows) / kSymCols; q < Q; ++q) {
            const float* rp = rowpart + (((size_t)rep * Q + q) * 3) * npad + row;
            Fx += rp[0]; Fy += rp[npad]; Fz += rp[2 * npad];
        }
        const int gmax = min(((qc + 1) * kSymCols - 1) / kSymRows, G - 1);   // the last row group whose tile row reaches column block qc
        for (int gg = 0; gg <= gmax; ++gg) {
            const float* cp = colpart + (((size_t)rep * G + gg) * 3) * npad + row;
            Fx += cp[0]; Fy += cp[npad]; Fz += cp[2 * npad];
        }
        // chain terms, (c-2 + c-1) + (c+1 + c+2) as everywhere
        float c0x, c0y, c0z, c1x, c1y, c1z, c2x, c2y, c2z, c3x, c3y, c3z;
        const float* Y = X + npad;
        const float* Z = Y + npad;
        chain_term(m, p, X, Y, Z, row, 0, true, c0x, c0y, c0z);
        chain_term(m, p, X, Y, Z, row, 1, true, c1x, c1y, c1z);
        chain_term(m, p, X, Y, Z, row, 2, true, c2x, c2y, c2z);
        chain_term(m, p, X, Y, Z, row, 3, true, c3x, c3y, c3z);
        Fx += (c0x + c1x) + (c2x + c3x); Fy += (c0y + c1y) + (c2y + c3y); Fz += (c0z + c1z) + (c2z + c3z);
    }
    float4 q = make_float4(0, 0, 0, 0);
    if (live) {
        const size_t ix = roff + row, iy = ix + npad, iz = iy + npad;
        float vx0 = 0.0f, vy0 = 0.0f, vz0 = 0.0f;
        if (p.kind != 3 && p.kind != 6) { const float* vsrc = p.kind == 4 ? vinit : vin; vx0 = vsrc[ix]; vy0 = vsrc[iy]; vz0 = vsrc[iz]; }
        float vx, vy, vz, xn, yn, zn;
        finish_row(m, p, fp, sc, st, Fx, Fy, Fz, X[row], X[npad + row], X[2 * npad + row], vx0, vy0, vz0, xn, yn, zn, vx, vy, vz, q);
        xout[ix] = xn; xout[iy] = yn; xout[iz] = zn;
        vout[ix] = vx; vout[iy] = vy; vout[iz] = vz;
    }
    // 8-row tile sums (tile_sum8's tree over eight consecutive lanes), one float4 per tile
    float4 t = q;
    t.x += dpp_mov<0xB1>(t.x); t.y += dpp_mov<0xB1>(t.y); t.z += dpp_mov<0xB1>(t.z); t.w += dpp_mov<0xB1>(t.w);
    t.x += dpp_mov<0x4E>(t.x); t.y += dpp_mov<0x4E>(t.y); t.z += dpp_mov<0x4E>(t.z); t.w += dpp_mov<0x4E>(t.w);
    t.x += dpp_mov<0x12C>(t.x); t.y += dpp_mov<0x12C>(t.y); t.z += dpp_mov<0x12C>(t.z); t.w += dpp_mov<0x12C>(t.w);
    const int tl = row >> 3;
    if ((lane & 7) == 0 && tl < m.ntiles) reinterpret_cast<float4*>(pout)[(size_t)rep * m.ntiles + tl] = t;
}

// ---- host side ---------------------------------------------------------------------------------------
void sym_geometry(const DevModel& m, int* Q, int* G, int* ntiles_offdiag, int* ntiles_diag) {
    *Q = m.npad / kSymCols;
    *G = (m.n + kSymRows - 1) / kSymRows;
    int od = 0, dg = 0;
    for (int g = 0; g < *G; ++g)
        for (int q = (g * kSymRows) / kSymCols; q < *Q; ++q) (q * kSymCols <= g * kSymRows + kSymRows - 1 ? dg : od) += 1;
    *ntiles_offdiag = od; *ntiles_diag = dg;
}
size_t sym_scratch_floats(const DevModel& m) {
    int Q, G, a, b;
    sym_geometry(m, &Q, &G, &a, &b);
    return (size_t)m.nrep * (Q + G) * 3 * m.npad;
}
// tile list: the off-diagonal tiles first, then the tiles that cross the diagonal (two launches, two kernels)
void sym_tile_list(const DevModel& m, int2* out) {
    int Q, G, od, dg;
    sym_geometry(m, &Q, &G, &od, &dg);
    int a = 0, b = od;
    for (int g = 0; g < G; ++g)
        for (int q = (g * kSymRows) / kSymCols; q < Q; ++q) {
            if (q * kSymCols <= g * kSymRows + kSymRows - 1) out[b++] = make_int2(g, q); else out[a++] = make_int2(g, q);
        }
}

template <int POT, bool RS1>
static hipError_t sym_go(const DevModel& m, const DevStep& p, const DevFire& fp, const DevBuffers& b, int par, const int2* tiles,
                         float* scratch, hipStream_t s) {
    int Q, G, od, dg;
    sym_geometry(m, &Q, &G, &od, &dg);
    float* rowpart = scratch;
    float* colpart = scratch + (size_t)m.nrep * Q * 3 * m.npad;
    const int q = par ^ 1;
    if (p.kind != 4) {
        if (od > 0)
            hipLaunchKernelGGL((k_pairs_sym<POT, RS1, false>), dim3(od, m.nrep_g), dim3(256), 0, s, b.X[par], b.tgt, tiles, rowpart, colpart, Q, G, m, p);
        hipLaunchKernelGGL((k_pairs_sym<POT, RS1, true>), dim3(dg, m.nrep_g), dim3(256), 0, s, b.X[par], b.tgt, tiles + od, rowpart, colpart, Q, G, m, p);
    }
    hipLaunchKernelGGL((k_update_sym<0>), dim3((m.n + 63) / 64, m.nrep_g), dim3(64), 0, s, b.P[par], b.X[par], b.V[par], b.Vinit, b.S[par],
                       rowpart, colpart, b.X[q], b.V[q], b.P[q], b.S[q], Q, G, m, p, fp);
    return hipGetLastError();
}

hipError_t launch_step_sym(const DevModel& m, const DevStep& p, const DevFire& fp, const DevBuffers& b, int parity, const void* tiles,
                           float* scratch, hipStream_t s) {
    const int2* t = reinterpret_cast<const int2*>(tiles);
    const bool rs1 = m.rs == 1.0f;
#define C3D_SYM(POT) return rs1 ? sym_go<POT, true>(m, p, fp, b, parity, t, scratch, s) : sym_go<POT, false>(m, p, fp, b, parity, t, scratch, s)
    switch (m.noe_pot) {
        case 0: C3D_SYM(0);
        case 1: C3D_SYM(1);
        case 3: C3D_SYM(3);
        case 4: C3D_SYM(4);
        default: C3D_SYM(2);
    }
#undef C3D_SYM
}

hipError_t preload_sym_unit() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_update_sym<0>));
}

}  // namespace c3d
