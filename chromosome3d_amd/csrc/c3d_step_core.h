// c3d_step_core.h — device-side building blocks of one SA step, shared by the per-step kernel
// (k_step, c3d_device.hip) and the multi-step cluster kernel (k_cluster, c3d_cluster.hip).  Both kernels
// form every row's force and every sum in the same order, so their trajectories are bit-identical.
#pragma once
#include "c3d_internal.h"

// Every fused multiply-add in this file is written as fmaf(): with implicit contraction the compiler picks
// which product of a*b + c*d to fuse from the surrounding code, and the two kernels would drift apart.
#pragma clang fp contract(off)

namespace c3d {

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
// Wave64 reductions on the VALU only (no ds_bpermute / LDS crossbar round trips):
//   lanes ^1, ^2      v_add_f32_dpp quad_perm
//   lanes -4, -8      v_add_f32_dpp row_ror:4 / row_ror:8   (sum of the 16-lane row, position kept)
//   rows ^1, halves   v_permlane16_swap / v_permlane32_swap  (gfx950) + v_add
// gfx9 DPP controls: quad_perm[a,b,c,d] = a|b<<2|c<<4|d<<6, row_ror:n = 0x120+n.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float xrow_sum(float v) {   // + the other three 16-lane rows, every lane
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__device__ __forceinline__ float wave_sum(float v) {   // total in every lane
    v += dpp_mov<0xB1>(v);          // lanes ^1
    v += dpp_mov<0x4E>(v);          // lanes ^2
    v += dpp_mov<0x124>(v);         // row_ror:4
    v += dpp_mov<0x128>(v);         // row_ror:8
    return xrow_sum(v);
}
// sum over the low RPW lanes of each quad (RPW in {1,2,4}); valid in lane 0
template <int RPW>
__device__ __forceinline__ float quad_sum(float v) {
    if constexpr (RPW >= 2) v += dpp_mov<0xB1>(v);
    if constexpr (RPW >= 4) v += dpp_mov<0x4E>(v);
    return v;
}
// Transposing reduction: a[r] is this lane's partial sum for row r.  Returns, in every lane l, the
// sum over all 64 lanes of a[l & (RPW-1)]: the row selection rides on the first butterfly levels
// (lane parity picks which row a lane keeps), every later level preserves lane & 3.
template <int RPW>
__device__ __forceinline__ float reduce_rows(const float (&a)[RPW], int lane) {
    static_assert(RPW >= 1 && RPW <= 4, "rows per wave must be 1..4");
    float k;
    if constexpr (RPW == 3) {   // the four-row tree with an empty fourth row: rows 0..2 get the bits they get there
        const float a4[4] = {a[0], a[1], a[2], 0.0f};
        return reduce_rows<4>(a4, lane);
    } else if constexpr (RPW == 1) {
        k = a[0];
        k += dpp_mov<0xB1>(k);
        k += dpp_mov<0x4E>(k);
    } else if constexpr (RPW == 2) {
        const bool b0 = lane & 1;
        k = b0 ? a[1] : a[0];
        const float s = b0 ? a[0] : a[1];
        k += dpp_mov<0xB1>(s);
        k += dpp_mov<0x4E>(k);
    } else {
        const bool b0 = lane & 1, b1 = lane & 2;
        float k01 = b0 ? a[1] : a[0];
        const float s01 = b0 ? a[0] : a[1];
        float k23 = b0 ? a[3] : a[2];
        const float s23 = b0 ? a[2] : a[3];
        k01 += dpp_mov<0xB1>(s01);
        k23 += dpp_mov<0xB1>(s23);
        k = b1 ? k23 : k01;
        const float s = b1 ? k01 : k23;
        k += dpp_mov<0x4E>(s);
    }
    k += dpp_mov<0x124>(k);
    k += dpp_mov<0x128>(k);
    return xrow_sum(k);
}

template <int POT, bool GEN>
__device__ __forceinline__ float noe_grad(float delta, const DevModel& m) {
    if constexpr (!GEN) {  // tail slope == 2*rs (and 2*mrs on the lower side of POT 3), b == 0
        if constexpr (POT == 1) return 2.0f * fminf(delta, m.rs);
        else if constexpr (POT == 0) return 2.0f * fminf(fmaxf(delta, -m.rs), m.rs);
        else if constexpr (POT == 4) { const float D = fmaxf(-delta, m.mrs); const float q = m.mrs / D; return 2.0f * fminf(fmaxf(delta, -m.mrs * q * q * q), m.rs); }
        else if constexpr (POT == 3) return 2.0f * fminf(fmaxf(delta, -m.mrs), m.rs);
        else return 2.0f * delta;
    } else {
        const float ad = fabsf(delta);
        const float tail = m.tail_c - m.tail_b / (ad * ad);
        if constexpr (POT == 1) return delta > m.rs ? tail : 2.0f * delta;
        else if constexpr (POT == 0) return ad > m.rs ? copysignf(tail, delta) : 2.0f * delta;
        else if constexpr (POT == 3 || POT == 4) {          // lower side: mtail_c - mtail_b / D^(mexp + 1)
            const float dp = m.mexp == 2 ? ad * ad * ad : ad * ad;
            return delta > m.rs ? tail : (delta < -m.mrs ? m.mtail_b / dp - m.mtail_c : 2.0f * delta);
        }
        else return 2.0f * delta;
    }
}

// ---------------------------------------------------------------------------------------------
// K2: forces on RPW consecutive rows, one wave, lanes across j.  The target row loads are
// software-pipelined one j-block ahead (tv = block being computed, tn = block in flight); the
// caller issues the first block before it stages xyz in LDS (tile_prefetch) so that their
// L2 latency overlaps the staging.  On return lane l holds the force on row row0 + (l & (RPW-1)).
// ---------------------------------------------------------------------------------------------
// One "column block" = 256 columns: lane l owns columns 256*jb + 4l .. 4l+3, so every target
// load is a 16-byte global_load_dwordx4 and every coordinate read a ds_read_b128 (dword loads
// are address-rate bound in the texture path: 1 pair per load instruction-lane starves the VALU).
// first column of lane `lane` in block jb, and how many consecutive columns it owns there (4; wl in the last block)
__device__ __forceinline__ int block_col0(const DevModel& m, int jb, int lane, int& width) {
    const bool last = 256 * (jb + 1) >= m.npad;
    width = last ? m.wl : 4;
    return 256 * jb + width * lane;
}
template <int RPW, bool NC = true>
__device__ __forceinline__ void tile_prefetch(const DevModel& m, const float* __restrict__ tgt, int row0, int lane,
                                              int jb, float4 (&tv)[RPW]) {
    // the four-column form first and unconditionally (always in bounds): callers put this load ahead of everything else, and
    // it is the right one unless block jb is a narrow last block
#pragma unroll
    for (int r = 0; r < RPW; ++r)
        tv[r] = *reinterpret_cast<const float4*>(tgt + (size_t)min(row0 + r, m.n - 1) * m.npad + 256 * jb + 4 * lane);
    if (NC && m.wl != 4 && 256 * (jb + 1) >= m.npad) {
        const int j = 256 * jb + m.wl * lane;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const float* t = tgt + (size_t)min(row0 + r, m.n - 1) * m.npad + j;
            tv[r] = make_float4(t[0], m.wl > 1 ? t[1] : 0.0f, m.wl > 2 ? t[2] : 0.0f, 0.0f);   // 4-byte aligned only
        }
    }
}
// coordinates of a lane's columns in block jb (a component beyond the lane's width is never used)
__device__ __forceinline__ void block_coords(const DevModel& m, const float* xs, const float* ys, const float* zs, int jb, int lane,
                                             int& width, float4& xj, float4& yj, float4& zj) {
    const int j = block_col0(m, jb, lane, width);
    if (width == 4) {
        xj = *reinterpret_cast<const float4*>(xs + j); yj = *reinterpret_cast<const float4*>(ys + j); zj = *reinterpret_cast<const float4*>(zs + j);
    } else {
        xj = make_float4(xs[j], width > 1 ? xs[j + 1] : 0.0f, width > 2 ? xs[j + 2] : 0.0f, 0.0f);
        yj = make_float4(ys[j], width > 1 ? ys[j + 1] : 0.0f, width > 2 ? ys[j + 2] : 0.0f, 0.0f);
        zj = make_float4(zs[j], width > 1 ? zs[j + 1] : 0.0f, width > 2 ? zs[j + 2] : 0.0f, 0.0f);
    }
}

// One pair term.  F_i += c * (x_i - x_j) with c = -(dE/dd)/d.  Written to minimise VALU issue slots: the solver is bound by
// the vector ALU (a wave issues one VALU instruction per ~5 ns, the pipe takes 1.2-3.5 ns per instruction by its form;
// tools/microbench/valu_forms, pair_loop_v2), so every instruction of this function costs ~1.5 ns x pairs / lanes on the chip.
//   r2 carries a +1e-12 guard inside the fma chain (no separate max);
//   repel: max(0, R2 - r2) = R2 * clamp01(1 - r2/R2) is ONE v_fma_f32 with the clamp output modifier.
// Clamp form (!GEN: tail slope 2 rs above, 2 mrs below for POT 3), 15 instructions:
//   with u = (d - t)/d = 1 - t/d the soft-square's clamp acts on u directly, -(dE/dd)/d = W clamp(u, -mrs/d, rs/d), W = -2 w S
//   (d itself is never formed).  Divided through by rs:  u/rs = a - b/d  with b = t/rs and a = 1/rs per pair (b = a = 0 where
//   no restraint exists: the term vanishes by itself, no mask multiply), bounds (-mrs/rs)/d and 1/d = rinv itself; and the
//   weight W rs is left out of the pair term altogether: c' = clamp(...) + kq * q01 with kq = (repel weight)/(W rs), the
//   row's sum of c' * (x_i - x_j) is multiplied by W rs ONCE after the reduction (row_total).  b and a are per-pair
//   constants of a whole launch (cluster kernel: registers / LDS) or two instructions from the streamed target (k_step).
// General tails (GEN): v = target in Angstrom, mw = W where restrained else 0, dE/dd from noe_grad; nothing is left out.
// the three run constants of the clamp form, held in VGPRs by the tile functions (an SGPR source costs the instruction a
// second issue cycle on gfx950: 2.25 against 1.17 ns, tools/microbench/valu_forms)
struct PairK {
    float inv_rep_r2, kq, nm_rs;
};
__device__ __forceinline__ PairK pair_k(const DevModel& m, const DevStep& p) {
    PairK k{p.inv_rep_r2, p.kq, m.nm_rs};
    asm volatile("" : "+v"(k.inv_rep_r2), "+v"(k.kq), "+v"(k.nm_rs));
    return k;
}
template <int POT, bool GEN>
__device__ __forceinline__ void pair_term(const DevModel& m, const DevStep& p, const PairK& k, float v, float mw, float dx, float dy,
                                          float dz, float& fx, float& fy, float& fz) {
    const float r2 = fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, 1e-12f)));
    const float rinv = __builtin_amdgcn_rsqf(r2);
    float q01;   // clamp01(1 - r2/R2): the clamp is an output modifier of the fma (hipcc has no builtin for it)
    asm("v_fma_f32 %0, -%1, %2, 1.0 clamp" : "=v"(q01) : "v"(r2), "v"(k.inv_rep_r2));
    // repel on EVERY column: padding beads are 1e4 A away (q = 0), the self term has dx = 0, and the
    // |i-j| < rep_sep neighbours are taken back out in the chain-term pass below
    float c;
    if constexpr (!GEN && POT == 4) {
        // Lower side square up to D = t - d = mrs, then the CNS soft form with exponent 2 and no asymptote: dE/dD = 2 mrs^4 / D^3.
        // This potential is written in Delta = d - t itself, divided through by MRS (round 4, second form: 19 instructions; the first
        // one bounded u = Delta / (rs d) and took 20): v = t / mrs, mw = 1 / mrs per pair (both 0: no restraint), so
        //   dl = (d - t) / mrs,  -(dE/dd) / (W mrs) = clamp(dl, -1 / |dl|^3, rs / mrs)      (W = -2 w S, applied once per row: w_rs = W mrs)
        // with CONSTANT upper bound (k.nm_rs holds rs / mrs for this potential) and the lower bound -|w|^3, w = 1 / dl: it lies below dl
        // while D < mrs and above it beyond; above the target it is negative and never binds.  dl = 0 (no restraint, or d = t): w = inf,
        // the bound -inf, the term 0.  The division by d rides in the fma that joins the repel term: d = r2 rinv, rcp, two
        // multiplications for the bound, v_med3_f32, one multiplication for the repel weight.  (Forming the bound only in waves that
        // hold a pair that deep — a compare and a scalar branch per pair term — was measured slower: 4.70 against 4.46 us per step at
        // chr1_500kb x 20, profiles/r04_lower_side_forms.txt.)
        const float dl = fmaf(r2 * rinv, mw, -v);
        const float w = __builtin_amdgcn_rcpf(fabsf(dl));
        const float g = __builtin_amdgcn_fmed3f(dl, -((w * w) * w), k.nm_rs);
        c = fmaf(g, rinv, k.kq * q01);
    } else if constexpr (!GEN) {
        const float u = fmaf(-v, rinv, mw);            // (d - t) / (rs d); v = t / rs, mw = 1 / rs (both 0: no restraint)
        float s;
        if constexpr (POT == 1) s = fminf(u, rinv);
        else if constexpr (POT == 0) s = __builtin_amdgcn_fmed3f(u, -rinv, rinv);
        else if constexpr (POT == 3) s = __builtin_amdgcn_fmed3f(u, k.nm_rs * rinv, rinv);   // clamp(u, -mrs/(rs d), 1/d): one v_med3_f32
        else if constexpr (POT == 4) s = 0.0f;          // (own form above)
        else s = u;
        c = fmaf(k.kq, q01, s);
    } else {
        const float s = 0.5f * noe_grad<POT, GEN>(r2 * rinv - v, m) * rinv;   // (dE/dd) / (2 d) without the weights
        c = fmaf(p.w_rep4r2, q01, mw * s);             // 4 w_vdw k_rep R2 * clamp01(1 - r2/R2)
    }
    fx = fmaf(c, dx, fx);
    fy = fmaf(c, dy, fy);
    fz = fmaf(c, dz, fz);
}
// a row's force from the reduced sum of its pair terms and its chain sum: the clamp form's common factor comes back here
// A row's force = its pair sum times the row factor PLUS its chain sum — as ONE explicit fma (round 5).  It used to be written as a
// product and a sum, and the library is built with -ffp-contract=fast, under which the BACKEND may fuse a multiplication with a following
// addition whatever `#pragma clang fp contract(off)` says about the source: every instantiation of every launch form did fuse it, which is
// why they agreed bit for bit — until a restructuring of the pair term changed the instruction selector's mind for the y and z components
// of ONE instantiation (k_cluster<4, 2, 2, 3, false>: 1-ulp differences in 40 % of the rows; profiles/r05_units_of_mrswitch_experiment.txt).
// Written as llvm.fma the fusion is what the source says, in every kernel, under any compiler.
template <bool GEN>
__device__ __forceinline__ float row_total(const DevStep& p, float pair_sum, float chain_sum) {
    if constexpr (GEN) return pair_sum + chain_sum; else return fmaf(p.w_rs, pair_sum, chain_sum);
}

// per-pair constants of four targets (Angstrom, 0 = none).  Clamp form: (t / rs, 1 / rs or 0); general tails: (t, W or 0)
template <bool GEN>
__device__ __forceinline__ float4 pair_b(const DevModel& m, const float4 tv) {
    if constexpr (GEN) return tv; else return make_float4(tv.x * m.inv_rs, tv.y * m.inv_rs, tv.z * m.inv_rs, tv.w * m.inv_rs);
}
template <bool GEN>
__device__ __forceinline__ float4 pair_a(const DevModel& m, const DevStep& p, const float4 tv) {
    const float on = GEN ? p.w_noe2n : m.inv_rs;
    return make_float4(tv.x > 0.0f ? on : 0.0f, tv.y > 0.0f ? on : 0.0f, tv.z > 0.0f ? on : 0.0f, tv.w > 0.0f ? on : 0.0f);
}
// (kept for the symmetric-tile kernels, which evaluate the round-2 form) -2 w S where a restraint exists (target > 0), else 0
__device__ __forceinline__ float4 noe_weights(const DevStep& p, const float4 tv) {
    return make_float4(tv.x > 0.0f ? p.w_noe2n : 0.0f, tv.y > 0.0f ? p.w_noe2n : 0.0f, tv.z > 0.0f ? p.w_noe2n : 0.0f,
                       tv.w > 0.0f ? p.w_noe2n : 0.0f);
}

// the four pair terms of one lane and one row against columns j .. j+3 (per-pair constants tb, ta: pair_b / pair_a)
template <int POT, bool GEN>
__device__ __forceinline__ void pair_quad(const DevModel& m, const DevStep& p, const PairK& k, const float4 tb, const float4 ta, float xi,
                                          float yi, float zi, const float4 xj, const float4 yj, const float4 zj, float& fx, float& fy,
                                          float& fz) {
    pair_term<POT, GEN>(m, p, k, tb.x, ta.x, xi - xj.x, yi - yj.x, zi - zj.x, fx, fy, fz);
    pair_term<POT, GEN>(m, p, k, tb.y, ta.y, xi - xj.y, yi - yj.y, zi - zj.y, fx, fy, fz);
    pair_term<POT, GEN>(m, p, k, tb.z, ta.z, xi - xj.z, yi - yj.z, zi - zj.z, fx, fy, fz);
    pair_term<POT, GEN>(m, p, k, tb.w, ta.w, xi - xj.w, yi - yj.w, zi - zj.w, fx, fy, fz);
}
// the same for a lane that owns only `width` (1..3) columns of the block (wave-uniform): the pair terms of the columns it
// does not own are not formed at all
template <int POT, bool GEN>
__device__ __forceinline__ void pair_quad_w(const DevModel& m, const DevStep& p, const PairK& k, int width, const float4 tb, const float4 ta,
                                            float xi, float yi, float zi, const float4 xj, const float4 yj, const float4 zj, float& fx,
                                            float& fy, float& fz) {
    pair_term<POT, GEN>(m, p, k, tb.x, ta.x, xi - xj.x, yi - yj.x, zi - zj.x, fx, fy, fz);
    if (width > 1) pair_term<POT, GEN>(m, p, k, tb.y, ta.y, xi - xj.y, yi - yj.y, zi - zj.y, fx, fy, fz);
    if (width > 2) pair_term<POT, GEN>(m, p, k, tb.z, ta.z, xi - xj.z, yi - yj.z, zi - zj.z, fx, fy, fz);
}
// Left-over columns (DevModel::nleft <= 8): row `row`'s pair term against column jl0 + c, one (row, c) per lane, eight lanes to
// a row; returns in the FIRST lane of the eight the sum ((t0 + t1) + (t2 + t3)) + ((t4 + t5) + (t6 + t7)) of the row's terms
// (a lane without a column contributes an exact 0).  `tgt` = the row's target row in global memory or nullptr when the
// caller passes the pair constants itself (tb, ta: cluster kernel, loaded once per launch).
template <int POT, bool GEN>
__device__ __forceinline__ void leftover_terms(const DevModel& m, const DevStep& p, const PairK& k, const float* xs, const float* ys,
                                               const float* zs, int row, int c, bool active, float tb, float ta, float& lx, float& ly,
                                               float& lz) {
    lx = ly = lz = 0.0f;
    if (active) {
        const int j = m.jl0 + c;
        pair_term<POT, GEN>(m, p, k, tb, ta, xs[row] - xs[j], ys[row] - ys[j], zs[row] - zs[j], lx, ly, lz);
    }
    lx += dpp_mov<0xB1>(lx); ly += dpp_mov<0xB1>(ly); lz += dpp_mov<0xB1>(lz);
    lx += dpp_mov<0x4E>(lx); ly += dpp_mov<0x4E>(ly); lz += dpp_mov<0x4E>(lz);
    lx += dpp_mov<0x12C>(lx); ly += dpp_mov<0x12C>(ly); lz += dpp_mov<0x12C>(lz);     // row_ror:12 = lane + 4
}

// One chain term: neighbour nb (0..3 = offsets -2,-1,+1,+2) of `row`: pseudo-bond (i,i+-1), pseudo-angle (i,i+-2) and the
// repel take-back for |i-j| < rep_sep (the pair loop applies the repel term to every column).  `active` = the lane
// has a row; returns the force contribution on `row` (0 where the neighbour does not exist).
__device__ __forceinline__ void chain_term(const DevModel& m, const DevStep& p, const float* xs, const float* ys, const float* zs,
                                           int row, int nb, bool active, float& cx, float& cy, float& cz) {
    const int off = nb < 2 ? nb - 2 : nb - 1;          // -2,-1,+1,+2
    const int sep = off < 0 ? -off : off;
    const int jn = row + off;
    cx = cy = cz = 0.0f;
    if (active && row < m.n && jn >= 0 && jn < m.n) {
        const float dx = xs[row] - xs[jn], dy = ys[row] - ys[jn], dz = zs[row] - zs[jn];
        const float r2 = fmaxf(fmaf(dx, dx, fmaf(dy, dy, dz * dz)), 1e-12f);
        const float rinv = __builtin_amdgcn_rsqf(r2);
        float d = r2 * rinv;
        asm("" : "+v"(d));                           // a product that "- r0" follows: pinned, see row_total
        const float k2 = sep == 1 ? m.k_bond2 : m.k_ang2;
        const float r0 = sep == 1 ? m.b0 : m.a0;
        const bool on = sep == 1 || (m.k_ang2 > 0.0f && (m.ang_mode == 1 || d < m.a0));
        float c = on ? -p.w_all * k2 * (d - r0) * rinv : 0.0f;
        if (sep < m.rep_sep) c = fmaf(-p.w_rep4, fmaxf(p.rep_r2 - r2, 0.0f), c);
        cx = c * dx; cy = c * dy; cz = c * dz;
        asm("" : "+v"(cx), "+v"(cy), "+v"(cz));      // products that the quad sum adds up next: pinned, see row_total
    }
}
// sum of the four chain terms of a row held by the four lanes of a quad, (c0 + c1) + (c2 + c3), in every lane of it
__device__ __forceinline__ float quad_chain_sum(float c) {
    c += dpp_mov<0xB1>(c);
    c += dpp_mov<0x4E>(c);
    return c;
}

// The force on a row = [butterfly sum over the 64 lanes of the pair terms, as it comes out in lane r < 4 (the tree's
// association differs from quad to quad)] + [(c-2 + c-1) + (c+1 + c+2)]: every kernel forms it in exactly this order, so a
// row's force has the same bits whatever the geometry (rows per wave, which wave or workgroup owns the row).
// Here (per-step kernel, forces hook): rows row0 .. row0+RPW-1 of one wave.  Chain terms: neighbour nb of row i is
// evaluated by lane 4 (i & 1) + nb, two rows per pass; the quad sums are then brought to lane r = i - row0, the row's
// finisher (lanes 0 .. RPW-1).
template <int POT, int RPW, bool GEN, bool NC = true>
__device__ __forceinline__ void reduce_and_chain(const DevModel& m, const DevStep& p, const float* __restrict__ tgt, const float* xs,
                                                 const float* ys, const float* zs, int row0, int lane, float (&fx)[RPW],
                                                 float (&fy)[RPW], float (&fz)[RPW], float& Fx, float& Fy, float& Fz) {
    Fx = reduce_rows<RPW>(fx, lane);
    Fy = reduce_rows<RPW>(fy, lane);
    Fz = reduce_rows<RPW>(fz, lane);
    if (NC && m.nleft > 0) {
        // left-over columns of this wave's rows: lane 8 r + c evaluates (row r, column jl0 + c); the row's sum comes out in lane
        // 8 r and goes to the row's finisher, lane r
        const int r = lane >> 3, c = lane & 7;
        const int row = row0 + r;
        const bool active = r < RPW && c < m.nleft && row < m.n;
        float tb = 0.0f, ta = 0.0f;
        if (active) {
            const float t = tgt[(size_t)row * m.npad + m.jl0 + c];
            tb = GEN ? t : t * m.inv_rs;
            ta = t > 0.0f ? (GEN ? p.w_noe2n : m.inv_rs) : 0.0f;
        }
        float lx, ly, lz;
        leftover_terms<POT, GEN>(m, p, pair_k(m, p), xs, ys, zs, min(row, m.n - 1), c, active, tb, ta, lx, ly, lz);
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const float ax = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(lx), 8 * q));
            const float ay = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ly), 8 * q));
            const float az = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(lz), 8 * q));
            if (lane == q) { Fx += ax; Fy += ay; Fz += az; }
        }
    }
    const int nb = lane & 3, half = (lane >> 2) & 1;     // lanes 0..3 serve even rows, 4..7 odd rows
    float Cx = 0.0f, Cy = 0.0f, Cz = 0.0f;               // this lane's row's chain sum (exactly one pass delivers it)
#pragma unroll
    for (int pass = 0; pass < (RPW + 1) / 2; ++pass) {
        // the row of {row0 + 2 pass, row0 + 2 pass + 1} whose parity is `half`
        const int rsel = 2 * pass + ((half ^ row0) & 1);
        float cx, cy, cz;
        chain_term(m, p, xs, ys, zs, row0 + rsel, nb, lane < 8 && rsel < RPW, cx, cy, cz);
        cx = quad_chain_sum(cx); cy = quad_chain_sum(cy); cz = quad_chain_sum(cz);
        // lane r < 4 takes its own quad's sum if that is its row's, else the sum of lanes 4..7 (row_ror:12 = lane + 4)
        const float ox = dpp_mov<0x12C>(cx), oy = dpp_mov<0x12C>(cy), oz = dpp_mov<0x12C>(cz);
        const bool own = lane == rsel, other = lane == (rsel ^ 1);   // in lanes 0..3 rsel is the even-quad row of this pass
        Cx = own ? cx : (other ? ox : Cx);
        Cy = own ? cy : (other ? oy : Cy);
        Cz = own ? cz : (other ? oz : Cz);
    }
    Fx = row_total<GEN>(p, Fx, Cx);
    Fy = row_total<GEN>(p, Fy, Cy);
    Fz = row_total<GEN>(p, Fz, Cz);
}

// ---------------------------------------------------------------------------------------------
// The same pair terms two ROWS at a time in the packed fp32 forms (device potential 4, cluster kernel).  gfx950 issues
// v_pk_add/mul/fma_f32 — two results — in 1.9-2.0 ns per SIMD where the scalar forms take 1.2-1.6 ns each and a square
// (v_fmac d,a,a: src0 = src1 in one bank) 2.2-2.45 ns (tools/microbench/valu_packed.hip, profiles/r04_valu_packed_microbench.txt):
// with three compute waves on a SIMD the pair loop is bound by instruction ISSUE, and a packed instruction carries two pair
// terms through one issue slot.  The pair is (row r, row r + 1) against ONE column: the row-side operands (coordinates,
// accumulators, per-pair constants) are natural register pairs, the column's coordinate is one half of the (x_j, x_j+1) pair a
// ds_read_b128 delivers, selected by op_sel.  Every component goes through exactly the operations of pair_term<4, false> in the
// same order (v_pk_fma_f32 is v_fma_f32 per half; the sums over a row's columns keep their order), so a row's force has the SAME
// BITS as from the scalar form: the per-step kernel, the forces hook and the left-over columns keep the scalar form, and the
// bit-identity tests between the launch forms are what checks this code.  16 packed + 6 scalar instructions (v_rsq, v_rcp,
// v_med3: no packed forms) for two pair terms, against 2 x 19.
#ifndef C3D_PAIR_NONE_AS_FAR
#define C3D_PAIR_NONE_AS_FAR 1     // cluster kernel, shipped potential: "no restraint" = a target of 1e30 A, no second per-pair constant (0: measurement builds)
#endif
#ifndef C3D_PACKED_STEP
#define C3D_PACKED_STEP 1          // 0: measurement / test builds with the per-step kernel's pair terms in the scalar form
#endif
typedef float float2v __attribute__((ext_vector_type(2)));
struct PairK2 {
    float2v k0, k1;              // (1e-12, 1 / rep_r2), (kq, rs / mrs): in VGPRs (an SGPR source costs an issue cycle)
};
__device__ __forceinline__ PairK2 pair_k2(const DevModel& m, const DevStep& p) {
    PairK2 k;
    k.k0 = float2v{1e-12f, p.inv_rep_r2}; k.k1 = float2v{p.kq, m.nm_rs};
    asm volatile("" : "+v"(k.k0), "+v"(k.k1));
    return k;
}
// v2 = (t / mrs) of (row r, row r + 1) against the column, mw2 = their (1 / mrs or 0); xi2 .. = the two rows' coordinates;
// xjp .. = the register pair that holds the column's coordinate in its SEL half.  Written as vector arithmetic: the compiler selects
// the packed instructions and folds every broadcast ({a, a} of one half of a pair) into op_sel itself — and, knowing the
// instructions, inserts exactly the wait states gfx950 wants (the result of a packed or transcendental instruction read by the next
// VALU instruction; an asm statement's operands are invisible to its hazard recogniser: a first version with d = r2 rinv written as
// asm read stale registers).  The one asm statement is the clamp modifier (left to itself the compiler clamps each half with a
// v_max_f32).  Handing the two chains of a column pair to one hand-zipped block of 44 instructions (no wait state at all, fixed
// temporaries v96-v127) was measured SLOWER on the same box: 3.92 against 3.82 us per step at chr1_500kb x 20.
template <int SEL>
__device__ __forceinline__ void pair_term2(const PairK2& k, float2v v2, float2v mw2, float2v xi2, float2v yi2, float2v zi2, float2v xjp,
                                           float2v yjp, float2v zjp, float2v& fx2, float2v& fy2, float2v& fz2) {
    const float xc = SEL ? xjp.y : xjp.x, yc = SEL ? yjp.y : yjp.x, zc = SEL ? zjp.y : zjp.x;
    const float2v dx = xi2 - float2v{xc, xc}, dy = yi2 - float2v{yc, yc}, dz = zi2 - float2v{zc, zc};
    float2v r2 = __builtin_elementwise_fma(dz, dz, float2v{k.k0.x, k.k0.x});                       // dz^2 + 1e-12
    r2 = __builtin_elementwise_fma(dy, dy, r2);
    r2 = __builtin_elementwise_fma(dx, dx, r2);
    const float2v rinv = float2v{__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
    float2v q01;                                                                                   // clamp01(1 - r2 / R2)
    asm("v_pk_fma_f32 %0, %1, %2, 1.0 op_sel:[0,1,0] op_sel_hi:[1,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0] clamp" : "=v"(q01) : "v"(r2), "v"(k.k0));
#ifdef C3D_EXP_FOLD_D
    // measurement build only (valid for mrswitch = 1, where mw2 = 1): what coordinates in units of mrswitch would save — d - t in ONE fma
    const float2v dl = __builtin_elementwise_fma(r2, rinv, -v2);
#else
    const float2v d = r2 * rinv;
    const float2v dl = __builtin_elementwise_fma(d, mw2, -v2);
#endif
    const float2v w = float2v{__builtin_amdgcn_rcpf(fabsf(dl.x)), __builtin_amdgcn_rcpf(fabsf(dl.y))};
    const float2v lo = -((w * w) * w);
    const float2v g = float2v{__builtin_amdgcn_fmed3f(dl.x, lo.x, k.k1.y), __builtin_amdgcn_fmed3f(dl.y, lo.y, k.k1.y)};
    const float2v rep = float2v{k.k1.x, k.k1.x} * q01;                                             // kq * q01
    const float2v c = __builtin_elementwise_fma(g, rinv, rep);
    fx2 = __builtin_elementwise_fma(c, dx, fx2);
    fy2 = __builtin_elementwise_fma(c, dy, fy2);
    fz2 = __builtin_elementwise_fma(c, dz, fz2);
}
// DevModel::tgs2 (the pre-scaled targets of row pairs): is this instantiation reading it, and its loads — block jb of the wave's row pair
// as two float4: (a.x, b.x, a.y, b.y), (a.z, b.z, a.w, b.w)
template <int POT, bool GEN, int RPW, bool NC>
__device__ __forceinline__ bool pair_targets_in_use(const DevModel& m) {
    if constexpr (POT == 4 && !GEN && (RPW == 2 || RPW == 4) && !NC) return m.tgs2 != nullptr; else return false;
}
// tv[2q], tv[2q + 1] = the two float4 of row pair q of the wave (a wave whose rows lie beyond the last bead re-reads the last pair: in bounds)
template <int RPW>
__device__ __forceinline__ void pair_targets_prefetch(const DevModel& m, int row0, int lane, int jb, float4 (&tv)[RPW]) {
    if constexpr (RPW == 2 || RPW == 4) {
        const int last = ((m.n + 1) >> 1) - 1;
#pragma unroll
        for (int q = 0; q < RPW / 2; ++q) {
            const int pr = min((row0 >> 1) + q, last);
            const float4* src = reinterpret_cast<const float4*>(m.tgs2 + (((size_t)pr * (m.npad >> 8) + jb) * 64 + lane) * 8);
            tv[2 * q] = src[0]; tv[2 * q + 1] = src[1];
        }
    }
}
// targets streamed from global memory, one column block ahead (per-step kernel)
// NC = false: the instantiation for problems whose last block is a full one and that leave no column over (m.wl == 4,
// m.nleft == 0: every N > 1024 among them) carries none of the narrow-column code — the per-step kernel is launched once per SA
// step and pays for every kilobyte of code it drags along (N = 2500: 26.5 against 27.2 us per step)
// PACKED4 = the packed form also at four rows per wave (two row pairs: the wide-tile step kernel of large problems, NC = false)
template <int POT, bool GEN, int RPW, bool NC = true, bool PACKED = true, bool PACKED4 = false>
__device__ __forceinline__ void tile_forces(const DevModel& m, const DevStep& p, const float* __restrict__ tgt,
                                            const float* xs, const float* ys, const float* zs, int row0, int lane,
                                            float4 (&tv)[RPW], float& Fx, float& Fy, float& Fz) {
    float fx[RPW], fy[RPW], fz[RPW];
    float xi[RPW], yi[RPW], zi[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = min(row0 + r, m.n - 1);
        xi[r] = xs[row]; yi[r] = ys[row]; zi[r] = zs[row];
        fx[r] = fy[r] = fz[r] = 0.0f;
    }
    const int nblk = m.npad >> 8;
    // full blocks: every lane owns four columns; the next block's targets are in flight while this one computes.  With a
    // narrow last block (m.wl < 4) the loop stops one block early and the last block follows with its own column map.
    const int nfull = (!NC || m.wl == 4) ? nblk : nblk - 1;
    if constexpr (POT == 4 && !GEN && RPW == 2 && PACKED && C3D_PACKED_STEP) {
        // device potential 4, two rows per wave: the wave's two rows are the row pair of the packed pair term (pair_term2: same bits
        // as the scalar form below, 11 instead of 19 instructions per pair term); the per-pair constants go straight into register pairs
        const PairK2 k2 = pair_k2(m, p);
        const float2v xi2 = float2v{xi[0], xi[1]}, yi2 = float2v{yi[0], yi[1]}, zi2 = float2v{zi[0], zi[1]};
        float2v fx2 = float2v{0.0f, 0.0f}, fy2 = fx2, fz2 = fx2;
        const float on = m.inv_rs;
#define C3D_PK_COL(C, SEL, XP, YP, ZP)                                                                                             \
        pair_term2<SEL>(k2, float2v{tv[0].C * m.inv_rs, tv[1].C * m.inv_rs}, float2v{tv[0].C > 0.0f ? on : 0.0f, tv[1].C > 0.0f ? on : 0.0f}, \
                        xi2, yi2, zi2, XP, YP, ZP, fx2, fy2, fz2)
        // the targets are streamed TWO column blocks ahead (tv = the block being computed, t1 = the next, the one after it in flight)
        const bool pairs = pair_targets_in_use<POT, GEN, RPW, NC>(m);
        const float* trow[RPW];
        float4 t1[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) trow[r] = tgt + (size_t)min(row0 + r, m.n - 1) * m.npad + 4 * lane;
        if (pairs) pair_targets_prefetch<RPW>(m, row0, lane, min(1, nblk - 1), t1);
        else {
#pragma unroll
            for (int r = 0; r < RPW; ++r) t1[r] = *reinterpret_cast<const float4*>(trow[r] + 256 * min(1, nblk - 1));
        }
        const float2v on2 = float2v{on, on};
        for (int jb = 0; jb < nfull; ++jb) {
            float4 t2[RPW];
            const int jn = min(jb + 2, nblk - 1);           // the last blocks re-read the last one (in bounds)
            if (pairs) pair_targets_prefetch<RPW>(m, row0, lane, jn, t2);
            else {
#pragma unroll
                for (int r = 0; r < RPW; ++r) t2[r] = *reinterpret_cast<const float4*>(trow[r] + 256 * jn);
            }
            const int j = 256 * jb + 4 * lane;
            const float4 xj = *reinterpret_cast<const float4*>(xs + j);
            const float4 yj = *reinterpret_cast<const float4*>(ys + j);
            const float4 zj = *reinterpret_cast<const float4*>(zs + j);
            const float2v x01 = float2v{xj.x, xj.y}, x23 = float2v{xj.z, xj.w}, y01 = float2v{yj.x, yj.y}, y23 = float2v{yj.z, yj.w};
            const float2v z01 = float2v{zj.x, zj.y}, z23 = float2v{zj.z, zj.w};
            if (pairs) {
                // resident per-pair constants of the row pair (DevModel::tgs2): t / mrs or 1e30, nothing to form per step
                pair_term2<0>(k2, float2v{tv[0].x, tv[0].y}, on2, xi2, yi2, zi2, x01, y01, z01, fx2, fy2, fz2);
                pair_term2<1>(k2, float2v{tv[0].z, tv[0].w}, on2, xi2, yi2, zi2, x01, y01, z01, fx2, fy2, fz2);
                pair_term2<0>(k2, float2v{tv[1].x, tv[1].y}, on2, xi2, yi2, zi2, x23, y23, z23, fx2, fy2, fz2);
                pair_term2<1>(k2, float2v{tv[1].z, tv[1].w}, on2, xi2, yi2, zi2, x23, y23, z23, fx2, fy2, fz2);
            } else {
                C3D_PK_COL(x, 0, x01, y01, z01); C3D_PK_COL(y, 1, x01, y01, z01); C3D_PK_COL(z, 0, x23, y23, z23); C3D_PK_COL(w, 1, x23, y23, z23);
            }
#pragma unroll
            for (int r = 0; r < RPW; ++r) { tv[r] = t1[r]; t1[r] = t2[r]; }
        }
        if (NC && m.wl != 4) {   // the narrow last block: lanes own m.wl (1..3) consecutive columns
            if (nblk > 1) tile_prefetch<RPW>(m, tgt, row0, lane, nblk - 1, tv);
            int width;
            float4 xj, yj, zj;
            block_coords(m, xs, ys, zs, nblk - 1, lane, width, xj, yj, zj);
            const float2v x01 = float2v{xj.x, xj.y}, x23 = float2v{xj.z, xj.w}, y01 = float2v{yj.x, yj.y}, y23 = float2v{yj.z, yj.w};
            const float2v z01 = float2v{zj.x, zj.y}, z23 = float2v{zj.z, zj.w};
            C3D_PK_COL(x, 0, x01, y01, z01);
            if (width > 1) C3D_PK_COL(y, 1, x01, y01, z01);
            if (width > 2) C3D_PK_COL(z, 0, x23, y23, z23);
        }
#undef C3D_PK_COL
        fx[0] = fx2.x; fx[1] = fx2.y; fy[0] = fy2.x; fy[1] = fy2.y; fz[0] = fz2.x; fz[1] = fz2.y;
        reduce_and_chain<POT, RPW, GEN, NC>(m, p, tgt, xs, ys, zs, row0, lane, fx, fy, fz, Fx, Fy, Fz);
        return;
    }
    if constexpr (POT == 4 && !GEN && RPW == 4 && !NC && PACKED4 && PACKED && C3D_PACKED_STEP) {
        // the same with four rows per wave: two row pairs sharing the column loads and the loop (the wide-tile step kernel; written out
        // apart from the two-row form above so that form's code stays what it was).  Resident pair targets only (DevModel::tgs2: the
        // launcher falls back to the two-row form without them), streamed ONE column block ahead: the register budget of four rows.
        constexpr int Q = RPW / 2;
        const PairK2 k2 = pair_k2(m, p);
        float2v xi2[Q], yi2[Q], zi2[Q], fx2[Q], fy2[Q], fz2[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            xi2[q] = float2v{xi[2 * q], xi[2 * q + 1]}; yi2[q] = float2v{yi[2 * q], yi[2 * q + 1]}; zi2[q] = float2v{zi[2 * q], zi[2 * q + 1]};
            fx2[q] = float2v{0.0f, 0.0f}; fy2[q] = fx2[q]; fz2[q] = fx2[q];
        }
        const float2v on2 = float2v{m.inv_rs, m.inv_rs};
        for (int jb = 0; jb < nfull; ++jb) {
            float4 t1[RPW];
            pair_targets_prefetch<RPW>(m, row0, lane, min(jb + 1, nblk - 1), t1);      // the last block re-reads itself (in bounds)
            const int j = 256 * jb + 4 * lane;
            const float4 xj = *reinterpret_cast<const float4*>(xs + j);
            const float4 yj = *reinterpret_cast<const float4*>(ys + j);
            const float4 zj = *reinterpret_cast<const float4*>(zs + j);
            const float2v x01 = float2v{xj.x, xj.y}, x23 = float2v{xj.z, xj.w}, y01 = float2v{yj.x, yj.y}, y23 = float2v{yj.z, yj.w};
            const float2v z01 = float2v{zj.x, zj.y}, z23 = float2v{zj.z, zj.w};
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                pair_term2<0>(k2, float2v{tv[2 * q].x, tv[2 * q].y}, on2, xi2[q], yi2[q], zi2[q], x01, y01, z01, fx2[q], fy2[q], fz2[q]);
                pair_term2<1>(k2, float2v{tv[2 * q].z, tv[2 * q].w}, on2, xi2[q], yi2[q], zi2[q], x01, y01, z01, fx2[q], fy2[q], fz2[q]);
                pair_term2<0>(k2, float2v{tv[2 * q + 1].x, tv[2 * q + 1].y}, on2, xi2[q], yi2[q], zi2[q], x23, y23, z23, fx2[q], fy2[q], fz2[q]);
                pair_term2<1>(k2, float2v{tv[2 * q + 1].z, tv[2 * q + 1].w}, on2, xi2[q], yi2[q], zi2[q], x23, y23, z23, fx2[q], fy2[q], fz2[q]);
                asm volatile("" : "+v"(fx2[q]), "+v"(fy2[q]), "+v"(fz2[q]));      // one row pair's terms in flight at a time (register budget)
            }
#pragma unroll
            for (int r = 0; r < RPW; ++r) tv[r] = t1[r];
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            fx[2 * q] = fx2[q].x; fx[2 * q + 1] = fx2[q].y; fy[2 * q] = fy2[q].x; fy[2 * q + 1] = fy2[q].y; fz[2 * q] = fz2[q].x; fz[2 * q + 1] = fz2[q].y;
        }
        reduce_and_chain<POT, RPW, GEN, NC>(m, p, tgt, xs, ys, zs, row0, lane, fx, fy, fz, Fx, Fy, Fz);
        return;
    }
    const PairK k = pair_k(m, p);
    for (int jb = 0; jb < nfull; ++jb) {
        float4 tn[RPW];
        const int jn = jb + 1 < nblk ? jb + 1 : jb;     // the last block re-reads itself (in bounds)
#pragma unroll
        for (int r = 0; r < RPW; ++r)      // always the four-column form (right unless the next block is a narrow last one: redone below)
            tn[r] = *reinterpret_cast<const float4*>(tgt + (size_t)min(row0 + r, m.n - 1) * m.npad + 256 * jn + 4 * lane);
        const int j = 256 * jb + 4 * lane;
        const float4 xj = *reinterpret_cast<const float4*>(xs + j);
        const float4 yj = *reinterpret_cast<const float4*>(ys + j);
        const float4 zj = *reinterpret_cast<const float4*>(zs + j);
#pragma unroll
        for (int r = 0; r < RPW; ++r)
            pair_quad<POT, GEN>(m, p, k, pair_b<GEN>(m, tv[r]), pair_a<GEN>(m, p, tv[r]), xi[r], yi[r], zi[r], xj, yj, zj, fx[r], fy[r], fz[r]);
#pragma unroll
        for (int r = 0; r < RPW; ++r) tv[r] = tn[r];
    }
    if (NC && m.wl != 4) {   // the narrow last block: lanes own m.wl (1..3) consecutive columns
        if (nblk > 1) tile_prefetch<RPW>(m, tgt, row0, lane, nblk - 1, tv);
        int width;
        float4 xj, yj, zj;
        block_coords(m, xs, ys, zs, nblk - 1, lane, width, xj, yj, zj);
#pragma unroll
        for (int r = 0; r < RPW; ++r)
            pair_quad_w<POT, GEN>(m, p, k, width, pair_b<GEN>(m, tv[r]), pair_a<GEN>(m, p, tv[r]), xi[r], yi[r], zi[r], xj, yj, zj, fx[r], fy[r], fz[r]);
    }
    reduce_and_chain<POT, RPW, GEN, NC>(m, p, tgt, xs, ys, zs, row0, lane, fx, fy, fz, Fx, Fy, Fz);
}

// clamp form with the per-pair constants resident for a whole launch (cluster kernel, compute waves): tv = pair_b in
// registers, mw_lds = pair_a in LDS; NB column blocks, fully unrolled.  Returns the butterfly sums of the PAIR terms only,
// WITHOUT the row factor (row_total): lane l holds the sum for row row0 + (l & 3) (RPW >= 3), row0 + (l & 1) (RPW = 2),
// row0 (RPW = 1); factor and chain terms are added by the finishing wave.
// NARROW = keep only four pair terms in flight (register budget).
template <int POT, int RPW, int NB, int WL, int NARROW>
__device__ __forceinline__ void tile_pair_sums_reg(const DevModel& m, const DevStep& p, const float4 (&tv)[RPW][NB],
                                                   const float4* mw_lds, const float* xs, const float* ys, const float* zs,
                                                   int row0, int lane, float& Fx, float& Fy, float& Fz) {
    // mw_lds: this wave's pair_a, [RPW * NB][64] float4 in LDS (lane-contiguous: ds_read_b128 at
    // full rate, no VALU slot); keeping them in registers next to the targets overflows the 128 a wave has at RPW x NB = 8
    float fx[RPW], fy[RPW], fz[RPW];
    float xi[RPW], yi[RPW], zi[RPW];
    const PairK k = pair_k(m, p);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = min(row0 + r, m.n - 1);
        xi[r] = xs[row]; yi[r] = ys[row]; zi[r] = zs[row];
        fx[r] = fy[r] = fz[r] = 0.0f;
    }
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) {
        constexpr int W4 = 4;
        const int width = jb == NB - 1 ? WL : W4;           // compile-time after unrolling: the last block's lanes own WL columns
        const int j = 256 * jb + width * lane;
        float4 xj, yj, zj;
        if (width == 4) {
            xj = *reinterpret_cast<const float4*>(xs + j); yj = *reinterpret_cast<const float4*>(ys + j); zj = *reinterpret_cast<const float4*>(zs + j);
        } else {
            xj = make_float4(xs[j], width > 1 ? xs[j + 1] : 0.0f, width > 2 ? xs[j + 2] : 0.0f, 0.0f);
            yj = make_float4(ys[j], width > 1 ? ys[j + 1] : 0.0f, width > 2 ? ys[j + 2] : 0.0f, 0.0f);
            zj = make_float4(zs[j], width > 1 ? zs[j + 1] : 0.0f, width > 2 ? zs[j + 2] : 0.0f, 0.0f);
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const float4 mw = mw_lds[(r * NB + jb) * 64 + lane];
            if (width == 4) pair_quad<POT, false>(m, p, k, tv[r][jb], mw, xi[r], yi[r], zi[r], xj, yj, zj, fx[r], fy[r], fz[r]);
            else pair_quad_w<POT, false>(m, p, k, width, tv[r][jb], mw, xi[r], yi[r], zi[r], xj, yj, zj, fx[r], fy[r], fz[r]);
            if constexpr (NARROW == 1) asm volatile("" : "+v"(fx[r]), "+v"(fy[r]), "+v"(fz[r]));   // four pair terms in flight, not 4 RPW NB
            if constexpr (NARROW == 2) { if (r & 1) asm volatile("" : "+v"(fx[r]), "+v"(fy[r]), "+v"(fz[r]), "+v"(fx[r - 1]), "+v"(fy[r - 1]), "+v"(fz[r - 1])); }   // eight
        }
        if constexpr (NARROW == 3) {
#pragma unroll
            for (int r = 0; r < RPW; ++r) asm volatile("" : "+v"(fx[r]), "+v"(fy[r]), "+v"(fz[r]));           // one block of all rows
        }
    }
    Fx = reduce_rows<RPW>(fx, lane);
    Fy = reduce_rows<RPW>(fy, lane);
    Fz = reduce_rows<RPW>(fz, lane);
}

// ---------------------------------------------------------------------------------------------
// the launch-resident pair constants of a compute wave in the layout the packed form wants: row pairs q = 0 .. RPW/2 - 1 hold
// (row 2q, row 2q + 1) per column, an odd last row stays a float4 per block
template <int RPW, int NB>
struct PairConsts2 {
    float2v p[RPW / 2 > 0 ? RPW / 2 : 1][NB][4];
    float4 s[NB];
};
// LDS layout of the 1 / mrs-or-0 constants: pair q, block jb: two float4 at ((q NB + jb) 2 + h) 64 + lane,
// h = 0: (a_r.x, a_r+1.x, a_r.y, a_r+1.y), h = 1: (.z, .z, .w, .w); the odd last row: one float4 at ((RPW - 1) NB + jb) 64 + lane
template <int RPW, int NB>
__device__ __forceinline__ void pair_consts2_build(const DevModel& m, const float4 (&traw)[RPW][NB], bool store, int lane, PairConsts2<RPW, NB>& pc,
                                                   float4* mw_lds) {
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) {
#pragma unroll
        for (int q = 0; q < RPW / 2; ++q) {
            const float4 ta = traw[2 * q][jb], tb = traw[2 * q + 1][jb];
#if C3D_PAIR_NONE_AS_FAR
            // "no restraint" as a target of 1e30 A (as in DevModel::tgs2): under the decaying lower bound such a pair feels exactly nothing, the
            // second per-pair constant is then 1 / mrs for every pair and need not exist — no 96 KB of LDS per workgroup, no 14 ds_read_b128
            // per pass of a compute wave
            auto enc = [&](float t) { return t > 0.0f ? t * m.inv_rs : 1e30f; };
            pc.p[q][jb][0] = float2v{enc(ta.x), enc(tb.x)}; pc.p[q][jb][1] = float2v{enc(ta.y), enc(tb.y)};
            pc.p[q][jb][2] = float2v{enc(ta.z), enc(tb.z)}; pc.p[q][jb][3] = float2v{enc(ta.w), enc(tb.w)};
            if (false) {
#else
            pc.p[q][jb][0] = float2v{ta.x * m.inv_rs, tb.x * m.inv_rs}; pc.p[q][jb][1] = float2v{ta.y * m.inv_rs, tb.y * m.inv_rs};
            pc.p[q][jb][2] = float2v{ta.z * m.inv_rs, tb.z * m.inv_rs}; pc.p[q][jb][3] = float2v{ta.w * m.inv_rs, tb.w * m.inv_rs};
            if (store) {
#endif
                const float on = m.inv_rs;
                mw_lds[((q * NB + jb) * 2) * 64 + lane] = make_float4(ta.x > 0.0f ? on : 0.0f, tb.x > 0.0f ? on : 0.0f, ta.y > 0.0f ? on : 0.0f, tb.y > 0.0f ? on : 0.0f);
                mw_lds[((q * NB + jb) * 2 + 1) * 64 + lane] = make_float4(ta.z > 0.0f ? on : 0.0f, tb.z > 0.0f ? on : 0.0f, ta.w > 0.0f ? on : 0.0f, tb.w > 0.0f ? on : 0.0f);
            }
        }
        if constexpr (RPW & 1) {
            DevStep p0{};
            pc.s[jb] = pair_b<false>(m, traw[RPW - 1][jb]);
            if (store) mw_lds[((RPW - 1) * NB + jb) * 64 + lane] = pair_a<false>(m, p0, traw[RPW - 1][jb]);
        }
    }
}
// tile_pair_sums_reg for device potential 4 in the packed form (same sums, same bits)
template <int RPW, int NB, int WL>
__device__ __forceinline__ void tile_pair_sums_pk(const DevModel& m, const DevStep& p, const PairConsts2<RPW, NB>& pc, const float4* mw_lds,
                                                  const float* xs, const float* ys, const float* zs, int row0, int lane, float& Fx, float& Fy,
                                                  float& Fz) {
    constexpr int NQ = RPW / 2;
    float2v fx2[NQ > 0 ? NQ : 1], fy2[NQ > 0 ? NQ : 1], fz2[NQ > 0 ? NQ : 1], xi2[NQ > 0 ? NQ : 1], yi2[NQ > 0 ? NQ : 1], zi2[NQ > 0 ? NQ : 1];
    float fxs = 0.0f, fys = 0.0f, fzs = 0.0f, xis = 0.0f, yis = 0.0f, zis = 0.0f;
    const PairK2 k2 = pair_k2(m, p);
    PairK k{};
    if constexpr (RPW & 1) k = pair_k(m, p);
#if C3D_PAIR_NONE_AS_FAR
    float2v on2 = float2v{m.inv_rs, m.inv_rs};
    asm volatile("" : "+v"(on2));
#endif
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        // the two rows of a pair are neighbours in LDS and the pair starts on an even row (RPW is even here, a workgroup's first row a
        // multiple of 8): ONE 8-byte read per coordinate instead of two reads and the moves that pair them up.  A row beyond the last bead
        // reads a padding bead (1e4 A away; the arrays hold 256 NB entries) — its sums are finite and nobody takes them (as before, when
        // such a row re-read the last bead)
        const int ra = min(row0 + 2 * q, 256 * NB - 2);
        if constexpr ((RPW & 1) == 0) {
            xi2[q] = *reinterpret_cast<const float2v*>(xs + ra); yi2[q] = *reinterpret_cast<const float2v*>(ys + ra); zi2[q] = *reinterpret_cast<const float2v*>(zs + ra);
        } else {
            const int rb = min(row0 + 2 * q + 1, 256 * NB - 1);
            xi2[q] = float2v{xs[ra], xs[rb]}; yi2[q] = float2v{ys[ra], ys[rb]}; zi2[q] = float2v{zs[ra], zs[rb]};
        }
        fx2[q] = fy2[q] = fz2[q] = float2v{0.0f, 0.0f};
    }
    if constexpr (RPW & 1) { const int row = min(row0 + RPW - 1, m.n - 1); xis = xs[row]; yis = ys[row]; zis = zs[row]; }
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) {
        constexpr int W4 = 4;
        const int width = jb == NB - 1 ? WL : W4;           // compile-time after unrolling: the last block's lanes own WL columns
        const int j = 256 * jb + width * lane;
        float4 xj, yj, zj;
        if (width == 4) {
            xj = *reinterpret_cast<const float4*>(xs + j); yj = *reinterpret_cast<const float4*>(ys + j); zj = *reinterpret_cast<const float4*>(zs + j);
        } else {
            xj = make_float4(xs[j], width > 1 ? xs[j + 1] : 0.0f, width > 2 ? xs[j + 2] : 0.0f, 0.0f);
            yj = make_float4(ys[j], width > 1 ? ys[j + 1] : 0.0f, width > 2 ? ys[j + 2] : 0.0f, 0.0f);
            zj = make_float4(zs[j], width > 1 ? zs[j + 1] : 0.0f, width > 2 ? zs[j + 2] : 0.0f, 0.0f);
        }
        const float2v x01 = float2v{xj.x, xj.y}, x23 = float2v{xj.z, xj.w}, y01 = float2v{yj.x, yj.y}, y23 = float2v{yj.z, yj.w};
        const float2v z01 = float2v{zj.x, zj.y}, z23 = float2v{zj.z, zj.w};
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#if C3D_PAIR_NONE_AS_FAR
            const float4 ma = make_float4(on2.x, on2.y, on2.x, on2.y), mb = ma;
#else
            const float4 ma = mw_lds[((q * NB + jb) * 2) * 64 + lane];
#endif
            pair_term2<0>(k2, pc.p[q][jb][0], float2v{ma.x, ma.y}, xi2[q], yi2[q], zi2[q], x01, y01, z01, fx2[q], fy2[q], fz2[q]);
            if (width > 1) pair_term2<1>(k2, pc.p[q][jb][1], float2v{ma.z, ma.w}, xi2[q], yi2[q], zi2[q], x01, y01, z01, fx2[q], fy2[q], fz2[q]);
            if (width > 2) {
#if !C3D_PAIR_NONE_AS_FAR
                const float4 mb = mw_lds[((q * NB + jb) * 2 + 1) * 64 + lane];
#endif
                pair_term2<0>(k2, pc.p[q][jb][2], float2v{mb.x, mb.y}, xi2[q], yi2[q], zi2[q], x23, y23, z23, fx2[q], fy2[q], fz2[q]);
                if (width > 3) pair_term2<1>(k2, pc.p[q][jb][3], float2v{mb.z, mb.w}, xi2[q], yi2[q], zi2[q], x23, y23, z23, fx2[q], fy2[q], fz2[q]);
            }
            asm volatile("" : "+v"(fx2[q]), "+v"(fy2[q]), "+v"(fz2[q]));      // one row pair's block in flight (register budget)
        }
        if constexpr (RPW & 1) {
            const float4 mw = mw_lds[((RPW - 1) * NB + jb) * 64 + lane];
            if (width == 4) pair_quad<4, false>(m, p, k, pc.s[jb], mw, xis, yis, zis, xj, yj, zj, fxs, fys, fzs);
            else pair_quad_w<4, false>(m, p, k, width, pc.s[jb], mw, xis, yis, zis, xj, yj, zj, fxs, fys, fzs);
            asm volatile("" : "+v"(fxs), "+v"(fys), "+v"(fzs));
        }
    }
    float fx[RPW], fy[RPW], fz[RPW];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { fx[2 * q] = fx2[q].x; fx[2 * q + 1] = fx2[q].y; fy[2 * q] = fy2[q].x; fy[2 * q + 1] = fy2[q].y; fz[2 * q] = fz2[q].x; fz[2 * q + 1] = fz2[q].y; }
    if constexpr (RPW & 1) { fx[RPW - 1] = fxs; fy[RPW - 1] = fys; fz[RPW - 1] = fzs; }
    Fx = reduce_rows<RPW>(fx, lane);
    Fy = reduce_rows<RPW>(fy, lane);
    Fz = reduce_rows<RPW>(fz, lane);
}

// ---------------------------------------------------------------------------------------------
// per-replica scalars of a step from the previous step's sums, identical in every wave
// ---------------------------------------------------------------------------------------------
struct StepScalars {
    float lam, cmx, cmy, cmz;   // MD: velocity scale, centre-of-mass velocity
    float keep, mix;            // FIRE: velocity mixing
};
// psum: MD kinds 0/1 = (sum v^2, sum vx, sum vy, sum vz) of the previous half step;
//       FIRE kind 2 = (v.F, F.F, v.v) of the previous evaluation; kind 3 (first step of a stage) = 0, fresh state
//       two-point step size kind 5 = (s.y, F.F, y.y, s.s) of the previous evaluation; kind 6 (first step of a stage) = nothing, fresh state
// TWO = false: a kernel that never runs the two-point minimiser's kinds (5 / 6) does not carry their branch (c3d_cluster.hip: k_cluster)
template <bool TWO = true>
__device__ __forceinline__ StepScalars step_scalars(const DevModel& m, const DevStep& p, const DevFire& fp, const float4 psum,
                                                    FireState& st) {
    StepScalars s;
    s.lam = 1.0f; s.cmx = s.cmy = s.cmz = 0.0f; s.keep = 0.0f; s.mix = 0.0f;
    if (p.kind == 0 || p.kind == 1) {
        // hardware reciprocal / square root (1 ulp): these few scalars sit on every workgroup's critical path
        // between the arrival of the sums and the pair loop, and the correctly rounded sequences are ~35
        // dependent instructions each
        const float tprev = fmaxf(m.t_fac * psum.x, 1e-2f);
        // (T0 / T - 1 as ONE explicit fma: what every instantiation's backend made of "product, then - 1" under -ffp-contract=fast; see row_total)
        const float rt = __builtin_amdgcn_rcpf(tprev);
        if (p.kind == 0) s.lam = __builtin_amdgcn_sqrtf(fmaxf(fmaf(p.dt * m.fbeta, fmaf(p.t_bath, rt, -1.0f), 1.0f), 0.0f));
        else s.lam = __builtin_amdgcn_sqrtf(p.t_bath * rt);
        s.cmx = psum.y * m.inv_n; s.cmy = psum.z * m.inv_n; s.cmz = psum.w * m.inv_n;
        // products that a subtraction follows (finish_row: v - v_cm): hidden from the instruction selector, so that "product, then difference"
        // is what every kernel computes whatever the backend would like to fuse under -ffp-contract=fast (see row_total)
        asm("" : "+v"(s.cmx), "+v"(s.cmy), "+v"(s.cmz));
    } else if (TWO && (p.kind == 5 || p.kind == 6)) {
        // two-point step size (Barzilai-Borwein) with the length one evaluation late (the CPU restatement: c3o_bb_step): lam = the length of the previous
        // move (to rebuild it from the force kept in the velocity slot), mix = the length of this one; st.dt / st.npos carry them on
        const int k = p.kind == 6 ? 0 : st.npos;
        const float a_prev = st.dt;
        float a = a_prev;
        if (k == 0) a = fp.dt_start * fp.dt_start * m.acc;
        else if (k >= 2) {
            const float sy = psum.x;
            if (sy > 0.0f) a = (k & 1) ? sy * __builtin_amdgcn_rcpf(psum.z) : psum.w * __builtin_amdgcn_rcpf(sy);
            else a = 2.0f * a_prev;
            a = fminf(fmaxf(a, 1e-7f), 1e2f);
        }
        s.lam = a_prev; s.mix = a;
        st.dt = a; st.npos = k + 1;
    } else if (p.kind == 2 || p.kind == 3) {
        // FIRE (Bitzek et al. 2006) with the power test on the previous step's sums
        if (psum.x > 0.0f) {
            s.keep = 1.0f - st.alpha;
            s.mix = st.alpha * __builtin_amdgcn_sqrtf(psum.z * __builtin_amdgcn_rcpf(fmaxf(psum.y, 1e-30f)));
            if (st.npos > fp.n_min) {
                st.dt = fminf(st.dt * fp.f_inc, fp.dt_max);
                st.alpha *= fp.f_alpha;
            }
            st.npos += 1;
        } else {
            st.alpha = fp.alpha_start;
            st.dt *= fp.f_dec;
            st.npos = 0;
        }
    }
    return s;
}

// one row's update from its force: new position, new velocity and the row's contribution q to the
// replica sums the NEXT step needs
__device__ __forceinline__ void finish_row(const DevModel& m, const DevStep& p, const DevFire& fp, const StepScalars& sc,
                                           const FireState& st, float Fx, float Fy, float Fz, float x0, float y0, float z0,
                                           float vx0, float vy0, float vz0, float& xn, float& yn, float& zn, float& vx,
                                           float& vy, float& vz, float4& q) {
    const bool is_md = p.kind == 0 || p.kind == 1 || p.kind == 4;
    if (is_md) {
        if (p.kind == 4) {            // MD begin: take the Maxwell velocities, no move
            vx = vx0; vy = vy0; vz = vz0;
            xn = x0; yn = y0; zn = z0;
        } else {
            const float a = p.dt * m.acc;
            vx = fmaf(a, Fx, sc.lam * (vx0 - sc.cmx));
            vy = fmaf(a, Fy, sc.lam * (vy0 - sc.cmy));
            vz = fmaf(a, Fz, sc.lam * (vz0 - sc.cmz));
            xn = fmaf(p.dt, vx, x0);
            yn = fmaf(p.dt, vy, y0);
            zn = fmaf(p.dt, vz, z0);
        }
        q = make_float4(fmaf(vx, vx, fmaf(vy, vy, vz * vz)), vx, vy, vz);
    } else if (p.kind == 5 || p.kind == 6) {
        // v0 = the force of the previous evaluation; the previous move is rebuilt from it (same clamp), y = F_old - F
        const float ms2 = fp.max_step * fp.max_step;
        const float ff = fmaf(Fx, Fx, fmaf(Fy, Fy, Fz * Fz));
        if (p.kind == 5) {
            float sx = sc.lam * vx0, sy = sc.lam * vy0, sz = sc.lam * vz0;
            const float s2 = fmaf(sx, sx, fmaf(sy, sy, sz * sz));
            const float scp = s2 > ms2 ? fp.max_step * __builtin_amdgcn_rsqf(s2) : 1.0f;
            sx *= scp; sy *= scp; sz *= scp;
            asm("" : "+v"(sx), "+v"(sy), "+v"(sz));          // (products, then sums of products: kept apart in every build, see row_total)
            const float yx = vx0 - Fx, yy = vy0 - Fy, yz = vz0 - Fz;
            q = make_float4(fmaf(sx, yx, fmaf(sy, yy, sz * yz)), ff, fmaf(yx, yx, fmaf(yy, yy, yz * yz)), fmaf(sx, sx, fmaf(sy, sy, sz * sz)));
        } else {
            q = make_float4(0.0f, ff, 0.0f, 0.0f);
        }
        const float dxs = sc.mix * Fx, dys = sc.mix * Fy, dzs = sc.mix * Fz;
        const float d2 = fmaf(dxs, dxs, fmaf(dys, dys, dzs * dzs));
        const float scl = d2 > ms2 ? fp.max_step * __builtin_amdgcn_rsqf(d2) : 1.0f;
        xn = fmaf(scl, dxs, x0); yn = fmaf(scl, dys, y0); zn = fmaf(scl, dzs, z0);
        vx = Fx; vy = Fy; vz = Fz;
    } else {
        // sums of THIS evaluation for the next step's test, with the velocity that led here
        q = make_float4(fmaf(vx0, Fx, fmaf(vy0, Fy, vz0 * Fz)), fmaf(Fx, Fx, fmaf(Fy, Fy, Fz * Fz)),
                        fmaf(vx0, vx0, fmaf(vy0, vy0, vz0 * vz0)), 0.0f);
        const float a = st.dt * m.acc;
        vx = fmaf(a, Fx, fmaf(sc.keep, vx0, sc.mix * Fx));
        vy = fmaf(a, Fy, fmaf(sc.keep, vy0, sc.mix * Fy));
        vz = fmaf(a, Fz, fmaf(sc.keep, vz0, sc.mix * Fz));
        const float dxs = st.dt * vx, dys = st.dt * vy, dzs = st.dt * vz;
        const float d2 = fmaf(dxs, dxs, fmaf(dys, dys, dzs * dzs));
        const float scl = d2 > fp.max_step * fp.max_step ? fp.max_step * __builtin_amdgcn_rsqf(d2) : 1.0f;
        xn = fmaf(scl, dxs, x0); yn = fmaf(scl, dys, y0); zn = fmaf(scl, dzs, z0);
    }
}

// one tile's (8 rows) sums from the rows' contributions q[0..7], a fixed tree: ((q0+q1) + (q2+q3)) + ((q4+q5) + (q6+q7))
__device__ __forceinline__ float4 tile_sum8(const float4* q) {
    float4 h[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float4 a = q[4 * k], b = q[4 * k + 1], c = q[4 * k + 2], d = q[4 * k + 3];
        h[k] = make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w));
    }
    return make_float4(h[0].x + h[1].x, h[0].y + h[1].y, h[0].z + h[1].z, h[0].w + h[1].w);
}

__device__ __forceinline__ float4 wave_sum4(float4 a) {
    return make_float4(wave_sum(a.x), wave_sum(a.y), wave_sum(a.z), wave_sum(a.w));
}

// async global -> LDS copy of `count` floats (count % 256 == 0), 16 bytes per lane per instruction
// (global_load_lds_dwordx4: LDS address = wave-uniform base + lane*16, no VGPR round trip).  The
// caller's __syncthreads() waits for it (hipcc emits s_waitcnt vmcnt(0) before the barrier).
template <int BLOCK>
__device__ __forceinline__ void lds_dma_copy(const float* __restrict__ src, float* dst, int count, int tid) {
    const int lane = tid & 63;
    for (int b = 4 * tid; b < count; b += 4 * BLOCK)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + b),
                                         (void __attribute__((address_space(3)))*)(dst + (b - 4 * lane)), 16, 0, 0);
}

}  // namespace c3d

#pragma clang fp contract(fast)
