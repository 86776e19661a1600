// aslr_backward_blk.inc.hpp -- Riccati backward pass for the larger state (nx = 28, 7-DoF SEA):
// ONE 128-THREAD BLOCK PER TRAJECTORY, every operand of the recursion resident in LDS.
//
// SolverDDP / SolverFDDP backwardPass + computeGains; same formulas, same summation
// order per entry as backward_kernel (aslr_backward.inc.hpp).  That kernel keeps a column of every matrix in
// registers, which at nx = 28 needs ~2x the register file and spills to scratch; here:
//
//   * thread (rr, g) = (tid >> 3, tid & 7) owns rows rr and rr + 16, columns 4g..4g+3 of every nx x nx product
//     (2 x 4 outputs per thread: 16 FMAs per 6 LDS reads); the left operands are read as pairs along the
//     contraction index, the right operand as the 4 contiguous doubles of a row (ds_read_b128, conflict-free);
//   * 128 threads = 2 waves per trajectory: the C5 shard (512 trajectories per GPU) is then 1024 waves, one per
//     SIMD, each with the whole register file -- nothing spills;
//   * products: C = P Fx (stored transposed: A = Fx^T P) and P Fu in one sweep over the contraction index,
//     then Qxx = Lxx + A Fx, and [Qux | Quu] = [Lxu^T | Luu] + (Fu^T P) [Fx | Fu] with one entry per thread;
//   * the nu x nu Cholesky, the nx + 1 triangular solves (one column of K per lane), Quu k, Vx and the
//     expected-improvement terms run in wave 0 while the other waves wait at the barrier;
//   * Vxx = Qxx - Qux^T K, its symmetrisation and the NaN / 1e30 test are again one row x 4 columns per thread;
//   * the next knot's record (16 KB) is prefetched into registers (8 x 16 B per thread) during the knot.
//
// SolverBoxDDP (BOX): wave 0 also runs computeGains' box QP for the block's trajectory (lane_gains<NU>: every lane the
// same nu x nu problem, wave-uniform control flow, the lane's own column of Qux solved with the final free-set factor);
// the node's u and stored k (the QP's bounds and warm start) come in with the record prefetch.
#pragma once
#include "aslr_backward.inc.hpp"

namespace aslr {

template <int NX, int NU>
struct BwdBlk {
  static_assert(NX % 4 == 0 && NX > 16 && NX <= 32 && NU <= 8, "thread map: (16 x 2) rows x 8 groups of 4 columns");
  static constexpr int NT = 128;
  static constexpr int NG = NX / 4;
  static constexpr int REC = rec_len_c(NX, NU);
  static constexpr int NPRE = (REC / 2 + NT - 1) / NT; // double2 prefetch registers per thread
  static constexpr int oFx = 0, oFu = oFx + NX * NX, oLxx = oFu + NX * NU, oLxu = oLxx + NX * NX,
                       oLuu = oLxu + NX * NU, oLx = oLuu + NU * NU, oLu = oLx + NX;
  static constexpr int even(int v) { return (v + 1) / 2 * 2; }
  // LDS arrays (doubles); every base is even (16-byte aligned)
  // Vx sits directly behind PT: it is row nx of the left operand of the MFMA pass 1, which then yields Fx^T Vx and
  // Fu^T Vx (for Qx, Qu) along with P Fx and P Fu
  static constexpr int sRec = 0, sPT = sRec + even(REC), sVx = sPT + NX * NX, sA = sVx + even(NX), sB = sA + NX * NX,
                       sQux = sB + even(NU * NX), sQuxT = sQux + even(NU * NX), sK = sQuxT + NX * 8,
                       sQuu = sK + 8 * NX, sQu = sQuu + 64, sQx = sQu + 8, sF = sQx + 32,
                       sRed = sF + 32, sCost = sRed + 64, sFlag = sCost + NT, sUK = sFlag + 2, sEnd = sUK + 16;
  static constexpr int LDS = even(sEnd);
};

typedef double double4_t __attribute__((ext_vector_type(4)));

// MFMA: the two nx x nx x nx products (and P Fu) through v_mfma_f64_16x16x4_f64 -- 16 x 16 output tiles, wave w owns
// rows 16w..16w+15 (lane l: operand row / column l & 15, contraction index l >> 4; results row (l >> 4) + 4 r,
// column l & 15).  The instruction accumulates exactly like the k-ordered fma chain of the vector path
// (tools/ubench/mfma_f64_layout.hip: 256 / 256 entries bit-equal), so both paths give the same bits.  It has the
// vector unit's FP64 rate; what it saves is issue slots: 35 MFMAs per wave and knot replace 504 FMAs + 196 LDS reads.
template <int NX, int NU, bool GAPS, bool MFMA, bool BOX = false>
__global__ void __launch_bounds__(128) backward_blk_kernel(KArgs a, SolverDev sp, ModelLimits lim) {
  using C = BwdBlk<NX, NU>;
  constexpr int REC = C::REC, NG = C::NG, NT = C::NT;
  __shared__ double sm[C::LDS];
  double *rec = sm + C::sRec, *PT = sm + C::sPT, *AL = sm + C::sA, *BL = sm + C::sB, *QuxL = sm + C::sQux,
         *QuxT = sm + C::sQuxT, *KL = sm + C::sK, *QuuL = sm + C::sQuu, *QuL = sm + C::sQu, *QxL = sm + C::sQx,
         *VxL = sm + C::sVx, *FL = sm + C::sF, *RedL = sm + C::sRed, *CostL = sm + C::sCost;
  int *FlagL = reinterpret_cast<int *>(sm + C::sFlag);
  double *UKL = sm + C::sUK; // [u (8) | stored k (8)] of the knot (box nodes)

  const int tid = threadIdx.x, rr = tid >> 3, g = tid & 7;
  const int gc = g < NG ? g : NG - 1, gu = g < NU ? g : NU - 1;
  const int row[2] = {rr, rr + 16 < NX ? rr + 16 : NX - 1};               // rows of this thread (second clamped)
  const bool cell[2] = {g < NG, g < NG && rr + 16 < NX};                  // it owns entries (row[h], 4g..4g+3)
  const bool wave0 = tid < 64;
  const int wv = tid >> 6, li = tid & 15, lk = (tid & 63) >> 4; // MFMA lane roles
  constexpr int NKS = NX / 4;
  const int B = a.B, T = a.T, b = a.b0 + blockIdx.x;
  int32_t *TI = a.traj_i;
  double *TF = a.traj_f;

  // ---- prologue: solver-state bookkeeping that Crocoddyl does inside calcDiff ----
  int done = 0, feasible = TI[ASLR_TI_FEASIBLE * B + b], status = TI[ASLR_TI_STATUS * B + b];
  if (!sp.standalone) {
    done = TI[ASLR_TI_DONE * B + b];
    const int recalc = TI[ASLR_TI_RECALC * B + b];
    if (!done && recalc) {
      if (!feasible) feasible = TI[ASLR_TI_GAPFLAG * B + b] ? 0 : 1;
      // cost_ = sum of node costs, in node order: loads spread over the block, adds by one thread
      double csum = 0.0;
      for (int t0 = 0; t0 <= T; t0 += NT) {
        __syncthreads();
        if (t0 + tid <= T) CostL[tid] = a.cost[(size_t)(t0 + tid) * B + b];
        __syncthreads();
        if (tid == 0) {
          const int n = T + 1 - t0 < NT ? T + 1 - t0 : NT;
          for (int q = 0; q < n; ++q) csum += CostL[q];
        }
      }
      if (tid == 0) TF[ASLR_TF_COST * B + b] = csum;
    }
    __syncthreads();
    if (tid == 0) {
      TI[ASLR_TI_FEASIBLE * B + b] = feasible;
      TI[ASLR_TI_ACCEPTED * B + b] = -1;
      TI[ASLR_TI_GAPFLAG * B + b] = 0;
    }
  }
  if (done) return;
  // zero padding of the 8-long contraction of the MFMA Vxx update (never overwritten: Qux^T and K hold nu < 8 entries)
  static_assert(NU < 8, "one padding slot per row");
  ASLR_UNROLL for (int c = NU; c < 8; ++c) {
    if (tid < NX) { QuxT[tid * 8 + c] = 0.0; KL[c * NX + tid] = 0.0; }
  }
  bool need = true;
  double xreg = TF[ASLR_TF_XREG * B + b];
  const bool fddp = GAPS && sp.solver == ASLR_SOLVER_FDDP;
  const bool gaps_on = GAPS && !feasible;
  const bool box = BOX && sp.solver == ASLR_SOLVER_BOXDDP && feasible; // (SolverBoxDDP::computeGains: plain gains while infeasible)

  // pass-2b tasks: entries e of [Qux (nu x nx) | Quu (nu x nu)], e = tid and tid + NT (compile-time strides on
  // each path: a per-thread stride makes the compiler keep one LDS address per contraction step)
  constexpr int NQX = NU * NX, NQ = NQX + NU * NU;
  static_assert(NT < NQX && NQ <= 2 * NT, "two tasks per thread, the first always in Qux");
  const int e1 = tid + NT;
  const bool e1x = e1 < NQX, e1u = !e1x && e1 < NQ;
  const int k2a = tid / NX, c2a = tid % NX;
  const int k2b = e1x ? e1 / NX : 0, c2b = e1x ? e1 % NX : 0;
  const int k3 = e1u ? (e1 - NQX) / NU : 0, c3 = e1u ? (e1 - NQX) % NU : 0;

  double d1 = 0.0, d2 = 0.0, stop = 0.0, dgf = 0.0, dqf = 0.0;
  while (need) {
    bool failed = false;
    d1 = d2 = stop = dgf = dqf = 0.0;
    const double xr = isnan(xreg) ? 0.0 : xreg;
    double Vx_own = 0.0; // threads tid < NX: entry tid of Vx
    // ---- terminal node: Vxx = Lxx (+xreg), Vx = Lx (+ Vxx f) ----
    {
      const double *rT = a.deriv + ((size_t)T * B + b) * REC;
      __syncthreads();
      // PT[c][l] = P(l, c): the record's Lxx is taken as stored (it is symmetric up to rounding only)
      for (int e = tid; e < NX * NX; e += NT) {
        const int l = e / NX, c = e % NX;
        PT[c * NX + l] = rT[C::oLxx + e] + (l == c ? xr : 0.0);
      }
      if (tid < NX) Vx_own = rT[C::oLx + tid];
      if (gaps_on && tid < NX) FL[tid] = a.gaps[((size_t)T * B + b) * NX + tid];
      __syncthreads();
      if (gaps_on && tid < NX) {
        double vf = 0.0;
        { const double2 *pr = reinterpret_cast<const double2 *>(PT + tid * NX), *fr = reinterpret_cast<const double2 *>(FL);
          ASLR_UNROLL for (int q = 0; q < NX / 2; ++q) { const double2 pv = pr[q], fv = fr[q]; vf += pv.x * fv.x; vf += pv.y * fv.y; } }
        Vx_own += vf;
        if (fddp) {
          dgf -= Vx_own * FL[tid];
          dqf += FL[tid] * vf;
          a.vxxf[((size_t)T * B + b) * NX + tid] = vf;
        }
      }
      if (sp.store_v) {
        if (tid < NX) a.vx[((size_t)T * B + b) * NX + tid] = Vx_own;
        double *o = a.vxx + ((size_t)T * B + b) * NX * NX;
        for (int e = tid; e < NX * NX; e += NT) o[e] = PT[(e % NX) * NX + e / NX];
      }
      if (tid < NX) VxL[tid] = Vx_own;
    }
    // ---- prefetch of knot T-1 ----
    double prx[C::NPRE], pry[C::NPRE], pre_f = 0.0, pre_uk = 0.0;
    int pre_m = 0;
#define ASLR_BLK_PREFETCH(tt)                                                                          \
    do {                                                                                               \
      const size_t tbp = (size_t)(tt) * B + b;                                                         \
      typedef double nt_double2 __attribute__((ext_vector_type(2)));                                   \
      const nt_double2 *src = reinterpret_cast<const nt_double2 *>(a.deriv + tbp * REC);               \
      ASLR_UNROLL for (int q = 0; q < C::NPRE; ++q) {                                                  \
        const int idx = tid + NT * q;                                                                  \
        if (C::NPRE * NT == REC / 2 || idx < REC / 2) { const nt_double2 v2 = __builtin_nontemporal_load(src + idx); prx[q] = v2.x; pry[q] = v2.y; } \
      }                                                                                                \
      if (gaps_on && tid < NX) pre_f = a.gaps[tbp * NX + tid];                                         \
      if (box && tid < 16) pre_uk = (tid & 7) < NU ? (tid < 8 ? a.us[tbp * NU + tid] : a.kff[tbp * NU + tid - 8]) : 0.0; \
      pre_m = node_model_at(a, tt);                                                                    \
    } while (0)
    ASLR_PROF_DECL;
    ASLR_BLK_PREFETCH(T - 1);
    for (int t = T - 1; t >= 0; --t) {
      const size_t tb = (size_t)t * B + b;
      ASLR_PROF(6);
      ASLR_PROF_COUNT(15);
      __syncthreads(); // every reader of the previous knot's record is done; PT / VxL of this knot are written
      {
        double2 *dst = reinterpret_cast<double2 *>(rec);
        ASLR_UNROLL for (int q = 0; q < C::NPRE; ++q) {
          const int idx = tid + NT * q;
          if (C::NPRE * NT == REC / 2 || idx < REC / 2) { double2 v2; v2.x = prx[q]; v2.y = pry[q]; dst[idx] = v2; }
        }
        if (gaps_on && tid < NX) FL[tid] = pre_f;
        if (box && tid < 16) UKL[tid] = pre_uk;
      }
      const int mi = pre_m;
      __syncthreads();
      ASLR_PROF(0);
      if (t > 0) ASLR_BLK_PREFETCH(t - 1);

      double bfx[2][MFMA ? NKS : 1]; // MFMA: the Fx operand fragments, reused by the second product
      if constexpr (MFMA) {
        // ---- pass 1 (MFMA): C = P Fx (row tile wv x column tiles 0, 1) and P Fu (column tile of Fu) ----
        double4_t c0 = {0.0, 0.0, 0.0, 0.0}, c1 = c0, cu = c0;
        const double *prow = PT + (16 * wv + li) * NX + lk; // (rows / columns >= nx read neighbouring LDS: unused results)
        ASLR_UNROLL for (int ks = 0; ks < NKS; ++ks) {
          const int k = 4 * ks + lk;
          const double pa = prow[4 * ks];
          bfx[0][ks] = rec[C::oFx + k * NX + li];
          bfx[1][ks] = rec[C::oFx + k * NX + 16 + li];
          const double bu = rec[C::oFu + k * NU + li];
          c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, bfx[0][ks], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, bfx[1][ks], c1, 0, 0, 0);
          cu = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, bu, cu, 0, 0, 0);
        }
        ASLR_UNROLL for (int q = 0; q < 4; ++q) {
          const int rw = 16 * wv + lk + 4 * q;
          if (rw < NX) {
            AL[li * NX + rw] = c0[q];                          // A = C^T
            if (16 + li < NX) AL[(16 + li) * NX + rw] = c1[q];
            if (li < NU) BL[li * NX + rw] = cu[q];
          }
        }
        // row nx of the left operand is Vx: its results are Fx^T Vx (two column tiles) and Fu^T Vx -- the same
        // products, summed in the same order, as the loops of the vector path below
        static_assert(NX % 16 == 12, "row nx = 16 + 12: wave 1, lanes lk = 0, result register 3");
        if (wv == 1 && lk == 0) {
          QxL[li] = rec[C::oLx + li] + c0[3];
          if (16 + li < NX) QxL[16 + li] = rec[C::oLx + 16 + li] + c1[3];
          if (li < NU) QuL[li] = rec[C::oLu + li] + cu[3];
        }
      } else {
      // ---- pass 1: C(r, 4g..) = sum_l P(r,l) Fx(l, 4g..)  [= A(4g.., r)],  (P Fu)(r, g) [= B(g, r)] ----
        {
          double c[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}, bu[2] = {0.0, 0.0};
          const double2 *prow0 = reinterpret_cast<const double2 *>(PT + row[0] * NX);
          const double2 *prow1 = reinterpret_cast<const double2 *>(PT + row[1] * NX);
          _Pragma("unroll 2") for (int l = 0; l < NX; l += 2) {
            const double2 p[2] = {prow0[l / 2], prow1[l / 2]};
            const double2 *f0 = reinterpret_cast<const double2 *>(rec + C::oFx + l * NX + 4 * gc);
            const double2 *f1 = reinterpret_cast<const double2 *>(rec + C::oFx + (l + 1) * NX + 4 * gc);
            const double2 f0a = f0[0], f0b = f0[1], f1a = f1[0], f1b = f1[1];
            const double u0 = rec[C::oFu + l * NU + gu], u1 = rec[C::oFu + (l + 1) * NU + gu];
            ASLR_UNROLL for (int h = 0; h < 2; ++h) {
              c[h][0] += p[h].x * f0a.x; c[h][1] += p[h].x * f0a.y; c[h][2] += p[h].x * f0b.x; c[h][3] += p[h].x * f0b.y;
              bu[h] += p[h].x * u0;
              c[h][0] += p[h].y * f1a.x; c[h][1] += p[h].y * f1a.y; c[h][2] += p[h].y * f1b.x; c[h][3] += p[h].y * f1b.y;
              bu[h] += p[h].y * u1;
            }
          }
          ASLR_UNROLL for (int h = 0; h < 2; ++h) {
            if (cell[h]) { ASLR_UNROLL for (int q = 0; q < 4; ++q) AL[(4 * g + q) * NX + row[h]] = c[h][q]; }
            if (g < NU && (h == 0 || rr + 16 < NX)) BL[g * NX + row[h]] = bu[h];
          }
        }
      }
      if (MFMA) {
      } else if (tid < NX) { // Qx = Lx + Fx^T Vx, Qu = Lu + Fu^T Vx (wave 0)
        double s = 0.0;
        ASLR_UNROLL for (int l = 0; l < NX; ++l) s += rec[C::oFx + l * NX + tid] * VxL[l];
        QxL[tid] = rec[C::oLx + tid] + s;
      } else if (tid < NX + NU) {
        double s = 0.0;
        ASLR_UNROLL for (int l = 0; l < NX; ++l) s += rec[C::oFu + l * NU + (tid - NX)] * VxL[l];
        QuL[tid - NX] = rec[C::oLu + (tid - NX)] + s;
      }
      __syncthreads();
      ASLR_PROF(1);
      // ---- pass 2a: Qxx = Lxx + A Fx ----
      double qxx[2][4]; // vector path: rows row[h], columns 4g..; MFMA path: column tile ct, rows 16 wv + lk + 4 q
      if constexpr (MFMA) {
        double4_t q0 = {0.0, 0.0, 0.0, 0.0}, q1 = q0;
        const double *arow = AL + (16 * wv + li) * NX + lk;
        ASLR_UNROLL for (int ks = 0; ks < NKS; ++ks) {
          const double pa = arow[4 * ks];
          q0 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, bfx[0][ks], q0, 0, 0, 0);
          q1 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, bfx[1][ks], q1, 0, 0, 0);
        }
        ASLR_UNROLL for (int q = 0; q < 4; ++q) {
          const int rw = 16 * wv + lk + 4 * q;
          qxx[0][q] = rec[C::oLxx + rw * NX + li] + q0[q];
          qxx[1][q] = rec[C::oLxx + rw * NX + 16 + li] + q1[q];
        }
      } else
      {
        double c[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
        const double2 *arow0 = reinterpret_cast<const double2 *>(AL + row[0] * NX);
        const double2 *arow1 = reinterpret_cast<const double2 *>(AL + row[1] * NX);
        _Pragma("unroll 2") for (int l = 0; l < NX; l += 2) {
          const double2 p[2] = {arow0[l / 2], arow1[l / 2]};
          const double2 *f0 = reinterpret_cast<const double2 *>(rec + C::oFx + l * NX + 4 * gc);
          const double2 *f1 = reinterpret_cast<const double2 *>(rec + C::oFx + (l + 1) * NX + 4 * gc);
          const double2 f0a = f0[0], f0b = f0[1], f1a = f1[0], f1b = f1[1];
          ASLR_UNROLL for (int h = 0; h < 2; ++h) {
            c[h][0] += p[h].x * f0a.x; c[h][1] += p[h].x * f0a.y; c[h][2] += p[h].x * f0b.x; c[h][3] += p[h].x * f0b.y;
            c[h][0] += p[h].y * f1a.x; c[h][1] += p[h].y * f1a.y; c[h][2] += p[h].y * f1b.x; c[h][3] += p[h].y * f1b.y;
          }
        }
        ASLR_UNROLL for (int h = 0; h < 2; ++h) {
          const double2 *lx = reinterpret_cast<const double2 *>(rec + C::oLxx + row[h] * NX + 4 * gc);
          const double2 la = lx[0], lb2 = lx[1];
          qxx[h][0] = la.x + c[h][0]; qxx[h][1] = la.y + c[h][1]; qxx[h][2] = lb2.x + c[h][2]; qxx[h][3] = lb2.y + c[h][3];
        }
      }
      ASLR_PROF(2);
      // ---- pass 2b: [Qux | Quu] = [Lxu^T | Luu] + B [Fx | Fu] ----
      if constexpr (MFMA) {
        // B = (P Fu)^T is the left operand (rows >= nu read neighbouring LDS: unused results); wave w takes column tile w
        // of Qux with the Fx fragments it already holds, wave 1 also Quu
        double4_t x0 = {0.0, 0.0, 0.0, 0.0}, u0 = x0;
        const double *brow = BL + li * NX + lk;
        ASLR_UNROLL for (int ks = 0; ks < NKS; ++ks) {
          const double pa = brow[4 * ks];
          x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, wv ? bfx[1][ks] : bfx[0][ks], x0, 0, 0, 0);
          if (wv == 1) u0 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, rec[C::oFu + (4 * ks + lk) * NU + li], u0, 0, 0, 0);
        }
        const int col = 16 * wv + li;
        ASLR_UNROLL for (int q = 0; q < 4; ++q) {
          const int gq = lk + 4 * q;
          if (gq < NU && col < NX) {
            const double v = rec[C::oLxu + col * NU + gq] + x0[q];
            QuxL[gq * NX + col] = v;
            QuxT[col * 8 + gq] = v;
          }
          if (wv == 1 && gq < NU && li < NU) QuuL[gq * NU + li] = rec[C::oLuu + gq * NU + li] + u0[q] + (gq == li ? xr : 0.0);
        }
      } else {
      {
        double s = 0.0;
        const double2 *brow = reinterpret_cast<const double2 *>(BL + k2a * NX);
        _Pragma("unroll 2") for (int l = 0; l < NX; l += 2) {
          const double2 p = brow[l / 2];
          s += p.x * rec[C::oFx + l * NX + c2a];
          s += p.y * rec[C::oFx + (l + 1) * NX + c2a];
        }
        const double v = rec[C::oLxu + c2a * NU + k2a] + s;
        QuxL[k2a * NX + c2a] = v;
        QuxT[c2a * 8 + k2a] = v;
      }
      if (e1x) {
        double s = 0.0;
        const double2 *brow = reinterpret_cast<const double2 *>(BL + k2b * NX);
        _Pragma("unroll 2") for (int l = 0; l < NX; l += 2) {
          const double2 p = brow[l / 2];
          s += p.x * rec[C::oFx + l * NX + c2b];
          s += p.y * rec[C::oFx + (l + 1) * NX + c2b];
        }
        const double v = rec[C::oLxu + c2b * NU + k2b] + s;
        QuxL[k2b * NX + c2b] = v;
        QuxT[c2b * 8 + k2b] = v;
      } else if (e1u) {
        double s = 0.0;
        const double2 *brow = reinterpret_cast<const double2 *>(BL + k3 * NX);
        _Pragma("unroll 2") for (int l = 0; l < NX; l += 2) {
          const double2 p = brow[l / 2];
          s += p.x * rec[C::oFu + l * NU + c3];
          s += p.y * rec[C::oFu + (l + 1) * NU + c3];
        }
        QuuL[k3 * NU + c3] = rec[C::oLuu + k3 * NU + c3] + s + (k3 == c3 ? xr : 0.0);
      }
      }
      __syncthreads();
      ASLR_PROF(3);
      // ---- gains (wave 0): K = Quu^-1 Qux (one column per lane), k = Quu^-1 Qu, Quu k, Vx, d1, d2, stop ----
      if (wave0) {
        double qu[NU], kv[NU], Kc[NU], Quuk[NU];
        const int col = tid < NX ? tid : NX - 1;
        bool bad;
        const bool boxed = BOX && box && lim.has[mi]; // block-uniform
        if (BOX && boxed) {
          double Quu[NU][NU], lb[NU], ub[NU], k0[NU];
          ASLR_UNROLL for (int c = 0; c < NU; ++c) {
            ASLR_UNROLL for (int e = 0; e < NU; ++e) Quu[c][e] = QuuL[c * NU + e];
            qu[c] = QuL[c]; Kc[c] = QuxL[c * NX + col];
            const double ut = UKL[c];
            k0[c] = UKL[8 + c];
            lb[c] = lim.lb[mi][c] - ut;
            ub[c] = lim.ub[mi][c] - ut;
          }
#ifdef ASLR_BWD_PROFILE
          bad = lane_gains<NU>(Quu, qu, Kc, kv, true, lb, ub, k0, sp, prof_acc, prof_last);
#else
          bad = lane_gains<NU>(Quu, qu, Kc, kv, true, lb, ub, k0, sp);
#endif
        } else {
          double L[NU][NU], rinv[NU];
          ASLR_UNROLL for (int c = 0; c < NU; ++c) // (chol_rs reads the lower triangle only)
            ASLR_UNROLL for (int e = 0; e < NU; ++e) L[c][e] = e <= c ? QuuL[c * NU + e] : 0.0;
          bad = chol_rs<NU>(L, rinv);
          ASLR_UNROLL for (int c = 0; c < NU; ++c) { qu[c] = QuL[c]; kv[c] = qu[c]; Kc[c] = QuxL[c * NX + col]; }
          chol_solve_r<NU>(L, rinv, kv);
          chol_solve_r<NU>(L, rinv, Kc);
        }
        ASLR_UNROLL for (int c = 0; c < NU; ++c) {
          double s = 0.0;
          ASLR_UNROLL for (int e = 0; e < NU; ++e) s += QuuL[c * NU + e] * kv[e];
          Quuk[c] = s;
        }
        ASLR_UNROLL for (int c = 0; c < NU; ++c) { d1 += qu[c] * kv[c]; d2 -= kv[c] * Quuk[c]; stop += qu[c] * qu[c]; }
        {
          double s = 0.0, s2 = 0.0;
          ASLR_UNROLL for (int c = 0; c < NU; ++c) { s += Kc[c] * Quuk[c]; s2 += Kc[c] * qu[c]; }
          Vx_own = QxL[col] + s - 2.0 * s2;
        }
        if (tid < NX) { ASLR_UNROLL for (int c = 0; c < NU; ++c) KL[c * NX + tid] = Kc[c]; }
        if (tid == 0) {
          FlagL[0] = bad ? 1 : 0;
          if (!bad) {
            ASLR_UNROLL for (int c = 0; c < NU; ++c) { a.kff[tb * NU + c] = kv[c]; a.qu[tb * NU + c] = qu[c]; }
          }
        }
      }
      __syncthreads();
      if (FlagL[0]) { failed = true; break; } // block-uniform
      ASLR_PROF(4);
      // ---- Vxx (unsymmetrised, state regularisation on the diagonal), K to HBM ----
      {
        if constexpr (MFMA) {
          // Qux^T K on the matrix unit too: contraction over the nu controls, padded to 8 with the zeros set at the top
          // (rows / columns >= nx read neighbouring LDS: unused results)
          double4_t a0 = {0.0, 0.0, 0.0, 0.0}, a1 = a0;
          const double *qrow = QuxT + (16 * wv + li) * 8 + lk;
          ASLR_UNROLL for (int ks = 0; ks < 2; ++ks) {
            const double pa = qrow[4 * ks];
            const double *krow = KL + (4 * ks + lk) * NX + li;
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, krow[0], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, krow[16], a1, 0, 0, 0);
          }
          double acc[2][4];
          ASLR_UNROLL for (int q = 0; q < 4; ++q) { acc[0][q] = a0[q]; acc[1][q] = a1[q]; }
          ASLR_UNROLL for (int ct = 0; ct < 2; ++ct)
            ASLR_UNROLL for (int q = 0; q < 4; ++q) {
              const int rw = 16 * wv + lk + 4 * q, cw = 16 * ct + li;
              if (rw < NX && cw < NX) AL[rw * NX + cw] = (qxx[ct][q] - acc[ct][q]) + (rw == cw ? xr : 0.0);
            }
        } else {
          double acc[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}, k8[2][8];
          ASLR_UNROLL for (int h = 0; h < 2; ++h) {
            const double2 *qt = reinterpret_cast<const double2 *>(QuxT + row[h] * 8);
            ASLR_UNROLL for (int c = 0; c < 8; c += 2) { const double2 v = qt[c / 2]; k8[h][c] = v.x; k8[h][c + 1] = v.y; }
          }
          ASLR_UNROLL for (int c = 0; c < NU; ++c) {
            const double2 *kr = reinterpret_cast<const double2 *>(KL + c * NX + 4 * gc);
            const double2 ka = kr[0], kb = kr[1];
            ASLR_UNROLL for (int h = 0; h < 2; ++h) {
              acc[h][0] += k8[h][c] * ka.x; acc[h][1] += k8[h][c] * ka.y; acc[h][2] += k8[h][c] * kb.x; acc[h][3] += k8[h][c] * kb.y;
            }
          }
          ASLR_UNROLL for (int h = 0; h < 2; ++h) {
            if (cell[h]) {
              double2 *vo = reinterpret_cast<double2 *>(AL + row[h] * NX + 4 * g);
              double2 va, vb;
              va.x = (qxx[h][0] - acc[h][0]) + (row[h] == 4 * g + 0 ? xr : 0.0);
              va.y = (qxx[h][1] - acc[h][1]) + (row[h] == 4 * g + 1 ? xr : 0.0);
              vb.x = (qxx[h][2] - acc[h][2]) + (row[h] == 4 * g + 2 ? xr : 0.0);
              vb.y = (qxx[h][3] - acc[h][3]) + (row[h] == 4 * g + 3 ? xr : 0.0);
              vo[0] = va; vo[1] = vb;
            }
          }
        }
        a.kgain[tb * NU * NX + tid] = KL[tid];
        if (tid + NT < NU * NX) a.kgain[tb * NU * NX + tid + NT] = KL[tid + NT];
      }
      __syncthreads();
      ASLR_PROF(5);
      // ---- symmetrise into PT, NaN / Inf / >= 1e30 test ("backward_error") ----
      bool bad = false;
      double2 pa[2], pb[2];
      ASLR_UNROLL for (int h = 0; h < 2; ++h) {
        const int rw = row[h];
        const double2 *vr = reinterpret_cast<const double2 *>(AL + rw * NX + 4 * gc);
        const double2 va = vr[0], vb = vr[1];
        pa[h].x = 0.5 * (va.x + AL[(4 * gc + 0) * NX + rw]); pa[h].y = 0.5 * (va.y + AL[(4 * gc + 1) * NX + rw]);
        pb[h].x = 0.5 * (vb.x + AL[(4 * gc + 2) * NX + rw]); pb[h].y = 0.5 * (vb.y + AL[(4 * gc + 3) * NX + rw]);
        if (cell[h]) {
          double2 *po = reinterpret_cast<double2 *>(PT + rw * NX + 4 * g);
          po[0] = pa[h]; po[1] = pb[h];
          const double ent[4] = {pa[h].x, pa[h].y, pb[h].x, pb[h].y};
          bad = bad || inf_norm_bad<4>(fabs(pa[h].x) + fabs(pa[h].y) + fabs(pb[h].x) + fabs(pb[h].y), ent);
        }
      }
      if (gaps_on) {
        __syncthreads();
        if (tid < NX) {
          double vf = 0.0;
          const double2 *pr = reinterpret_cast<const double2 *>(PT + tid * NX), *fr = reinterpret_cast<const double2 *>(FL);
          ASLR_UNROLL for (int q = 0; q < NX / 2; ++q) { const double2 pv = pr[q], fv = fr[q]; vf += pv.x * fv.x; vf += pv.y * fv.y; }
          Vx_own += vf;
          if (fddp) {
            dgf -= Vx_own * FL[tid];
            dqf += FL[tid] * vf;
            a.vxxf[tb * NX + tid] = vf;
          }
        }
      }
      if (tid < NX) bad = bad || is_bad(fabs(Vx_own));
      if (__syncthreads_or(bad ? 1 : 0)) { failed = true; break; }
      if (sp.store_v) {
        if (tid < NX) a.vx[tb * NX + tid] = Vx_own;
        ASLR_UNROLL for (int h = 0; h < 2; ++h) {
          if (cell[h]) {
            double2 *o = reinterpret_cast<double2 *>(a.vxx + tb * NX * NX + row[h] * NX + 4 * g);
            o[0] = pa[h]; o[1] = pb[h];
          }
        }
      }
      if (tid < NX) VxL[tid] = Vx_own;
    }
#undef ASLR_BLK_PREFETCH
    ASLR_PROF(6);
    ASLR_PROF_FLUSH;
    // ---- end of sweep: publish or regularise and retry ----
    if (!failed) {
      if (fddp) { // per-column gap terms summed in column order by one thread
        __syncthreads();
        if (tid < NX) { RedL[tid] = dgf; RedL[32 + tid] = dqf; }
        __syncthreads();
        if (tid == 0) {
          dgf = 0.0; dqf = 0.0;
          for (int q = 0; q < NX; ++q) { dgf += RedL[q]; dqf += RedL[32 + q]; }
        }
      }
      if (tid == 0) {
        TF[ASLR_TF_STOP * B + b] = stop;
        if (fddp) {
          TF[ASLR_TF_DG * B + b] = d1 + dgf;
          TF[ASLR_TF_DQ * B + b] = d2 + dqf;
        }
        TF[ASLR_TF_D1 * B + b] = d1;
        TF[ASLR_TF_D2 * B + b] = d2;
        TF[ASLR_TF_XREG * B + b] = xreg;
        TI[ASLR_TI_STATUS * B + b] = status;
      }
      need = false;
    } else {
      status |= ASLR_ST_BACKWARD_ERR;
      if (sp.standalone) {
        if (tid == 0) TI[ASLR_TI_STATUS * B + b] = status;
        need = false;
      } else {
        const double grown = xreg * sp.reg_incfactor; // (a regularisation that cannot grow counts as the ceiling: the kernel must terminate)
        xreg = (grown > xreg) ? grown : sp.reg_max;
        if (xreg > sp.reg_max) xreg = sp.reg_max;
        if (!(xreg < sp.reg_max)) { // (== reg_max; also a NaN ceiling ends the retries)
          status |= ASLR_ST_REG_MAX;
          if (tid == 0) {
            TF[ASLR_TF_XREG * B + b] = xreg;
            TI[ASLR_TI_STATUS * B + b] = status;
            TI[ASLR_TI_DONE * B + b] = 1;
          }
          need = false;
        }
      }
    }
  }
}

template <int NX, int NU>
int launch_backward_blk(const KArgs &k, const SolverDev &sd, const ModelLimits &lim, bool all_feasible, bool mfma, hipStream_t st) {
  // mfma = false (ASLR_BLK_MFMA=0 when the handle was created) selects the vector-FMA products (comparison runs; both
  // paths give the same bits)
  const dim3 grid(k.b1 - k.b0), block(BwdBlk<NX, NU>::NT);
  if (sd.solver == ASLR_SOLVER_BOXDDP) { // (gap terms compiled in: a cold-started BoxDDP solve is infeasible at first)
    if (mfma) hipLaunchKernelGGL((backward_blk_kernel<NX, NU, true, true, true>), grid, block, 0, st, k, sd, lim);
    else hipLaunchKernelGGL((backward_blk_kernel<NX, NU, true, false, true>), grid, block, 0, st, k, sd, lim);
  } else if (mfma) {
    if (all_feasible) hipLaunchKernelGGL((backward_blk_kernel<NX, NU, false, true>), grid, block, 0, st, k, sd, lim);
    else hipLaunchKernelGGL((backward_blk_kernel<NX, NU, true, true>), grid, block, 0, st, k, sd, lim);
  } else {
    if (all_feasible) hipLaunchKernelGGL((backward_blk_kernel<NX, NU, false, false>), grid, block, 0, st, k, sd, lim);
    else hipLaunchKernelGGL((backward_blk_kernel<NX, NU, true, false>), grid, block, 0, st, k, sd, lim);
  }
  HIP_TRY(hipGetLastError());
  return ASLR_OK;
}

} // namespace aslr
