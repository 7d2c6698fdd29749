// Implicit-GEMM convolution / dense engine on the f32 matrix cores of gfx950.
//
//   y[m, co] = epilogue( sum_{kh,kw,ci} prologue(x[n, ho*s+kh-pt, wo*s+kw-pl, ci]) * w[(kh*KW+kw)*Cin+ci, co] )
//
// GEMM view: M = N*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin.  One 256-thread
// workgroup (4 wave64 in a 2x2 arrangement) owns a BM x BN output tile and walks
// K in BK=16 slabs.  A (im2col rows, gathered on the fly, never materialised) and
// B (HWIO weights) slabs are staged global -> registers -> LDS with one barrier per
// slab (register double buffering: the global loads of slab k+1 are in flight
// while the MFMAs of slab k run).  A is stored K-major in LDS ([BK][BM+2]) so that
// the v_mfma_f32_32x32x2_f32 operand fetch (lane l: A[i=l&31][k=l>>5]) is a
// conflict-free ds_read_b32 and the transposing stores hit 32 distinct banks.
// The f32 MFMA is an exact fmaf chain (cdna_hip_programming.md section 3), so the
// result is bitwise a k-ordered fp32 dot product: parity-grade numerics.
//
// Fusions (all optional): GroupNorm+ReLU / ReLU+GroupNorm / ReLU / affine applied
// to A while staging (padding zeros are inserted AFTER the prologue, as in the
// reference where the conv pads the normalised tensor); bias, residual add,
// bilinear x2 up-sample-add (FPN), ReLU and a row mask in the epilogue.
//
// Replaces: flax.linen.Conv / StdConv / Dense call sites of
// snap/models/resnet.py:83-132,200-215, image_encoder.py:67-94, layers.py:66-77,
// streetview_encoder.py:228,281, bev_mapper.py:285 and the direct correlation of
// pose_exhaustive_voting.py:86-91.
#include "conv_common.h"

namespace {

template <int BM, int BN, bool VEC, int PRO, int BK, bool DMA = false>
__device__ __forceinline__ void conv_igemm_body(const ConvArgs& a) {
  constexpr int AS = BM + 2;    // LDS row stride of the K-major A slab
  constexpr int TM = BM / 64;   // 32x32 MFMA tiles per wave along M
  constexpr int TN = BN / 64;   // ... along N
  constexpr int QPR = BK / 4;             // VEC: float4 quads per A row of the slab
  constexpr int RPP = 256 / QPR;          // VEC: rows staged per pass
  constexpr int AROWS = BM / RPP;         // VEC: float4 rows per thread
  constexpr int SRPP = 256 / BK;          // SCALAR: rows staged per pass
  constexpr int AELEMS = BM / SRPP;       // SCALAR: scalars per thread
  constexpr int BQ = BN / 4;              // float4 per B row
  constexpr int BROWS_PER_PASS = 256 / BQ;
  constexpr int BPASS = BK / BROWS_PER_PASS;  // float4 per thread for B

  // one LDS block: the A/B slab ring during the main loop, then the staging buffer
  // of the epilogue (64 x BN fp32) -- a single __shared__ object by design.
  constexpr int kDmaStages = 3;
  constexpr int kSlabFloats = DMA ? kDmaStages * (BK * BM + BK * BN) : 2 * BK * AS + 2 * BK * BN;
  constexpr int kStageFloats = 64 * BN;
  constexpr int kSmemFloats = kSlabFloats > kStageFloats ? kSlabFloats : kStageFloats;
  __shared__ __attribute__((aligned(16))) float smem[kSmemFloats];
  float* const As0 = smem;
  float* const Bs0 = smem + 2 * BK * AS;

  const SnapConvDesc& d = a.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  // XCD-aware tile order (blocks are dispatched round-robin over the 8 XCDs, each
  // with a private L2): row-tile r lives on XCD r % 8 and ALL its column tiles run
  // back to back on that XCD, so the A rows (and 3x3 halos of the next row tile of
  // the same XCD) are re-read from its L2 instead of HBM.  A pure speed mapping.
  // split-K (small-M, deep-K layers that cannot fill 256 CUs with output tiles): the
  // outermost grid dimension walks slices of the K slabs; partial tiles go to a workspace
  // and a second kernel sums them in fixed order and applies the epilogue.
  const int ncol = a.ncol;
  const int split = a.ksplit > 1 ? blockIdx.x / a.tiles_per_split : 0;
  const int bid = a.ksplit > 1 ? blockIdx.x - split * a.tiles_per_split : blockIdx.x;
  const int xcd = bid & 7;
  const int seq = bid >> 3;
  const int col_t = seq % ncol;
  const int row_t = (seq / ncol) * 8 + xcd;
  // Row-indexed mode (compacted voxel lists): the launch covers the worst case M and the
  // workgroups past the device-side count leave at once.
  const int Meff = a.row_count ? min(*a.row_count, a.M) : a.M;
  if (row_t * BM >= Meff) return;  // padding blocks of the last row group (uniform per block)
  const int m0 = row_t * BM;
  const int n0 = col_t * BN;
  const int HoWo = d.Ho * d.Wo;
  constexpr bool need_gn = (PRO == SNAP_PRO_GN_RELU || PRO == SNAP_PRO_RELU_GN);

  // ---- per-thread A row bookkeeping -------------------------------------
  constexpr int NR = VEC ? AROWS : AELEMS;
  int r_n[NR], r_hb[NR], r_wb[NR];
  bool r_ok[NR];
  const float* r_px[NR];   // &x[n, hb, wb, 0]  (may point outside the image: used only when in-bounds)
  int64_t r_gn[NR];        // n * Cin  (GroupNorm statistics row)
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int row = VEC ? (tid / QPR) + RPP * i : (tid / BK) + SRPP * i;
    const int m = m0 + row;
    r_ok[i] = m < Meff;
    int mm = r_ok[i] ? m : 0;
    if (a.rows_in) mm = a.rows_in[mm];
    const int n = mm / HoWo;
    const int r = mm - n * HoWo;
    const int ho = r / d.Wo;
    const int wo = r - ho * d.Wo;
    r_n[i] = n;
    r_hb[i] = ho * d.stride - d.pad_t;
    r_wb[i] = wo * d.stride - d.pad_l;
    r_px[i] = a.x + (((int64_t)n * d.H + r_hb[i]) * d.W + r_wb[i]) * d.Cin_stride;
    r_gn[i] = (int64_t)n * d.Cin;
  }
  const int akq = tid % QPR;  // VEC: which float4 of the BK-wide K slab
  const int akid = tid % BK;  // SCALAR: which k of the slab

  // B loader coordinates
  const int bcq = tid % BQ;
  const int bkr = tid / BQ;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging registers
  f32x4 xa[VEC ? AROWS : 1], xmu[VEC ? AROWS : 1], xsc[VEC ? AROWS : 1], xbeta;
  bool xin[VEC ? AROWS : 1];
  float sa[VEC ? 1 : AELEMS], smu[VEC ? 1 : AELEMS], ssc[VEC ? 1 : AELEMS], sbeta = 0.f;
  bool sin_[VEC ? 1 : AELEMS];
  f32x4 xb[BPASS];
  bool xbok[BPASS];
  int cur_c = 0;  // channel of element 0 of this thread's quad (VEC)

  const int kt_begin = a.ksplit > 1 ? split * a.slabs_per_split : 0;
  const int kt_end = a.ksplit > 1 ? min(a.nk, kt_begin + a.slabs_per_split) : a.nk;
  // K-slab walk state (VEC): (kpos = kh*KW+kw, ct)
  int kpos = 0, ct = 0, kh = 0, kw = 0;
  if (VEC && kt_begin > 0) {
    kpos = kt_begin / a.ctiles;
    ct = kt_begin - kpos * a.ctiles;
    kh = kpos / d.KW;
    kw = kpos - kh * d.KW;
  }

  // Per-tap state of the VEC loader: whether row i's (kh, kw) tap is inside the image and the
  // pointer to its channel 0.  Recomputed only when the walk moves to the next tap (once per
  // Cin/BK slabs; never again for 1x1 convs) instead of in every slab.
  const float* tap_px[VEC ? AROWS : 1];
  bool tap_in[VEC ? AROWS : 1];
  auto set_tap = [&]() {
    if constexpr (VEC) {
      const int64_t delta = ((int64_t)kh * d.W + kw) * d.Cin_stride;
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        const int hi = r_hb[i] + kh, wi = r_wb[i] + kw;
        tap_in[i] = r_ok[i] && hi >= 0 && hi < d.H && wi >= 0 && wi < d.W;
        tap_px[i] = r_px[i] + delta;
      }
    }
  };
  set_tap();

  auto load_slab = [&](int kt) {
    if constexpr (VEC) {
      const int c = ct * BK + 4 * akq;
      cur_c = c;
      const bool cvalid = c < d.Cin;
      // Loads are unconditional from a clamped (always mapped) address; invalid
      // lanes are zeroed when the slab is stored.  No divergent branches.
      if constexpr (need_gn) xbeta = *reinterpret_cast<const f32x4*>(a.gn_beta + (cvalid ? c : 0));
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        // (kh, kw) bounds test and tap pointer are per-TAP state (set_tap), not per slab
        const bool inb = tap_in[i] && cvalid;
        xin[i] = inb;
        const float* px = inb ? tap_px[i] + c : a.x;
        xa[i] = *reinterpret_cast<const f32x4*>(px);
        if constexpr (need_gn) {
          const int64_t so = inb ? r_gn[i] + c : (int64_t)0;
          xmu[i] = *reinterpret_cast<const f32x4*>(a.gn_mu + so);
          xsc[i] = *reinterpret_cast<const f32x4*>(a.gn_sc + so);
        }
      }
      // B rows of this slab
      const int wrow0 = kpos * d.Cin + ct * BK;
#pragma unroll
      for (int p = 0; p < BPASS; ++p) {
        const int kr = bkr + p * BROWS_PER_PASS;
        const int col = n0 + 4 * bcq;
        const bool ok = (ct * BK + kr) < d.Cin && col < d.Cout;
        const int64_t wo = ok ? (int64_t)(wrow0 + kr) * d.Cout + col : (int64_t)0;
        // NB: nothing may consume the loaded registers before the MFMA block (a
        // select here would pull the s_waitcnt vmcnt(0) in front of the MFMAs).
        xb[p] = *reinterpret_cast<const f32x4*>(a.w + wo);
        xbok[p] = ok;
      }
    } else {
      const int k = kt * BK + akid;
      const bool kvalid = k < a.K;
      const int kk = kvalid ? k : 0;
      const int kp = kk / d.Cin;
      const int c = kk - kp * d.Cin;
      const int skh = kp / d.KW;
      const int skw = kp - skh * d.KW;
      sbeta = 0.f;
      if constexpr (need_gn) sbeta = a.gn_beta[c];
#pragma unroll
      for (int i = 0; i < AELEMS; ++i) {
        const int hi = r_hb[i] + skh, wi = r_wb[i] + skw;
        const bool inb = r_ok[i] && kvalid && hi >= 0 && hi < d.H && wi >= 0 && wi < d.W;
        sin_[i] = inb;
        const int64_t sdelta = ((int64_t)skh * d.W + skw) * d.Cin_stride + c;
        const float* px = inb ? r_px[i] + sdelta : a.x;
        sa[i] = *px;
        if constexpr (need_gn) {
          const int64_t so = inb ? r_gn[i] + c : (int64_t)0;
          smu[i] = a.gn_mu[so];
          ssc[i] = a.gn_sc[so];
        }
      }
#pragma unroll
      for (int p = 0; p < BPASS; ++p) {
        const int kr = bkr + p * BROWS_PER_PASS;
        const int col = n0 + 4 * bcq;
        const bool ok = (kt * BK + kr) < a.K && col < d.Cout;
        const int64_t wo = ok ? (int64_t)(kt * BK + kr) * d.Cout + col : (int64_t)0;
        // NB: nothing may consume the loaded registers before the MFMA block (a
        // select here would pull the s_waitcnt vmcnt(0) in front of the MFMAs).
        xb[p] = *reinterpret_cast<const f32x4*>(a.w + wo);
        xbok[p] = ok;
      }
    }
  };

  auto advance = [&]() {
    if constexpr (VEC) {
      if (++ct == a.ctiles) {
        ct = 0;
        ++kpos;
        if (++kw == d.KW) { kw = 0; ++kh; }
        set_tap();
      }
    }
  };

  auto store_slab = [&](int buf) {
    float* as = As0 + buf * (BK * AS);
    float* bs = Bs0 + buf * (BK * BN);
    if constexpr (VEC) {
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        const int row = (tid / QPR) + RPP * i;
        f32x4 v = xa[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float pv;
          if constexpr (need_gn)
            pv = apply_pro<PRO>(v[e], xmu[i][e], xsc[i][e], xbeta[e], d.in_scale, d.in_shift);
          else
            pv = apply_pro<PRO>(v[e], 0.f, 0.f, 0.f, d.in_scale, d.in_shift);
          v[e] = (xin[i] && (cur_c + e < d.Cin)) ? pv : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) as[(4 * akq + e) * AS + row] = v[e];
      }
    } else {
#pragma unroll
      for (int i = 0; i < AELEMS; ++i) {
        const int row = (tid / BK) + SRPP * i;
        float pv;
        if constexpr (need_gn)
          pv = apply_pro<PRO>(sa[i], smu[i], ssc[i], sbeta, d.in_scale, d.in_shift);
        else
          pv = apply_pro<PRO>(sa[i], 0.f, 0.f, 0.f, d.in_scale, d.in_shift);
        as[akid * AS + row] = sin_[i] ? pv : 0.f;
      }
    }
#pragma unroll
    for (int p = 0; p < BPASS; ++p) {
      const int kr = bkr + p * BROWS_PER_PASS;
      *reinterpret_cast<f32x4*>(bs + kr * BN + 4 * bcq) =
          xbok[p] ? xb[p] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };

  // ---- main loop ---------------------------------------------------------
  const int l31 = lane & 31, lhi = lane >> 5;
  if constexpr (DMA) {
    // LDS-DMA pipeline (prologues without per-(image, channel) operands: NONE / RELU).
    // Both operands go global -> LDS by global_load_lds, three stages deep, no staging
    // registers and no store phase: the wave only issues 4 DMA instructions per slab and
    // runs MFMAs.  A stage = [row][16 k] with the 16-byte k-quads XOR-swizzled by
    // (row >> 2) & 3: four lanes fetch one row's contiguous 64 bytes (coalesced) and the
    // operand fetch is a conflict-free ds_read_b128 per 32-row block and lane group; lane
    // group g consumes k = 8g .. 8g+7 of the slab, so MFMA j pairs k = j with k = 8 + j (B rows
    // are read to match; the fp32 sum of a slab is re-associated, nothing else changes).
    static_assert(VEC && BK == 16 && (PRO == SNAP_PRO_NONE || PRO == SNAP_PRO_RELU), "DMA path");
    constexpr int A_ST = BK * BM, B_ST = BK * BN;
    float* const Ad = smem;
    float* const Bd = smem + kDmaStages * A_ST;
    const int nslab = kt_end - kt_begin;
    auto issue = [&](int stage) {
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        const int row = (tid / QPR) + RPP * i;
        const int q = akq ^ ((row >> 2) & 3);            // logical k-quad held by this slot
        const int c = ct * BK + 4 * q;
        const bool ok = tap_in[i] && c < d.Cin;          // (Cin % 4 == 0 on the VEC path)
        const float* src = ok ? tap_px[i] + c : kZeroChunk;
        __builtin_amdgcn_global_load_lds((cglobal_void_t*)src,
                                         (lds_void_t*)(Ad + stage * A_ST + 4 * (tid + 256 * i)), 16, 0, 0);
      }
#pragma unroll
      for (int p = 0; p < BPASS; ++p) {
        const int kr = bkr + p * BROWS_PER_PASS;
        const int col = n0 + 4 * bcq;
        const bool ok = (ct * BK + kr) < d.Cin && col < d.Cout;
        const float* src = ok ? a.w + ((int64_t)(kpos * d.Cin + ct * BK + kr) * d.Cout + col) : kZeroChunk;
        __builtin_amdgcn_global_load_lds((cglobal_void_t*)src,
                                         (lds_void_t*)(Bd + stage * B_ST + 4 * (tid + 256 * p)), 16, 0, 0);
      }
    };
    if (nslab > 0) { issue(0); advance(); }
    if (nslab > 1) { issue(1); advance(); }
    for (int t = 0; t < nslab; ++t) {
      if (t + 1 < nslab) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AROWS + BPASS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();   // slab t landed for every wave; stage (t+2)%3 (slab t-1) is free
      if (t + 2 < nslab) { issue((t + 2) % kDmaStages); advance(); }
      const float* as = Ad + (t % kDmaStages) * A_ST;
      const float* bs = Bd + (t % kDmaStages) * B_ST;
      f32x4 av[TM][2];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int R = wr * (BM / 2) + i * 32 + l31;
          const int pq = (lhi * 2 + h) ^ ((R >> 2) & 3);
          av[i][h] = *reinterpret_cast<const f32x4*>(as + 4 * (R * 4 + pq));
          if constexpr (PRO == SNAP_PRO_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) av[i][h][e] = snap_relu(av[i][h][e]);
          }
        }
      // B operand fetch runs one k-pair ahead of the MFMAs (pinned below)
      float bv[2][TN];
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) bv[0][jn] = bs[(8 * lhi) * BN + wc * (BN / 2) + jn * 32 + l31];
#pragma unroll
      for (int j = 0; j < BK / 2; ++j) {
        const int cu = j & 1, nx = cu ^ 1;
        if (j + 1 < BK / 2) {
#pragma unroll
          for (int jn = 0; jn < TN; ++jn)
            bv[nx][jn] = bs[(8 * lhi + j + 1) * BN + wc * (BN / 2) + jn * 32 + l31];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jn = 0; jn < TN; ++jn)
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][j >> 2][j & 3], bv[cu][jn], acc[i][jn], 0, 0, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * TM + TN, 0);   // A quads + first B pair
#pragma unroll
      for (int j = 0; j < BK / 2; ++j) {
        if (j + 1 < BK / 2) __builtin_amdgcn_sched_group_barrier(0x100, TN, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
      }
    }
    __syncthreads();     // the epilogue reuses the ring
  } else {
  load_slab(kt_begin);
  advance();
  store_slab(0);
  __syncthreads();

  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int cur = (kt - kt_begin) & 1;
    const bool more = kt + 1 < kt_end;
    if (more && !SNAP_IGEMM_ABL(1)) {
      load_slab(kt + 1);
      advance();
    }
    const float* as = As0 + cur * (BK * AS);
    const float* bs = Bs0 + cur * (BK * BN);
    if (a.prio) __builtin_amdgcn_s_setprio(1);
    // LDS -> register operand fetch runs one k-pair ahead of the MFMAs.
    float av[2][TM], bv[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) av[0][i] = as[lhi * AS + wr * (BM / 2) + i * 32 + l31];
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[0][j] = bs[lhi * BN + wc * (BN / 2) + j * 32 + l31];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int cu = kk & 1, nx = cu ^ 1;
      if (kk + 1 < BK / 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          av[nx][i] = as[(2 * (kk + 1) + lhi) * AS + wr * (BM / 2) + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bv[nx][j] = bs[(2 * (kk + 1) + lhi) * BN + wc * (BN / 2) + j * 32 + l31];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] =
              __builtin_amdgcn_mfma_f32_32x32x2f32(av[cu][i], bv[cu][j], acc[i][j], 0, 0, 0);
    }
    // Pin the schedule: the ds_reads of k-pair kk+1 are issued BEFORE the MFMAs of
    // k-pair kk, so the LDS latency hides under 4 x 64 MFMA cycles.
    __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      if (kk + 1 < BK / 2) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
    }
    if (a.prio) __builtin_amdgcn_s_setprio(0);
    if (!SNAP_IGEMM_ABL(2)) {
      if (more) store_slab(cur ^ 1);
      if (!SNAP_IGEMM_ABL(8)) __syncthreads();   // bit3: keep the stores, drop only the barrier
    }
  }

  }

  conv_epilogue<BM, BN>(a, acc, smem, m0, n0, Meff, row_t, split);
}


// Two entry points over one body.  Variants WITHOUT a GroupNorm prologue fit 128 VGPRs
// and are held to 4 waves per SIMD (measured +3-4 % on the big Dense layers: more waves to
// cover the global-load latency); the GroupNorm variants carry the statistics operands
// and would spill at that budget, so they keep the default allocation (3 waves).
template <int BM, int BN, bool VEC, int PRO, int BK>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
  conv_igemm_body<BM, BN, VEC, PRO, BK>(a);
}
template <int BM, int BN, bool VEC, int PRO, int BK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void conv_igemm_kernel_occ4(const ConvArgs a) {
  conv_igemm_body<BM, BN, VEC, PRO, BK>(a);
}
template <int BM, int BN, int PRO>
__global__ __launch_bounds__(256) void conv_igemm_kernel_dma(const ConvArgs a) {
  conv_igemm_body<BM, BN, true, PRO, 16, true>(a);
}

inline bool conv_dma_enabled() { return true; }   // (the LDS-DMA loader of the NONE / RELU prologues: settled)


template <int BM, int BN, bool VEC, int PRO, int BK>
int launch(ConvArgs a, hipStream_t s) {
  if (VEC) {
    a.ctiles = (a.d.Cin + BK - 1) / BK;
    a.nk = a.d.KH * a.d.KW * a.ctiles;
  } else {
    a.ctiles = 0;
    a.nk = (a.K + BK - 1) / BK;
  }
  const int64_t nrow = snap_cdiv(a.M, BM);
  a.ncol = (int)snap_cdiv(a.d.Cout, BN);
  a.gn_slabs = (a.d.Ho * a.d.Wo) / BM + 2;
  int64_t nblocks = snap_cdiv(nrow, 8) * 8 * a.ncol;
  if (nblocks > 0x7fffffffLL) return SNAP_ERR_BAD_SHAPE;
  // split-K decision (needs a caller-provided workspace; plain epilogues only)
  a.ksplit = 1;
  a.tiles_per_split = (int)nblocks;
  a.slabs_per_split = a.nk;
  const int64_t tiles = nrow * a.ncol;
  const int target = splitk_target();
  if (a.kpartial && target > 0 && tiles <= splitk_max_tiles() && a.nk >= 16 && !a.rows_in &&
      !a.rows_out && !a.row_count && !a.gn_partial &&
      !(a.d.epilogue & SNAP_EPI_UPSAMPLE2X_ADD)) {
    int64_t S = (target + tiles - 1) / tiles;
    S = S < a.nk / 8 ? S : a.nk / 8;                       // >= 8 slabs per split
    const int64_t fit = (int64_t)(a.kpartial_bytes / ((size_t)a.M * a.d.Cout * sizeof(float)));
    S = S < fit ? S : fit;
    if (S >= 2) {
      a.slabs_per_split = (int)((a.nk + S - 1) / S);
      a.ksplit = (a.nk + a.slabs_per_split - 1) / a.slabs_per_split;
      nblocks *= a.ksplit;
    }
  }
  constexpr bool kGn = PRO == SNAP_PRO_GN_RELU || PRO == SNAP_PRO_RELU_GN;
  constexpr bool kDmaOk = VEC && BK == 16 && (PRO == SNAP_PRO_NONE || PRO == SNAP_PRO_RELU);
  bool launched = false;
  if constexpr (kDmaOk) {
    if (conv_dma_enabled()) {
      hipLaunchKernelGGL((conv_igemm_kernel_dma<BM, BN, PRO>), dim3((unsigned)nblocks), dim3(256), 0,
                         s, a);
      launched = true;
    }
  }
  if (launched) {
  } else if constexpr (!kGn && BK == 16 && (VEC || BM * BN < 128 * 128)) {  // (scalar 128x128 would spill)
    hipLaunchKernelGGL((conv_igemm_kernel_occ4<BM, BN, VEC, PRO, BK>), dim3((unsigned)nblocks),
                       dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, VEC, PRO, BK>), dim3((unsigned)nblocks),
                       dim3(256), 0, s, a);
  }
  SNAP_CHECK_LAUNCH();
  if (a.ksplit > 1) {
    const int64_t total4 = (int64_t)a.M * (a.d.Cout / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)snap_cdiv(total4, 256)), dim3(256), 0, s,
                       (const float*)a.kpartial, a.ksplit, (int64_t)a.M, a.d.Cout, a.d.Cout_stride,
                       a.d.epilogue, a.bias, a.residual, a.row_mask, a.y);
    SNAP_CHECK_LAUNCH();
  }
  return SNAP_OK;
}

template <int BM, int BN, bool VEC, int BK>
int launch_pro(const ConvArgs& a, hipStream_t s) {
  switch (a.d.prologue) {
    case SNAP_PRO_NONE: return launch<BM, BN, VEC, SNAP_PRO_NONE, BK>(a, s);
    case SNAP_PRO_AFFINE: return launch<BM, BN, VEC, SNAP_PRO_AFFINE, BK>(a, s);
    case SNAP_PRO_GN_RELU:
      // the scalar (unaligned / Cin % 4 != 0) path carries no GroupNorm variant.
      if constexpr (VEC) return launch<BM, BN, VEC, SNAP_PRO_GN_RELU, BK>(a, s);
      return SNAP_ERR_UNSUPPORTED;
    case SNAP_PRO_RELU_GN:
      if constexpr (VEC) return launch<BM, BN, VEC, SNAP_PRO_RELU_GN, BK>(a, s);
      return SNAP_ERR_UNSUPPORTED;
    case SNAP_PRO_RELU: return launch<BM, BN, VEC, SNAP_PRO_RELU, BK>(a, s);
    default: return SNAP_ERR_UNSUPPORTED;
  }
}

// K-slab depth of the big tiles (SnapConvExtras.bk_hint = 16 | 32; default 16, set by
// measurement on MI355X)


template <bool VEC>
int launch_tile(const ConvArgs& a, hipStream_t s) {
  const TileChoice t = choose_tile(a.M, a.d.Cout, a.d.tile_hint, desc_k(a.d));
  const bool deep = VEC && a.bk == 32 && a.d.Cin >= 32;
  if (t.bm == 128 && t.bn == 128) {
    if constexpr (VEC) {
      if (deep) return launch_pro<128, 128, VEC, 32>(a, s);
    }
    return launch_pro<128, 128, VEC, 16>(a, s);
  }
  if (t.bm == 128) {
    if constexpr (VEC) {
      if (deep) return launch_pro<128, 64, VEC, 32>(a, s);
    }
    return launch_pro<128, 64, VEC, 16>(a, s);
  }
  if (t.bn == 128) return launch_pro<64, 128, VEC, 16>(a, s);
  return launch_pro<64, 64, VEC, 16>(a, s);
}

}  // namespace

extern "C" int snap_conv2d_nhwc_f32(const SnapConvDesc* desc, const float* x,
                                    const float* w, float* y, const float* gn_mu,
                                    const float* gn_sc, const float* gn_beta,
                                    const float* bias, const float* residual,
                                    const float* up_prev, const uint8_t* row_mask,
                                    void* stream) {
  return snap_conv2d_nhwc_ex_f32(desc, x, w, y, gn_mu, gn_sc, gn_beta, bias, residual, up_prev,
                                 row_mask, nullptr, stream);
}

// (split_parts: SnapConvExtras.w_split_parts of the launch -- the weights-stationary kernel of the
//  two-part split engine emits its statistics per 32-row slab; 0 = any other engine)
extern "C" int32_t snap_conv2d_tile_rows_ex(const SnapConvDesc* desc, int32_t split_parts) {
  if (!desc) return 0;
  const int kind = snapconv::stationary_kind(*desc, split_parts, false);
  if (kind == 3) return -(desc->H * ((desc->W + 29) / 30));   // (< 0: that many slabs per image, all live)
  if (kind == 2) return 32;
  return choose_tile((int64_t)desc->N * desc->Ho * desc->Wo, desc->Cout, desc->tile_hint, desc_k(*desc)).bm;
}

extern "C" size_t snap_conv2d_gn_partial_bytes_ex(const SnapConvDesc* desc, int32_t split_parts) {
  if (!desc) return 0;
  const SnapConvDesc& d = *desc;
  const int64_t HoWo = (int64_t)d.Ho * d.Wo;
  const int bm = snap_conv2d_tile_rows_ex(desc, split_parts);
  if (bm < 0) return (size_t)d.N * (size_t)(-bm) * d.Cout * 2 * sizeof(float);
  if (HoWo < bm) return 0;  // a tile would straddle more than two images: not produced
  return (size_t)d.N * (HoWo / bm + 2) * d.Cout * 2 * sizeof(float);
}

extern "C" size_t snap_conv2d_gn_partial_bytes(const SnapConvDesc* desc) {
  return snap_conv2d_gn_partial_bytes_ex(desc, 0);
}

// ... of a SPLIT-K launch of the split-operand engine (snap_conv2d_workspace_bytes(desc) > 0 and the
// workspace passed): partial sums per 32-row slab out of the reduce pass (SnapConvExtras.gn_partial_rows
// = 32; snap_group_norm_stats_from_partial_f32 with tile_rows = 32).  0 = not available for this shape.
extern "C" size_t snap_conv2d_splitk_gn_partial_bytes(const SnapConvDesc* desc) {
  if (!desc || snap_conv2d_workspace_bytes(desc) == 0) return 0;
  const SnapConvDesc& d = *desc;
  const int64_t HoWo = (int64_t)d.Ho * d.Wo;
  const int q = d.Cout >> 2;
  if (HoWo < 32 || (d.Cout & 3) || d.Cout_stride != d.Cout || 256 % (q < 256 ? q : 256) != 0) return 0;
  return (size_t)d.N * (HoWo / 32 + 2) * d.Cout * 2 * sizeof(float);
}

extern "C" size_t snap_conv2d_workspace_bytes(const SnapConvDesc* desc) {
  if (!desc) return 0;
  const SnapConvDesc& d = *desc;
  const int64_t M = (int64_t)d.N * d.Ho * d.Wo;
  const TileChoice t = choose_tile(M, d.Cout, d.tile_hint, desc_k(d));
  const int64_t tiles = snap_cdiv(M, t.bm) * snap_cdiv((int64_t)d.Cout, t.bn);
  const int target = splitk_target();
  const int64_t nk = (int64_t)d.KH * d.KW * ((d.Cin + 15) / 16);
  if (target <= 0 || tiles > splitk_max_tiles() || nk < 16 || (d.epilogue & SNAP_EPI_UPSAMPLE2X_ADD)) return 0;
  int64_t S = (target + tiles - 1) / tiles;
  S = S < nk / 8 ? S : nk / 8;
  if (S < 2) return 0;
  return (size_t)S * M * d.Cout * sizeof(float);
}

extern "C" int32_t snap_conv2d_tile_rows(const SnapConvDesc* desc) {
  return snap_conv2d_tile_rows_ex(desc, 0);
}

extern "C" int32_t snap_conv2d_stationary_kind(const SnapConvDesc* desc, int32_t parts) {
  if (!desc) return 0;
  return snapconv::stationary_kind(*desc, parts, false);
}

// ---- pre-split launches (conv_ps.hip) ---------------------------------------------------------
extern "C" int32_t snap_conv2d_presplit_tile_rows(const SnapConvDesc* desc, int32_t ps_tile) {
  if (!desc) return 0;
  return snapconv::ps_choose_tile((int64_t)desc->N * desc->Ho * desc->Wo, desc->Cout, ps_tile).bm;
}

extern "C" size_t snap_conv2d_presplit_gn_partial_bytes(const SnapConvDesc* desc, int32_t ps_tile) {
  if (!desc) return 0;
  const SnapConvDesc& d = *desc;
  const int64_t HoWo = (int64_t)d.Ho * d.Wo;
  const snapconv::PsTile t = snapconv::ps_choose_tile((int64_t)d.N * HoWo, d.Cout, ps_tile);
  if (HoWo < t.bm) return 0;
  return (size_t)d.N * (HoWo / t.bm + 2) * d.Cout * 2 * sizeof(float);
}

extern "C" size_t snap_conv2d_presplit_workspace_bytes(const SnapConvDesc* desc, int32_t ps_tile) {
  if (!desc) return 0;
  const SnapConvDesc& d = *desc;
  if (d.Cin % 16 != 0 || (d.epilogue & SNAP_EPI_UPSAMPLE2X_ADD)) return 0;
  const int64_t M = (int64_t)d.N * d.Ho * d.Wo;
  const snapconv::PsTile t = snapconv::ps_choose_tile(M, d.Cout, ps_tile);
  const int64_t nk = (int64_t)d.KH * d.KW * (d.Cin / 16);
  const int S = snapconv::ps_ksplit(M, d.Cout, nk, t.bm, t.bn, (size_t)-1);
  return S >= 2 ? (size_t)S * M * d.Cout * sizeof(float) : 0;
}

extern "C" int snap_conv2d_nhwc_ex_f32(const SnapConvDesc* desc, const float* x,
                                       const float* w, float* y, const float* gn_mu,
                                       const float* gn_sc, const float* gn_beta,
                                       const float* bias, const float* residual,
                                       const float* up_prev, const uint8_t* row_mask,
                                       const SnapConvExtras* ex, void* stream) {
  if (!desc || !x || !w || (!y && !(ex && ex->y_half))) return SNAP_ERR_NULL;
  const int32_t* rows_in = ex ? ex->rows_in : nullptr;
  const int32_t* rows_out = ex ? ex->rows_out : nullptr;
  const int32_t* row_count = ex ? ex->row_count : nullptr;
  float* gn_partial = ex ? ex->gn_partial : nullptr;
  if ((rows_in || rows_out) &&
      (desc->epilogue & (SNAP_EPI_RESIDUAL | SNAP_EPI_UPSAMPLE2X_ADD | SNAP_EPI_ROWMASK)))
    return SNAP_ERR_UNSUPPORTED;  // row-indexed launches carry bias / ReLU only
  const bool presplit = ex && ex->x_presplit;
  const bool split_vec = ex && ex->w_bf16 && !ex->w_split_root && !rows_in && !rows_out && !row_count;
  // gn_partial_rows = 32: a split-K launch of the split engine or of the bf16 / fp16 engine (workspace
  // given), whose reduce pass emits the sums per 32-row slab
  const bool rows32 = gn_partial && ex->gn_partial_rows == 32;
  if (gn_partial && ex->gn_partial_rows != 0 && ex->gn_partial_rows != 32) return SNAP_ERR_BAD_SHAPE;
  if (rows32 && (presplit || !split_vec || ex->w_split_parts == 1 || !ex->workspace || ex->gn_partial2 ||
                 ex->y_half || ex->gnb_mode))
    return SNAP_ERR_UNSUPPORTED;
  const size_t gn_need = !gn_partial ? 0
                         : rows32    ? snap_conv2d_splitk_gn_partial_bytes(desc)
                         : presplit  ? snap_conv2d_presplit_gn_partial_bytes(desc, ex->ps_tile)
                                     : snap_conv2d_gn_partial_bytes_ex(desc, split_vec ? ex->w_split_parts : 0);
  if (gn_partial) {
    if (rows_in || rows_out || row_count) return SNAP_ERR_UNSUPPORTED;
    if (desc->Cout_stride != desc->Cout) return SNAP_ERR_UNSUPPORTED;
    if (ex->gn_partial_bytes < gn_need || gn_need == 0) return SNAP_ERR_WORKSPACE;
  }
  const SnapConvDesc& d = *desc;
  if (d.N <= 0 || d.H <= 0 || d.W <= 0 || d.Cin <= 0 || d.Cout <= 0 || d.KH <= 0 ||
      d.KW <= 0 || d.stride <= 0 || d.Ho <= 0 || d.Wo <= 0)
    return SNAP_ERR_BAD_SHAPE;
  if (d.Cin_stride < d.Cin || d.Cout_stride < d.Cout) return SNAP_ERR_BAD_SHAPE;
  if (d.Cout % 4 != 0 || d.Cout_stride % 4 != 0) return SNAP_ERR_BAD_SHAPE;
  if ((reinterpret_cast<uintptr_t>(y) & 15) || (reinterpret_cast<uintptr_t>(w) & 15) ||
      (reinterpret_cast<uintptr_t>(bias) & 15) || (reinterpret_cast<uintptr_t>(residual) & 15) ||
      (reinterpret_cast<uintptr_t>(up_prev) & 15))
    return SNAP_ERR_BAD_SHAPE;
  if ((int64_t)d.N * d.Ho * d.Wo > 0x7fffffffLL) return SNAP_ERR_BAD_SHAPE;
  const bool gn = d.prologue == SNAP_PRO_GN_RELU || d.prologue == SNAP_PRO_RELU_GN;
  if (d.prologue < 0 || d.prologue > SNAP_PRO_RELU) return SNAP_ERR_UNSUPPORTED;
  if (gn && (!gn_mu || !gn_sc || !gn_beta)) return SNAP_ERR_NULL;
  if ((d.epilogue & SNAP_EPI_BIAS) && !bias) return SNAP_ERR_NULL;
  if ((d.epilogue & SNAP_EPI_RESIDUAL) && !residual) return SNAP_ERR_NULL;
  if ((d.epilogue & SNAP_EPI_ROWMASK) && !row_mask) return SNAP_ERR_NULL;
  if (d.epilogue & SNAP_EPI_UPSAMPLE2X_ADD) {
    if (!up_prev) return SNAP_ERR_NULL;
    if ((d.Ho & 1) || (d.Wo & 1)) return SNAP_ERR_BAD_SHAPE;
  }
  ConvArgs a;
  a.d = d;
  a.x = x; a.w = w; a.y = y;
  a.gn_mu = gn_mu; a.gn_sc = gn_sc; a.gn_beta = gn_beta;
  a.bias = bias; a.residual = residual; a.up_prev = up_prev; a.row_mask = row_mask;
  a.rows_in = rows_in; a.rows_out = rows_out; a.row_count = row_count;
  a.gn_partial = gn_partial;
  a.gn_relu = ex ? ex->gn_partial_relu : 0;
  a.gn_rows32 = rows32 ? 1 : 0;
  a.gn_partial2 = nullptr;
  a.gn_partial2_done = nullptr;
  if (gn_partial && ex->gn_partial2) {
    if (a.gn_relu || ex->gn_partial2_bytes < gn_need) return SNAP_ERR_WORKSPACE;
    a.gn_partial2 = ex->gn_partial2;
    // (an output field of the caller's struct: set by the engine that honours the request)
    a.gn_partial2_done = const_cast<int32_t*>(&ex->gn_partial2_done);
    *a.gn_partial2_done = 0;
  }
  a.kpartial = ex ? static_cast<float*>(ex->workspace) : nullptr;
  a.kpartial_bytes = ex ? ex->workspace_bytes : 0;
  if (reinterpret_cast<uintptr_t>(a.kpartial) & 15) a.kpartial = nullptr;
  a.ksplit = 1; a.slabs_per_split = 0; a.tiles_per_split = 0;
  a.gn_slabs = 0;
  a.M = d.N * d.Ho * d.Wo;
  a.K = d.KH * d.KW * d.Cin;
  // float4 path: channel runs must be 16-byte addressable.
  const bool vec = (d.Cin_stride % 4 == 0) && (d.Cin >= 4) &&
                   ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                   (!gn || (d.Cin % 4 == 0));
  a.ctiles = 0;
  a.nk = 0;  // set per K-slab depth in launch<>
  a.prio = 0;
  a.ablate = snap_alt_ablate_bits();   // alt builds only (timing experiments, wrong results); 0 in the product build
  a.bk = (ex && ex->bk_hint == 32) ? 32 : 16;
  a.no_halo = (ex && (ex->tune_flags & SNAP_TUNE_NO_HALO)) ? 1 : 0;
  a.rs_nsplit = ex ? (ex->tune_flags >> SNAP_TUNE_RS_NSPLIT_SHIFT) & 15 : 0;
  a.no_plain = (ex && (ex->tune_flags & SNAP_TUNE_NO_PLAIN)) ? 1 : 0;
  a.use_raw = (ex && (ex->tune_flags & SNAP_TUNE_RAW_RING)) ? 1 : 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // bf16-operand engine (training precision): needs the packed bf16 weights and the float4
  // loader's alignment; anything else runs on the (more precise) f32 engine below.
  a.w_bf16 = ex ? ex->w_bf16 : nullptr;
  a.half = (ex && ex->w_half) ? 1 : 0;
  if (a.half && (ex->w_split_parts != 0 || presplit || ex->w_split_root)) return SNAP_ERR_UNSUPPORTED;
  a.x_half = nullptr;
  a.y_half = nullptr;
  const bool ps1 = presplit && ex->w_split_parts == 1;   // the one-part pre-split engine: plain bf16 x, bf16 arithmetic
  if (ex && ex->y_half) {       // half (also / only) output: training-precision engine, no split-K, no statistics
    if (!a.w_bf16 || (ex->w_split_parts != 0 && !ps1) || (presplit && !ps1) || ex->w_split_root || gn_partial ||
        (d.epilogue & SNAP_EPI_UPSAMPLE2X_ADD) || d.Cout_stride % 4 != 0 ||
        (reinterpret_cast<uintptr_t>(ex->y_half) & 7))
      return SNAP_ERR_UNSUPPORTED;
    a.y_half = ex->y_half;
    a.kpartial = nullptr;
  }
  a.gnb_x = a.gnb_mu = a.gnb_rstd = a.gnb_gamma = a.gnb_beta = nullptr;
  a.gnb_mode = 0;
  if (ex && ex->gnb_mode) {   // GroupNorm-VJP statistics in the epilogue: half-input data-gradient launches only
    if ((ex->gnb_mode != SNAP_PRO_GN_RELU && ex->gnb_mode != SNAP_PRO_RELU_GN) || !ex->x_half || !gn_partial ||
        ex->gn_partial_rows != 0 || ex->gn_partial2 || ex->gn_partial_relu || ex->workspace ||
        (d.epilogue & ~(SNAP_EPI_RESIDUAL | SNAP_EPI_BIAS)))
      return SNAP_ERR_UNSUPPORTED;
    if (!ex->gnb_x || !ex->gnb_mu || !ex->gnb_rstd || !ex->gnb_gamma || !ex->gnb_beta) return SNAP_ERR_NULL;
    if ((reinterpret_cast<uintptr_t>(ex->gnb_x) | reinterpret_cast<uintptr_t>(ex->gnb_mu) |
         reinterpret_cast<uintptr_t>(ex->gnb_rstd) | reinterpret_cast<uintptr_t>(ex->gnb_gamma) |
         reinterpret_cast<uintptr_t>(ex->gnb_beta)) & 15)
      return SNAP_ERR_BAD_SHAPE;
    a.gnb_x = ex->gnb_x; a.gnb_mu = ex->gnb_mu; a.gnb_rstd = ex->gnb_rstd;
    a.gnb_gamma = ex->gnb_gamma; a.gnb_beta = ex->gnb_beta;
    a.gnb_mode = ex->gnb_mode;
  }
  a.cin8 = (d.Cin + 7) / 8 * 8;
  a.x_ps = nullptr;
  a.ps_tile = 0;
  a.ps_res_init = 0;
  a.ps_parts = 2;
  if (ex && ex->x_half) {     // the input is already bf16 / f16: training-precision engine, both operands by DMA
    if (!a.w_bf16 || ex->w_split_parts != 0 || presplit || ex->w_split_root || d.prologue != SNAP_PRO_NONE ||
        rows_in || d.Cin_stride % 8 != 0 || d.Cin % 8 != 0 || ex->y_half)
      return SNAP_ERR_UNSUPPORTED;       // (rows_out / row_count: the compact buffers of the masked MLP)
    if (ex->w_bf16_bytes < snap_conv2d_packed_weights_bytes(d.KH * d.KW, d.Cin, d.Cout)) return SNAP_ERR_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(a.w_bf16) | reinterpret_cast<uintptr_t>(x)) & 15) return SNAP_ERR_BAD_SHAPE;
    a.x_half = x;
    a.x = nullptr;
    return snapconv::launch_bf16(a, s);
  }
  if (presplit) {                            // both operands pre-split: conv_ps.hip
    if (!a.w_bf16 || (ex->w_split_parts != 2 && ex->w_split_parts != 1) || ex->w_split_root) return SNAP_ERR_UNSUPPORTED;
    if (ps1 && d.Cin_stride != d.Cin) return SNAP_ERR_UNSUPPORTED;
    a.ps_parts = ex->w_split_parts;
    const size_t need = snap_conv2d_packed_weights_split_bytes(d.KH * d.KW, d.Cin, d.Cout, ex->w_split_parts);
    if (need == 0) return SNAP_ERR_UNSUPPORTED;
    if (ex->w_bf16_bytes < need) return SNAP_ERR_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(a.w_bf16) | reinterpret_cast<uintptr_t>(x)) & 15) return SNAP_ERR_BAD_SHAPE;
    a.x_ps = x;
    a.x = nullptr;
    a.ps_tile = ex->ps_tile;
    a.ps_res_init = ex->ps_res_init;
    a.cin8 = d.Cin;
    return snapconv::launch_ps(a, s);
  }
  if (a.w_bf16 && ex->w_split_root) {        // RGB root convolution on the split engine
    const int parts = ex->w_split_parts;
    if (ex->w_bf16_bytes < snap_conv2d_packed_weights_split_root_bytes(d.Cout, parts) ||
        snap_conv2d_packed_weights_split_root_bytes(d.Cout, parts) == 0)
      return SNAP_ERR_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(a.w_bf16) | reinterpret_cast<uintptr_t>(x)) & 15) return SNAP_ERR_BAD_SHAPE;
    a.cin8 = 16;
    return snapconv::launch_split_root(a, parts, s);
  }
  if (a.w_bf16 && vec) {
    const int parts = ex->w_split_parts;
    if (parts < 0 || parts == 1 || parts > 3) return SNAP_ERR_UNSUPPORTED;
    const size_t need = parts ? snap_conv2d_packed_weights_split_bytes(d.KH * d.KW, d.Cin, d.Cout, parts)
                              : snap_conv2d_packed_weights_bytes(d.KH * d.KW, d.Cin, d.Cout);
    if (need == 0) return SNAP_ERR_UNSUPPORTED;
    if (ex->w_bf16_bytes < need) return SNAP_ERR_WORKSPACE;
    if (parts) a.cin8 = (d.Cin + 15) / 16 * 16;   // split image: channel axis padded to the slab
    if (reinterpret_cast<uintptr_t>(a.w_bf16) & 15) return SNAP_ERR_BAD_SHAPE;
    return parts ? snapconv::launch_split(a, parts, s) : snapconv::launch_bf16(a, s);
  }
  return vec ? launch_tile<true>(a, s) : launch_tile<false>(a, s);
}
