// Kernel gradient of the 3 x 3 / stride 1 / pad 1 convolutions on the half-precision engines, all NINE taps in
// one workgroup (reference analogue: jax.grad through the StdConv 3 x 3 of every bottleneck unit,
// snap/models/resnet.py:112-132, under the float16 train config snap/configs/train_localization.py:25).
//
// wgrad_bf16.hip gives every tap its own workgroup: Z and dY cross the L2 -> CU boundary once per tap and once
// per tile of the other operand (18 x the tensors at 64 x 64 tiles; the launches sit at 2.6-3.6 TB/s of that
// traffic whatever their shape, 57-124 TFLOP/s).  Here the reduction slab is a PATCH of 4 x 8 output pixels
// (k = 8 row + col -- any pixel order is a valid reduction order as long as both operands use it):
//   * dY of the patch is staged once and is the B fragment of all nine taps;
//   * the input window of the patch (6 x 10 pixels) is normalised / rounded ONCE per horizontal shift kw and
//     written as three copies Zs[kw][channel][6 rows][8 px]; the fragment of tap (kh, kw) for output row r is
//     the 16 aligned bytes Zs[kw][c][r + kh] -- the vertical shift is an address, the horizontal one a copy;
//   * 64 input channels x 128 (64) output channels x 9 taps per workgroup of eight waves: 144 accumulator
//     registers per wave, 18 MFMAs per wave and slab behind one barrier (8 in the per-tap kernel).
// The partial tiles, the fixed-order reduction over the chunks and the operand rounding are wgrad_bf16.hip's:
// same products, another summation order (patch order instead of row-major pixels).
#include "wgrad_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <bool F16> struct Elem;
template <> struct Elem<false> {
  typedef bf16x8 x8; typedef bf16x4 x4;
  static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Elem<true> {
  typedef f16x8 x8; typedef f16x4 x4;
  static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

constexpr int W3_NT = 512;
constexpr int W3_ZC = 64;            // input channels per workgroup
constexpr int W3_ZROW = 112;         // bytes per (kw, channel): 6 rows x 16 B + 16 B pad (conflict-free b128 reads)
constexpr int W3_DROW = 80;          // bytes per output column: 32 k x 2 B + 16 B pad
constexpr int W3_ZU = 3 * 6 * 2 * 16;   // loader units of the input window: (kw, row, 4-px group, channel quad)

// BN = 128: waves 2 (channels) x 4 (columns), every wave both k-steps.  BN = 64: 2 x 2 x 2 -- the third factor
// splits the two k-steps of a slab between wave sets, which write separate partial slots (KS = 2).
template <int BN, int PRO, bool F16>
__global__ __launch_bounds__(W3_NT) void wgrad3x3_kernel(const WgradArgs a) {
  typedef Elem<F16> E;
  typedef typename E::x8 etx8;
  typedef typename E::x4 etx4;
  constexpr bool need_gn = PRO == SNAP_PRO_GN_RELU;
  constexpr int KS = BN == 128 ? 1 : 2;
  constexpr int WC = BN / 32;                       // wave columns
  constexpr int DQ = BN / 4;                        // dY column quads
  constexpr int DU = 4 * 2 * DQ;                    // dY loader units
  constexpr int Z_ST = 3 * W3_ZC * W3_ZROW, D_ST = BN * W3_DROW;
  __shared__ __attribute__((aligned(16))) char smem[2 * Z_ST + 2 * D_ST];
  char* const Zs0 = smem;
  char* const Ds0 = smem + 2 * Z_ST;

  const SnapConvDesc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int ks = KS == 2 ? (wid >> 2) : 0;
  const int w4 = KS == 2 ? (wid & 3) : wid;
  const int wr = w4 / WC, wc = w4 - wr * WC;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int ct = blockIdx.x / a.ncol, col_t = blockIdx.x - ct * a.ncol;
  const int c0 = ct * W3_ZC, n0 = col_t * BN;
  const int PY = (d.Ho + 3) >> 2, PX = (d.Wo + 7) >> 3;
  const int NP = d.N * PY * PX;
  const int p_begin = blockIdx.y * a.slabs_per_chunk;
  const int p_end = min(NP, p_begin + a.slabs_per_chunk);
  const int nslab = max(0, p_end - p_begin);

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // loader units of this thread: two of the input window, (at most) one of dY
  int zkw[2], zr[2], zg[2], zq[2];
  bool zon[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int u = tid + j * W3_NT;
    zon[j] = u < W3_ZU;
    const int uu = zon[j] ? u : 0;
    zkw[j] = uu / 192;
    const int rem = uu - zkw[j] * 192;
    zr[j] = rem >> 5;
    zg[j] = (rem >> 4) & 1;
    zq[j] = rem & 15;
    zon[j] = zon[j] && (c0 + 4 * zq[j] < d.Cin);
  }
  const bool don_u = tid < DU;
  const int dq = tid % DQ, dg = (tid / DQ) & 1, drw = (tid / (2 * DQ)) & 3;
  const bool don = don_u && (n0 + 4 * dq < d.Cout);

  f32x4 zv[2][4], zmu[2], zsc[2], zbeta[2];
  bool zin[2][4];
  u32x2 dv[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    zbeta[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    zmu[j] = zsc[j] = zbeta[j];
    if constexpr (need_gn)
      if (zon[j]) zbeta[j] = *reinterpret_cast<const f32x4*>(a.gn_beta + c0 + 4 * zq[j]);
  }

  auto load_slab = [&](int P) {
    const int n = P / (PY * PX);
    const int rem = P - n * (PY * PX);
    const int py = rem / PX, px = rem - py * PX;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int hi = py * 4 + zr[j] - 1;
      const int wi0 = px * 8 + 4 * zg[j] + zkw[j] - 1;
      const bool rok = zon[j] && hi >= 0 && hi < d.H;
      const int64_t rbase = ((int64_t)n * d.H + (rok ? hi : 0)) * d.W;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int wi = wi0 + e;
        const bool inb = rok && wi >= 0 && wi < d.W;
        zin[j][e] = inb;
        const int64_t off = inb ? (rbase + wi) * d.Cin_stride + c0 + 4 * zq[j] : (int64_t)0;
        zv[j][e] = *reinterpret_cast<const f32x4*>(a.x + off);
      }
      if constexpr (need_gn) {
        const int64_t so = zon[j] ? (int64_t)n * d.Cin + c0 + 4 * zq[j] : (int64_t)0;
        zmu[j] = *reinterpret_cast<const f32x4*>(a.gn_mu + so);
        zsc[j] = *reinterpret_cast<const f32x4*>(a.gn_sc + so);
      }
    }
    {
      const int ho = py * 4 + drw;
      const int wo0 = px * 8 + 4 * dg;
      const bool rok = don && ho < d.Ho;
      const int64_t rbase = ((int64_t)n * d.Ho + (rok ? ho : 0)) * d.Wo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = rok && wo0 + e < d.Wo;
        dv[e] = ok ? *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(a.dy) +
                                                      (rbase + wo0 + e) * d.Cout_stride + n0 + 4 * dq)
                   : u32x2{0u, 0u};
      }
    }
  };

  auto store_slab = [&](int buf) {
    char* zs = Zs0 + buf * Z_ST;
    char* ds = Ds0 + buf * D_ST;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (!zon[j]) continue;
      f32x4 pv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float v;
          if constexpr (need_gn) v = wg_pro<PRO>(zv[j][e][c], zmu[j][c], zsc[j][c], zbeta[j][c], d.in_scale, d.in_shift);
          else v = wg_pro<PRO>(zv[j][e][c], 0.f, 0.f, 0.f, d.in_scale, d.in_shift);
          pv[e][c] = zin[j][e] ? v : 0.f;              // (the reference pads the normalised tensor)
        }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 t = {pv[0][c], pv[1][c], pv[2][c], pv[3][c]};   // 4 consecutive pixels of channel c
        *reinterpret_cast<etx4*>(zs + (zkw[j] * W3_ZC + 4 * zq[j] + c) * W3_ZROW + zr[j] * 16 + zg[j] * 8) =
            __builtin_convertvector(t, etx4);
      }
    }
    if (don_u) {
      // element e of the four pixels' packed quads -> one 8-byte (4 consecutive k) store per column
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const unsigned sel = (c & 1) ? 0x07060302u : 0x05040100u;
        const int w = c >> 1;
        u32x2 o;
        o[0] = __builtin_amdgcn_perm(dv[1][w], dv[0][w], sel);
        o[1] = __builtin_amdgcn_perm(dv[3][w], dv[2][w], sel);
        *reinterpret_cast<u32x2*>(ds + (4 * dq + c) * W3_DROW + (drw * 8 + 4 * dg) * 2) = o;
      }
    }
  };

  if (nslab > 0) {
    load_slab(p_begin);
    store_slab(0);
  }
  __syncthreads();
  for (int sl = 0; sl < nslab; ++sl) {
    const int cur = sl & 1;
    const bool more = sl + 1 < nslab;
    if (more) load_slab(p_begin + sl + 1);
    const char* zs = Zs0 + cur * Z_ST + (wr * 32 + l31) * W3_ZROW;
    const char* ds = Ds0 + cur * D_ST + (wc * 32 + l31) * W3_DROW;
#pragma unroll
    for (int s0 = 0; s0 < 2 / KS; ++s0) {
      const int s = KS == 2 ? ks : s0;
      const int prow = 2 * s + lhi;                    // output patch row of this lane's 8 k values
      const etx8 bv = *reinterpret_cast<const etx8*>(ds + prow * 16);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const etx8 av = *reinterpret_cast<const etx8*>(zs + kw * (W3_ZC * W3_ZROW) + (prow + kh) * 16);
          acc[kh * 3 + kw] = E::mfma(av, bv, acc[kh * 3 + kw]);
        }
    }
    if (more) store_slab(cur ^ 1);
    __syncthreads();
  }

  // partial tiles -> workspace [slot][K][Cout], slot = chunk * KS + ks
  float* out = a.partial + ((int64_t)blockIdx.y * KS + ks) * a.K * d.Cout;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ri = (r & 3) + 8 * (r >> 2) + 4 * lhi;
      const int c = c0 + wr * 32 + ri;
      const int col = n0 + wc * 32 + l31;
      if (c < d.Cin && col < d.Cout) out[((int64_t)t * d.Cin + c) * d.Cout + col] = acc[t][r];
    }
}

template <int BN, int PRO>
int w3_launch(const WgradArgs& a, const WgPlan& p, bool half, hipStream_t s) {
  constexpr int KS = BN == 128 ? 1 : 2;
  const dim3 grid((unsigned)(p.ktiles * p.ncol), (unsigned)(p.S / KS));
  if (half) hipLaunchKernelGGL((wgrad3x3_kernel<BN, PRO, true>), grid, dim3(W3_NT), 0, s, a);
  else hipLaunchKernelGGL((wgrad3x3_kernel<BN, PRO, false>), grid, dim3(W3_NT), 0, s, a);
  SNAP_CHECK_LAUNCH();
  return SNAP_OK;
}

}  // namespace

int snapwg::launch_3x3(const WgradArgs& a, const WgPlan& p, bool half, hipStream_t s) {
  if (!a.dy_is_half || a.x_is_half) return SNAP_ERR_UNSUPPORTED;
  if (a.d.prologue == SNAP_PRO_GN_RELU)
    return p.bn == 128 ? w3_launch<128, SNAP_PRO_GN_RELU>(a, p, half, s) : w3_launch<64, SNAP_PRO_GN_RELU>(a, p, half, s);
  if (a.d.prologue == SNAP_PRO_NONE)
    return p.bn == 128 ? w3_launch<128, SNAP_PRO_NONE>(a, p, half, s) : w3_launch<64, SNAP_PRO_NONE>(a, p, half, s);
  return SNAP_ERR_UNSUPPORTED;
}
