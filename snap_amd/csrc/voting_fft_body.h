// Frequency-domain exhaustive (x, y, theta) voting: kernel bodies and launch sequence.
//
// Replaces the direct-form correlation of snap/models/pose_exhaustive_voting.py:72-104
// (jax.scipy.signal.convolve over every template, summed over channels, the overlap count and the
// -inf / normalisation pass) by its frequency-domain equivalent, which needs ~1e-3 of the direct
// form's multiply-adds:
//
//   scores[r,a,b] = sum_{i,j,d} q[r,i,j,d] m_pad[a+i, b+j, d]
//                 = Re IFFT2( sum_p conj(FFT2(z_q[r,p])) * FFT2(z_m[p]) )[a, b]
//
// with the channel PAIRS (2p, 2p+1) packed as the real / imaginary part of one complex signal
// (z = c_2p + i c_2p+1: the channels-last f32 tensors ARE complex arrays [.., D/2] in memory; the
// cross terms of a pair land in the imaginary part of the result, which is dropped).  The overlap
// count packs the ROTATIONS r and r + R/2 the same way: Re -> count of r, -Im -> count of r + R/2,
// rounded to the nearest integer (operands are 0/1: the count is exact).
//
// Transforms: in-place mixed-radix (3, 2, 4...) FFTs of kCols (= 8) columns at a time in LDS, layout
// [n][kCols] complex64 -- the columns are the channel pairs of one 2 kCols-channel group (one 64-byte
// run of a channels-last pixel: every global access is a whole run).  Forward = decimation in
// frequency (natural order in, digit-reversed out); the spectra stay in digit-reversed order (the
// map's, produced by the same routine, are permuted identically); inverse = decimation in time, the
// exact conjugate transpose of the forward stages in reverse order (digit-reversed in, natural
// out).  No reordering pass exists.  tools/fft_voting_model.py is the index-exact numpy model.
//
// This header is compiled twice: by hipcc into libsnap_hip.so (voting_fft.hip), and by g++ with
// VF_EMU into the CPU emulation harness of the tests (tests/emu/voting_fft_emu.cpp: one pthread per
// GPU thread, a pthread barrier per __syncthreads) -- test infrastructure that lets the index
// arithmetic be checked against oracle/voting.py without a GPU.  The product never loads the
// emulation.
#ifndef SNAP_CSRC_VOTING_FFT_BODY_H_
#define SNAP_CSRC_VOTING_FFT_BODY_H_

#include <stdint.h>
#include <stddef.h>
#include <math.h>

#include "rotate_sample.h"

#ifdef VF_EMU
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
#define VF_DEV static inline
void vf_emu_barrier();
float vf_emu_shfl_xor(float v, int mask, int tid);
#define VF_SYNC() vf_emu_barrier()
#define VF_SHFL_XOR(v, m, tid) vf_emu_shfl_xor(v, m, tid)
#define VF_KEEP(v) ((void)(v))
#else
#define VF_DEV __device__ __forceinline__
// LDS-only workgroup barrier: every hand-off between threads of these kernels goes through LDS, so the
// barrier waits for this wave's LDS traffic (lgkmcnt) and NOT for its outstanding global loads / stores
// (__syncthreads() drains vmcnt too: a load issued before a transform to be consumed after it would be
// waited for at the transform's first barrier)
#define VF_SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define VF_SHFL_XOR(v, m, tid) __shfl_xor(v, m, 16)
#define VF_KEEP(v) asm volatile("" ::"v"(v))      // a use the optimiser cannot drop or move
#endif

namespace vfft {

// The query plane a rotated call samples its templates from (null feat: explicit templates in `q`).
struct RotSource { const float* feat; const uint8_t* valid; const float* tfm; float cell; };

constexpr int kMaxStages = 8;
#ifndef VF_COLS
#define VF_COLS 8
#endif
// Columns transformed together = channel pairs of a group (a group = 2 kCols channels: one 64-byte run
// of a channels-last pixel at kCols = 8).  8 columns: 49 KB of LDS per transform at N = 768, so TWO
// workgroups share a CU and the staging / transform / product phases of one run under the other's
// (16 columns = whole 128-byte lines but one workgroup per CU: measured slower).
constexpr int kCols = VF_COLS;
constexpr int kColShift = kCols == 16 ? 4 : (kCols == 8 ? 3 : 2);
static_assert((1 << kColShift) == kCols, "kCols: 4, 8 or 16");
constexpr int kGroupCh = 2 * kCols;     // channels per group
constexpr int kMaxN = 1024;        // (kCols + 2) * N * 8 B of LDS
constexpr int kZPre = 6;           // 16-byte map-spectrum loads a thread keeps in flight across the transform

struct Plan {                      // one axis: N = prod radix[s]; stage s works on spans of L[s]
  int N, nst;
  int radix[kMaxStages];
  int L[kMaxStages];               // span entering stage s (forward order): L[0] = N
  int logM[kMaxStages];            // M = L / radix is a power of two (the 3, if any, goes first)
  int tstep[kMaxStages];           // N / L: twiddle-table step of the stage
};

// smallest N >= n of the form 2^a or 3 * 2^a (a >= 2)
static inline int fft_size(int n) {
  int best = 0;
  for (int base = 1; base <= 3; base += 2) {
    int N = base * 4;
    while (N < n) N *= 2;
    if (!best || N < best) best = N;
  }
  return best;
}

static inline bool make_plan(int N, Plan* pl) {
  pl->N = N; pl->nst = 0;
  int n = N, rad[kMaxStages + 8], ns = 0;
  if (n % 3 == 0) { rad[ns++] = 3; n /= 3; }
  int k = 0;
  while (n > 1) { if (n & 1) return false; n >>= 1; ++k; }
  if (k & 1) rad[ns++] = 2;
  for (int i = 0; i < k / 2; ++i) rad[ns++] = 4;
  if (ns > kMaxStages || ns == 0) return false;
  int L = N;
  for (int s = 0; s < ns; ++s) {
    const int M = L / rad[s];
    int lg = 0;
    while ((1 << lg) < M) ++lg;
    if ((1 << lg) != M) return false;
    pl->radix[s] = rad[s]; pl->L[s] = L; pl->logM[s] = lg; pl->tstep[s] = N / L;
    L = M;
  }
  pl->nst = ns;
  return true;
}

// LDS rows are stored swizzled: logical row n of a C-column buffer lives at row vf_phys<C>(n).  The last fused pass
// of a transform (spans 16 and 4) has every thread walk the 16 consecutive rows of its block -- 1024 bytes at
// C = 8 -- so the lanes of a wave, 16 rows apart, met on the same banks (8 lanes per 64-byte row, 8 rows per
// wave on the same 16 banks; the single-column buffer: 128 bytes apart, 2 banks for 32 lanes).  XOR-ing the
// row's low bits with bits 4.. of its index (the block number of that pass) spreads neighbouring blocks over
// the banks; the earlier passes see a lane-uniform XOR of their low bits, a permutation inside a run they
// already read whole.  A bijection of [0, N) (N = 2^a or 3 * 2^a, a >= 2).  VF_NO_SWIZZLE: A/B builds.
template <int C> VF_DEV int vf_phys(int n) {
#ifdef VF_NO_SWIZZLE
  return n;
#else
  return C == 1 ? (n ^ ((n >> 4) & 15)) : (n ^ ((n >> 4) & 3));
#endif
}
// 16-byte item i of the kCols-column buffer (kCols / 2 items per row) -> its swizzled item index
VF_DEV int vf_phys4(int i) { return vf_phys<kCols>(i / (kCols / 2)) * (kCols / 2) + i % (kCols / 2); }

#if defined(VF_EMU) || defined(VF_SCALAR_MATH)     // (VF_SCALAR_MATH: A/B builds of the HIP library)
VF_DEV float2 cadd(float2 a, float2 b) { float2 r; r.x = a.x + b.x; r.y = a.y + b.y; return r; }
VF_DEV float2 csub(float2 a, float2 b) { float2 r; r.x = a.x - b.x; r.y = a.y - b.y; return r; }
VF_DEV float2 cmul(float2 a, float2 b) {            // a * b
  float2 r; r.x = a.x * b.x - a.y * b.y; r.y = a.x * b.y + a.y * b.x; return r;
}
VF_DEV float2 cmulc(float2 a, float2 b) {           // a * conj(b)
  float2 r; r.x = a.x * b.x + a.y * b.y; r.y = a.y * b.x - a.x * b.y; return r;
}
// a + (-i) b (forward) / a + i b (inverse), and a - (-i) b / a - i b
template <bool INV> VF_DEV float2 cadd_rot(float2 a, float2 b) {
  float2 r;
  if (INV) { r.x = a.x - b.y; r.y = a.y + b.x; } else { r.x = a.x + b.y; r.y = a.y - b.x; }
  return r;
}
template <bool INV> VF_DEV float2 csub_rot(float2 a, float2 b) {
  float2 r;
  if (INV) { r.x = a.x + b.y; r.y = a.y - b.x; } else { r.x = a.x - b.y; r.y = a.y + b.x; }
  return r;
}
VF_DEV float2 cscale(float2 a, float s) { float2 r; r.x = s * a.x; r.y = s * a.y; return r; }
#else
// The same IEEE operations, two lanes per instruction (v_pk_add_f32 / v_pk_mul_f32 with their op_sel
// swizzles and per-lane neg modifiers; -ffp-contract=off and no FMA: every product and sum is rounded
// exactly as in the scalar form above, so the transforms are bit-identical to it).  The per-lane
// negations and the (re, im) swap of a multiplication by +-i ride on the instruction's source
// modifiers, which the compiler does not select by itself: a radix-4 butterfly with its three twiddle
// products is 17 instructions instead of 34.
typedef float vf_v2 __attribute__((ext_vector_type(2)));
VF_DEV vf_v2 vf_pk(float2 a) { vf_v2 r = {a.x, a.y}; return r; }
VF_DEV float2 vf_un(vf_v2 a) { float2 r; r.x = a.x; r.y = a.y; return r; }
VF_DEV float2 cadd(float2 a, float2 b) { return vf_un(vf_pk(a) + vf_pk(b)); }
VF_DEV float2 csub(float2 a, float2 b) { return vf_un(vf_pk(a) - vf_pk(b)); }
VF_DEV float2 cscale(float2 a, float s) { vf_v2 sv = {s, s}; return vf_un(sv * vf_pk(a)); }
VF_DEV float2 cmul(float2 a, float2 b) {            // a * b = (ax bx - ay by, ax by + ay bx)
  const vf_v2 av = vf_pk(a), bv = vf_pk(b);
  const vf_v2 t1 = av.xx * bv;                      // (ax bx, ax by)
  const vf_v2 t2 = av.yy * bv.yx;                   // (ay by, ay bx)
  vf_v2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(t1), "v"(t2));
  return vf_un(r);
}
VF_DEV float2 cmulc(float2 a, float2 b) {           // a * conj(b) = (ax bx + ay by, ay bx - ax by)
  const vf_v2 av = vf_pk(a), bv = vf_pk(b);
  const vf_v2 t1 = av.xx * bv;                      // (ax bx, ax by)
  const vf_v2 t2 = av.yy * bv.yx;                   // (ay by, ay bx)
  vf_v2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_hi:[1,0]" : "=v"(r) : "v"(t1), "v"(t2));
  return vf_un(r);
}
// a + (-i) b = (ax + by, ay - bx) (forward) / a + i b = (ax - by, ay + bx) (inverse)
template <bool INV> VF_DEV float2 cadd_rot(float2 a, float2 b) {
  const vf_v2 av = vf_pk(a), bv = vf_pk(b);
  vf_v2 r;
  if (INV) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(av), "v"(bv));
  else     asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(av), "v"(bv));
  return vf_un(r);
}
// a - (-i) b = (ax - by, ay + bx) (forward) / a - i b = (ax + by, ay - bx) (inverse)
template <bool INV> VF_DEV float2 csub_rot(float2 a, float2 b) { return cadd_rot<!INV>(a, b); }
#endif
// two complex products a * conj(b) packed in 16 bytes
VF_DEV float4 cmulc2(float4 a, float4 b) {
  float2 a0, a1, b0, b1;
  a0.x = a.x; a0.y = a.y; a1.x = a.z; a1.y = a.w;
  b0.x = b.x; b0.y = b.y; b1.x = b.z; b1.y = b.w;
  const float2 r0 = cmulc(a0, b0), r1 = cmulc(a1, b1);
  float4 r; r.x = r0.x; r.y = r0.y; r.z = r1.x; r.w = r1.y;
  return r;
}

// One radix-4 stage butterfly on four registers.  Forward (DIF): butterfly, then the stage twiddles
// w_m = exp(-2 pi i k m / L) on outputs 1..3; inverse (DIT, the conjugate transpose): conjugate
// twiddles on inputs 1..3, then the conjugate butterfly.  tw = false (k == 0): the twiddles are 1.
template <bool INV>
VF_DEV void bfly4(float2& x0, float2& x1, float2& x2, float2& x3, bool tw, float2 w1, float2 w2, float2 w3) {
  if (INV && tw) { x1 = cmulc(x1, w1); x2 = cmulc(x2, w2); x3 = cmulc(x3, w3); }
  const float2 t0 = cadd(x0, x2), u1 = csub(x0, x2), u2 = cadd(x1, x3);
  const float2 d = csub(x1, x3);                    // x1 = u1 + (-+i) d, x3 = u1 - (-+i) d
  x0 = cadd(t0, u2); x1 = cadd_rot<INV>(u1, d); x2 = csub(t0, u2); x3 = csub_rot<INV>(u1, d);
  if (!INV && tw) { x1 = cmul(x1, w1); x2 = cmul(x2, w2); x3 = cmul(x3, w3); }
}

// One radix-3 stage butterfly (forward: butterfly, then the twiddles on outputs 1, 2; inverse: conjugate
// twiddles on inputs 1, 2 first).
template <bool INV>
VF_DEV void bfly3(float2& x0, float2& x1, float2& x2, bool tw, float2 w1, float2 w2) {
  if (INV && tw) { x1 = cmulc(x1, w1); x2 = cmulc(x2, w2); }
  const float2 sm = cadd(x1, x2), df = csub(x1, x2);
  const float2 h = csub(x0, cscale(sm, 0.5f));
  const float2 e = cscale(df, 0.86602540378443864676f);
  // y1 = h + (-+ i) e, y2 = h - (-+ i) e   (-+ i (sqrt 3 / 2) (x1 - x2))
  const float2 y0 = cadd(x0, sm), y1 = cadd_rot<INV>(h, e), y2 = csub_rot<INV>(h, e);
  x0 = y0; x1 = y1; x2 = y2;
  if (!INV && tw) { x1 = cmul(x1, w1); x2 = cmul(x2, w2); }
}

// In-place transform of C interleaved columns, buf[n * C + p].  tw[t] = exp(-2 pi i t / N).
// Forward: DIF, stages 0..nst-1; inverse: DIT, stages nst-1..0, each the conjugate transpose of the
// forward stage (conjugate twiddles BEFORE the conjugate butterfly) => N * ifft, natural order out.
// Two consecutive radix-4 stages run as ONE pass over LDS: the 16 elements {b0 + k + q' M/4 + m M}
// that the four butterflies of the outer stage and the four of the inner stage share stay in a
// thread's registers (the same butterflies with the same operands: bit-identical to two passes, with
// one barrier, one round of index arithmetic and 15 instead of 24 twiddle loads per 16 elements).
// N = 768: three passes (3 | 4,4 | 4,4) instead of five.
// The data must be visible (barrier) on entry; it is on exit.
// skip_first (forward only): stage 0 has been applied by the caller (stage0_zero_tail below).
template <int C, bool INV>
VF_DEV void fft_lds(float2* buf, const float2* tw, const Plan& pl, int tid, int nt, bool skip_first = false) {
  int ss = (!INV && skip_first) ? 1 : 0;
  while (ss < pl.nst) {
    const int s = INV ? pl.nst - 1 - ss : ss;                 // the stage this pass starts with
    const int s2 = INV ? s - 1 : s + 1;                       // ... and its partner, if both are radix 4
    const bool pair = pl.radix[s] == 4 && s2 >= 0 && s2 < pl.nst && pl.radix[s2] == 4;
    if (pair) {
      const int so = INV ? s2 : s, si = INV ? s : s2;         // outer (larger span) / inner stage
      const int logM = pl.logM[so], logMi = pl.logM[si];      // M = L / 4, M' = M / 4
      const int M = 1 << logM, Mi = 1 << logMi;
      const int tso = pl.tstep[so], tsi = pl.tstep[si];
      const int total = (pl.N >> 4) * C;
      for (int item = tid; item < total; item += nt) {
        const int p = item % C, ti = item / C;
        const int kk = ti & (Mi - 1), blk = ti >> logMi;
        const int n0 = (blk << (logM + 2)) + kk;
        float2* b = buf + p;
        float2 v[4][4];                                        // [q' : inner position][m : outer position]
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int m = 0; m < 4; ++m) v[q][m] = b[vf_phys<C>(n0 + q * Mi + m * M) * C];
        float2 w1, w2, w3;
        w1.x = w2.x = w3.x = 1.f; w1.y = w2.y = w3.y = 0.f;
        if (!INV) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {                        // outer stage: k = kk + q M'
            const int t1 = (kk + q * Mi) * tso;
            const bool has = (kk + q * Mi) != 0;
            if (has) { w1 = tw[t1]; w2 = tw[2 * t1]; w3 = tw[3 * t1]; }
            bfly4<false>(v[q][0], v[q][1], v[q][2], v[q][3], has, w1, w2, w3);
          }
          const int t1 = kk * tsi;
          if (kk) { w1 = tw[t1]; w2 = tw[2 * t1]; w3 = tw[3 * t1]; }
#pragma unroll
          for (int m = 0; m < 4; ++m) bfly4<false>(v[0][m], v[1][m], v[2][m], v[3][m], kk != 0, w1, w2, w3);
        } else {
          const int t1 = kk * tsi;
          if (kk) { w1 = tw[t1]; w2 = tw[2 * t1]; w3 = tw[3 * t1]; }
#pragma unroll
          for (int m = 0; m < 4; ++m) bfly4<true>(v[0][m], v[1][m], v[2][m], v[3][m], kk != 0, w1, w2, w3);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int t1o = (kk + q * Mi) * tso;
            const bool has = (kk + q * Mi) != 0;
            if (has) { w1 = tw[t1o]; w2 = tw[2 * t1o]; w3 = tw[3 * t1o]; }
            bfly4<true>(v[q][0], v[q][1], v[q][2], v[q][3], has, w1, w2, w3);
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int m = 0; m < 4; ++m) b[vf_phys<C>(n0 + q * Mi + m * M) * C] = v[q][m];
      }
      VF_SYNC();
      ss += 2;
      continue;
    }
    const int R = pl.radix[s], logM = pl.logM[s], M = 1 << logM, L = pl.L[s], ts = pl.tstep[s];
    const int total = (pl.N / R) * C;
    for (int item = tid; item < total; item += nt) {
      const int p = item % C, bi = item / C;
      const int k = bi & (M - 1), blk = bi >> logM;
      const int n0 = blk * L + k;
      float2* b = buf + p;
      const int a0 = vf_phys<C>(n0) * C, a1 = vf_phys<C>(n0 + M) * C;
      const int t1 = k * ts;
      if (R == 4) {
        const int a2 = vf_phys<C>(n0 + 2 * M) * C, a3 = vf_phys<C>(n0 + 3 * M) * C;
        float2 x0 = b[a0], x1 = b[a1], x2 = b[a2], x3 = b[a3];
        float2 w1, w2, w3;
        w1.x = w2.x = w3.x = 1.f; w1.y = w2.y = w3.y = 0.f;
        if (k) { w1 = tw[t1]; w2 = tw[2 * t1]; w3 = tw[3 * t1]; }
        bfly4<INV>(x0, x1, x2, x3, k != 0, w1, w2, w3);
        b[a0] = x0; b[a1] = x1; b[a2] = x2; b[a3] = x3;
      } else if (R == 3) {
        const int a2 = vf_phys<C>(n0 + 2 * M) * C;
        float2 x0 = b[a0], x1 = b[a1], x2 = b[a2];
        float2 w1, w2;
        w1.x = w2.x = 1.f; w1.y = w2.y = 0.f;
        if (k) { w1 = tw[t1]; w2 = tw[2 * t1]; }
        bfly3<INV>(x0, x1, x2, k != 0, w1, w2);
        b[a0] = x0; b[a1] = x1; b[a2] = x2;
      } else {
        float2 x0 = b[a0], x1 = b[a1];
        float2 w1; w1.x = 1.f; w1.y = 0.f;
        if (k) w1 = tw[t1];
        if (INV && k) x1 = cmulc(x1, w1);
        float2 y0 = cadd(x0, x1), y1 = csub(x0, x1);
        if (!INV && k) y1 = cmul(y1, w1);
        b[a0] = y0; b[a1] = y1;
      }
    }
    VF_SYNC();
    ss += 1;
  }
}

// Timing ablations exist only in alt builds (-DVF_ABLATE=<bits>, WRONG results).  Dot launch: 1 = no X1 row
// load, 2 = no forward transform, 4 = no map-spectrum loads, 8 = no inverse, 16 = no column sums; slow-axis launch
// of a rotated call: 32 = no tap loads, 64 = no transform, 128 = no store of the half-transformed rows
#ifndef VF_ABLATE
#define VF_ABLATE 0
#endif
// The first forward stage where its partners are zero padding.  N = 3 * 2^a and n_in <= N / 3 (templates
// of the map's own size: 256 rows in 768 transform points): the radix-3 butterfly {x[k], x[k + N/3],
// x[k + 2N/3]} has x1 = x2 = 0, so the thread that STAGES x[k] runs the butterfly on the value it holds
// and writes the three outputs -- the zero rows are never staged and the stage's own pass over LDS (a
// read and a write of the whole buffer, one barrier) does not exist.  The same operations on the same
// operands as that pass, zeros included: bit-identical to it.  `tw` must be visible (barrier) on entry.
VF_DEV bool stage0_fusable(const Plan& pl, int n_in) {
  return pl.nst > 1 && pl.radix[0] == 3 && n_in <= pl.N / 3;
}
VF_DEV void stage0_zero_tail(float2 x0, int k, const float2* tw, const Plan& pl, float2* y) {
  float2 x1, x2, w1, w2;
  x1.x = x1.y = x2.x = x2.y = 0.f;
  w1.x = w2.x = 1.f; w1.y = w2.y = 0.f;
  const int t1 = k * pl.tstep[0];
  if (k) { w1 = tw[t1]; w2 = tw[2 * t1]; }
  bfly3<false>(x0, x1, x2, k != 0, w1, w2);
  y[0] = x0; y[1] = x1; y[2] = x2;
}

// ------------------------------------------------------------------------------------------------
// Kernel A: forward transform along the SLOW spatial axis (rows) of one column of 16-pair vectors.
//   grid (ncols, nbatch); dst[batch][k1'][col][16]
// ------------------------------------------------------------------------------------------------
enum { kSlowTemplate = 0, kSlowMap = 1, kSlowCount = 2, kSlowMvalid = 3, kSlowRotate = 4 };

struct SlowArgs {
  Plan pl;
  const float2* tw;
  int mode;
  const float* srcf;        // TEMPLATE: q [R, sH, sW, D]; MAP: m [sH, sW, D]
  const uint8_t* srcb;      // COUNT: tvalid [R, sH, sW]; MVALID: mvalid [sH, sW]
  int n_in;                 // logical input rows (H, or 3 sH - 2 for the padded map)
  int ncols;                // logical columns = gridDim.x
  int sH, sW, D, G;         // source dims; G = 32-channel groups (batch = r * G + g / g)
  int R, R2;                // COUNT: rotations, ceil(R / 2)
  // ROTATE: the templates are sampled on the fly from the query plane (srcf = feat [sH, sW, D], srcb =
  // its validity): rotation r = k * R/4 + r0 is rot90^k of the bilinear sample under tfm[r0]
  // (pose_exhaustive_voting.py:37-69) -- the [R, H, W, D] template tensor is never materialised
  const float* tfm;         // [R / 4, 4] (cos, sin, tx, ty)
  float cell;
  float2* dst;
};

VF_DEV void slow_body(const SlowArgs& a, int bx, int by, int tid, int nt, float2* buf, float2* twl) {
  const int N = a.pl.N;
  for (int t = tid; t < N; t += nt) twl[t] = a.tw[t];
  const int p = tid & (kCols - 1), slot = tid >> kColShift, nslot = nt >> kColShift;
  const int col = bx, batch = by;
  const bool fuse0 = stage0_fusable(a.pl, a.n_in);
  const int M0 = N / 3;
  if (fuse0) VF_SYNC();                                   // (the staging reads the twiddles)
  int row_first = slot;
  if (a.mode == kSlowRotate && fuse0 && !(a.D & 1)) {
    // Templates sampled on the fly: kRotB rows of the thread at a time, the four taps' features AND validity
    // bytes of all of them requested before the first is blended (the taps are clamped into the plane, so
    // nothing waits for the validity verdict: one round trip per kRotB rows instead of two per row; 2 rows:
    // slow-axis launch 0.56 -> 0.46 ms, 4 rows: another 1-2 % of the voting).  The
    // arithmetic per sample is snap_rot_sample / snap_rot_mix: the same bits.
#ifndef VF_ROTB
#define VF_ROTB 4
#endif
    constexpr int kRotB = VF_ROTB;
    const int r = batch / a.G, g = batch - r * a.G, RQ = a.R >> 2;
    const int k = r / RQ, r0 = r - k * RQ;
    const int c = kGroupCh * g + 2 * p;
    const bool cok = c < a.D && !(VF_ABLATE & 32);
    const float2 zero2 = {0.f, 0.f};
    for (; row_first + (kRotB - 1) * nslot < M0; row_first += kRotB * nslot) {
      SnapRotSample rs[kRotB];
      float2 t[kRotB][4];
      uint8_t ok4[kRotB][4];
#pragma unroll
      for (int j = 0; j < kRotB; ++j) {
        const int row = row_first + j * nslot;
        int si = 0, sj = 0;
        if (row < a.n_in) snap_rot90_source(k, row, col, a.sH, a.sW, &si, &sj);
        rs[j] = snap_rot_geom(a.tfm + r0 * 4, si, sj, a.sH, a.sW, a.cell);
        if (row >= a.n_in) rs[j].ok = false;
        const int o00 = rs[j].i0 * a.sW + rs[j].j0, o01 = rs[j].i0 * a.sW + rs[j].j1;
        const int o10 = rs[j].i1 * a.sW + rs[j].j0, o11 = rs[j].i1 * a.sW + rs[j].j1;
        ok4[j][0] = a.srcb[o00]; ok4[j][1] = a.srcb[o01]; ok4[j][2] = a.srcb[o10]; ok4[j][3] = a.srcb[o11];
        t[j][0] = t[j][1] = t[j][2] = t[j][3] = zero2;
        if (cok) {
          t[j][0] = *reinterpret_cast<const float2*>(a.srcf + (int64_t)o00 * a.D + c);
          t[j][1] = *reinterpret_cast<const float2*>(a.srcf + (int64_t)o01 * a.D + c);
          t[j][2] = *reinterpret_cast<const float2*>(a.srcf + (int64_t)o10 * a.D + c);
          t[j][3] = *reinterpret_cast<const float2*>(a.srcf + (int64_t)o11 * a.D + c);
        }
      }
#pragma unroll
      for (int j = 0; j < kRotB; ++j) {
        const int row = row_first + j * nslot;
        const bool ok = rs[j].ok && ok4[j][0] && ok4[j][1] && ok4[j][2] && ok4[j][3] && cok;
        float2 v = zero2;
        if (ok) {
          v.x = snap_rot_mix(rs[j], t[j][0].x, t[j][1].x, t[j][2].x, t[j][3].x);
          v.y = snap_rot_mix(rs[j], t[j][0].y, t[j][1].y, t[j][2].y, t[j][3].y);
        }
        float2 y[3];
        stage0_zero_tail(v, row, twl, a.pl, y);
        buf[vf_phys<kCols>(row) * kCols + p] = y[0];
        buf[vf_phys<kCols>(row + M0) * kCols + p] = y[1];
        buf[vf_phys<kCols>(row + 2 * M0) * kCols + p] = y[2];
      }
    }
  }
  for (int row = row_first; row < (fuse0 ? M0 : N); row += nslot) {
    float2 v; v.x = 0.f; v.y = 0.f;
    if (row < a.n_in) {
      if (a.mode == kSlowTemplate || a.mode == kSlowMap) {
        int g, si, sj; int64_t img = 0;
        if (a.mode == kSlowTemplate) {
          const int r = batch / a.G; g = batch - r * a.G; si = row; sj = col;
          img = (int64_t)r * a.sH * a.sW;
        } else {                                          // edge padding folded into the load
          g = batch;
          si = row - (a.sH - 1); si = si < 0 ? 0 : (si > a.sH - 1 ? a.sH - 1 : si);
          sj = col - (a.sW - 1); sj = sj < 0 ? 0 : (sj > a.sW - 1 ? a.sW - 1 : sj);
        }
        const int c = kGroupCh * g + 2 * p;
        if (c < a.D) {
          const float* s = a.srcf + (img + (int64_t)si * a.sW + sj) * a.D + c;
          if (c + 1 < a.D && !(a.D & 1)) v = *reinterpret_cast<const float2*>(s);
          else { v.x = s[0]; if (c + 1 < a.D) v.y = s[1]; }
        }
      } else if (a.mode == kSlowRotate) {
        const int r = batch / a.G, g = batch - r * a.G, RQ = a.R >> 2;
        const int k = r / RQ, r0 = r - k * RQ;
        int si, sj;
        snap_rot90_source(k, row, col, a.sH, a.sW, &si, &sj);
        const SnapRotSample rs = snap_rot_sample(a.tfm + r0 * 4, si, sj, a.sH, a.sW, a.cell, a.srcb);
        const int c = kGroupCh * g + 2 * p;
        if (rs.ok && c < a.D && !(VF_ABLATE & 32)) {
          const float* f00 = a.srcf + ((int64_t)rs.i0 * a.sW + rs.j0) * a.D + c;
          const float* f01 = a.srcf + ((int64_t)rs.i0 * a.sW + rs.j1) * a.D + c;
          const float* f10 = a.srcf + ((int64_t)rs.i1 * a.sW + rs.j0) * a.D + c;
          const float* f11 = a.srcf + ((int64_t)rs.i1 * a.sW + rs.j1) * a.D + c;
          if (c + 1 < a.D && !(a.D & 1)) {
            const float2 a00 = *reinterpret_cast<const float2*>(f00), a01 = *reinterpret_cast<const float2*>(f01);
            const float2 a10 = *reinterpret_cast<const float2*>(f10), a11 = *reinterpret_cast<const float2*>(f11);
            v.x = snap_rot_mix(rs, a00.x, a01.x, a10.x, a11.x);
            v.y = snap_rot_mix(rs, a00.y, a01.y, a10.y, a11.y);
          } else {
            v.x = snap_rot_mix(rs, f00[0], f01[0], f10[0], f11[0]);
            if (c + 1 < a.D) v.y = snap_rot_mix(rs, f00[1], f01[1], f10[1], f11[1]);
          }
        }
      } else if (a.mode == kSlowCount) {
        // count filter = 180-degree rotated template mask (pose_exhaustive_voting.py:97-99 passes
        // q_valid UN-flipped to a true convolution); rotations rp and rp + R2 as re / im
        const int rp = kCols * batch + p;
        if (rp < a.R2) {
          const int64_t idx = (int64_t)(a.sH - 1 - row) * a.sW + (a.sW - 1 - col);
          const int64_t hw = (int64_t)a.sH * a.sW;
          v.x = a.srcb[rp * hw + idx] ? 1.f : 0.f;
          if (rp + a.R2 < a.R) v.y = a.srcb[(rp + a.R2) * hw + idx] ? 1.f : 0.f;
        }
      } else {                                            // map validity, ZERO outside the map
        if (p == 0) {
          const int si = row - (a.sH - 1), sj = col - (a.sW - 1);
          if (si >= 0 && si < a.sH && sj >= 0 && sj < a.sW)
            v.x = a.srcb[(int64_t)si * a.sW + sj] ? 1.f : 0.f;
        }
      }
    }
    if (fuse0) {
      float2 y[3];
      stage0_zero_tail(v, row, twl, a.pl, y);
      buf[vf_phys<kCols>(row) * kCols + p] = y[0];
      buf[vf_phys<kCols>(row + M0) * kCols + p] = y[1];
      buf[vf_phys<kCols>(row + 2 * M0) * kCols + p] = y[2];
    } else {
      buf[vf_phys<kCols>(row) * kCols + p] = v;
    }
  }
  VF_SYNC();
  if (!(VF_ABLATE & 64)) fft_lds<kCols, false>(buf, twl, a.pl, tid, nt, fuse0);
  float2* d = a.dst + ((int64_t)batch * N * a.ncols + col) * kCols + p;
  for (int row = slot; row < N && !(VF_ABLATE & 128); row += nslot) d[(int64_t)row * a.ncols * kCols] = buf[vf_phys<kCols>(row) * kCols + p];
}

// ------------------------------------------------------------------------------------------------
// Kernel B: forward transform along the FAST spatial axis of one k1 row, then by mode
//   STORE16 -> Z[batch][k1][k2'][16]        (map spectrum)
//   STORE1  -> Z[k1][k2']  (column 0 only)  (map-validity spectrum)
//   DOT     -> S[k2'] = sum_g sum_p conj(X[k2'][p]) Zm[g][k1][k2'][p]; inverse along k2 -> Y[outer][k1][b]
//   MUL     -> X[k2'][p] = conj(X) Zv[k1][k2']; inverse along k2 (16 columns) -> Yc[batch][k1][b][16]
//   grid (N1, nouter); x1[outer * gloop + g][k1][n_in][16]
// ------------------------------------------------------------------------------------------------
enum { kFastStore16 = 0, kFastStore1 = 1, kFastDot = 2, kFastMul = 3 };

struct FastArgs {
  Plan pl;
  const float2* tw;
  int mode;
  const float2* x1;
  int n_in, N1, gloop;
  const float2* z;
  float2* out;
  int ld_out;               // DOT / MUL: row stride of the output in b (multiple of 16)
  int nb_out;               // DOT / MUL: columns b written (<= min(ld_out, N))
};

VF_DEV void fast_body(const FastArgs& a, int bx, int by, int tid, int nt, float2* buf, float2* twl,
                      float2* sbuf) {
  const int N = a.pl.N;
  for (int t = tid; t < N; t += nt) twl[t] = a.tw[t];
  if (a.mode == kFastDot)
    for (int t = tid; t < N; t += nt) { sbuf[t].x = 0.f; sbuf[t].y = 0.f; }
  const int k1 = bx, outer = by;
  float4* b4 = reinterpret_cast<float4*>(buf);
  const int n4_in = a.n_in * (kCols / 2), n4 = N * (kCols / 2);
  const bool fuse0 = stage0_fusable(a.pl, a.n_in);
  const int n4_st = fuse0 ? (N / 3) * (kCols / 2) : n4;      // 16-byte items the staging walks
  const int n4_m = (N / 3) * (kCols / 2);
  // the row of the NEXT channel group is requested while this one is transformed (the barriers of these
  // kernels wait for LDS traffic only): kXPre 16-byte loads per thread stay in flight across the passes
  constexpr int kXPre = 2;
  float4 pre[kXPre];
  {
    const float4* src0 = reinterpret_cast<const float4*>(a.x1 + (((int64_t)outer * a.gloop) * a.N1 + k1) * a.n_in * kCols);
#pragma unroll
    for (int u = 0; u < kXPre; ++u) {
      const int i = tid + u * nt;
      pre[u].x = pre[u].y = pre[u].z = pre[u].w = 0.f;
      if (i < n4_in && i < n4_st && !(VF_ABLATE & 1)) pre[u] = src0[i];
    }
  }
  if (fuse0) VF_SYNC();                                      // (the staging reads the twiddles)
  for (int g = 0; g < a.gloop; ++g) {
    const int64_t batch = (int64_t)outer * a.gloop + g;
    const float4* src = reinterpret_cast<const float4*>(a.x1 + (batch * a.N1 + k1) * a.n_in * kCols);
    for (int u = 0; tid + u * nt < n4_st; ++u) {
      const int i = tid + u * nt;
      float4 v; v.x = v.y = v.z = v.w = 0.f;
      if (u < kXPre) {
        v = u == 0 ? pre[0] : pre[kXPre - 1];                // (kXPre == 2)
      } else if (i < n4_in && !(VF_ABLATE & 1)) {
        v = src[i];
      }
      if (fuse0) {
        const int row = i / (kCols / 2);
        float2 x, y[3];
        float4 o0, o1, o2;
        x.x = v.x; x.y = v.y;
        stage0_zero_tail(x, row, twl, a.pl, y);
        o0.x = y[0].x; o0.y = y[0].y; o1.x = y[1].x; o1.y = y[1].y; o2.x = y[2].x; o2.y = y[2].y;
        x.x = v.z; x.y = v.w;
        stage0_zero_tail(x, row, twl, a.pl, y);
        o0.z = y[0].x; o0.w = y[0].y; o1.z = y[1].x; o1.w = y[1].y; o2.z = y[2].x; o2.w = y[2].y;
        b4[vf_phys4(i)] = o0; b4[vf_phys4(i + n4_m)] = o1; b4[vf_phys4(i + 2 * n4_m)] = o2;
      } else {
        b4[vf_phys4(i)] = v;
      }
    }
    if (g + 1 < a.gloop) {
      const float4* nsrc = reinterpret_cast<const float4*>(a.x1 + ((batch + 1) * a.N1 + k1) * a.n_in * kCols);
#pragma unroll
      for (int u = 0; u < kXPre; ++u) {
        const int i = tid + u * nt;
        pre[u].x = pre[u].y = pre[u].z = pre[u].w = 0.f;
        if (i < n4_in && i < n4_st && !(VF_ABLATE & 1)) pre[u] = nsrc[i];
      }
    }
    // DOT: this row of the map spectrum (98 KB, shared by every rotation: L2 / Infinity-Cache resident
    // after the first) is TOUCHED before the transform -- one dword per 128-byte line, a register each
    // -- so that its trip from the Infinity Cache / HBM into this XCD's L2 runs under the LDS passes;
    // the 16-byte loads that follow the transform then hit L2.  (Holding the row itself in registers
    // across the transform spilled: the fused radix-16 passes need the registers.)
    const float4* z4 = nullptr;
    float touch = 0.f;
    if (a.mode == kFastDot) {
      z4 = reinterpret_cast<const float4*>(a.z + ((int64_t)g * a.N1 + k1) * N * kCols);
      const float* zt = reinterpret_cast<const float*>(z4);
      const int nlines = N * kCols / 16;                // 128-byte lines of the row
      if (tid < nlines) touch = zt[tid * 32];          // ONE un-waited load per thread (nt >= nlines on the GPU) ...
      for (int line = tid + nt; line < nlines; line += nt) touch += zt[line * 32];   // ... (smaller workgroups: the rest)
    }
    VF_SYNC();
    if (!(VF_ABLATE & 2) || a.mode != kFastDot) fft_lds<kCols, false>(buf, twl, a.pl, tid, nt, fuse0);
    if (a.mode == kFastStore16) {
      float4* d = reinterpret_cast<float4*>(a.out + (batch * a.N1 + k1) * N * kCols);
      for (int i = tid; i < n4; i += nt) d[i] = b4[vf_phys4(i)];
    } else if (a.mode == kFastStore1) {
      for (int t = tid; t < N; t += nt) a.out[(int64_t)k1 * N + t] = buf[vf_phys<kCols>(t) * kCols];
    } else if (a.mode == kFastDot) {
      // products Zm * conj(X) in place (kZPre coalesced 16-byte loads in flight per thread), then one
      // thread per k2' sums its 16 pairs (rotated start: the 128-byte row stride would put every lane
      // on the same banks)
      for (int i0 = tid; i0 < n4; i0 += kZPre * nt) {
        float4 zreg[kZPre];
#pragma unroll
        for (int u = 0; u < kZPre; ++u) {
          const int i = i0 + u * nt;
          zreg[u].x = zreg[u].y = zreg[u].z = zreg[u].w = 0.f;
          if (i < n4 && !(VF_ABLATE & 4)) zreg[u] = z4[i];
        }
#pragma unroll
        for (int u = 0; u < kZPre; ++u) {
          const int i = i0 + u * nt;
          if (i < n4) { const int ip = vf_phys4(i); b4[ip] = cmulc2(zreg[u], b4[ip]); }
        }
      }
      VF_KEEP(touch);                                  // the touch loads' only use: after the transform
      VF_SYNC();
      for (int k2 = tid; k2 < N && !(VF_ABLATE & 16); k2 += nt) {
        const int ks = vf_phys<1>(k2), kb = vf_phys<kCols>(k2) * kCols;
        float2 acc = sbuf[ks];
#pragma unroll
        for (int j = 0; j < kCols; ++j) acc = cadd(acc, buf[kb + ((j + tid) & (kCols - 1))]);
        sbuf[ks] = acc;
      }
    } else {
      for (int i = tid; i < N * kCols; i += nt) {
        const int ip = vf_phys<kCols>(i >> kColShift) * kCols + (i & (kCols - 1));
        buf[ip] = cmulc(a.z[(int64_t)k1 * N + (i >> kColShift)], buf[ip]);
      }
    }
    VF_SYNC();
  }
  if (a.mode == kFastDot) {
    if (!(VF_ABLATE & 8)) fft_lds<1, true>(sbuf, twl, a.pl, tid, nt);
    float2* d = a.out + ((int64_t)outer * a.N1 + k1) * a.ld_out;
    for (int t = tid; t < a.nb_out; t += nt) d[t] = sbuf[vf_phys<1>(t)];
  } else if (a.mode == kFastMul) {
    fft_lds<kCols, true>(buf, twl, a.pl, tid, nt);
    float2* d = a.out + ((int64_t)outer * a.N1 + k1) * a.ld_out * kCols;
    for (int i = tid; i < a.nb_out * kCols; i += nt) d[i] = buf[vf_phys<kCols>(i >> kColShift) * kCols + (i & (kCols - 1))];
  }
}

// ------------------------------------------------------------------------------------------------
// Kernel C: inverse transform along k1 of 16 columns, then
//   SCORE: columns = 16 adjacent b of Y[r][k1][b]; scores[r][a][b] = Re / (N1 N2), -inf where the
//          overlap flag is 0, / tcount[r]   (pose_exhaustive_voting.py:101-104)      grid (ld / 16, R)
//   COUNT: columns = 16 rotation pairs of Yc[gc][k1][b][16]; flag[r][a][b] = rint(count) > thr
//                                                                                    grid (Wo, GC)
// ------------------------------------------------------------------------------------------------
enum { kInvScore = 0, kInvCount = 1 };

struct InvArgs {
  Plan pl;
  const float2* tw;
  int mode;
  const float2* y;
  int ld, nb_valid;         // row stride of y in b; columns b that were written
  int Ho, Wo, R, R2;
  float scale, thr;
  int use_overlap;
  const uint8_t* flags_in;
  uint8_t* flags_out;
  const float* tcount;
  float* scores;
};

VF_DEV void inv_body(const InvArgs& a, int bx, int by, int tid, int nt, float2* buf, float2* twl) {
  const int N = a.pl.N;
  for (int t = tid; t < N; t += nt) twl[t] = a.tw[t];
  const int p = tid & (kCols - 1), slot = tid >> kColShift, nslot = nt >> kColShift;
  if (a.mode == kInvScore) {
    const int r = by, b = bx * kCols + p;
    const float2* y = a.y + (int64_t)r * N * a.ld + b;
    for (int row = slot; row < N; row += nslot) {
      float2 v; v.x = 0.f; v.y = 0.f;
      if (b < a.nb_valid) v = y[(int64_t)row * a.ld];
      buf[vf_phys<kCols>(row) * kCols + p] = v;
    }
  } else {
    const float2* y = a.y + ((int64_t)by * N * a.ld + bx) * kCols + p;
    for (int row = slot; row < N; row += nslot) buf[vf_phys<kCols>(row) * kCols + p] = y[(int64_t)row * a.ld * kCols];
  }
  VF_SYNC();
  fft_lds<kCols, true>(buf, twl, a.pl, tid, nt);
  if (a.mode == kInvScore) {
    const int r = by, b = bx * kCols + p;
    if (b < a.Wo) {
      const float tc = a.tcount[r];
      for (int row = slot; row < a.Ho; row += nslot) {
        const int64_t o = ((int64_t)r * a.Ho + row) * a.Wo + b;
        float v = buf[vf_phys<kCols>(row) * kCols + p].x * a.scale;
        if (a.use_overlap && !a.flags_in[o]) v = -INFINITY;
        a.scores[o] = v / tc;
      }
    }
  } else {
    const int rp = kCols * by + p, b = bx;
    if (rp < a.R2) {
      for (int row = slot; row < a.Ho; row += nslot) {
        const float2 c = buf[vf_phys<kCols>(row) * kCols + p];
        a.flags_out[((int64_t)rp * a.Ho + row) * a.Wo + b] = rintf(c.x * a.scale) > a.thr ? 1 : 0;
        if (rp + a.R2 < a.R)
          a.flags_out[((int64_t)(rp + a.R2) * a.Ho + row) * a.Wo + b] = rintf(-c.y * a.scale) > a.thr ? 1 : 0;
      }
    }
  }
}

// Template masks of a rotated call: cell idx of the first quadrant (r0, si, sj) -> tvalid of its four
// rot90 copies; returns the validity (the caller counts: tcount[k * RQ + r0] += ok).
VF_DEV bool rot_mask_body(const RotSource& rs, int H, int W, int R, int64_t idx, uint8_t* tvalid, int* r0_out) {
  const int RQ = R >> 2;
  const int sj = (int)(idx % W);
  const int64_t t = idx / W;
  const int si = (int)(t % H);
  const int r0 = (int)(t / H);
  *r0_out = r0;
  const bool ok = snap_rot_sample(rs.tfm + r0 * 4, si, sj, H, W, rs.cell, rs.valid).ok;
  for (int k = 0; k < 4; ++k) {
    int di, dj;                           // destination of (si, sj) under rot90(., k, axes=(2,1))
    if (k == 0) { di = si; dj = sj; }
    else if (k == 1) { di = sj; dj = H - 1 - si; }
    else if (k == 2) { di = H - 1 - si; dj = W - 1 - sj; }
    else { di = H - 1 - sj; dj = si; }
    tvalid[((int64_t)(k * RQ + r0) * H + di) * W + dj] = ok ? 1 : 0;
  }
  return ok;
}

// tw[t] = exp(-2 pi i t / N), evaluated in double precision
VF_DEV void twiddle_body(float2* tw, int N, int t) {
  if (t < N) {
    const double ang = -2.0 * 3.14159265358979323846 * (double)t / (double)N;
    float2 w; w.x = (float)cos(ang); w.y = (float)sin(ang);
    tw[t] = w;
  }
}

// ------------------------------------------------------------------------------------------------
// Problem geometry, workspace carving and the launch sequence (shared with the emulation).
// ------------------------------------------------------------------------------------------------
struct Geometry {
  int R, H, W, D, Hm, Wm;
  int Hp, Wp, Ho, Wo, N1, N2, G, R2, GC, ld, nb;
  Plan p1, p2;
  // workspace offsets in bytes
  size_t o_tw1, o_tw2, o_xm1, o_zm, o_x1, o_y, o_zv, o_xc1, o_yc, o_flags, o_tvalid, o_tcount, total;
};

static inline size_t vf_align(size_t v) { return (v + 255) & ~(size_t)255; }

static inline bool make_geometry(int R, int H, int W, int D, int Hm, int Wm, Geometry* g) {
  if (R <= 0 || H <= 0 || W <= 0 || D <= 0 || Hm <= 0 || Wm <= 0) return false;
  g->R = R; g->H = H; g->W = W; g->D = D; g->Hm = Hm; g->Wm = Wm;
  g->Hp = 3 * Hm - 2; g->Wp = 3 * Wm - 2;
  g->Ho = g->Hp - H + 1; g->Wo = g->Wp - W + 1;
  if (g->Ho <= 0 || g->Wo <= 0) return false;
  g->N1 = fft_size(g->Hp); g->N2 = fft_size(g->Wp);
  if (g->N1 > kMaxN || g->N2 > kMaxN) return false;
  if (!make_plan(g->N1, &g->p1) || !make_plan(g->N2, &g->p2)) return false;
  g->G = (D + kGroupCh - 1) / kGroupCh;
  g->R2 = (R + 1) / 2; g->GC = (g->R2 + kCols - 1) / kCols;
  g->ld = (g->Wo + kCols - 1) / kCols * kCols;
  g->nb = g->ld < g->N2 ? g->ld : g->N2;
  const size_t c = sizeof(float2);
  size_t o = 0;
  g->o_tw1 = o; o += vf_align(c * g->N1);
  g->o_tw2 = o; o += vf_align(c * g->N2);
  g->o_xm1 = o; o += vf_align(c * kCols * (size_t)g->G * g->N1 * g->Wp);     // also the validity pass
  g->o_zm = o;  o += vf_align(c * kCols * (size_t)g->G * g->N1 * g->N2);
  g->o_x1 = o;  o += vf_align(c * kCols * (size_t)R * g->G * g->N1 * W);
  g->o_y = o;   o += vf_align(c * (size_t)R * g->N1 * g->ld);
  g->o_zv = o;  o += vf_align(c * (size_t)g->N1 * g->N2);
  g->o_xc1 = o; o += vf_align(c * kCols * (size_t)g->GC * g->N1 * W);
  g->o_yc = o;  o += vf_align(c * kCols * (size_t)g->GC * g->N1 * g->ld);
  g->o_flags = o; o += vf_align((size_t)R * g->Ho * g->Wo);
  g->o_tvalid = o; o += vf_align((size_t)R * H * W);      // (rotated entry point: the template masks)
  g->o_tcount = o; o += vf_align(sizeof(float) * (size_t)R);
  g->total = o;
  return true;
}

// LAUNCH: a functor with  twiddle(float2*, N), slow(SlowArgs, gx, gy), fast(FastArgs, gx, gy),
// inv(InvArgs, gx, gy), each returning false on a launch failure.
template <class LAUNCH>
static inline bool run_voting(const Geometry& g, const float* q, const uint8_t* q_valid, const float* m,
                              const uint8_t* m_valid, const float* tcount, float thr, int use_overlap,
                              char* ws, float* scores, LAUNCH& L, const RotSource* rot = nullptr) {
  float2* tw1 = reinterpret_cast<float2*>(ws + g.o_tw1);
  float2* tw2 = reinterpret_cast<float2*>(ws + g.o_tw2);
  float2* xm1 = reinterpret_cast<float2*>(ws + g.o_xm1);
  float2* zm = reinterpret_cast<float2*>(ws + g.o_zm);
  float2* x1 = reinterpret_cast<float2*>(ws + g.o_x1);
  float2* y = reinterpret_cast<float2*>(ws + g.o_y);
  float2* zv = reinterpret_cast<float2*>(ws + g.o_zv);
  float2* xc1 = reinterpret_cast<float2*>(ws + g.o_xc1);
  float2* yc = reinterpret_cast<float2*>(ws + g.o_yc);
  uint8_t* flags = reinterpret_cast<uint8_t*>(ws + g.o_flags);
  if (!L.twiddle(tw1, g.N1) || !L.twiddle(tw2, g.N2)) return false;
  const float scale = (float)(1.0 / ((double)g.N1 * (double)g.N2));

  SlowArgs sa; FastArgs fa; InvArgs ia;
  if (use_overlap) {
    // map validity -> Zv[k1'][k2']
    sa = SlowArgs(); sa.pl = g.p1; sa.tw = tw1; sa.mode = kSlowMvalid; sa.srcf = nullptr; sa.srcb = m_valid;
    sa.n_in = g.Hp; sa.ncols = g.Wp; sa.sH = g.Hm; sa.sW = g.Wm; sa.D = g.D; sa.G = 1; sa.R = g.R; sa.R2 = g.R2;
    sa.dst = xm1;
    if (!L.slow(sa, g.Wp, 1)) return false;
    fa = FastArgs(); fa.pl = g.p2; fa.tw = tw2; fa.mode = kFastStore1; fa.x1 = xm1; fa.n_in = g.Wp; fa.N1 = g.N1;
    fa.gloop = 1; fa.z = nullptr; fa.out = zv; fa.ld_out = 0; fa.nb_out = 0;
    if (!L.fast(fa, g.N1, 1)) return false;
    // template masks (rotation pairs) -> counts -> flags
    sa.mode = kSlowCount; sa.srcb = q_valid; sa.n_in = g.H; sa.ncols = g.W; sa.sH = g.H; sa.sW = g.W; sa.dst = xc1;
    if (!L.slow(sa, g.W, g.GC)) return false;
    fa.mode = kFastMul; fa.x1 = xc1; fa.n_in = g.W; fa.z = zv; fa.out = yc; fa.ld_out = g.ld; fa.nb_out = g.nb;
    if (!L.fast(fa, g.N1, g.GC)) return false;
    ia = InvArgs(); ia.pl = g.p1; ia.tw = tw1; ia.mode = kInvCount; ia.y = yc; ia.ld = g.ld; ia.nb_valid = g.nb;
    ia.Ho = g.Ho; ia.Wo = g.Wo; ia.R = g.R; ia.R2 = g.R2; ia.scale = scale; ia.thr = thr; ia.use_overlap = 1;
    ia.flags_in = nullptr; ia.flags_out = flags; ia.tcount = tcount; ia.scores = nullptr;
    if (!L.inv(ia, g.Wo, g.GC)) return false;
  }
  // map features -> Zm[g][k1'][k2'][16]
  sa = SlowArgs(); sa.pl = g.p1; sa.tw = tw1; sa.mode = kSlowMap; sa.srcf = m; sa.srcb = nullptr;
  sa.n_in = g.Hp; sa.ncols = g.Wp; sa.sH = g.Hm; sa.sW = g.Wm; sa.D = g.D; sa.G = g.G; sa.R = g.R; sa.R2 = g.R2;
  sa.dst = xm1;
  if (!L.slow(sa, g.Wp, g.G)) return false;
  fa = FastArgs(); fa.pl = g.p2; fa.tw = tw2; fa.mode = kFastStore16; fa.x1 = xm1; fa.n_in = g.Wp; fa.N1 = g.N1;
  fa.gloop = 1; fa.z = nullptr; fa.out = zm; fa.ld_out = 0; fa.nb_out = 0;
  if (!L.fast(fa, g.N1, g.G)) return false;
  // templates -> X1[r * G + g][k1'][j][16] -> Y[r][k1'][b] -> scores
  sa.mode = kSlowTemplate; sa.srcf = q; sa.n_in = g.H; sa.ncols = g.W; sa.sH = g.H; sa.sW = g.W; sa.dst = x1;
  if (rot && rot->feat) {
    sa.mode = kSlowRotate; sa.srcf = rot->feat; sa.srcb = rot->valid; sa.tfm = rot->tfm; sa.cell = rot->cell;
  }
  if (!L.slow(sa, g.W, g.R * g.G)) return false;
  fa.mode = kFastDot; fa.x1 = x1; fa.n_in = g.W; fa.gloop = g.G; fa.z = zm; fa.out = y; fa.ld_out = g.ld;
  fa.nb_out = g.nb;
  if (!L.fast(fa, g.N1, g.R)) return false;
  ia = InvArgs(); ia.pl = g.p1; ia.tw = tw1; ia.mode = kInvScore; ia.y = y; ia.ld = g.ld; ia.nb_valid = g.nb;
  ia.Ho = g.Ho; ia.Wo = g.Wo; ia.R = g.R; ia.R2 = g.R2; ia.scale = scale; ia.thr = thr; ia.use_overlap = use_overlap;
  ia.flags_in = flags; ia.flags_out = nullptr; ia.tcount = tcount; ia.scores = scores;
  if (!L.inv(ia, g.ld / kCols, g.R)) return false;
  return true;
}

}  // namespace vfft

#endif  // SNAP_CSRC_VOTING_FFT_BODY_H_
