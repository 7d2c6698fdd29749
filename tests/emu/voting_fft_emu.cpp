// CPU emulation harness of snap_amd/csrc/voting_fft_body.h -- TEST INFRASTRUCTURE ONLY.
//
// Compiles the very kernel bodies the HIP library runs (same header, VF_EMU) with g++: one pthread
// per GPU thread of a workgroup, a pthread barrier per __syncthreads(), the 16-lane xor shuffle
// through a shared exchange array.  tests/test_host_logic.py builds it into tests/_build/ and
// compares emu_voting_fft_f32 with oracle/voting.py, so that the index arithmetic, the plan and the
// launch sequence are checked without a GPU.  Nothing in snap_amd/ loads this file.
#define VF_EMU
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

static thread_local pthread_barrier_t* tl_bar = nullptr;
static thread_local float* tl_xch = nullptr;

#include "voting_fft_body.h"

void vf_emu_barrier() { pthread_barrier_wait(tl_bar); }
float vf_emu_shfl_xor(float v, int mask, int tid) {
  tl_xch[tid] = v;
  pthread_barrier_wait(tl_bar);
  const float r = tl_xch[tid ^ mask];
  pthread_barrier_wait(tl_bar);
  return r;
}

namespace {

struct Job {
  int kind, gx, gy, nt, tid;
  const void* args;
  char* smem;
  pthread_barrier_t* bar;
  float* xch;
};

void* worker(void* pv) {
  Job* j = static_cast<Job*>(pv);
  tl_bar = j->bar; tl_xch = j->xch;
  for (int by = 0; by < j->gy; ++by)
    for (int bx = 0; bx < j->gx; ++bx) {
      float2* buf = reinterpret_cast<float2*>(j->smem);
      if (j->kind == 0) {
        const vfft::SlowArgs& a = *static_cast<const vfft::SlowArgs*>(j->args);
        vfft::slow_body(a, bx, by, j->tid, j->nt, buf, buf + a.pl.N * vfft::kCols);
      } else if (j->kind == 1) {
        const vfft::FastArgs& a = *static_cast<const vfft::FastArgs*>(j->args);
        float2* twl = buf + a.pl.N * vfft::kCols;
        vfft::fast_body(a, bx, by, j->tid, j->nt, buf, twl, twl + a.pl.N);
      } else {
        const vfft::InvArgs& a = *static_cast<const vfft::InvArgs*>(j->args);
        vfft::inv_body(a, bx, by, j->tid, j->nt, buf, buf + a.pl.N * vfft::kCols);
      }
      pthread_barrier_wait(j->bar);     // workgroup boundary: LDS is reused
    }
  return nullptr;
}

struct EmuLaunch {
  int nt;
  bool run(int kind, const void* args, int N, int gx, int gy) {
    std::vector<char> smem(sizeof(float2) * (size_t)N * (vfft::kCols + 2));
    std::vector<float> xch(nt);
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, nullptr, nt);
    std::vector<Job> jobs(nt);
    std::vector<pthread_t> th(nt);
    for (int t = 0; t < nt; ++t) {
      jobs[t] = Job{kind, gx, gy, nt, t, args, smem.data(), &bar, xch.data()};
      if (pthread_create(&th[t], nullptr, worker, &jobs[t]) != 0) return false;
    }
    for (int t = 0; t < nt; ++t) pthread_join(th[t], nullptr);
    pthread_barrier_destroy(&bar);
    return true;
  }
  bool twiddle(float2* tw, int N) { for (int t = 0; t < N; ++t) vfft::twiddle_body(tw, N, t); return true; }
  bool slow(const vfft::SlowArgs& a, int gx, int gy) { return run(0, &a, a.pl.N, gx, gy); }
  bool fast(const vfft::FastArgs& a, int gx, int gy) { return run(1, &a, a.pl.N, gx, gy); }
  bool inv(const vfft::InvArgs& a, int gx, int gy) { return run(2, &a, a.pl.N, gx, gy); }
};

}  // namespace

extern "C" size_t emu_voting_fft_workspace_bytes(int R, int H, int W, int D, int Hm, int Wm) {
  vfft::Geometry g;
  return vfft::make_geometry(R, H, W, D, Hm, Wm, &g) ? g.total : 0;
}

extern "C" int emu_voting_fft_f32(const float* templates, const uint8_t* tvalid, const float* map,
                                  const uint8_t* mvalid, const float* tcount, int R, int H, int W, int D,
                                  int Hm, int Wm, float thr, int use_overlap, void* ws, float* scores,
                                  int nt) {
  vfft::Geometry g;
  if (!vfft::make_geometry(R, H, W, D, Hm, Wm, &g)) return -2;
  EmuLaunch L{nt};
  return vfft::run_voting(g, templates, tvalid, map, mvalid, tcount, thr, use_overlap, static_cast<char*>(ws),
                          scores, L) ? 0 : -4;
}

extern "C" int emu_voting_fft_rotated_f32(const float* feat, const uint8_t* valid, const float* tfm, float cell,
                                          const float* map, const uint8_t* mvalid, int R, int H, int D, int Hm,
                                          int Wm, float min_overlap, void* ws_, float* scores, int nt) {
  vfft::Geometry g;
  if (!vfft::make_geometry(R, H, H, D, Hm, Wm, &g)) return -2;
  char* ws = static_cast<char*>(ws_);
  uint8_t* tvalid = reinterpret_cast<uint8_t*>(ws + g.o_tvalid);
  float* tcount = reinterpret_cast<float*>(ws + g.o_tcount);
  for (int r = 0; r < R; ++r) tcount[r] = 0.f;
  vfft::RotSource rs{feat, valid, tfm, cell};
  const int RQ = R / 4;
  for (int64_t idx = 0; idx < (int64_t)RQ * H * H; ++idx) {
    int r0;
    if (vfft::rot_mask_body(rs, H, H, R, idx, tvalid, &r0))
      for (int k = 0; k < 4; ++k) tcount[k * RQ + r0] += 1.f;
  }
  EmuLaunch L{nt};
  return vfft::run_voting(g, nullptr, tvalid, map, mvalid, tcount, min_overlap * (float)H * (float)H, 1, ws, scores,
                          L, &rs) ? 0 : -4;
}
