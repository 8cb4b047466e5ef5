// Live per-kernel-family profiler: one HIP event pair around each instrumented launch, recorded on the stream the
// kernel is launched on (bench.py's `roofline` objects: achieved FLOP/s or bytes/s = algorithmic work of the launch /
// its measured duration).  Off by default; nothing is recorded inside a timed throughput window.
#include <hip/hip_runtime.h>

#include "../../include/uwu_hip.h"

void uwu_set_error(const char* fmt, ...);

namespace {
constexpr int MAXP = 16384;
struct Prof {
  bool on = false, created = false;
  int n = 0;
  hipEvent_t ev[2 * MAXP];
  double flops[MAXP], bytes[MAXP];
  short tag[MAXP], kind[MAXP];
};
Prof g;
}  // namespace

// internal (common.h: UwuProfScope)
int uwu_prof_begin(void* st) {
  if (!g.on || g.n >= MAXP) return -1;
  const int slot = g.n++;
  (void)hipEventRecord(g.ev[2 * slot], static_cast<hipStream_t>(st));
  g.tag[slot] = -1;
  return slot;
}
void uwu_prof_end(int slot, int tag, int kind, double flops, double bytes, void* st) {
  if (slot < 0) return;
  (void)hipEventRecord(g.ev[2 * slot + 1], static_cast<hipStream_t>(st));
  g.tag[slot] = (short)tag;
  g.kind[slot] = (short)kind;
  g.flops[slot] = flops;
  g.bytes[slot] = bytes;
}

extern "C" int uwu_prof_enable(int on) {
  if (on && !g.created) {
    for (int i = 0; i < 2 * MAXP; ++i)
      if (hipEventCreate(&g.ev[i]) != hipSuccess) {
        uwu_set_error("prof: hipEventCreate failed");
        return UWU_ELAUNCH;
      }
    g.created = true;
  }
  g.on = on != 0;
  if (on) g.n = 0;
  return UWU_OK;
}

// Sums over the launches recorded since the last enable with this tag (tag < 0: every GEMM tag) and operand kind
// (0 = bf16 / fp8 operands, 1 = fp32, < 0 = any).  Waits for the recorded events (host side, outside timed regions).
extern "C" int uwu_prof_collect(int tag, int kind, double* ms, double* flops, double* bytes, int* launches) {
  double t = 0.0, f = 0.0, b = 0.0;
  int c = 0;
  for (int i = 0; i < g.n; ++i) {
    if (g.tag[i] < 0) continue;
    const bool is_gemm = g.tag[i] <= UWU_PROF_GEMM_FC2_DGELU;
    if (tag < 0 ? !is_gemm : g.tag[i] != tag) continue;
    if (kind >= 0 && g.kind[i] != kind) continue;
    float e = 0.f;
    if (hipEventSynchronize(g.ev[2 * i + 1]) != hipSuccess ||
        hipEventElapsedTime(&e, g.ev[2 * i], g.ev[2 * i + 1]) != hipSuccess) {
      uwu_set_error("prof: event query failed");
      return UWU_ELAUNCH;
    }
    t += e;
    f += g.flops[i];
    b += g.bytes[i];
    ++c;
  }
  if (ms) *ms = t;
  if (flops) *flops = f;
  if (bytes) *bytes = b;
  if (launches) *launches = c;
  return UWU_OK;
}

// round-1 names (GEMM family only)
extern "C" int uwu_gemm_prof_enable(int on) { return uwu_prof_enable(on); }
extern "C" int uwu_gemm_prof_collect(int kind, double* ms, double* flops, int* launches) {
  return uwu_prof_collect(-1, kind, ms, flops, nullptr, launches);
}
