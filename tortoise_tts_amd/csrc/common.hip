#include "common.h"
#include <stdarg.h>

namespace tt {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* last_error() { return g_err; }

}  // namespace tt

// ---------------------------------------------------------------- kernel-class profiler
#include <vector>
namespace tt {
bool g_prof_on = false;
bool g_graph_replay = true;
struct ProfClass {
  std::vector<hipEvent_t> start, stop;
  double flops = 0, bytes = 0;
};
static ProfClass g_prof[PROF_COUNT];
static const char* g_prof_names[PROF_COUNT] = {
    "gemm_glds<64,64,EpiStd,1x1>", "gemm_glds<64,64,EpiStd,conv>", "gemm_glds<64,64,EpiQkvHeads>", "gemm_glds<64,64,EpiQkvDecode>",
    "gemm_glds<128,64,EpiStd,1x1>", "gemm_glds<128,64,EpiStd,conv>", "gemm_glds<128,64,EpiQkvHeads>", "gemm_glds<128,64,EpiQkvDecode>",
    "gemm_glds<128,128,EpiStd,1x1>", "gemm_glds<128,128,EpiStd,conv>", "gemm_glds<128,128,EpiQkvHeads>", "gemm_glds<128,128,EpiQkvDecode>",
    "gemm_glds<256,256,EpiStd,1x1>", "gemm_glds<256,256,EpiStd,conv>", "gemm_glds<256,256,EpiQkvHeads>", "gemm_glds<256,256,EpiQkvDecode>",
    "flash_kernel", "decode_attn_kernel", "rownorm_kernel", "gn_apply_kernel(+gn_stats)", "sample_kernel", "glue",
    "conv1d_direct_kernel", "convt1d_kernel", "lvc_kernel", "gemm_glds<64,64,EpiStd,1x1,stats>",
    "gemm_gna<32,256,EpiStd,stats>",
    "gemm_glds<32,16,EpiStd,1x1>", "gemm_glds<32,16,EpiQkvDecode>", "gemm_glds<64,16,EpiStd,1x1>", "gemm_glds<64,16,EpiQkvDecode>", "gemv_kernel"};
const char* prof_name(int id) { return id >= 0 && id < PROF_COUNT ? g_prof_names[id] : "?"; }

void prof_record(int id, hipStream_t s, bool begin, double flops, double bytes) {
  ProfClass& c = g_prof[id];
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, s);
  if (begin) {
    c.start.push_back(e);
    c.flops += flops;
    c.bytes += bytes;
  } else {
    c.stop.push_back(e);
  }
}
void prof_pair(int id, double flops, double bytes, hipEvent_t* e0, hipEvent_t* e1) {
  ProfClass& c = g_prof[id];
  *e0 = *e1 = nullptr;
  if (hipEventCreate(e0) != hipSuccess || hipEventCreate(e1) != hipSuccess) return;
  c.start.push_back(*e0);
  c.stop.push_back(*e1);
  c.flops += flops;
  c.bytes += bytes;
}
}  // namespace tt

extern "C" {
int tt_prof_enable(int on) {
  using namespace tt;
  if (on) {
    for (int i = 0; i < PROF_COUNT; ++i) {
      for (auto e : g_prof[i].start) (void)hipEventDestroy(e);
      for (auto e : g_prof[i].stop) (void)hipEventDestroy(e);
      g_prof[i] = ProfClass();
    }
  }
  g_prof_on = on != 0;
  return 0;
}
int tt_graph_replay(int on) {
  const int prev = tt::g_graph_replay ? 1 : 0;
  tt::g_graph_replay = on != 0;
  return prev;
}
int tt_prof_classes(void) { return tt::PROF_COUNT; }
const char* tt_prof_class_name(int id) { return tt::prof_name(id); }
// Synchronises the device; out[0..3] = {launches, total_ms, algorithmic_flops, algorithmic_bytes} of class `id`.
int tt_prof_read(int id, double* out) {
  using namespace tt;
  if (id < 0 || id >= PROF_COUNT || !out) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  ProfClass& c = g_prof[id];
  double ms = 0;
  const size_t n = c.start.size() < c.stop.size() ? c.start.size() : c.stop.size();
  for (size_t i = 0; i < n; ++i) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, c.start[i], c.stop[i]) == hipSuccess) ms += t;
  }
  out[0] = (double)n; out[1] = ms; out[2] = c.flops; out[3] = c.bytes;
  return 0;
}
}
