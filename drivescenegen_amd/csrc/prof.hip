// Optional per-kernel-class timing with HIP events on the launch stream (bench.py's roofline leg).
// Off by default; when on, every convolution launch is bracketed by two events and tagged with its
// algorithmic FLOPs / bytes.  Not part of the reference's surface: measurement plumbing only.
#include "dsg_common.h"

#include <mutex>
#include <vector>

namespace dsg {

struct ProfRec {
  hipEvent_t a, b;
  double flops, bytes;
  int kid;
};

static std::mutex g_mu;
static bool g_on = false;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;

bool prof_on() { return g_on; }

static hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

// returns an index to pass to prof_end, or -1
int prof_begin(int kid, double flops, double bytes, hipStream_t st) {
  if (!g_on) return -1;
  std::lock_guard<std::mutex> lk(g_mu);
  ProfRec r{get_event(), get_event(), flops, bytes, kid};
  if (!r.a || !r.b) return -1;
  (void)hipEventRecord(r.a, st);
  g_recs.push_back(r);
  return (int)g_recs.size() - 1;
}

void prof_end(int idx, hipStream_t st) {
  if (idx < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  (void)hipEventRecord(g_recs[idx].b, st);
}

}  // namespace dsg

// Kernel classes: 0 conv3x3 stride-1 (plain or [x||skip] gather), 1 conv3x3 on nearest-x2 upsampled
// input, 2 conv3x3 stride-2, 3 conv1x1, 4 direct (VALU) conv, 11 conv_in.hip, 12 conv_out.hip (either mode).
// on: 1 start (drops earlier records), 0 stop (drops records); 2 pause, 3 resume -- keep the records (bench.py
// brackets only every n-th step of its timed region, so that the event records cost it next to nothing)
DSG_API int dsg_prof_enable(int32_t on) {
  std::lock_guard<std::mutex> lk(dsg::g_mu);
  if (on == 2 || on == 3) {
    dsg::g_on = on == 3;
    return DSG_OK;
  }
  for (auto& r : dsg::g_recs) {
    dsg::g_pool.push_back(r.a);
    dsg::g_pool.push_back(r.b);
  }
  dsg::g_recs.clear();
  dsg::g_on = on != 0;
  return DSG_OK;
}

// Sums over the launches of one kernel class recorded since dsg_prof_enable(1).  Waits for the events.
DSG_API int dsg_prof_summary(int32_t kid, double* total_ms, double* total_flops, double* total_bytes,
                             int64_t* launches) {
  std::lock_guard<std::mutex> lk(dsg::g_mu);
  double ms = 0, fl = 0, by = 0;
  int64_t n = 0;
  for (auto& r : dsg::g_recs) {
    if (r.kid != kid) continue;
    DSG_HIP(hipEventSynchronize(r.b));
    float t = 0.f;
    DSG_HIP(hipEventElapsedTime(&t, r.a, r.b));
    ms += t;
    fl += r.flops;
    by += r.bytes;
    ++n;
  }
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (total_bytes) *total_bytes = by;
  if (launches) *launches = n;
  return DSG_OK;
}

// Per-launch records (class, algorithmic flops, bytes, ms) as CSV -- for profiles/ and layer-level analysis.
DSG_API int dsg_prof_dump(const char* path) {
  DSG_CHECK_ARG(path != nullptr, "dsg_prof_dump: path is NULL");
  std::lock_guard<std::mutex> lk(dsg::g_mu);
  FILE* f = fopen(path, "w");
  if (!f) return dsg::fail(DSG_ERR_INVALID_ARG, "dsg_prof_dump: cannot open %s", path);
  fprintf(f, "index,kernel_class,alg_flops,alg_bytes,ms,tflops\n");
  int i = 0;
  for (auto& r : dsg::g_recs) {
    float t = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess)
      fprintf(f, "%d,%d,%.0f,%.0f,%.6f,%.3f\n", i, r.kid, r.flops, r.bytes, t, r.flops / (t * 1e-3) / 1e12);
    ++i;
  }
  fclose(f);
  return DSG_OK;
}
