// Error text, view validation, and the plan (record -> eager replay / hipGraph replay).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "common.h"

namespace glsdet {

static thread_local char g_err[512] = "";
static thread_local glsdet_plan* g_recording = nullptr;
static thread_local int g_branch = 0;

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

int check_view(const glsdet_view& v, const char* what, bool need16) {
  if (!v.base) GLS_FAIL(GLSDET_E_ARG, "%s: null base", what);
  if (v.dtype != GLSDET_F16 && v.dtype != GLSDET_F32) GLS_FAIL(GLSDET_E_ARG, "%s: bad dtype %d", what, v.dtype);
  if (v.n < 1 || v.h < 1 || v.w < 1 || v.c < 1) GLS_FAIL(GLSDET_E_ARG, "%s: empty extent [%d,%d,%d,%d]", what, v.n, v.h, v.w, v.c);
  if (v.sn < 0 || v.sh < 0 || v.sw < 0) GLS_FAIL(GLSDET_E_ARG, "%s: negative stride", what);
  const int es = dtype_size(v.dtype);
  if (need16) {
    const int vec = 16 / es;
    if (((uintptr_t)v.base & 15) || (v.sn % vec) || (v.sh % vec) || (v.sw % vec))
      GLS_FAIL(GLSDET_E_ALIGN, "%s: base/strides not 16-byte compatible", what);
  }
  if (v.sw < v.c && v.w > 1) GLS_FAIL(GLSDET_E_ARG, "%s: pixel stride %ld < channels %d", what, (long)v.sw, v.c);
  const char* lo = (const char*)v.base;
  const char* hi = lo + ((int64_t)(v.n - 1) * v.sn + (int64_t)(v.h - 1) * v.sh + (int64_t)(v.w - 1) * v.sw + v.c) * es;
  if (!v.alloc_lo || !v.alloc_hi || lo < (const char*)v.alloc_lo || hi > (const char*)v.alloc_hi)
    GLS_FAIL(GLSDET_E_BOUNDS, "%s: view [%p,%p) leaves its allocation [%p,%p)", what, (const void*)lo,
             (const void*)hi, v.alloc_lo, v.alloc_hi);
  return 0;
}

}  // namespace glsdet

#define GLS_MAX_BRANCH 8
struct glsdet_plan {
  std::vector<glsdet::OpRecord> ops;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  hipStream_t side[GLS_MAX_BRANCH + 1] = {};
  hipEvent_t ev_fork[GLS_MAX_BRANCH + 1] = {}, ev_join[GLS_MAX_BRANCH + 1] = {};
};

namespace glsdet {
// Replay the op list on `st`; ops of side branches go to the plan's side streams, forked from
// and joined back into `st` with events (works the same eagerly and under stream capture,
// where it produces parallel graph branches).
static int replay(glsdet_plan* p, hipStream_t st, bool use_branches) {
  bool active[GLS_MAX_BRANCH + 1] = {};
  auto join_all = [&]() -> int {
    for (int b = 1; b <= GLS_MAX_BRANCH; ++b)
      if (active[b]) {
        GLS_HIP(hipEventRecord(p->ev_join[b], p->side[b]));
        GLS_HIP(hipStreamWaitEvent(st, p->ev_join[b], 0));
        active[b] = false;
      }
    return 0;
  };
  for (auto& o : p->ops) {
    int rc;
    const int b = use_branches ? o.branch : 0;
    if (b <= 0 || b > GLS_MAX_BRANCH) {
      if ((rc = join_all())) return rc;
      if ((rc = o.launch(st))) return rc;
      continue;
    }
    if (!p->side[b]) {
      GLS_HIP(hipStreamCreateWithFlags(&p->side[b], hipStreamNonBlocking));
      GLS_HIP(hipEventCreateWithFlags(&p->ev_fork[b], hipEventDisableTiming));
      GLS_HIP(hipEventCreateWithFlags(&p->ev_join[b], hipEventDisableTiming));
    }
    if (!active[b]) {
      GLS_HIP(hipEventRecord(p->ev_fork[b], st));
      GLS_HIP(hipStreamWaitEvent(p->side[b], p->ev_fork[b], 0));
      active[b] = true;
    }
    if ((rc = o.launch(p->side[b]))) return rc;
  }
  return join_all();
}
}  // namespace glsdet

namespace glsdet {
int submit(OpRecord&& op, void* stream) {
  if (g_recording) {
    // timing experiments (tools/probe/knockout.sh): ops whose name contains one of the comma-separated tokens of
    // GLSDET_SKIP_OPS are recorded as no-ops -- the results of such a plan are garbage, only its wall clock is of interest
    if (const char* skip = getenv("GLSDET_SKIP_OPS")) {
      std::string all(skip);
      size_t pos = 0;
      while (pos <= all.size()) {
        const size_t e = all.find(',', pos);
        const std::string tok = all.substr(pos, e == std::string::npos ? std::string::npos : e - pos);
        if (!tok.empty() && op.name.find(tok) != std::string::npos) {
          op.launch = [](hipStream_t) -> int { return 0; };
          op.name = "SKIPPED " + op.name;
          break;
        }
        if (e == std::string::npos) break;
        pos = e + 1;
      }
    }
    op.branch = g_branch;
    g_recording->ops.emplace_back(std::move(op));
    return 0;
  }
  return op.launch((hipStream_t)stream);
}
}  // namespace glsdet

using namespace glsdet;

extern "C" const char* glsdet_last_error(void) { return g_err; }
extern "C" int32_t glsdet_abi_version(void) { return GLSDET_ABI_VERSION; }

extern "C" glsdet_plan* glsdet_plan_create(void) { return new glsdet_plan(); }
extern "C" void glsdet_plan_destroy(glsdet_plan* p) {
  if (!p) return;
  if (g_recording == p) g_recording = nullptr;
  if (p->exec) (void)hipGraphExecDestroy(p->exec);
  if (p->graph) (void)hipGraphDestroy(p->graph);
  for (int b = 0; b <= GLS_MAX_BRANCH; ++b) {
    if (p->ev_fork[b]) (void)hipEventDestroy(p->ev_fork[b]);
    if (p->ev_join[b]) (void)hipEventDestroy(p->ev_join[b]);
    if (p->side[b]) (void)hipStreamDestroy(p->side[b]);
  }
  delete p;
}
extern "C" int glsdet_plan_begin(glsdet_plan* p) {
  if (!p) GLS_FAIL(GLSDET_E_ARG, "plan_begin: null plan");
  if (g_recording) GLS_FAIL(GLSDET_E_ARG, "plan_begin: this thread is already recording");
  g_recording = p;
  return 0;
}
extern "C" int glsdet_plan_end(glsdet_plan* p) {
  if (g_recording != p) GLS_FAIL(GLSDET_E_ARG, "plan_end: plan is not the one being recorded");
  g_recording = nullptr;
  g_branch = 0;
  return 0;
}
extern "C" int glsdet_plan_set_branch(int32_t branch) {
  if (branch < 0 || branch > GLS_MAX_BRANCH) GLS_FAIL(GLSDET_E_ARG, "plan_set_branch: branch %d not in [0,%d]", branch, GLS_MAX_BRANCH);
  if (g_recording && branch == 0 && g_branch != 0) {
    // leaving a fork region: an explicit join point, so that a following region's branches
    // start only after ALL branches of this one (there may be no main-sequence op in between)
    OpRecord j;
    j.kind = -1;
    j.flops = j.bytes = 0;
    j.name = "join";
    j.branch = 0;
    j.launch = [](hipStream_t) -> int { return 0; };
    g_recording->ops.emplace_back(std::move(j));
  }
  g_branch = branch;
  return 0;
}
extern "C" int32_t glsdet_plan_num_ops(const glsdet_plan* p) { return p ? (int32_t)p->ops.size() : 0; }
extern "C" int glsdet_plan_op_info(const glsdet_plan* p, int32_t i, int32_t* kind, double* flops, double* bytes,
                                   char* name, int32_t name_cap) {
  if (!p || i < 0 || i >= (int32_t)p->ops.size()) GLS_FAIL(GLSDET_E_ARG, "plan_op_info: bad index");
  const OpRecord& o = p->ops[i];
  if (kind) *kind = o.kind;
  if (flops) *flops = o.flops;
  if (bytes) *bytes = o.bytes;
  if (name && name_cap > 0) {
    strncpy(name, o.name.c_str(), name_cap - 1);
    name[name_cap - 1] = 0;
  }
  return 0;
}
extern "C" int glsdet_plan_run(glsdet_plan* p, void* stream) {
  if (!p) GLS_FAIL(GLSDET_E_ARG, "plan_run: null plan");
  return replay(p, (hipStream_t)stream, true);
}
extern "C" int glsdet_plan_capture(glsdet_plan* p, void* stream) {
  if (!p) GLS_FAIL(GLSDET_E_ARG, "plan_capture: null plan");
  hipStream_t st = (hipStream_t)stream;
  if (!st) GLS_FAIL(GLSDET_E_ARG, "plan_capture: needs a non-default stream");
  if (p->exec) { (void)hipGraphExecDestroy(p->exec); p->exec = nullptr; }
  if (p->graph) { (void)hipGraphDestroy(p->graph); p->graph = nullptr; }
  GLS_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  int rc = replay(p, st, true);
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(st, &g);
  if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
  if (e != hipSuccess) GLS_FAIL(GLSDET_E_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
  p->graph = g;
  GLS_HIP(hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0));
  return 0;
}
extern "C" int glsdet_plan_launch(glsdet_plan* p, void* stream) {
  if (!p || !p->exec) GLS_FAIL(GLSDET_E_ARG, "plan_launch: plan not captured");
  GLS_HIP(hipGraphLaunch(p->exec, (hipStream_t)stream));
  return 0;
}
extern "C" int glsdet_plan_run_timed(glsdet_plan* p, void* stream, float* ms) {
  if (!p || !ms) GLS_FAIL(GLSDET_E_ARG, "plan_run_timed: null argument");
  hipStream_t st = (hipStream_t)stream;
  const size_t n = p->ops.size();
  std::vector<hipEvent_t> ev(n + 1);
  for (auto& e : ev) GLS_HIP(hipEventCreate(&e));
  int rc = 0;
  GLS_HIP(hipEventRecord(ev[0], st));
  for (size_t i = 0; i < n && !rc; ++i) {
    rc = p->ops[i].launch(st);
    if (!rc && hipEventRecord(ev[i + 1], st) != hipSuccess) rc = GLSDET_E_HIP;
  }
  if (!rc && hipEventSynchronize(ev[n]) != hipSuccess) rc = GLSDET_E_HIP;
  for (size_t i = 0; i < n && !rc; ++i) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, ev[i], ev[i + 1]) != hipSuccess) { rc = GLSDET_E_HIP; break; }
    ms[i] += t;
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  if (rc == GLSDET_E_HIP) set_error("plan_run_timed: HIP event error");
  return rc;
}
