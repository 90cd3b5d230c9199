// runtime.hip — library plumbing of libgcd_amd: error string, device probe, hipGraph capture /
// replay of a launch sequence, and HIP events on the caller's stream.
#include "common.h"

#include <stdlib.h>
#include <string.h>

#include "gemm_common.h"

static thread_local char g_err[1024] = "";

void gcd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* gcd_last_error(void) { return g_err; }
extern "C" int gcd_abi_version(void) { return GCD_AMD_ABI_VERSION; }

extern "C" int gcd_device_info(int device, char* name, int cap, int* num_cus, size_t* hbm_bytes) {
  hipDeviceProp_t prop;
  GCD_CHECK_HIP(hipGetDeviceProperties(&prop, device));
  if (name && cap > 0) {
    strncpy(name, prop.gcnArchName, cap - 1);
    name[cap - 1] = 0;
  }
  if (num_cus) *num_cus = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
  return 0;
}

// ---- tuning knobs --------------------------------------------------------------------------------
static int g_tune[GCD_TUNE_COUNT];
static bool g_tune_init = false;
static void tune_init() {
  if (g_tune_init) return;
  g_tune_init = true;
  static const char* env_names[GCD_TUNE_COUNT] = {"GCD_GEMM_IMPL", "GCD_ATTN_IMPL", "GCD_PP_MIN_TILES", "GCD_STREAM"};
  for (int i = 0; i < GCD_TUNE_COUNT; ++i) {
    const char* e = getenv(env_names[i]);
    g_tune[i] = e ? atoi(e) : 0;
  }
}
int gcd_tune_get(int knob) {
  tune_init();
  return (knob >= 0 && knob < GCD_TUNE_COUNT) ? g_tune[knob] : 0;
}
extern "C" int gcd_tune_set(int knob, int value) {
  tune_init();
  GCD_CHECK_ARG(knob >= 0 && knob < GCD_TUNE_COUNT, "gcd_tune_set: unknown knob %d", knob);
  g_tune[knob] = value;
  return 0;
}

// ---- hipGraph capture of a launch sequence ------------------------------------------------------
extern "C" int gcd_graph_begin_capture(void* stream) {
  GCD_CHECK_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return 0;
}

extern "C" int gcd_graph_end_capture(void* stream, void** graph_exec_out) {
  GCD_CHECK_ARG(graph_exec_out, "gcd_graph_end_capture: null output");
  hipGraph_t graph = nullptr;
  GCD_CHECK_HIP(hipStreamEndCapture((hipStream_t)stream, &graph));
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) {
    gcd_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
    return 1;
  }
  *graph_exec_out = (void*)exec;
  return 0;
}

extern "C" int gcd_graph_launch(void* graph_exec, void* stream) {
  GCD_CHECK_ARG(graph_exec, "gcd_graph_launch: null graph");
  GCD_CHECK_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return 0;
}

extern "C" int gcd_graph_destroy(void* graph_exec) {
  if (graph_exec) GCD_CHECK_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return 0;
}

// ---- events -------------------------------------------------------------------------------------
extern "C" int gcd_event_create(void** ev) {
  GCD_CHECK_ARG(ev, "gcd_event_create: null output");
  hipEvent_t e;
  GCD_CHECK_HIP(hipEventCreate(&e));
  *ev = (void*)e;
  return 0;
}
extern "C" int gcd_event_record(void* ev, void* stream) {
  GCD_CHECK_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
  return 0;
}
extern "C" int gcd_event_sync(void* ev) {
  GCD_CHECK_HIP(hipEventSynchronize((hipEvent_t)ev));
  return 0;
}
extern "C" int gcd_event_elapsed_ms(void* start, void* stop, float* ms) {
  GCD_CHECK_ARG(ms, "gcd_event_elapsed_ms: null output");
  GCD_CHECK_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return 0;
}
extern "C" int gcd_event_destroy(void* ev) {
  if (ev) GCD_CHECK_HIP(hipEventDestroy((hipEvent_t)ev));
  return 0;
}
extern "C" int gcd_stream_sync(void* stream) {
  GCD_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}
