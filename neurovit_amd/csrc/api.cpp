// C-ABI housekeeping: error string, version / architecture probes.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void nv_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* nv_last_error(void) { return g_err; }

extern "C" int nv_version(void) { return 1; }

#include "../../include/neurovit_hip.h"
extern "C" int nv_abi_version(void) { return NV_ABI_VERSION; }   // see INTEGRATION.md "ABI revisions"

// ---- 16-bit operand format of this process: what every `void*` "16-bit" buffer of the C-ABI holds and which MFMA the contractions
// issue (common.h: kernels are templates over the element type; every launcher reads this switch).  bf16 is the default and what
// BASELINE.json's metric names; fp16 is the reference's own training arithmetic (autocast(float16) + GradScaler, src/Trainer.py:29,68,
// 74-76): 3 more mantissa bits (logits within 1e-3 of the fp32 CPU forward) at the same MFMA rate, 5 instead of 8 exponent bits
// (the train step then scales the loss: nv_loss_scale_*).  A caller that mixes models of both formats sets it before each call.
static int g_operand_format = NV_OPERAND_BF16;
extern "C" int nv_operand_format(void) { return g_operand_format; }
extern "C" int nv_set_operand_format(int fmt) {
  if (fmt != NV_OPERAND_BF16 && fmt != NV_OPERAND_FP16) { nv_set_error("nv_set_operand_format: %d is neither NV_OPERAND_BF16 (0) nor NV_OPERAND_FP16 (1)", fmt); return NV_ERR_ARG; }
  g_operand_format = fmt;
  return NV_OK;
}

// 1 when the current HIP device is gfx950 (MI355X), 0 otherwise, negative on HIP error.
extern "C" int nv_arch_ok(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { nv_set_error("nv_arch_ok: no HIP device"); return -2; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { nv_set_error("nv_arch_ok: hipGetDeviceProperties failed"); return -2; }
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

// ---- optional per-launch event profiler (bench.py's roofline leg) -------------------------------------
// When enabled, kernel launchers bracket each launch of a profiled kind with hipEvents on the launch stream.
// Kinds: 0 gemm NT, 1 gemm NN, 2 gemm TN, 3 attention fwd, 4 attention bwd (dQ + dK/dV).
#include <stdlib.h>
#include <vector>
namespace {
struct Rec { hipEvent_t a, b; int kind; double work, bytes; };
std::vector<Rec> g_pool;
size_t g_used = 0;
bool g_on = false;
}  // namespace

extern "C" int nv_prof_enable(int on) {
  g_on = on != 0;
  g_used = 0;
  return 0;
}

extern "C" int nv_prof_begin(int kind, double work, void* stream) {
  if (!g_on) return -1;
  if (g_used == g_pool.size()) {
    if (g_pool.size() >= 65536) return -1;
    Rec r; r.kind = kind; r.work = work;
    // timing-only events: without the system-scope fence a default event carries, which would both lengthen the measured interval
    // and disturb the work that follows (hip_runtime_api.h, hipEventDisableSystemFence)
    if (hipEventCreateWithFlags(&r.a, hipEventDisableSystemFence) != hipSuccess || hipEventCreateWithFlags(&r.b, hipEventDisableSystemFence) != hipSuccess) return -1;
    g_pool.push_back(r);
  }
  Rec& r = g_pool[g_used];
  r.kind = kind; r.work = work; r.bytes = 0.0;
  (void)hipEventRecord(r.a, (hipStream_t)stream);
  return (int)g_used++;
}

// algorithmic bytes of the launch in `slot` (operands read once + outputs written once): the HBM-side check of the roofline leg
extern "C" void nv_prof_bytes(int slot, double bytes) {
  if (slot >= 0 && (size_t)slot < g_pool.size()) g_pool[slot].bytes = bytes;
}

extern "C" void nv_prof_end(int slot, void* stream) {
  if (slot >= 0 && (size_t)slot < g_pool.size()) (void)hipEventRecord(g_pool[slot].b, (hipStream_t)stream);
}

// Sums over the records of `kind` since nv_prof_enable(1): total milliseconds, total work (flops), launch count.
// Synchronises with the recorded events (call it outside any timed region).
extern "C" int nv_prof_summary(int kind, double* ms, double* work, long* count) {
  double tms = 0, tw = 0; long c = 0;
  for (size_t i = 0; i < g_used; ++i) {
    Rec& r = g_pool[i];
    if (r.kind != kind) continue;
    if (hipEventSynchronize(r.b) != hipSuccess) { nv_set_error("nv_prof_summary: event sync failed"); return -2; }
    float e = 0.f;
    if (hipEventElapsedTime(&e, r.a, r.b) != hipSuccess) { nv_set_error("nv_prof_summary: elapsed failed"); return -2; }
    tms += e; tw += r.work; ++c;
  }
  if (ms) *ms = tms; if (work) *work = tw; if (count) *count = c;
  return 0;
}

// total algorithmic bytes of the records of `kind` (see nv_prof_bytes); 0 for kinds whose launchers do not state them
extern "C" int nv_prof_summary_bytes(int kind, double* bytes) {
  double tb = 0;
  for (size_t i = 0; i < g_used; ++i)
    if (g_pool[i].kind == kind) tb += g_pool[i].bytes;
  if (bytes) *bytes = tb;
  return 0;
}

// ---- pooled events for fork / join between two streams (created once, timing disabled) -----------------------
namespace {
std::vector<hipEvent_t> g_sync_events;
size_t g_sync_next = 0;
}  // namespace

// Make stream `to` wait for everything enqueued so far on stream `from`.  Returns 0 or NV_ERR_HIP (-2).
// Events that only order streams of this device against each other (the host never waits on them): no timing and no
// system-scope fence.  A default event's record costs 6-10 us of queue time on MI355X (four extra records per layer lengthen
// the 4.27 ms step by 0.3-0.5 ms); without the system fence the step is 2 % shorter.  Kernel boundaries keep their own
// device-scope release / acquire, which is what stream-to-stream ordering on one device needs.
unsigned nv_sync_event_flags() { return hipEventDisableTiming | hipEventDisableSystemFence; }

extern "C" int nv_stream_sync(void* from, void* to) {
  hipEvent_t e;
  if (g_sync_events.size() < 64) {
    if (hipEventCreateWithFlags(&e, nv_sync_event_flags()) != hipSuccess) { nv_set_error("nv_stream_sync: hipEventCreate failed"); return -2; }
    g_sync_events.push_back(e);
  } else {
    e = g_sync_events[g_sync_next++ % g_sync_events.size()];
  }
  if (hipEventRecord(e, (hipStream_t)from) != hipSuccess || hipStreamWaitEvent((hipStream_t)to, e, 0) != hipSuccess) {
    nv_set_error("nv_stream_sync: record / wait failed");
    return -2;
  }
  return 0;
}

// ---- a kernel that only occupies its stream for a given time (one wave): the stream-placement probes of the data-parallel
// start-up (neurovit_amd/parallel.py) need "work that is still running" on one stream while they watch another.  Timed on the
// constant 100 MHz s_memrealtime counter, so the duration does not depend on the shader clock; bounded at 50 ms.
__global__ void nv_spin_kernel(unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

extern "C" int nv_spin_us(int microseconds, void* stream) {
  if (microseconds < 0 || microseconds > 50000) { nv_set_error("nv_spin_us: 0 .. 50000 us"); return -1; }
  hipLaunchKernelGGL(nv_spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)microseconds * 100ull);
  if (hipGetLastError() != hipSuccess) { nv_set_error("nv_spin_us: launch failed"); return -2; }
  return 0;
}

// ---- placement census (diagnostic; tools/cu_mask_probe.py, tests of the XCD-aware grids): every workgroup of a `blocks`-wide grid
// records where it ran - out[2 b] = HW_REG_HW_ID (wave / simd / cu / sh / se fields), out[2 b + 1] = HW_REG_XCC_ID - and then holds its
// CU for `hold_us` so that the grid spreads over the CUs its stream may use (a CU-masked stream: hipExtStreamCreateWithCUMask).
__global__ void nv_census_kernel(unsigned* out, unsigned long long ticks) {
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID, 32 bits
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

extern "C" int nv_cu_census(unsigned* out, int blocks, int threads, int lds_bytes, int hold_us, void* stream) {
  if (!out || blocks < 1 || threads < 64 || threads > 1024 || lds_bytes < 0 || lds_bytes > 160 * 1024 || hold_us < 0 || hold_us > 50000) {
    nv_set_error("nv_cu_census: bad arguments");
    return -1;
  }
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nv_census_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(nv_census_kernel, dim3(blocks), dim3(threads), lds_bytes, (hipStream_t)stream, out, (unsigned long long)hold_us * 100ull);
  if (hipGetLastError() != hipSuccess) { nv_set_error("nv_cu_census: launch failed"); return -2; }
  return 0;
}
