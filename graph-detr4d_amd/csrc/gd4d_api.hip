// libgd4d.so: ABI version, error strings, per-thread last-HIP-error text.
#include "gd4d_common.h"

namespace gd4d {
static thread_local hipError_t g_last = hipSuccess;
void set_last_hip_error(hipError_t e) { g_last = e; }
}  // namespace gd4d

extern "C" int gd4d_abi_version(void) { return GD4D_ABI_VERSION; }

extern "C" const char* gd4d_error_string(int code) {
  switch (code) {
    case GD4D_OK: return "ok";
    case GD4D_EINVAL: return "invalid argument (NULL pointer or non-positive size)";
    case GD4D_EUNSUPPORTED: return "shape/dtype not supported by the gfx950 kernels";
    case GD4D_EALIGN: return "pointer not 16-byte aligned";
    case GD4D_ELAUNCH: return "HIP kernel launch failed (see gd4d_last_hip_error)";
    case GD4D_EWORKSPACE: return "workspace too small";
    default: return "unknown gd4d error";
  }
}

extern "C" const char* gd4d_last_hip_error(void) {
  return gd4d::g_last == hipSuccess ? "" : hipGetErrorString(gd4d::g_last);
}

// dev: device-side timeline of the kernels of a (replayed) step - see gd4d_common.h / tools/trace_step.py
extern "C" void gd4d_trace_set_rowchain(unsigned long long*);
extern "C" void gd4d_trace_set_mha(unsigned long long*);
extern "C" void gd4d_trace_set_late(unsigned long long*);
extern "C" void gd4d_trace_set_sliced(unsigned long long*);
extern "C" int gd4d_trace_enable(void* buffer) {
  unsigned long long* p = static_cast<unsigned long long*>(buffer);
  gd4d_trace_set_rowchain(p);
  gd4d_trace_set_mha(p);
  gd4d_trace_set_late(p);
  gd4d_trace_set_sliced(p);
  return GD4D_OK;
}
