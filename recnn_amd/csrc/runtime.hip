// runtime.hip -- error plumbing and ABI version of librecnn_hip.
#include <stdarg.h>
#include "common.h"

static thread_local char g_err[512] = "";

void recnn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int recnn_check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  recnn_set_error("%s: %s", what, hipGetErrorString(e));
  return (int)e;
}

extern "C" const char* recnn_last_error(void) { return g_err; }
extern "C" int recnn_abi_version(void) { return RECNN_ABI_VERSION; }

// sizeof() of the ABI structs, so that a binding can verify its struct declarations.
extern "C" int64_t recnn_abi_sizeof(int which) {
  switch (which) {
    case 0: return (int64_t)sizeof(recnn_gemm_args);
    case 1: return (int64_t)sizeof(recnn_engine_config);
    case 2: return (int64_t)sizeof(recnn_hyper);
    case 3: return (int64_t)sizeof(recnn_engine_sizes);
    case 4: return (int64_t)sizeof(recnn_sampler);
    case 5: return (int64_t)sizeof(recnn_engine_tuning);
    case 6: return (int64_t)sizeof(recnn_shadow_out);
    default: return -1;
  }
}
