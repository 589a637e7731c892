// Why an entry point refused its arguments: the C-ABI returns hipErrorInvalidValue, the reason is kept per host thread.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace icrl {

static thread_local char g_last_error[512] = "";

int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
  return (int)hipErrorInvalidValue;
}

}  // namespace icrl

extern "C" const char* icrl_last_error(void) { return icrl::g_last_error; }
extern "C" void icrl_clear_error(void) { icrl::g_last_error[0] = 0; }
