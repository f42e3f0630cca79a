#include "common.hpp"

#include <cstring>

namespace s2a {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }
}  // namespace s2a

extern "C" const char* s2a_last_error(void) { return s2a::get_error(); }
extern "C" const char* s2a_version(void) { return "s2anet_hip 0.1 (gfx950)"; }
