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

// Non-zero when an object of this library was compiled with a measurement / ablation switch (scripts/abl.sh,
// scripts/nms_debug.sh rebuild single objects with EXTRA=-D...): such a build may skip work or print diagnostics and
// must never produce a reported number.  s2anet_amd/_lib.py refuses to load it unless S2A_ALLOW_MEASURE_BUILD=1.
namespace s2a {
int build_flags_dcn();
int build_flags_rotated();
int build_flags_dcn_bwd();
int build_flags_wino();
}  // namespace s2a
extern "C" int s2a_build_flags(void) {
  return s2a::build_flags_dcn() | (s2a::build_flags_rotated() << 16) | (s2a::build_flags_dcn_bwd() << 24) | (s2a::build_flags_wino() << 30);
}
