// Test scaffolding only: the two symbols group_host.cpp expects from the rest of libposehip, so that the host-side grouping code can
// be compiled ALONE by g++ with -fsanitize=address,undefined (tests/test_host_cpu.py::test_host_grouping_under_sanitizers).
#include <cstdarg>
#include <cstdio>

namespace {
char g_err[512] = "";
}

namespace ph {
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace ph

extern "C" const char* ph_last_error() { return g_err; }
