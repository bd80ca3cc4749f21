// vt_host.h — host-side error plumbing shared by the drivers (thread-local last-error string).
#pragma once
#include <stdarg.h>
#include <stdio.h>

inline char* vt_errbuf() { static thread_local char buf[512] = "ok"; return buf; }
inline int vt_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(vt_errbuf(), 512, fmt, ap);
  va_end(ap);
  return code;
}
// annotate a failing launch code from a lower layer
inline int vt_wrap(int code, const char* what) {
  if (code) {
    char tmp[400];
    snprintf(tmp, sizeof(tmp), "%s", vt_errbuf());
    snprintf(vt_errbuf(), 512, "%s failed (%d): %s", what, code, tmp);
  }
  return code;
}
