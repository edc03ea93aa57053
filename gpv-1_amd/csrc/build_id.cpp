// gpv_build_id: the sha256 prefix of the sources this library was built from (Makefile: BUILD_ID), so that a caller can tell a
// stale libgpv_hip.so from one compiled from the tree next to it (__graft_entry__.build() does).
#include "build_id.h"
extern "C" int gpv_build_id(char* buf, int cap) {
  const char* id = GPV_BUILD_ID;
  int n = 0;
  while (id[n]) ++n;
  if (!buf || cap <= n) return 1;          // hipErrorInvalidValue
  for (int i = 0; i <= n; ++i) buf[i] = id[i];
  return 0;
}
