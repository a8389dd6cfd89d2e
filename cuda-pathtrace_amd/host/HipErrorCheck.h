// HipErrorCheck.h -- status checking for the look-alike classes.
//
// The reference wraps every CUDA runtime call in gpuErrchk(...) (include/CudaErrorCheck.h:6-14):
// on failure it writes "GPUassert: <message> <file> <line>" to stderr and terminates the
// process with the error code; no status ever reaches the caller.  The ptcore C ABI returns
// an int status and keeps the message in pt_last_error(); this header turns that back into the
// reference's print-and-exit behaviour so code written against Renderer/OutputBuffer/Scene
// keeps its error semantics.
#ifndef PT_HOST_ERROR_CHECK_H
#define PT_HOST_ERROR_CHECK_H
#include <cstdio>
#include <cstdlib>

#include "../../include/ptcore.h"

namespace pthost {
// Returns normally only when status == PT_OK (or when terminate is false).
inline void require_ok(int status, const char* where_file, int where_line, bool terminate = true) {
  if (status == PT_OK) return;
  std::fprintf(stderr, "GPUassert: %s %s %d\n", pt_last_error(), where_file, where_line);
  if (terminate) std::exit(status);
}
}  // namespace pthost

// same spelling as the reference so call sites read alike
#define gpuErrchk(status_expr) ::pthost::require_ok((status_expr), __FILE__, __LINE__)
#endif
