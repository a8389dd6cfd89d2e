// HipErrorCheck.h -- look-alike of include/CudaErrorCheck.h:6-14.
// gpuErrchk(x) takes the int status of a ptcore call; on failure it prints
// "GPUassert: <message> <file> <line>" to stderr and exits with the code, exactly the
// reference's behaviour (no error codes are surfaced to the caller).
#ifndef HIP_ERROR_CHECK_H
#define HIP_ERROR_CHECK_H
#include <stdio.h>
#include <stdlib.h>

#include "../../include/ptcore.h"

#define gpuErrchk(ans) \
  { gpuAssert((ans), __FILE__, __LINE__); }
inline void gpuAssert(int code, const char* file, int line, bool abort = true) {
  if (code != PT_OK) {
    fprintf(stderr, "GPUassert: %s %s %d\n", pt_last_error(), file, line);
    if (abort) exit(code);
  }
}
#endif
