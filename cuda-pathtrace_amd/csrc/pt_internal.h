// pt_internal.h -- error plumbing shared by the host-side translation units of libptcore.
#pragma once
#include "../../include/ptcore.h"

// Records `msg` (printf-style) as the calling thread's last error and returns `code`.
int pt_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
