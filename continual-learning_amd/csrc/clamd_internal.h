// Internal glue shared by the translation units of libclamd.so (error reporting, dtype codes).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/clamd.h"

// Records a message retrievable through clamd_last_error() and returns a negative status.
int clamd_fail(const char* msg);
// hipGetLastError() after a launch -> 0 or a negative status with the HIP error string recorded.
int clamd_check_launch(const char* what);
