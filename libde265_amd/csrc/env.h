// env.h -- the library's tuning, experiment and diagnostic switches (DE265HIP_* environment variables: A/B forms of kernels and
// schedules behind the parity tests, stream / grid / thread counts of the pipeline, timing print-outs).
//
// They are honoured ONLY in a process that sets DE265HIP_TUNING=1 (the tests, bench.py and the tools do).  In any other process
// every switch reads as unset: a variable left behind in a service's environment must not be able to change what a decoder
// does - some of the switches make its results invalid (DE265HIP_PIPE_NO_RUN, DE265HIP_OUT_COPY=none, DE265HIP_DEBUG).
#pragma once
#include <cstdlib>

inline const char* d265_env(const char* name)
{
  static const bool on = [] { const char* e = getenv("DE265HIP_TUNING"); return e && atoi(e) != 0; }();
  return on ? getenv(name) : nullptr;
}
