// Translation unit of libzkstark_hip.so: the witness-table generators (tracegen.cuh, memtrace.cuh,
// arithtrace.cuh and their host sides).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <vector>

#include <rocprim/rocprim.hpp>   // radix sort + scans of the Memory-table generator (memtrace_host.inc)

#include "internal.hpp"
#include "merkle.cuh"
#include "tracegen.cuh"
#include "memtrace.cuh"
#include "arithtrace.cuh"

#include "tracegen_host.inc"
#include "memtrace_host.inc"
