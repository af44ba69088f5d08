// Translation unit of libzkstark_hip.so: the PLONK prover of the recursion layer (plonk.cuh, plonk_host.inc).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <chrono>
#include <cstring>
#include <memory>
#include <vector>

#include "internal.hpp"
#include "merkle.cuh"
#include "fri.cuh"
#include "stark.cuh"
#include "quotient.cuh"   // split_quotient_chunks_kernel
#include "plonk.cuh"

#include "plonk_host.inc"
