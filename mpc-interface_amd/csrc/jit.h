// jit.h -- the persistent kernel specialised per plan by hiprtc (jit.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "plan_dev.h"

namespace mpcasm {

extern int g_jit;         // MPCASM_OPT_JIT
extern int g_phase_mask;  // fused.hip

// "plan_spec.h" of a plan: its sizes and trip lists as constants
std::string jit_spec_header(const PlanDev& d, const int32_t* h_itab);
// resident.hip + that header -> gfx950 code object (needs libhiprtc.so, no device)
// (phases >= 0: the phase mask as a constant of the build; -1: the kernel argument)
int jit_compile(const std::string& header, std::vector<char>* code, std::string* log,
                bool stamps = false, int phases = 0xBF);
// the compiled kernel of (plan structure, device), or nullptr: use the ahead-of-time kernel
const void* jit_kernel_for(const PlanDev& d, const int32_t* h_itab, int device, int batch,
                           size_t lds_bytes);
// the code object of a generated header: from the disk cache ($MPCASM_CACHE_DIR, $XDG_CACHE_HOME/mpcasm
// or ~/.cache/mpcasm; MPCASM_NO_DISK_CACHE=1: never), else compiled and stored there
// (distrust_disk: the cached file was read and did not load -- delete it and compile)
bool jit_code_for(const std::string& header, bool stamps, int phases, std::vector<char>* code,
                  std::string* log, bool distrust_disk = false);
bool jit_available();
// compilations, code objects read from / written to the disk cache, in this process so far
void jit_stats(long out[3]);
// (called when a plan is destroyed: compiled kernels are remembered by the plan's device tables)
void jit_forget(const void* itab);
int jit_launch(const void* kernel, const PlanDev& p, const SrcTable& src, const double* params,
               const double* given, double* P, double* q, double* G, double* h, int batch,
               int num_cus, int per_cu_limit, int grid_limit, void* work, hipStream_t stream, hipError_t* err);

}  // namespace mpcasm
