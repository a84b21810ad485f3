// Lean small-M kernels (w4a16_lean.hpp): instantiations, part c (one translation unit per group of builds: they compile in parallel).
#include "w4a16_common.hpp"
#include "w4a16_lean_inst.hpp"

namespace quick_amd {
QA_LEAN_INSTANTIATE(16, 4, 1)
QA_LEAN_INSTANTIATE(16, 4, 2)
QA_LEAN_INSTANTIATE(16, 8, 1)
}  // namespace quick_amd
